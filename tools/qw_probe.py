#!/usr/bin/env python3
"""In-kernel clock of the diagnostic four-wave 256 x 256 GEMM (tools/diag/gemm_qw.hip, builds made by tools/qw_probe.sh): shader cycles / real time around the
K loop of every workgroup -> the frequency the chip sustains under this kernel, and the matrix-pipe utilisation inside the loop
(64 MFMAs of 16 cycles per 32-deep stage).  One child process per build (the library is chosen at import)."""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import ctypes, os, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(HERE))
from neurovit_amd import ops
M, N, K = (int(v) for v in os.environ["QW_SHAPE"].split("x"))
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
dbg = ctypes.CDLL(os.environ["QW_LIB"])
dbg.nv_debug_gemm_qw.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_long] * 3 + [ctypes.c_void_p]
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def run():
    assert dbg.nv_debug_gemm_qw(M, N, K, A.data_ptr(), K, B.data_ptr(), K, out.data_ptr(), N, torch.cuda.current_stream().cuda_stream) == 0
run(); torch.cuda.synchronize()
if os.environ["QW_MODE"] == "0":                             # the full kernel must agree with the library's GEMM
    ref = ops.gemm(ops.NT, ops.EPI_STORE_BF16, A, B)
    err = float((out.float() - ref.float()).abs().max() / ref.float().abs().max())
    assert err < 1e-2, err
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 2.5:                       # >= 2 s of back-to-back launches: the clock has settled
    for _ in range(50):
        run()
    torch.cuda.synchronize(); n += 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
blocks = min(4096, ((M + 255) // 256) * ((N + 255) // 256))
buf = (ctypes.c_ulonglong * (4 * blocks))()
f = dbg.nv_debug_qw_probe
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert f(buf, blocks) == 0
mhz, cyc = [], []
for b in range(blocks):
    dc, dr, nk = buf[4 * b], buf[4 * b + 1], buf[4 * b + 3]
    if dr:
        mhz.append(dc / dr * 100.0); cyc.append(dc / nk)
c = statistics.median(cyc)
print(f"{os.environ['QW_TAG']:34s} {M}x{N}x{K}: {us:8.1f} us/launch = {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s; in-loop clock {statistics.median(mhz):6.0f} MHz "
      f"(min {min(mhz):.0f}, max {max(mhz):.0f}); {c:6.0f} cycles per 32-deep stage = {100 * 1024 / c:4.1f} % matrix-pipe utilisation")
'''


def main():
    shapes = sys.argv[1:] or ["8192x8192x4096"]
    for shape in shapes:
        modes = {0: "full kernel", 1: "no staging in the loop", 2: "matrix instructions only", 3: "staging with every load killed", 4: "staging + reads, 1/8 of the MFMAs",
                 5: "ds_write only", 6: "global loads only"}
        for d in [int(v) for v in os.environ.get("QW_MODES", "0 1 2 3 4 5 6").split()]:
            tag = modes[d]
            lib = os.path.join(HERE, "_ab", f"lib_qwprobe{d}{os.environ.get('QW_SUFFIX', '')}.so")
            if not os.path.exists(lib):
                sys.exit(f"{lib} missing: run tools/qw_probe.sh in the build container first")
            env = dict(os.environ, QW_LIB=lib, QW_MODE=str(d), QW_SHAPE=shape, QW_TAG=tag + os.environ.get("QW_SUFFIX", ""))
            subprocess.run([sys.executable, "-c", "HERE = %r\n" % HERE + CHILD], env=env, check=True)


if __name__ == "__main__":
    main()
