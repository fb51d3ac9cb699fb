import torch, sys
sys.path.insert(0, '.')
from neurovit_amd import ops
from neurovit_amd._cabi import lib
for (B, n, heads) in ((4, 513, 12), (20, 513, 12), (1, 65, 3)):
    qkv = torch.randn(B * n, 3 * heads * 64, device='cuda')
    ref = None
    for aw in (4, 2):
        lib.nv_gemm_f32_set_tile(-1, aw)
        out = ops.attn_fwd_f32(qkv, B, n, heads)
        for _ in range(3): ops.attn_fwd_f32(qkv, B, n, heads)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10): ops.attn_fwd_f32(qkv, B, n, heads)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100
        if ref is None: ref = out
        print(f"B{B} n{n} h{heads} waves {aw}: {us:8.1f} us {4.0*B*heads*n*n*64/us/1e6:6.1f} TF  max diff vs 4-wave {float((out-ref).abs().max()):.2e}")
lib.nv_gemm_f32_set_tile(-1, 0)
