#!/usr/bin/env python3
"""Static check of the emitted gfx950 ISA for the data hazards the hardware does not interlock and that hipcc cannot see through an
`asm` statement (cdna_hip_programming.md 5.7): the attention kernels keep their cross-lane maxima / sums in inline asm
(v_permlane16_swap / v_permlane32_swap behind an `s_nop 1`, bare v_max_f32) next to MFMA results.

Rules checked on the linear instruction stream of every kernel of a code object (wait states = instructions issued in between, an
`s_nop N` counting N + 1; the counts are those of LLVM's GCNHazardRecognizer for gfx940 / gfx950):
  H1  v_mfma / v_smfmac write of VGPR r  ->  any non-matrix instruction that reads r        >= passes + 4 wait states (XDL: bf16 / f16 /
      fp8 / scaled inputs; passes + 3, +1 on gfx950), >= passes + 2 for the f32-input forms (not XDL); passes = issue cycles / 4:
      16x16x32 bf16 = 4, 32x32x16 bf16 = 8, scaled 16x16x128 f8f6f4 = 8, 16x16x4 f32 = 8.  Calibrated against what hipcc itself
      emits where no asm is involved (gemm.o: 8 behind v_mfma_f32_16x16x32_bf16; precise.o: 10 behind v_mfma_f32_16x16x4_f32)
  H2  VALU write of VGPR r  ->  v_permlane16_swap / v_permlane32_swap that reads r           >= 2 wait states
  H3  v_permlane*_swap write of VGPR r  ->  VALU read of r                                   >= 1 wait state
A hazard is only looked for inside a straight-line window of WINDOW instructions (no tracking across branches: the kernels' loops are
fully unrolled around the asm statements).  usage: isa_hazards.py <object.o | code object> [...]; exit status 1 on a violation."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
WINDOW = 24
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def mfma_need(op):
    """wait states between this matrix instruction and a non-matrix reader of its result"""
    f32_in = bool(re.search(r"x\d+_f32$|x\d+f32$", op.split("_e64")[0]))       # v_mfma_f32_16x16x4_f32, v_mfma_f32_32x32x2_f32
    if "32x32" in op:
        passes = 16 if f32_in else 8
    elif "16x16x128" in op or "16x16x64" in op or f32_in:
        passes = 8
    else:
        passes = 4
    return passes + (2 if f32_in else 4)


def device_code(path, workdir):
    """path of the gfx950 code object inside a host object (or the path itself when it already is one)."""
    head = subprocess.run([f"{LLVM}/llvm-readelf", "-h", path], capture_output=True, text=True).stdout
    if "AMDGPU" in head or "AMD GPU" in head:
        return path
    local = os.path.join(workdir, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], capture_output=True, cwd=workdir, check=True)
    for f in sorted(os.listdir(workdir)):
        if f.startswith(os.path.basename(path) + ".") and "gfx950" in f:
            return os.path.join(workdir, f)
    return None          # a host-only translation unit (engine.hip: launch sequencing, no kernels)


def parse(code_object):
    """{kernel: [(mnemonic, operand text)]}"""
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", code_object], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return kernels


def split_operands(op, text):
    """(written registers, read registers) of one instruction - the first operand is the destination for everything looked at here;
    v_permlane*_swap reads AND writes both of its operands."""
    ops = [o.strip() for o in text.split(",")]
    if not ops or not ops[0]:
        return set(), set()
    if op.startswith(("v_permlane16_swap", "v_permlane32_swap")):
        both = regs(text)
        return both, both
    if op.startswith(("ds_write", "ds_store", "buffer_store", "global_store", "scratch_store", "flat_store", "s_", "buffer_atomic", "global_atomic")):
        return set(), regs(text)
    if op.startswith(("v_cmp", "v_cmpx")):
        return set(), regs(text)
    wr = regs(ops[0])
    rd = regs(",".join(ops[1:]))
    if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_mfma", "v_smfmac")) or "accum" in op:
        rd |= wr if not op.startswith(("v_mfma", "v_smfmac")) else set()
    return wr, rd


def check(kernels):
    problems, stats = [], {"H1": None, "H2": None, "H3": None}

    def note(rule, dist):
        if stats[rule] is None or dist < stats[rule]:
            stats[rule] = dist

    for name, ins in kernels.items():
        for i, (op, text) in enumerate(ins):
            is_mfma = op.startswith(("v_mfma", "v_smfmac"))
            is_swap = op.startswith(("v_permlane16_swap", "v_permlane32_swap"))
            is_valu = op.startswith("v_") and not is_mfma
            _, rd = split_operands(op, text)
            if is_mfma:
                # an MFMA's own operand / accumulator hazards are the compiler's (no asm MFMA in the tree): only its consumers are checked
                continue
            if not rd:
                continue
            waits = 0
            for j in range(i - 1, max(-1, i - 1 - WINDOW), -1):
                pop, ptext = ins[j]
                if pop.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_barrier", "s_setpc", "s_swappc")):
                    break
                pwr, _ = split_operands(pop, ptext)
                hit = pwr & rd
                if hit:
                    if pop.startswith(("v_mfma", "v_smfmac")):
                        need = mfma_need(pop)
                        note("H1", waits - need)
                        if waits < need:
                            problems.append(f"H1 {name}: +{j} `{pop} {ptext}` -> +{i} `{op} {text}`: {waits} wait states, {need} needed ({sorted(hit)[:2]})")
                    elif is_swap and pop.startswith("v_"):
                        note("H2", waits - 2)
                        if waits < 2:
                            problems.append(f"H2 {name}: +{j} `{pop} {ptext}` -> +{i} `{op} {text}`: {waits} wait states, 2 needed")
                    elif is_valu and pop.startswith(("v_permlane16_swap", "v_permlane32_swap")):
                        note("H3", waits - 1)
                        if waits < 1:
                            problems.append(f"H3 {name}: +{j} `{pop} {ptext}` -> +{i} `{op} {text}`: {waits} wait states, 1 needed")
                    rd = rd - hit
                    if not rd:
                        break
                if pop == "s_nop":
                    waits += int(ptext.strip() or 0, 0) + 1
                elif not pop.startswith(("s_waitcnt", ".")):
                    waits += 1
                else:
                    waits += 1
    return problems, stats


def main(paths):
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for p in paths:
            code = device_code(p, td)
            if code is None:
                print(f"{os.path.basename(p)}: no device code")
                continue
            kernels = parse(code)
            problems, stats = check(kernels)
            n_ins = sum(len(v) for v in kernels.values())
            print(f"{os.path.basename(p)}: {len(kernels)} functions, {n_ins} instructions; smallest slack (wait states beyond the requirement): {stats}")
            for line in problems[:40]:
                print("  " + line)
            bad += len(problems)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
