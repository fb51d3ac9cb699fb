"""Readable kernel names for the rocprofv3 CSVs.  Since round 5 every 16-bit kernel is a template over its operand element type
(__bf16 / _Float16); rocprofv3's demangler leaves names with those types (Itanium codes DF16b / DF16_) mangled.  They are rewritten
to codes every demangler knows - the vendor type `u6__bf16` and `Dh` (half) - and passed through c++filt."""
import re
import subprocess


def pretty_many(names):
    """[raw Kernel_Name] -> [demangled name] (unchanged when it was readable already)"""
    todo = sorted({n for n in names if n.startswith("_Z")})
    table = {}
    if todo:
        fixed = [n.replace("DF16b", "u6__bf16").replace("DF16_", "Dh") for n in todo]
        try:
            out = subprocess.run(["c++filt"], input="\n".join(fixed), capture_output=True, text=True, check=True).stdout.split("\n")
            table = {n: (o.replace("half", "_Float16") if o else n) for n, o in zip(todo, out)}
        except (OSError, subprocess.CalledProcessError):
            table = {}
    return [table.get(n, n) for n in names]


def kernel_key(name: str) -> str:
    """'void (anonymous namespace)::gemm_pp_kernel<__bf16, 256, 128, 4, 2, false, true, 6>(GemmArgs)' -> 'gemm_pp_kernel<__bf16, 256, 128, 4, 2, false, true, 6>'"""
    s = pretty_many([name.strip()])[0]
    if s.endswith(")"):                      # strip the final balanced argument list
        depth, i = 0, len(s) - 1
        while i >= 0:
            if s[i] == ")":
                depth += 1
            elif s[i] == "(":
                depth -= 1
                if depth == 0:
                    break
            i -= 1
        if i > 0:
            s = s[:i]
    if s.startswith("void "):
        s = s[5:]
    return s.replace("(anonymous namespace)::", "").strip()


_T = r"(?:(?:__bf16|_Float16), )?"          # the operand element type leads the template arguments of the GEMM kernels


def kind_of(key: str):
    """kernel instantiation -> nv_prof kind (include/neurovit_hip.h, nv_prof_enable): the classes bench.py's roofline leg times"""
    def tn(m, base):
        a_t, b_t = m.group(1) == "true", m.group(2) == "true"
        return base + (2 if a_t else (1 if b_t else 0))
    m = re.match(r"gemm_ws_kernel<" + _T + r"\d+, \d+, \d+, \d+, (true|false), (true|false), \d+>", key)
    if m:
        return tn(m, 0)
    if key.startswith("gemm_ws_grouped_kernel"):
        return 2
    m = re.match(r"gemm_pp_kernel<" + _T + r"\d+, \d+, \d+, \d+, (true|false), (true|false), \d+(, (true|false))?>", key)
    if m:
        return 5 if m.group(4) == "true" else tn(m, 10)
    if key.startswith("gemm_pp_f8_kernel"):
        return 5
    if key.startswith("gemm_pp_grouped_tn_adamw_kernel"):
        return 14
    if key.startswith("gemm_pp_grouped_tn_kernel"):
        return 13
    m = re.match(r"gemm_pq_kernel<(true|false), (true|false), \d+, (true|false)(?:, (?:__bf16|_Float16))?>", key)
    if m:
        return 5 if m.group(3) == "true" else tn(m, 20)
    if key.startswith("attn_fwd"):
        return 3
    if key.startswith("attn_bwd"):
        return 4
    if key.startswith("gemm_f32_nt_kernel"):
        return 30
    if key.startswith("attn_f32_fwd_kernel"):
        return 31
    return None
