# Diagnostic builds of tools/diag/gemm_qw.hip (four-wave 256 x 256 GEMM with clock stamps and ablation modes): run HERE (hipcc
# cross-compiles), then on the GPU box:  python tools/qw_probe.py [MxNxK ...]   (loads tools/_ab/lib_qwprobe<mode>.so)
#   QW_MODES="0 1 3 4 5 6"  modes to build     QW_EXTRA="-DQW_ST64"  extra defines     QW_SUFFIX=_st64  name suffix
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_ab
for d in ${QW_MODES:-0 1 2 3 4 5 6}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Ineurovit_amd/csrc -DQW_PROBE -DQW_DBG=$d ${QW_EXTRA} \
      -shared -o tools/_ab/lib_qwprobe$d${QW_SUFFIX}.so tools/diag/gemm_qw.hip
done
ls -la tools/_ab/
