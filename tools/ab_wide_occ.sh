# Same-box A/B of the wide attention forward at two register budgets (2 vs 3 workgroups per CU).  Build HERE:  bash tools/ab_wide_occ.sh build
# then on the GPU box:  bash tools/ab_wide_occ.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  mkdir -p tools/_ab
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Ineurovit_amd/csrc -DNV_WIDE_FWD_BLOCKS=3 -c neurovit_amd/csrc/attention.hip -o /tmp/attn_occ3.o
  objs=$(ls neurovit_amd/lib/obj/*.o | grep -v "/attention.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/lib_occ3.so $objs /tmp/attn_occ3.o
  ls -la tools/_ab/lib_occ3.so
else
  for i in 1 2; do
    python bench.py --preset large --forward-only --steps 8 --warmup 2 2>/dev/null | cut -c75-110
    NEUROVIT_HIP_LIB=$PWD/tools/_ab/lib_occ3.so python bench.py --preset large --forward-only --steps 8 --warmup 2 2>/dev/null | cut -c75-110
  done
fi
