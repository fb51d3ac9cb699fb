// Micro-benchmark: L2 -> LDS streaming rate per CU in the GEMM access pattern (tuning aid, not part of the product).
// Each workgroup repeatedly DMA-loads [rows x 128 B] tiles of a small (L2-resident) matrix into an LDS ring.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_stream_bench l2_stream_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;

template <int DEPTH, int PIECES, bool REG>   // PIECES = 1-KiB wave-instructions per wave per tile
__global__ __launch_bounds__(256) void stream_kernel(const char* src, long ld_bytes, int rows_total, int ktiles, int iters, float* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned bytes = (unsigned)((long)rows_total * ld_bytes);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // tile = 4 waves x PIECES pieces x 8 rows x 128 B ; block b starts at row (b * 37) % rows
  const int rows_per_tile = 4 * PIECES * 8;
  const int row0 = (blockIdx.x * 64) % (rows_total - rows_per_tile);
  int voff[PIECES];
  for (int i = 0; i < PIECES; ++i) voff[i] = (int)((long)(row0 + (wid * PIECES + i) * 8 + (lane >> 3)) * ld_bytes + (lane & 7) * 16);
  const int tile_bytes = rows_per_tile * 128;
  float acc = 0.f;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 regs[PIECES];
  int t = 0;
  for (int it = 0; it < iters; ++it) {
    const int soff = (t % ktiles) * 128;
    char* dst = smem + (it % DEPTH) * tile_bytes;
    if constexpr (REG) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) regs[i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff[i], soff, 0);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) *(u32x4*)(dst + (wid * PIECES + i) * 1024 + lane * 16) = regs[i];
    } else {
#pragma unroll
      for (int i = 0; i < PIECES; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(dst + (wid * PIECES + i) * 1024), 16, voff[i], soff, 0, 0);
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (DEPTH == 2) { if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
      else if (DEPTH == 3) { if (PIECES == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
      else { if (PIECES == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
    }
    __builtin_amdgcn_s_barrier();
    acc += *(float*)(smem + ((it + 1) % DEPTH) * tile_bytes + threadIdx.x * 4);
    ++t;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 123.456f) sink[0] = acc;
#endif
}

template <int DEPTH, int PIECES, bool REG>
void run(const char* name, const char* d, long ld, int rows, int ktiles, int blocks, float* sink) {
  const int iters = 2000;
  const int tile_bytes = 4 * PIECES * 8 * 128;
  const int lds = DEPTH * tile_bytes;
  auto k = stream_kernel<DEPTH, PIECES, REG>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<<<blocks, 256, lds>>>(d, ld, rows, ktiles, 100, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<<<blocks, 256, lds>>>(d, ld, rows, ktiles, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * iters * tile_bytes;
  printf("%-34s blocks=%4d tile=%2d KiB depth=%d lds=%3d KiB : %7.2f TB/s  %6.1f GB/s per CU (256 CUs)\n", name, blocks, tile_bytes / 1024, DEPTH,
         lds / 1024, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  const int rows = 2304, K = 768;            // a [2304, 768] bf16 matrix (3.5 MB): the qkv weight
  const long ld = K * 2;
  char* d; float* sink;
  hipMalloc(&d, (size_t)rows * ld); hipMemset(d, 1, (size_t)rows * ld); hipMalloc(&sink, 4);
  const int kt = K / 64;
  for (int blocks : {256, 512, 1024}) {
    run<1, 4, false>("dma 16K tile", d, ld, rows, kt, blocks, sink);
    run<2, 4, false>("dma 16K tile", d, ld, rows, kt, blocks, sink);
    run<3, 4, false>("dma 16K tile", d, ld, rows, kt, blocks, sink);
    run<4, 4, false>("dma 16K tile", d, ld, rows, kt, blocks, sink);
    run<2, 8, false>("dma 32K tile", d, ld, rows, kt, blocks, sink);
    run<3, 8, false>("dma 32K tile", d, ld, rows, kt, blocks, sink);
    run<4, 8, false>("dma 32K tile", d, ld, rows, kt, blocks, sink);
    run<2, 4, true>("regs 16K tile", d, ld, rows, kt, blocks, sink);
    run<2, 8, true>("regs 32K tile", d, ld, rows, kt, blocks, sink);
  }
  return 0;
}
