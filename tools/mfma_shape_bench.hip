// Micro-benchmark for VERDICT r3 item 4 ("32 x 32 x 16 MFMA in the big-tile GEMM: half the operand register reads per FLOP"): the two
// bf16 MFMA shapes of gfx950 in the loop a 64 x 64 wave tile runs per 64-deep K tile, on RANDOM data (zeros clock higher and hide the
// effect), every CU busy:
//   mode 0  operands in registers (bare loop)                16 x v_mfma_f32_32x32x16_bf16   vs   32 x v_mfma_f32_16x16x32_bf16
//   mode 1  operand fragments re-read from LDS every K tile  (16 ds_read_b128 per K tile either way: the same bytes)
// Both forms do the same FLOPs per iteration (64 x 64 x 64 MACs per wave) on the same accumulator count (64 registers).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_bench mfma_shape_bench.hip ; run on the GPU box.  Prints TFLOP/s chip-wide.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int MODE>   // SHAPE 16 / 32
__global__ __launch_bounds__(256) void k_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ bf16x8 lds[2048];                       // 32 KiB: one 64-deep K tile of a 128 x 128 workgroup tile
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += 256) lds[i] = src[(blockIdx.x * 2048 + i) & 0xFFFF];
  __syncthreads();
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = lds[(lane + 64 * i) & 2047]; b[i] = lds[(lane + 64 * i + 512) & 2047]; }
  if constexpr (SHAPE == 16) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = lds[(lane + 64 * i + it) & 2047]; b[i] = lds[(lane + 64 * i + 512 + it) & 2047]; }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[4 * ks + i], b[4 * ks + j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
  } else {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = lds[(lane + 64 * i + it) & 2047]; b[i] = lds[(lane + 64 * i + 512 + it) & 2047]; }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)          // 64-deep K tile = 4 x 16
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * ks + i], b[2 * ks + j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 256 + tid] = s;
  }
}

template <int SHAPE, int MODE>
static double run(const bf16x8* src, float* out, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k_loop<SHAPE, MODE><<<blocks, 256>>>(src, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) k_loop<SHAPE, MODE><<<blocks, 256>>>(src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 64 * 64 * 64 * iters * 4.0 * blocks * reps;   // per wave 64 x 64 x 64 MACs per iteration, 4 waves
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  const int n = 65536;
  unsigned short* h = (unsigned short*)malloc(n * 16);
  srand(7);
  for (int i = 0; i < n * 8; ++i) {   // random bf16 in roughly [-2, 2): sign, exponent 125..128, random mantissa
    h[i] = (unsigned short)(((rand() & 1) << 15) | ((125 + (rand() & 3)) << 7) | (rand() & 0x7F));
  }
  bf16x8* src; float* out;
  hipMalloc(&src, n * 16); hipMalloc(&out, 4096 * 256 * 4);
  hipMemcpy(src, h, n * 16, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {          // one / two waves per SIMD
    for (int round = 0; round < 2; ++round) {
      printf("blocks %4d (%d wave(s) per SIMD)  registers: 16x16x32 %7.1f  32x32x16 %7.1f TFLOP/s   LDS re-read: 16x16x32 %7.1f  32x32x16 %7.1f TFLOP/s\n", blocks,
             blocks / 256, run<16, 0>(src, out, blocks, 20000), run<32, 0>(src, out, blocks, 20000), run<16, 1>(src, out, blocks, 20000), run<32, 1>(src, out, blocks, 20000));
      fflush(stdout);
    }
  }
  return 0;
}
