// Does the matrix pipe of a gfx950 SIMD run beside its vector ALU - and from WHICH waves?  The question behind the n = 4097 attention forward
// (per 16 x 64 score block 16 MFMAs = 256 cycles, 16 v_exp_f32 + ~44 other vector instructions = 432; measured throughput = the SUM).
// One iteration = what one wave does for one 16-row group and one 64-key tile: 16 x v_mfma_f32_16x16x32_bf16 (eight accumulators, two rounds)
// and a softmax-shaped vector block (16 v_exp_f32, 8 v_pk_fma_f32, 8 v_pk_add_f32, 8 v_max3_f32, 8 v_pk_mul_f32, 8 v_cvt_pk_bf16_f32).
//   mode 0  MFMAs only                      mode 1  vector block only
//   mode 2  phases: 16 MFMAs, then the vector block (what the kernels do today; overlap can only come from OTHER waves of the SIMD)
//   mode 3  interleaved in one wave: after every MFMA one v_exp_f32 and 2.5 other vector instructions (independent of the MFMAs)
// at 1, 2 and 3 waves per SIMD (workgroups of four waves; LDS sized so that exactly k fit a CU).  Instruction order is pinned with asm volatile.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; prints ns per iteration per SIMD and the ratio to mode 0 + mode 1.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define PKFMA(x, c, d) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
#define PKADD(x, y) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define PKMUL(x, y) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define MAX3(m, x, y) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(x), "v"(y))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))

template <int MODE>
__global__ __launch_bounds__(256) void k_overlap(const s16x8* __restrict__ src, float* __restrict__ out, int iters, long long* __restrict__ clk, unsigned long long* __restrict__ rec) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x;
  s16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(tid + 256 * i) & 4095]; b[i] = src[(tid + 256 * i + 1024) & 4095]; }
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 x[8], ps = {0.f, 0.f}, o[8];
  f32x2 c2 = {1.0001f, 1.0001f}, m2 = {-0.5f, -0.5f}, al = {0.999f, 0.999f};
  asm volatile("" : "+v"(c2), "+v"(m2), "+v"(al));      // loop-invariant REGISTERS (not constants rematerialised inside the loop)
  float mx = -1e30f;
  unsigned pk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = f32x2{0.001f * (tid & 31), 0.002f * (tid & 15)}; o[i] = f32x2{1.f, 2.f}; pk[i] = 0; }

  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int rnd = 0; rnd < 2; ++rnd)
#pragma unroll
        for (int i = 0; i < 8; ++i) MFMA(acc[i], a[(i + rnd) & 3], b[i & 3]);
    }
    if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MAX3(mx, x[i][0], x[i][1]);
#pragma unroll
      for (int i = 0; i < 8; ++i) { PKFMA(x[i], c2, m2); EXP(x[i][0]); EXP(x[i][1]); PKADD(ps, x[i]); }
#pragma unroll
      for (int i = 0; i < 8; ++i) PKMUL(o[i], al);
#pragma unroll
      for (int i = 0; i < 8; ++i) CVT(pk[i], x[i][0], x[i][1]);
    }
    if constexpr (MODE == 4 || MODE == 5) {       // per slot: [MFMA +] four independent v_pk_fma_f32 (no transcendental)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if constexpr (MODE == 4) MFMA(acc[s & 7], a[(s + (s >> 3)) & 3], b[s & 3]);
        PKFMA(x[(s) & 7], c2, m2); PKFMA(x[(s + 2) & 7], c2, m2); PKFMA(o[s & 7], al, m2); PKFMA(o[(s + 2) & 7], al, m2);
      }
    }
    if constexpr (MODE == 6 || MODE == 7) {       // per slot: [MFMA +] one v_exp_f32
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if constexpr (MODE == 6) MFMA(acc[s & 7], a[(s + (s >> 3)) & 3], b[s & 3]);
        EXP(x[s >> 1][s & 1]);
      }
    }
    if constexpr (MODE == 8) {
      // the schedule the finding suggests: 16 MFMAs with ONLY co-issuing vector work between them (v_exp_f32, v_max3_f32, v_cvt_pk_bf16_f32),
      // then the packed arithmetic (v_pk_fma / v_pk_add / v_pk_mul) in one block while no MFMA is in flight
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int i = s >> 1, h = s & 1;
        MFMA(acc[s & 7], a[(s + (s >> 3)) & 3], b[s & 3]);
        EXP(x[i][h]);
        if (h == 0) MAX3(mx, x[i][0], x[i][1]); else CVT(pk[i], x[i][0], x[i][1]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) PKFMA(x[i], c2, m2);
#pragma unroll
      for (int i = 0; i < 8; ++i) PKADD(ps, x[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) PKMUL(o[i], al);
    }
    if constexpr (MODE == 3) {
      // 16 slots: slot s = one MFMA, one v_exp_f32, and the other 40 vector instructions spread 2-3 per slot
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int i = s >> 1, h = s & 1;
        MFMA(acc[s & 7], a[(s + (s >> 3)) & 3], b[s & 3]);
        if (h == 0) { MAX3(mx, x[i][0], x[i][1]); PKFMA(x[i], c2, m2); }
        EXP(x[i][h]);
        if (h == 1) { PKADD(ps, x[i]); PKMUL(o[i], al); CVT(pk[i], x[i][0], x[i][1]); }
      }
    }
  }
  if (tid == 0) {      // where and when this workgroup ran: (XCC_ID << 32 | HW_ID), start, end on the 100 MHz counter
    rec[3 * blockIdx.x] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
    rec[3 * blockIdx.x + 1] = (unsigned long long)w0; rec[3 * blockIdx.x + 2] = (unsigned long long)wall_clock64();
  }
  if (blockIdx.x == 7 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }      // shader cycles and 100 MHz ticks of one wave's loop
  float s = mx + ps[0] + ps[1];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + o[i][0] + o[i][1] + (float)pk[i] + x[i][0];
  out[blockIdx.x * 256 + tid] = s + (float)lds[tid & 15];
}

#define FMA1(x, c, d) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
#define MUL1(x, c) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c))
#define MAXB(m, x, y) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(m) : "v"(x), "v"(y))
// per slot: [one MFMA +] four independent vector instructions of ONE kind: which kinds run beside the matrix pipe?
//   OP 0 v_fma_f32   1 v_pk_fma_f32   2 v_max3_f32   3 v_cvt_pk_bf16_f32   4 v_pk_mul_f32   5 v_pk_add_f32   6 v_mul_f32   7 v_exp_f32 (x 2)
template <int OP, bool WITH>
__global__ __launch_bounds__(256) void k_kind(const s16x8* __restrict__ src, float* __restrict__ out, int iters, long long* __restrict__ clk) {
  const int tid = threadIdx.x;
  s16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(tid + 256 * i) & 4095]; b[i] = src[(tid + 256 * i + 1024) & 4095]; }
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 x[8], c2 = {1.0001f, 1.0001f}, m2 = {-0.5f, -0.5f};
  unsigned pk[4] = {0, 0, 0, 0};
  asm volatile("" : "+v"(c2), "+v"(m2));
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f32x2{0.001f * (tid & 31), 0.002f * (tid & 15)};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if constexpr (WITH) MFMA(acc[s & 7], a[(s + (s >> 3)) & 3], b[s & 3]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = (2 * s + u) & 7;
        if constexpr (OP == 0) FMA1(x[i][0], c2[0], m2[0]);
        if constexpr (OP == 1) PKFMA(x[i], c2, m2);
        if constexpr (OP == 2) MAXB(x[i][0], c2[0], m2[1]);
        if constexpr (OP == 3) CVT(pk[u], x[i][0], x[i][1]);
        if constexpr (OP == 4) PKMUL(x[i], c2);
        if constexpr (OP == 5) PKADD(x[i], m2);
        if constexpr (OP == 6) MUL1(x[i][0], c2[0]);
        if constexpr (OP == 7) { if (u < 2) EXP(x[i][u]); }
      }
    }
  }
  if (blockIdx.x == 7 && tid == 0) clk[0] = clock64() - c0;
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3] + x[i][0] + x[i][1];
  out[blockIdx.x * 256 + tid] = r + (float)(pk[0] + pk[1] + pk[2] + pk[3]);
}
template <int OP, bool WITH>
static double run_kind(const s16x8* src, float* out, int iters, long long* clk) {
  hipLaunchKernelGGL((k_kind<OP, WITH>), dim3(256), dim3(256), 0, 0, src, out, iters, clk);
  hipLaunchKernelGGL((k_kind<OP, WITH>), dim3(256), dim3(256), 0, 0, src, out, iters, clk);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  return (double)h / iters;
}
template <int OP>
static void kind_line(const char* name, int per_slot, const s16x8* src, float* out, int iters, long long* clk, double mfma) {
  const double alone = run_kind<OP, false>(src, out, iters, clk), with = run_kind<OP, true>(src, out, iters, clk);
  printf("%-20s x %d per MFMA: alone %6.1f cycles (%.1f each)   beside 16 MFMAs %6.1f   = %.2f of the sum, %.2f of the larger\n", name, per_slot, alone, alone / (16 * per_slot), with,
         with / (alone + mfma), with / (alone > mfma ? alone : mfma));
}

template <int MODE>
static double run(const s16x8* src, float* out, int per_cu, int iters, long long* clk, double* cyc, double* mhz, double* conc = nullptr) {
  static unsigned long long* rec = nullptr;
  if (!rec) hipMalloc(&rec, 768 * 3 * 8);
  const int lds = per_cu == 1 ? 150 * 1024 : per_cu == 2 ? 76 * 1024 : 50 * 1024;     // exactly per_cu workgroups fit 160 KiB
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_overlap<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * per_cu;
  hipLaunchKernelGGL(k_overlap<MODE>, dim3(blocks), dim3(256), lds, 0, src, out, iters, clk, rec);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_overlap<MODE>, dim3(blocks), dim3(256), lds, 0, src, out, iters, clk, rec);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  if (conc) {     // mean number of workgroups in flight per CU = sum of the workgroups' lifetimes / (span x CUs that ran any)
    static unsigned long long hr[768 * 3];
    hipMemcpy(hr, rec, blocks * 24, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0; double life = 0; int ncu = 0; static unsigned long long seen[1024];
    for (int b = 0; b < blocks; ++b) {
      lo = hr[3 * b + 1] < lo ? hr[3 * b + 1] : lo; hi = hr[3 * b + 2] > hi ? hr[3 * b + 2] : hi; life += (double)(hr[3 * b + 2] - hr[3 * b + 1]);
      const unsigned long long where = (hr[3 * b] >> 32 << 32) | (hr[3 * b] & 0x0000FF00u) | ((hr[3 * b] >> 12 & 1) << 16);     // XCC, SE / SH / CU fields of HW_ID (bits 8..15: cu 8-11, sh 12, se 13-15)
      int k = 0; while (k < ncu && seen[k] != where) ++k;
      if (k == ncu) seen[ncu++] = where;
    }
    *conc = life / ((double)(hi - lo) * ncu);
    conc[1] = ncu;
  }
  *cyc = (double)h[0] / iters / per_cu; *mhz = (double)h[0] / ((double)h[1] / 100.0);
  return best * 1e6 / iters / per_cu;      // ns of SIMD time per (iteration of one wave)
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  s16x8* src; float* out; long long* clk; hipMalloc(&clk, 16);
  hipMalloc(&src, 4096 * sizeof(s16x8)); hipMalloc(&out, 256 * 3 * 256 * sizeof(float));
  short* h = (short*)malloc(4096 * 16);
  srand(1);
  for (int i = 0; i < 4096 * 8; ++i) { const float f = (rand() % 2001 - 1000) * 1e-3f; unsigned u; memcpy(&u, &f, 4); h[i] = (short)(u >> 16); }
  hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
  printf("# per iteration of one wave (16 MFMAs and / or one softmax-shaped vector block), divided by the waves per SIMD: ns from events, shader cycles\n");
  printf("# from s_memtime, clock from s_memtime / s_memrealtime.  phases = 16 MFMAs then the vector block; interleaved = one MFMA, 3-4 vector instructions, ...\n");
  const char* name[9] = {"mfma_only", "valu_only", "phases", "interleaved", "mfma+4pkfma", "4pkfma_only", "mfma+exp", "exp_only", "co-issue+packed"};
  for (int k = 1; k <= 3; ++k) {
    double t[9], c[9], f[9], occ[2];
    t[0] = run<0>(src, out, k, iters, clk, &c[0], &f[0], occ);
    printf("waves/SIMD %d asked for: %.2f workgroups in flight per CU on average over %d CUs (mfma_only run)\n", k, occ[0], (int)occ[1]); t[1] = run<1>(src, out, k, iters, clk, &c[1], &f[1]);
    t[2] = run<2>(src, out, k, iters, clk, &c[2], &f[2]); t[3] = run<3>(src, out, k, iters, clk, &c[3], &f[3]);
    t[4] = run<4>(src, out, k, iters, clk, &c[4], &f[4]); t[5] = run<5>(src, out, k, iters, clk, &c[5], &f[5]);
    t[6] = run<6>(src, out, k, iters, clk, &c[6], &f[6]); t[7] = run<7>(src, out, k, iters, clk, &c[7], &f[7]);
    t[8] = run<8>(src, out, k, iters, clk, &c[8], &f[8]);
    for (int m = 0; m < 9; ++m) printf("waves/SIMD %d  %-12s %7.1f ns of SIMD time   %7.1f cycles of one wave   %6.0f MHz\n", k, name[m], t[m], c[m] * k, f[m]);
    printf("waves/SIMD %d  ns: phases / (mfma + valu) = %.2f   interleaved / (mfma + valu) = %.2f   interleaved / max(mfma, valu) = %.2f\n", k, t[2] / (t[0] + t[1]),
           t[3] / (t[0] + t[1]), t[3] / (t[0] > t[1] ? t[0] : t[1]));
  }
  printf("# one wave per SIMD, cycles of one wave per iteration: 16 MFMAs (one per slot) with vector instructions of ONE kind in every slot\n");
  const double mfma = run_kind<0, true>(src, out, iters, clk) * 0 + 284.0;
  kind_line<0>("v_fma_f32", 4, src, out, iters, clk, mfma);
  kind_line<6>("v_mul_f32", 4, src, out, iters, clk, mfma);
  kind_line<2>("v_max3_f32", 4, src, out, iters, clk, mfma);
  kind_line<3>("v_cvt_pk_bf16_f32", 4, src, out, iters, clk, mfma);
  kind_line<1>("v_pk_fma_f32", 4, src, out, iters, clk, mfma);
  kind_line<4>("v_pk_mul_f32", 4, src, out, iters, clk, mfma);
  kind_line<5>("v_pk_add_f32", 4, src, out, iters, clk, mfma);
  kind_line<7>("v_exp_f32", 2, src, out, iters, clk, mfma);
  return 0;
}
