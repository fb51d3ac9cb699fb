// fp32 inference path ("precise" mode) of the ViT3D encoder, gfx950.
//
// The reference validates in fp32 with no autocast (src/Trainer.py:101-118) and BASELINE.json asks for logits within 1e-3 of the
// reference's CPU forward.  bf16 MFMA operands cannot hold that (profiles/r02_cast_point_ablation.txt: bf16 weights alone cost
// 1e-3 ... 6e-3 on the logits), so this path keeps EVERY operand in fp32 and runs the contractions on the fp32 matrix instruction
// v_mfma_f32_16x16x4_f32 (64 FLOP/clk/SIMD = 1/16 of the bf16 rate, bit-equivalent to an fmaf chain - MI355X_MICROARCH.md,
// "Matrix cores"): weights are read straight from the fp32 parameter arena (no shadow copy), activations stay fp32 end to end.
//
//   gemm_f32_nt_kernel   every nn.Linear of the path (vit_3d.py:19,22,41,44,94): C = epi(A[M,K] . W[N,K]^T)
//   attn_f32_fwd_kernel  softmax(q k^T * scale) v, flash-style (vit_3d.py:53-59)
//   nv_vit_forward_f32   ViT.forward (vit_3d.py:112-126) sequenced over them + the fp32 row kernels of norm.hip
//
// Design notes.  At 1/16 of the bf16 MFMA rate the matrix pipe is the only limiter: one 16x16x4 MFMA (32 cycles per SIMD) consumes
// 8 B per lane, so operands come straight from global memory / L1 in 16-byte pieces with a register double buffer - no LDS, no
// barriers in the GEMM.  Both operands of an NT product are K-contiguous, and a dot product does not care about the order of k:
// lane (i, g) loads the float4 A[i][k0 + 4g .. 4g+3] and the t-th of four MFMAs takes component t of every lane, i.e. the k set
// {4g + t}; the four together cover k0 .. k0+15.  The operands are passed swapped (weights as the MFMA's A, activations as its B)
// so that a lane ends up with four CONSECUTIVE output columns of one row: float4 epilogue loads and stores.
#include <math.h>

#include "common.h"

namespace {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float gelu_exact(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }

// ------------------------------------------------------------------------------------------------ GEMM
// Workgroup tile (32 WM) x (32 WN) x 32, 2 x 2 waves of (16 WM) x (16 WN).  EPI: 0 store, 2 + bias, 3 gelu(+ bias), 4 resid + (+ bias).
// Operands are staged through LDS in whole 128-byte lines (eight lanes per row: 8 x 16 B), one register set in flight under the
// MFMAs of the current step, two LDS buffers, one barrier per 32-deep step.  (First built with the fragments loaded straight from
// global memory - no LDS, no barrier: 37 % of the fp32 MFMA peak.  A fragment-shaped load touches 16 rows x 64 B = sixteen half
// cache lines per instruction, and the CU's one L1 / TA path serves four SIMDs: address-path time per MFMA time came out as
// (WM + WN) / (WM WN) = 1 for the 2 x 2 tile the load balance wants.  Through LDS every line is fetched once per workgroup, whole.)
// ALIGNED: lda, ldb multiples of 4, 16-byte aligned bases, K % 4 == 0 (float4 loads); otherwise scalar loads with per-element
// bounds (the reference's default patch_dim 729 = 9^3, configs/config.yaml:39-40).
constexpr int F32_BK = 32, F32_LD = F32_BK + 4;      // LDS row pitch in floats: 144 B = 9 x 16 B, sixteen rows hit sixteen different 16-byte bank groups
template <int WM, int WN, int EPI, bool ALIGNED>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(int M, int N, int K, const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                                          float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                          const float* __restrict__ resid, long ldr, int tiles_n) {
  constexpr int BM = 32 * WM, BN = 32 * WN;
  extern __shared__ __attribute__((aligned(16))) float fsm[];      // [2][(BM + BN)][F32_LD]
  constexpr int BUF = (BM + BN) * F32_LD;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  // staging: thread -> (row t / 8 + 32 u, chunk t % 8): eight lanes read one whole 128-byte line of a row
  const int srow = tid >> 3, sch = tid & 7;
  const float* ap[WM];
  const float* bp[WN];
#pragma unroll
  for (int u = 0; u < WM; ++u) ap[u] = A + (long)min(m0 + srow + 32 * u, M - 1) * lda + 4 * sch;
#pragma unroll
  for (int u = 0; u < WN; ++u) bp[u] = B + (long)min(n0 + srow + 32 * u, N - 1) * ldb + 4 * sch;
  f32x4 ra[WM], rb[WN];
  auto ld4 = [&](const float* p, int k0) -> f32x4 {     // elements k0 + 4 sch .. + 3 of the thread's row, zero beyond K
    const int k = k0 + 4 * sch;
    if constexpr (ALIGNED) return (k < K) ? *reinterpret_cast<const f32x4*>(p + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k + e < K) ? p[k0 + e] : 0.f;
    return v;
  };
  auto gload = [&](int k0) {
#pragma unroll
    for (int u = 0; u < WM; ++u) ra[u] = ld4(ap[u], k0);
#pragma unroll
    for (int u = 0; u < WN; ++u) rb[u] = ld4(bp[u], k0);
  };
  auto swrite = [&](float* buf) {
#pragma unroll
    for (int u = 0; u < WM; ++u) *reinterpret_cast<f32x4*>(buf + (srow + 32 * u) * F32_LD + 4 * sch) = ra[u];
#pragma unroll
    for (int u = 0; u < WN; ++u) *reinterpret_cast<f32x4*>(buf + (BM + srow + 32 * u) * F32_LD + 4 * sch) = rb[u];
  };
  const int wm = wid >> 1, wn = wid & 1;
  const int i = lane & 15, g = lane >> 4;
  f32x4 acc[WM][WN];
#pragma unroll
  for (int bm = 0; bm < WM; ++bm)
#pragma unroll
    for (int bn = 0; bn < WN; ++bn) acc[bm][bn] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const float* buf) {
    const float* a0 = buf + (wm * 16 * WM + i) * F32_LD + 4 * g;
    const float* b0 = buf + (BM + wn * 16 * WN + i) * F32_LD + 4 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 af[WM], bf[WN];
#pragma unroll
      for (int bm = 0; bm < WM; ++bm) af[bm] = *reinterpret_cast<const f32x4*>(a0 + 16 * bm * F32_LD + 16 * ks);
#pragma unroll
      for (int bn = 0; bn < WN; ++bn) bf[bn] = *reinterpret_cast<const f32x4*>(b0 + 16 * bn * F32_LD + 16 * ks);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int bm = 0; bm < WM; ++bm)
#pragma unroll
          for (int bn = 0; bn < WN; ++bn) acc[bm][bn] = mfma4(bf[bn][t], af[bm][t], acc[bm][bn]);
    }
  };
  const int nk = (K + F32_BK - 1) / F32_BK;
  gload(0);
  swrite(fsm);
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    float* cur = fsm + (s & 1) * BUF;
    if (s + 1 < nk) gload((s + 1) * F32_BK);           // in flight under this step's MFMAs
    compute(cur);
    if (s + 1 < nk) swrite(fsm + ((s + 1) & 1) * BUF);  // that buffer was last read in step s - 1: every wave has passed its barrier
    __syncthreads();
  }
  // lane (j, q), register r of block (bm, bn) holds C[m0 + 16 (WM wm + bm) + j][n0 + 16 (WN wn + bn) + 4 q + r]
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int bm = 0; bm < WM; ++bm) {
    const int row = m0 + 16 * (WM * wm + bm) + j;
    if (row >= M) continue;
#pragma unroll
    for (int bn = 0; bn < WN; ++bn) {
      const int col = n0 + 16 * (WN * wn + bn) + 4 * q;
      if (col >= N) continue;                // N % 4 == 0: whole float4 in or out
      f32x4 v = acc[bm][bn];
      if constexpr (EPI >= 2) v += *reinterpret_cast<const f32x4*>(bias + col);
      if constexpr (EPI == 3) { v[0] = gelu_exact(v[0]); v[1] = gelu_exact(v[1]); v[2] = gelu_exact(v[2]); v[3] = gelu_exact(v[3]); }
      if constexpr (EPI == 4) v += *reinterpret_cast<const f32x4*>(resid + (long)row * ldr + col);
      *reinterpret_cast<f32x4*>(C + (long)row * ldc + col) = v;
    }
  }
}

// A few rows (the cls rows of the last block: M = batch): weight streaming.  One wave per output column, lanes across K in 16-byte
// pieces, up to SK_ROWS rows of A accumulated per pass; fp32 FMAs, fixed-order wave reduction.
constexpr int SK_ROWS = 8;
template <int EPI, bool ALIGNED>
__global__ __launch_bounds__(256) void skinny_f32_nt_kernel(int M, int N, int K, const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                                            float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                            const float* __restrict__ resid, long ldr) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* w = B + (long)n * ldb;
  for (int r0 = 0; r0 < M; r0 += SK_ROWS) {
    float acc[SK_ROWS];
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) acc[r] = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
      f32x4 wv;
      if constexpr (ALIGNED) wv = *reinterpret_cast<const f32x4*>(w + k);
      else { for (int e = 0; e < 4; ++e) wv[e] = (k + e < K) ? w[k + e] : 0.f; }
#pragma unroll
      for (int r = 0; r < SK_ROWS; ++r) {
        if (r0 + r < M) {
          const float* a = A + (long)(r0 + r) * lda + k;
          f32x4 av;
          if constexpr (ALIGNED) av = *reinterpret_cast<const f32x4*>(a);
          else { for (int e = 0; e < 4; ++e) av[e] = (k + e < K) ? a[e] : 0.f; }
          acc[r] = __builtin_fmaf(av[0], wv[0], __builtin_fmaf(av[1], wv[1], __builtin_fmaf(av[2], wv[2], __builtin_fmaf(av[3], wv[3], acc[r]))));
        }
      }
    }
#pragma unroll
    for (int r = 0; r < SK_ROWS; ++r) {
      if (r0 + r < M) {
        float v = wave_sum(acc[r]);
        if (lane == 0) {
          if constexpr (EPI >= 2) v += bias[n];
          if constexpr (EPI == 3) v = gelu_exact(v);
          if constexpr (EPI == 4) v += resid[(long)(r0 + r) * ldr + n];
          C[(long)(r0 + r) * ldc + n] = v;
        }
      }
    }
  }
}

// Tile choice between the 64 x 64 (2 x 2) and the 128 x 128 (4 x 4) workgroup tile, by a residency model calibrated on MI355X
// (tools/gemm_f32_bench.py, profiles/r03_gemm_f32_shapes.log): a CU holds R workgroups at once (LDS: 4 of the small tile, 2 of the
// large one), i.e. R waves per SIMD, and the matrix pipe's utilisation depends on how many waves share it - a lone 2 x 2 wave keeps
// it 55 % busy, four keep it full; a lone 4 x 4 wave 48 %, two 100 % - so the time is the full rounds at residency R plus the tail
// round at its own residency, times the tile's blocks per wave and its cost per block at full residency (1.30 / 1.12 of the MFMA
// rate).  The 2 x 4 / 4 x 2 tiles never won a measured shape; they stay selectable for the tests.
struct TileChoice { int wm, wn; };
static double tile_cost(long wgs, int blocks_per_wave, int R, const double* f, double c) {
  const double q = (double)wgs / 256.0;                 // workgroups per CU
  const long full = (long)(q / R);
  const double rem = q - (double)full * R;
  int t = (int)(rem + 0.999999);
  if (t > R) t = R;
  double rounds = (double)full * R / f[R - 1];
  if (t > 0) rounds += (double)t / f[t - 1];
  return blocks_per_wave * c * rounds;
}
TileChoice pick_tile(int M, int N) {
  static const double f22[4] = {0.55, 0.75, 0.90, 1.00}, f44[2] = {0.48, 1.00};
  const long n22 = (long)((M + 63) / 64) * ((N + 63) / 64), n44 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const double c22 = tile_cost(n22, 4, 4, f22, 1.30), c44 = tile_cost(n44, 16, 2, f44, 1.12);
  return c44 < c22 ? TileChoice{4, 4} : TileChoice{2, 2};
}

int g_force_wm = 0, g_force_wn = 0;   // tuning aid (nv_gemm_f32_set_tile)

template <int WM, int WN, bool ALIGNED>
void launch_gemm_f32(int epi, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, const float* bias,
                     const float* resid, long ldr, hipStream_t s) {
  const int tiles_m = (M + 32 * WM - 1) / (32 * WM), tiles_n = (N + 32 * WN - 1) / (32 * WN);
  const dim3 grid(tiles_m * tiles_n), block(256);
  constexpr size_t lds = (size_t)2 * (32 * WM + 32 * WN) * F32_LD * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  if (lds > 64 * 1024) {
    static bool attr_set = false;              // per (WM, WN, ALIGNED) instantiation of this function: covers its four epilogues
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_nt_kernel<WM, WN, 0, ALIGNED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_nt_kernel<WM, WN, 2, ALIGNED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_nt_kernel<WM, WN, 3, ALIGNED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_nt_kernel<WM, WN, 4, ALIGNED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
  }
#define NV_F32_LAUNCH(E) hipLaunchKernelGGL((gemm_f32_nt_kernel<WM, WN, E, ALIGNED>), grid, block, lds, s, M, N, K, A, lda, B, ldb, C, ldc, bias, resid, ldr, tiles_n)
  switch (epi) {
    case 0: NV_F32_LAUNCH(0); break;
    case 2: NV_F32_LAUNCH(2); break;
    case 3: NV_F32_LAUNCH(3); break;
    default: NV_F32_LAUNCH(4); break;
  }
#undef NV_F32_LAUNCH
}

// ------------------------------------------------------------------------------------------------ attention
// One workgroup = 4 waves = 64 query rows of one (batch, head); keys in tiles of 64 staged through LDS (K row-major, V transposed),
// online softmax in base e.  Orientation: S^T = K Q^T (MFMA A = K rows, B = Q rows) leaves lane (j, q) with the scores of QUERY j
// against keys 4q .. 4q+3 of each 16-key block - exactly the B-operand layout of the second product O^T = V^T P^T, whose result
// gives lane (j, q) four consecutive head-dim columns of query j: row statistics are per lane, the output store is a float4.
// DHB = ceil(dim_head / 16); columns beyond dim_head are zero-filled.
constexpr int ATK = 64;                      // keys per tile
// AW waves (16 query rows each) per workgroup: 4 (64 rows); 2 is a tuning aid (more, smaller workgroups: measured slower).
template <int DHB, int AW>
__global__ __launch_bounds__(64 * AW) void attn_f32_fwd_kernel(const float* __restrict__ qkv, long ld, int n, int heads, int dh, float scale,
                                                           float* __restrict__ out, long ldo) {
  constexpr int DHP = 16 * DHB + 4;          // K tile row stride (floats)
  constexpr int KP = ATK + 4;                // V^T tile row stride
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sk = smem;                          // [ATK][DHP]
  float* svt = smem + ATK * DHP;             // [16 DHB][KP]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NT = 64 * AW, TQR = 16 * AW;                   // threads, query rows per workgroup
  const Grid2 gb = grid2d_xcd((n + TQR - 1) / TQR);            // 1-D launch: the row blocks of one head share an XCD (its K / V in one L2)
  const int bh = gb.by, b = bh / heads, h = bh - b * heads, inner = heads * dh;
  const float* base = qkv + (long)b * n * ld + (long)h * dh;
  const int q0 = gb.bx * TQR + wid * 16;
  const int j = lane & 15, g = lane >> 4;
  // Q fragments: lane (j, g) holds Q[q0 + j][16 s + 4 g .. + 3], s < DHB
  f32x4 qf[DHB];
  {
    const float* qrow = base + (long)min(q0 + j, n - 1) * ld;
#pragma unroll
    for (int s = 0; s < DHB; ++s) {
      const int c = 16 * s + 4 * g;
      qf[s] = (c < dh) ? *reinterpret_cast<const f32x4*>(qrow + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  f32x4 o[DHB];
#pragma unroll
  for (int s = 0; s < DHB; ++s) o[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_part = 0.f;     // l_part: this lane's share of the row sum (its own key columns)
  const int c4n = dh >> 2;                   // float4 chunks per row that exist
  // staging of one 64-key tile: thread t handles elements t + NT u (u < 256 DHB / NT) of K (key e / (4 DHB), chunk e % (4 DHB): whole rows
  // per eight lanes) and of V (chunk e / 64, key e % 64: consecutive lanes = consecutive keys -> conflict-free transposed writes).
  // The loads of tile k + 1 are in flight under the MFMAs of tile k (one register set).
  constexpr int NU = 256 * DHB / NT;
  f32x4 kreg[NU], vreg[NU];
  auto gload = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = tid + NT * u;
      {
        const int kk = e / (4 * DHB), c = e - kk * (4 * DHB);
        const bool ok = (k0 + kk < n) && (c < c4n);
        kreg[u] = ok ? *reinterpret_cast<const f32x4*>(base + (long)(k0 + kk) * ld + 4 * c + inner) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      {
        const int c = e / ATK, kk = e - c * ATK;
        const bool ok = (k0 + kk < n) && (c < c4n);
        vreg[u] = ok ? *reinterpret_cast<const f32x4*>(base + (long)(k0 + kk) * ld + 4 * c + 2 * inner) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto swrite = [&]() {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = tid + NT * u;
      {
        const int kk = e / (4 * DHB), c = e - kk * (4 * DHB);
        *reinterpret_cast<f32x4*>(sk + kk * DHP + 4 * c) = kreg[u];
      }
      {
        const int c = e / ATK, kk = e - c * ATK;
#pragma unroll
        for (int x = 0; x < 4; ++x) svt[(4 * c + x) * KP + kk] = vreg[u][x];
      }
    }
  };
  gload(0);
  for (int k0 = 0; k0 < n; k0 += ATK) {
    __syncthreads();                         // previous tile consumed
    swrite();
    __syncthreads();
    if (k0 + ATK < n) gload(k0 + ATK);
    // S^T blocks: st[kb][r] = score of query j against key k0 + 16 kb + 4 g + r
    f32x4 st[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < DHB; ++s) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(sk + (16 * kb + j) * DHP + 16 * s + 4 * g);   // lane (i = key, g)
#pragma unroll
        for (int t = 0; t < 4; ++t) a = mfma4(kf[t], qf[s][t], a);
      }
      st[kb] = a;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * kb + 4 * g + r;
        const float v = (key < n) ? st[kb][r] * scale : -INFINITY;
        st[kb][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);    // finite: every tile holds at least one valid key
    // exp(x) as v_exp_f32(x log2 e): the hardware exp2 is good to about one ulp, the product rounds the exponent to 2^-24 |x|
    // (|x| <= ~100): 1e-6 relative on a probability at worst, against ~20 VALU instructions per libm expf beside a matrix pipe that
    // waits for them
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * 1.44269504088896340736f);   // first tile: exp2(-inf) = 0
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f((st[kb][r] - m_new) * 1.44269504088896340736f);
        st[kb][r] = p;
        ps += p;
      }
    l_part = l_part * alpha + ps;
#pragma unroll
    for (int s = 0; s < DHB; ++s) o[s] *= alpha;
    // O^T += V^T P^T: MFMA A = V^T rows (lane (i = head-dim column, g): keys 16 kb + 4 g .. + 3), B = P (lane (j, g): the same keys)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int s = 0; s < DHB; ++s) {
        const f32x4 vf = *reinterpret_cast<const f32x4*>(svt + (16 * s + j) * KP + 16 * kb + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) o[s] = mfma4(vf[t], st[kb][t], o[s]);
      }
  }
  float l = l_part + __shfl_xor(l_part, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  const int row = q0 + j;
  if (row < n) {
    float* orow = out + ((long)b * n + row) * ldo + (long)h * dh;
#pragma unroll
    for (int s = 0; s < DHB; ++s) {
      const int c = 16 * s + 4 * g;
      if (c < dh) *reinterpret_cast<f32x4*>(orow + c) = o[s] * inv;
    }
  }
}

int g_attn_f32_waves = 0;     // tuning aid: 0 = heuristic, 2 / 4 = forced
template <int DHB>
void launch_attn_f32(const float* qkv, long ld, int B, int n, int heads, int dh, float scale, float* out, long ldo, hipStream_t s) {
  const size_t lds = (size_t)(ATK * (16 * DHB + 4) + 16 * DHB * (ATK + 4)) * sizeof(float);
  const long wg4 = (long)((n + 63) / 64) * B * heads;
  const int aw = g_attn_f32_waves ? g_attn_f32_waves : 4;      // (two-wave workgroups measured slower at every size: 70 vs 63 us at batch 4)
  if (aw == 4) hipLaunchKernelGGL((attn_f32_fwd_kernel<DHB, 4>), dim3((unsigned)wg4), dim3(256), lds, s, qkv, ld, n, heads, dh, scale, out, ldo);
  else hipLaunchKernelGGL((attn_f32_fwd_kernel<DHB, 2>), dim3(((n + 31) / 32) * B * heads), dim3(128), lds, s, qkv, ld, n, heads, dh, scale, out, ldo);
}

}  // namespace

extern "C" int nv_gemm_f32_set_tile(int wm, int wn) {
  if (wm == -1) { g_attn_f32_waves = (wn == 2 || wn == 4) ? wn : 0; return NV_OK; }      // (-1, 2 | 4 | 0): waves per workgroup of the fp32 attention
  NV_CHECK_ARG((wm == 0 && wn == 0) || ((wm == 2 || wm == 4) && (wn == 2 || wn == 4)), "nv_gemm_f32_set_tile: (0, 0) or wm, wn in {2, 4}");
  g_force_wm = wm; g_force_wn = wn;
  return NV_OK;
}

extern "C" int nv_gemm_f32(int epi, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                           const float* bias, const float* resid, long ldr, void* stream) {
  NV_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "nv_gemm_f32: bad shape / null pointer");
  NV_CHECK_ARG(epi == 0 || epi == 2 || epi == 3 || epi == 4, "nv_gemm_f32: epilogue %d (0 store, 2 bias, 3 bias + GELU, 4 bias + residual)", epi);
  NV_CHECK_ARG((N % 4) == 0 && (ldc % 4) == 0 && nv_aligned16(C) && lda >= K && ldb >= K && ldc >= N, "nv_gemm_f32: N, ldc must be multiples of 4, C 16-byte aligned");
  NV_CHECK_ARG(epi < 2 || (bias && nv_aligned16(bias)), "nv_gemm_f32: epilogue %d needs a 16-byte aligned bias", epi);
  NV_CHECK_ARG(epi != 4 || (resid && nv_aligned16(resid) && (ldr % 4) == 0 && ldr >= N), "nv_gemm_f32: epilogue 4 needs an aligned residual");
  const bool aligned = (K % 4) == 0 && (lda % 4) == 0 && (ldb % 4) == 0 && nv_aligned16(A) && nv_aligned16(B);
  TileChoice t = pick_tile(M, N);
  if (g_force_wm) t = TileChoice{g_force_wm, g_force_wn};
  hipStream_t s = (hipStream_t)stream;
  const int slot = nv_prof_begin(30, 2.0 * M * N * (double)K, stream);
  if (M <= 2 * SK_ROWS && !g_force_wm) {          // a few rows: stream the weight once, one wave per output column
    const dim3 grid((N + 3) / 4), block(256);
#define NV_F32_SKINNY(E)                                                                                                              \
    do {                                                                                                                              \
      if (aligned) hipLaunchKernelGGL((skinny_f32_nt_kernel<E, true>), grid, block, 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, resid, ldr);   \
      else hipLaunchKernelGGL((skinny_f32_nt_kernel<E, false>), grid, block, 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, resid, ldr);          \
    } while (0)
    switch (epi) {
      case 0: NV_F32_SKINNY(0); break;
      case 2: NV_F32_SKINNY(2); break;
      case 3: NV_F32_SKINNY(3); break;
      default: NV_F32_SKINNY(4); break;
    }
#undef NV_F32_SKINNY
    nv_prof_end(slot, stream);
    NV_CHECK_LAUNCH("nv_gemm_f32/skinny");
    return NV_OK;
  }
  nv_prof_bytes(slot, 4.0 * ((double)M * K + (double)N * K + (double)M * N * (epi == 4 ? 2 : 1)));
#define NV_F32_TILE(WM, WN)                                                                             \
  do {                                                                                                  \
    if (aligned) launch_gemm_f32<WM, WN, true>(epi, M, N, K, A, lda, B, ldb, C, ldc, bias, resid, ldr, s);    \
    else launch_gemm_f32<WM, WN, false>(epi, M, N, K, A, lda, B, ldb, C, ldc, bias, resid, ldr, s);           \
  } while (0)
  if (t.wm == 4 && t.wn == 4) NV_F32_TILE(4, 4);
  else if (t.wm == 2 && t.wn == 4) NV_F32_TILE(2, 4);
  else if (t.wm == 4 && t.wn == 2) NV_F32_TILE(4, 2);
  else NV_F32_TILE(2, 2);
#undef NV_F32_TILE
  nv_prof_end(slot, stream);
  NV_CHECK_LAUNCH("nv_gemm_f32");
  return NV_OK;
}

extern "C" int nv_attn_fwd_f32(const float* qkv, long ld_qkv, int B, int n, int heads, int dim_head, float scale, float* out, long ld_out,
                               void* stream) {
  NV_CHECK_ARG(qkv && out && B > 0 && n > 0 && heads > 0, "nv_attn_fwd_f32: bad shape / null pointer");
  NV_CHECK_ARG(dim_head >= 4 && dim_head <= 128 && (dim_head % 4) == 0, "nv_attn_fwd_f32: dim_head=%d must be a multiple of 4 up to 128", dim_head);
  NV_CHECK_ARG((ld_qkv % 4) == 0 && (ld_out % 4) == 0 && nv_aligned16(qkv) && nv_aligned16(out) && ld_qkv >= 3L * heads * dim_head && ld_out >= (long)heads * dim_head,
               "nv_attn_fwd_f32: leading dimensions must be multiples of 4 and cover the heads, buffers 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int slot = nv_prof_begin(31, 4.0 * B * heads * (double)n * n * dim_head, stream);
  switch ((dim_head + 15) / 16) {
    case 1: launch_attn_f32<1>(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, s); break;
    case 2: launch_attn_f32<2>(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, s); break;
    case 3: launch_attn_f32<3>(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, s); break;
    case 4: launch_attn_f32<4>(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, s); break;
    case 5: launch_attn_f32<5>(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, s); break;
    case 6: launch_attn_f32<6>(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, s); break;
    case 7: launch_attn_f32<7>(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, s); break;
    default: launch_attn_f32<8>(qkv, ld_qkv, B, n, heads, dim_head, scale, out, ld_out, s); break;
  }
  nv_prof_end(slot, stream);
  NV_CHECK_LAUNCH("nv_attn_fwd_f32");
  return NV_OK;
}
