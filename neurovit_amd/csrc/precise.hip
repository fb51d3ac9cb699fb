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
// Wave tile (16 WM) x (16 WN), workgroup = 2 x 2 waves.  EPI: 0 store, 2 + bias, 3 gelu(+ bias), 4 resid + (+ bias).
// ALIGNED: lda, ldb multiples of 4, 16-byte aligned bases, K % 4 == 0 (float4 operand loads); otherwise scalar loads with
// per-element bounds (the reference's default patch_dim 729 = 9^3, configs/config.yaml:39-40).
template <int WM, int WN, int EPI, bool ALIGNED>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(int M, int N, int K, const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                                          float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                          const float* __restrict__ resid, long ldr, int tiles_n) {
  constexpr int BM = 32 * WM, BN = 32 * WN;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM + (wid >> 1) * 16 * WM, n0 = tn * BN + (wid & 1) * 16 * WN;
  if (m0 >= M || n0 >= N) return;            // whole wave outside (no barriers in this kernel)
  const int i = lane & 15, g = lane >> 4;
  const float* ap[WM];
  const float* bp[WN];
#pragma unroll
  for (int bm = 0; bm < WM; ++bm) { const int r = min(m0 + 16 * bm + i, M - 1); ap[bm] = A + (long)r * lda + 4 * g; }
#pragma unroll
  for (int bn = 0; bn < WN; ++bn) { const int r = min(n0 + 16 * bn + i, N - 1); bp[bn] = B + (long)r * ldb + 4 * g; }
  f32x4 acc[WM][WN];
#pragma unroll
  for (int bm = 0; bm < WM; ++bm)
#pragma unroll
    for (int bn = 0; bn < WN; ++bn) acc[bm][bn] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 af[2][WM], bf[2][WN];
  auto ld4 = [&](const float* p, int k0) -> f32x4 {     // elements k0 + 4g .. + 3 of the lane's row, zero beyond K
    const int k = k0 + 4 * g;
    if constexpr (ALIGNED) return (k < K) ? *reinterpret_cast<const f32x4*>(p + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k + e < K) ? p[k0 + e] : 0.f;
    return v;
  };
  auto load = [&](int buf, int k0) {
#pragma unroll
    for (int bm = 0; bm < WM; ++bm) af[buf][bm] = ld4(ap[bm], k0);
#pragma unroll
    for (int bn = 0; bn < WN; ++bn) bf[buf][bn] = ld4(bp[bn], k0);
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int bm = 0; bm < WM; ++bm)
#pragma unroll
        for (int bn = 0; bn < WN; ++bn) acc[bm][bn] = mfma4(bf[buf][bn][t], af[buf][bm][t], acc[bm][bn]);
  };
  const int nk = (K + 15) >> 4;
  load(0, 0);
  for (int s = 0; s < nk; s += 2) {
    load(1, (s + 1) * 16);                   // beyond K: zeros, no memory access
    compute(0);
    if (s + 1 < nk) {
      load(0, (s + 2) * 16);
      compute(1);
    }
  }
  // lane (j, q), register r of block (bm, bn) holds C[m0 + 16 bm + j][n0 + 16 bn + 4 q + r]
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int bm = 0; bm < WM; ++bm) {
    const int row = m0 + 16 * bm + j;
    if (row >= M) continue;
#pragma unroll
    for (int bn = 0; bn < WN; ++bn) {
      const int col = n0 + 16 * bn + 4 * q;
      if (col >= N) continue;                // N % 4 == 0: whole float4 in or out
      f32x4 v = acc[bm][bn];
      if constexpr (EPI >= 2) v += *reinterpret_cast<const f32x4*>(bias + col);
      if constexpr (EPI == 3) { v[0] = gelu_exact(v[0]); v[1] = gelu_exact(v[1]); v[2] = gelu_exact(v[2]); v[3] = gelu_exact(v[3]); }
      if constexpr (EPI == 4) v += *reinterpret_cast<const f32x4*>(resid + (long)row * ldr + col);
      *reinterpret_cast<f32x4*>(C + (long)row * ldc + col) = v;
    }
  }
}

// tile choice: the machine has 1024 SIMDs and every wave tile is an indivisible unit of matrix-pipe time, so what counts is the
// fill of the last round; larger wave tiles load fewer operand bytes per FLOP and win ties.
struct TileChoice { int wm, wn; };
TileChoice pick_tile(int M, int N) {
  static const int cand[4][2] = {{4, 4}, {2, 4}, {4, 2}, {2, 2}};
  double best = -1.0;
  TileChoice c{2, 2};
  for (int t = 0; t < 4; ++t) {
    const int wm = cand[t][0], wn = cand[t][1];
    const long waves = 4L * ((M + 32 * wm - 1) / (32 * wm)) * ((N + 32 * wn - 1) / (32 * wn));
    const long useful = (long)((M + 15) / 16) * ((N + 15) / 16);               // 16 x 16 blocks that hold output
    const long rounds = (waves + 1023) / 1024;
    const double eff = (double)useful / ((double)rounds * 1024.0 * wm * wn);   // useful blocks per block slot of the rounds taken
    const double score = eff * (1.0 + 0.02 * (wm * wn) / 16.0);                // ties go to the larger tile
    if (score > best) { best = score; c = TileChoice{wm, wn}; }
  }
  return c;
}

int g_force_wm = 0, g_force_wn = 0;   // tuning aid (nv_gemm_f32_set_tile)

template <int WM, int WN, bool ALIGNED>
void launch_gemm_f32(int epi, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, const float* bias,
                     const float* resid, long ldr, hipStream_t s) {
  const int tiles_m = (M + 32 * WM - 1) / (32 * WM), tiles_n = (N + 32 * WN - 1) / (32 * WN);
  const dim3 grid(tiles_m * tiles_n), block(256);
#define NV_F32_LAUNCH(E) hipLaunchKernelGGL((gemm_f32_nt_kernel<WM, WN, E, ALIGNED>), grid, block, 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, resid, ldr, tiles_n)
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
template <int DHB>
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(const float* __restrict__ qkv, long ld, int n, int heads, int dh, float scale,
                                                           float* __restrict__ out, long ldo) {
  constexpr int DHP = 16 * DHB + 4;          // K tile row stride (floats)
  constexpr int KP = ATK + 4;                // V^T tile row stride
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sk = smem;                          // [ATK][DHP]
  float* svt = smem + ATK * DHP;             // [16 DHB][KP]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads, inner = heads * dh;
  const float* base = qkv + (long)b * n * ld + (long)h * dh;
  const int q0 = blockIdx.x * 64 + wid * 16;
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
  for (int k0 = 0; k0 < n; k0 += ATK) {
    __syncthreads();                         // previous tile consumed
    // stage K (row-major) and V (transposed): chunk c of key kk
    for (int e = tid; e < ATK * 4 * DHB; e += 256) {
      const int kk = e / (4 * DHB), c = e - kk * (4 * DHB);
      const bool ok = (k0 + kk < n) && (c < c4n);
      const float* src = base + (long)min(k0 + kk, n - 1) * ld + 4 * c;
      const f32x4 kv = ok ? *reinterpret_cast<const f32x4*>(src + inner) : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(sk + kk * DHP + 4 * c) = kv;
    }
    for (int e = tid; e < ATK * 4 * DHB; e += 256) {
      const int c = e / ATK, kk = e - c * ATK;          // consecutive lanes: consecutive keys -> conflict-free transposed writes
      const bool ok = (k0 + kk < n) && (c < c4n);
      const float* src = base + (long)min(k0 + kk, n - 1) * ld + 4 * c;
      const f32x4 vv = ok ? *reinterpret_cast<const f32x4*>(src + 2 * inner) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int x = 0; x < 4; ++x) svt[(4 * c + x) * KP + kk] = vv[x];
    }
    __syncthreads();
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
    const float alpha = expf(m_run - m_new); // first tile: exp(-inf) = 0
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = expf(st[kb][r] - m_new);
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

template <int DHB>
void launch_attn_f32(const float* qkv, long ld, int B, int n, int heads, int dh, float scale, float* out, long ldo, hipStream_t s) {
  const size_t lds = (size_t)(ATK * (16 * DHB + 4) + 16 * DHB * (ATK + 4)) * sizeof(float);
  hipLaunchKernelGGL(attn_f32_fwd_kernel<DHB>, dim3((n + 63) / 64, B * heads), dim3(256), lds, s, qkv, ld, n, heads, dh, scale, out, ldo);
}

}  // namespace

extern "C" int nv_gemm_f32_set_tile(int wm, int wn) {
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
