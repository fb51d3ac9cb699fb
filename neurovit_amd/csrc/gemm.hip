// bf16 MFMA GEMM for gfx950 with fused epilogues - the workhorse of the ViT3D hot path.
//
//   C[M,N] = op(A) . op(B)   (fp32 accumulate, v_mfma_f32_16x16x32_bf16)
//     layout NT: A [M,K] row-major, B [N,K] row-major   (y = x W^T      : forward linears, vit_3d.py:19,22,41,44)
//     layout NN: A [M,K] row-major, B [K,N] row-major   (dx = dy W      : data gradients)
//     layout TN: A [K,M] row-major, B [K,N] row-major   (dW = dy^T x    : weight gradients)
//
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.  Operands are staged
// global -> registers -> LDS (issue early / write late, one barrier per K tile, two LDS buffers).
// K-contiguous operands use the IMG128 image + ds_read_b128; K-strided ("T") operands keep their
// memory layout in an IMG256 image and are transposed for free by ds_read_b64_tr_b16.
// The MFMA is issued with swapped operands (D = B_frag x A_frag) so every lane ends up holding four
// CONSECUTIVE output columns of one row: epilogue loads/stores are 8-16 bytes per lane.
// Ragged M/N/K are handled by clamped loads + zero select and predicated stores.
#include "common.h"

enum {
  EPI_STORE_BF16 = 0,   // C(bf16) = acc
  EPI_STORE_F32 = 1,    // C(f32)  = acc (+ C if accumulate)
  EPI_BIAS_F32 = 2,     // C(f32)  = acc + bias[n]
  EPI_BIAS_GELU = 3,    // aux_out(bf16) = u = acc + bias[n];  C(bf16) = gelu(u)
  EPI_BIAS_RESID = 4,   // C(f32)  = aux_in(f32)[m,n] + acc + bias[n]
  EPI_DGELU = 5,        // C(bf16) = acc * gelu'(aux_in(bf16)[m,n])
  EPI_SCALE_BF16 = 6,   // C(bf16) = alpha * acc
};

struct GemmArgs {
  const bf16* A;
  const bf16* B;
  void* C;
  const float* bias;
  const void* aux_in;
  void* aux_out;
  long lda, ldb, ldc, ld_aux_in, ld_aux_out;
  int M, N, K;
  int accumulate;
  float alpha;
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;   // 16 KiB per operand tile (both images)

// ---- staging: global -> registers ---------------------------------------------------------------
// rows x K operand (K contiguous): 128 rows x 8 chunks of 8 bf16.
__device__ __forceinline__ void gload_rowmajor(const bf16* X, long ld, int R, int K, int r0, int k0, int tid, uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    const int row = r0 + (c >> 3), kk = k0 + ((c & 7) << 3);
    const bool ok = (row < R) && (kk < K);
    const uint4 v = *reinterpret_cast<const uint4*>(X + (ok ? (long)row * ld + kk : 0));
    reg[i] = ok ? v : make_uint4(0, 0, 0, 0);
  }
}
// K x cols operand (cols contiguous): 64 k-rows x 16 chunks.
__device__ __forceinline__ void gload_kmajor(const bf16* X, long ld, int Ccols, int K, int c0, int k0, int tid, uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    const int kk = k0 + (c >> 4), col = c0 + ((c & 15) << 3);
    const bool ok = (kk < K) && (col < Ccols);
    const uint4 v = *reinterpret_cast<const uint4*>(X + (ok ? (long)kk * ld + col : 0));
    reg[i] = ok ? v : make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void swrite_rowmajor(char* img, int tid, const uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    *reinterpret_cast<uint4*>(img + img128_off(c >> 3, c & 7)) = reg[i];
  }
}
__device__ __forceinline__ void swrite_kmajor(char* img, int tid, const uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    *reinterpret_cast<uint4*>(img + img256_off(c >> 4, c & 15)) = reg[i];
  }
}

// ---- fragment reads -----------------------------------------------------------------------------
// Operand fragment of v_mfma_f32_16x16x32_bf16: lane (r = lane&15, g = lane>>4) holds the 8 values
// k = 8g .. 8g+7 of row/column r.
template <bool T>
__device__ __forceinline__ bf16x8 read_frag(const char* img, int rc0, int ks, int lane) {
  const int r = lane & 15, g = lane >> 4;
  if constexpr (!T) {
    return *reinterpret_cast<const bf16x8*>(img + img128_off(rc0 + r, 4 * ks + g));
  } else {
    const int q = r >> 2, p = r & 3;
    const int k0 = 32 * ks + 8 * g + q;
    const int ch = (rc0 >> 3) + (p >> 1), sub = (p & 1) << 3;
    const bf16x4 lo = lds_read_tr(img + img256_off(k0, ch) + sub);
    const bf16x4 hi = lds_read_tr(img + img256_off(k0 + 4, ch) + sub);
    return cat4(lo, hi);
  }
}

template <bool A_T, bool B_T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int nk = (g.K + BK - 1) / BK;

  char* sA0 = smem;
  char* sB0 = smem + TILE_BYTES;
  char* sA1 = smem + 2 * TILE_BYTES;
  char* sB1 = smem + 3 * TILE_BYTES;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    if constexpr (A_T) gload_kmajor(g.A, g.lda, g.M, g.K, m0, k0, tid, ra);
    else gload_rowmajor(g.A, g.lda, g.M, g.K, m0, k0, tid, ra);
    if constexpr (B_T) gload_kmajor(g.B, g.ldb, g.N, g.K, n0, k0, tid, rb);
    else gload_rowmajor(g.B, g.ldb, g.N, g.K, n0, k0, tid, rb);
  };
  auto swrite = [&](char* a, char* b) {
    if constexpr (A_T) swrite_kmajor(a, tid, ra); else swrite_rowmajor(a, tid, ra);
    if constexpr (B_T) swrite_kmajor(b, tid, rb); else swrite_rowmajor(b, tid, rb);
  };

  gload(0);
  swrite(sA0, sB0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const char* a = (kt & 1) ? sA1 : sA0;
    const char* b = (kt & 1) ? sB1 : sB0;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag<A_T>(a, wm * 64 + 16 * i, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag<B_T>(b, wn * 64 + 16 * j, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      if (kt & 1) swrite(sA0, sB0); else swrite(sA1, sB1);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3] ---------------------
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + 16 * i + lr;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + 16 * j + 4 * lg;
      if (n >= g.N) continue;
      f32x4 v = acc[i][j];
      if constexpr (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n);
        v += bv;
      }
      if constexpr (EPI == EPI_STORE_BF16) {
        *reinterpret_cast<bf16x4*>((bf16*)g.C + (long)m * g.ldc + n) = cvt4(v[0], v[1], v[2], v[3]);
      } else if constexpr (EPI == EPI_SCALE_BF16) {
        v *= g.alpha;
        *reinterpret_cast<bf16x4*>((bf16*)g.C + (long)m * g.ldc + n) = cvt4(v[0], v[1], v[2], v[3]);
      } else if constexpr (EPI == EPI_STORE_F32) {
        float* c = (float*)g.C + (long)m * g.ldc + n;
        if (g.accumulate) v += *reinterpret_cast<const f32x4*>(c);
        *reinterpret_cast<f32x4*>(c) = v;
      } else if constexpr (EPI == EPI_BIAS_F32) {
        *reinterpret_cast<f32x4*>((float*)g.C + (long)m * g.ldc + n) = v;
      } else if constexpr (EPI == EPI_BIAS_GELU) {
        *reinterpret_cast<bf16x4*>((bf16*)g.aux_out + (long)m * g.ld_aux_out + n) = cvt4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<bf16x4*>((bf16*)g.C + (long)m * g.ldc + n) =
            cvt4(gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3]));
      } else if constexpr (EPI == EPI_BIAS_RESID) {
        v += *reinterpret_cast<const f32x4*>((const float*)g.aux_in + (long)m * g.ld_aux_in + n);
        *reinterpret_cast<f32x4*>((float*)g.C + (long)m * g.ldc + n) = v;
      } else if constexpr (EPI == EPI_DGELU) {
        const bf16x4 u = *reinterpret_cast<const bf16x4*>((const bf16*)g.aux_in + (long)m * g.ld_aux_in + n);
        *reinterpret_cast<bf16x4*>((bf16*)g.C + (long)m * g.ldc + n) =
            cvt4(v[0] * gelu_grad_f((float)u[0]), v[1] * gelu_grad_f((float)u[1]),
                 v[2] * gelu_grad_f((float)u[2]), v[3] * gelu_grad_f((float)u[3]));
      }
    }
  }
}

template <bool A_T, bool B_T, int EPI>
static int launch(const GemmArgs& a, hipStream_t s) {
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  auto kern = gemm_bf16_kernel<A_T, B_T, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    attr_set = true;
  }
  const int slot = nv_prof_begin((A_T ? 2 : (B_T ? 1 : 0)), 2.0 * a.M * a.N * a.K, s);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 4 * TILE_BYTES, s, a);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm_bf16");
  return NV_OK;
}

extern "C" int nv_gemm_bf16(int layout, int epi, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                            void* C, long ldc, const float* bias, const void* aux_in, long ld_aux_in, void* aux_out,
                            long ld_aux_out, int accumulate, float alpha, void* stream) {
  NV_CHECK_ARG(M > 0 && N > 0 && K > 0, "nv_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
  NV_CHECK_ARG(A && B && C, "nv_gemm_bf16: null operand");
  NV_CHECK_ARG(nv_aligned16(A) && nv_aligned16(B) && nv_aligned16(C), "nv_gemm_bf16: operands must be 16-byte aligned");
  NV_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 4) == 0, "nv_gemm_bf16: lda/ldb must be multiples of 8, ldc of 4");
  NV_CHECK_ARG((N % 8) == 0, "nv_gemm_bf16: N=%d must be a multiple of 8", N);
  if (layout == 0) NV_CHECK_ARG((K % 8) == 0 && lda >= K && ldb >= K, "nv_gemm_bf16[NT]: K=%d must be a multiple of 8 and <= lda, ldb", K);
  if (layout == 1) NV_CHECK_ARG((K % 8) == 0 && lda >= K && ldb >= N, "nv_gemm_bf16[NN]: bad K/lda/ldb");
  if (layout == 2) NV_CHECK_ARG((M % 8) == 0 && lda >= M && ldb >= N, "nv_gemm_bf16[TN]: M=%d must be a multiple of 8; lda>=M, ldb>=N", M);
  NV_CHECK_ARG(ldc >= N, "nv_gemm_bf16: ldc < N");
  GemmArgs a;
  a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = C; a.bias = bias; a.aux_in = aux_in; a.aux_out = aux_out;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ld_aux_in = ld_aux_in; a.ld_aux_out = ld_aux_out;
  a.M = M; a.N = N; a.K = K; a.accumulate = accumulate; a.alpha = alpha;
  hipStream_t s = (hipStream_t)stream;
  const bool need_bias = (epi == EPI_BIAS_F32 || epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID);
  NV_CHECK_ARG(!need_bias || (bias && nv_aligned16(bias)), "nv_gemm_bf16: epilogue %d needs a 16-byte aligned bias", epi);
  NV_CHECK_ARG(!(epi == EPI_BIAS_RESID || epi == EPI_DGELU) || (aux_in && nv_aligned16(aux_in) && (ld_aux_in % 4) == 0),
               "nv_gemm_bf16: epilogue %d needs aux_in", epi);
  NV_CHECK_ARG(epi != EPI_BIAS_GELU || (aux_out && nv_aligned16(aux_out) && (ld_aux_out % 4) == 0),
               "nv_gemm_bf16: EPI_BIAS_GELU needs aux_out");
  switch (layout * 16 + epi) {
    case 0 * 16 + EPI_STORE_BF16: return launch<false, false, EPI_STORE_BF16>(a, s);
    case 0 * 16 + EPI_STORE_F32: return launch<false, false, EPI_STORE_F32>(a, s);
    case 0 * 16 + EPI_BIAS_F32: return launch<false, false, EPI_BIAS_F32>(a, s);
    case 0 * 16 + EPI_BIAS_GELU: return launch<false, false, EPI_BIAS_GELU>(a, s);
    case 0 * 16 + EPI_BIAS_RESID: return launch<false, false, EPI_BIAS_RESID>(a, s);
    case 1 * 16 + EPI_STORE_BF16: return launch<false, true, EPI_STORE_BF16>(a, s);
    case 1 * 16 + EPI_STORE_F32: return launch<false, true, EPI_STORE_F32>(a, s);
    case 1 * 16 + EPI_DGELU: return launch<false, true, EPI_DGELU>(a, s);
    case 2 * 16 + EPI_STORE_F32: return launch<true, true, EPI_STORE_F32>(a, s);
    default: break;
  }
  nv_set_error("nv_gemm_bf16: unsupported layout/epilogue combination (%d, %d)", layout, epi);
  return NV_ERR_ARG;
}
