// bf16 MFMA GEMM for gfx950 with fused epilogues - the workhorse of the ViT3D hot path.
//
//   C[M,N] = op(A) . op(B)   (fp32 accumulate, v_mfma_f32_16x16x32_bf16)
//     layout NT: A [M,K] row-major, B [N,K] row-major   (y = x W^T      : forward linears, vit_3d.py:19,22,41,44)
//     layout NN: A [M,K] row-major, B [K,N] row-major   (dx = dy W      : data gradients)
//     layout TN: A [K,M] row-major, B [K,N] row-major   (dW = dy^T x    : weight gradients)
//
// Workgroup tile BM x BN x 64 with 2x2 waves; BM, BN in {64, 128} chosen per problem so that the grid fills the
// 256 CUs several times over (the ViT3D-base shapes have only M = 2052 rows: 128x128 tiles would leave most CUs
// with a single wave per SIMD and nothing to hide latency behind).  Operands are staged by LDS-DMA
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass; the swizzle is applied on the per-lane SOURCE
// address because the LDS destination of one wave-instruction is linear), two LDS buffers, one barrier per
// K tile; only a ragged K tail goes through registers.  K-contiguous operands use a
// [rows][64] image + ds_read_b128; K-strided ("T") operands keep their memory layout in a [64][cols] image and
// are transposed for free by ds_read_b64_tr_b16.  All images are XOR-swizzled: zero LDS bank conflicts measured
// (SQ_LDS_BANK_CONFLICT = 0).  The MFMA is issued with swapped operands (D = B_frag x A_frag) so every lane ends
// up holding four CONSECUTIVE output columns of one row: epilogue loads/stores are 8-16 bytes per lane.
// Ragged M/N need no masking on the load side (out-of-range rows/columns are clamped to valid memory and their
// results never stored); only a ragged K tail is zero-filled, in a peeled last iteration.
#include "common.h"

enum {
  EPI_STORE_BF16 = 0,   // C(bf16) = acc
  EPI_STORE_F32 = 1,    // C(f32)  = acc (+ C if accumulate)
  EPI_BIAS_F32 = 2,     // C(f32)  = acc + bias[n]
  EPI_BIAS_GELU = 3,    // aux_out(bf16) = u = acc + bias[n];  C(bf16) = gelu(u)
  EPI_BIAS_RESID = 4,   // C(f32)  = aux_in(f32)[m,n] + acc + bias[n]
  EPI_DGELU = 5,        // C(bf16) = acc * gelu'(aux_in(bf16)[m,n])
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

constexpr int BK = 64;
constexpr int NTHREADS = 256;

// [64 k-rows][64 cols] transposed-read image with 128-byte rows: chunk XOR so that the four same-parity rows a
// 32-lane half touches in one ds_read_b64_tr_b16 ({0,2,8,10} + multiples) land on four different chunk pairs.
__device__ __forceinline__ int img128t_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1)) << 4);
}

template <int BX>
__device__ __forceinline__ int imgT_off(int krow, int chunk) {
  if constexpr (BX == 128) return img256_off(krow, chunk);
  else return img128t_off(krow, chunk);
}

// ---- staging: global -> registers ---------------------------------------------------------------
// rows x K operand (K contiguous): BX rows x 8 chunks of 8 bf16.  TAIL: zero-fill k >= K.
template <int BX, bool TAIL>
__device__ __forceinline__ void gload_rowmajor(const bf16* X, long ld, int R, int K, int r0, int k0, int tid, uint4 (&reg)[BX / 32]) {
#pragma unroll
  for (int i = 0; i < BX / 32; ++i) {
    const int c = tid + NTHREADS * i;
    const int row = min(r0 + (c >> 3), R - 1), kk = k0 + ((c & 7) << 3);
    if constexpr (TAIL) {
      const bool ok = kk < K;
      const uint4 v = *reinterpret_cast<const uint4*>(X + (long)row * ld + (ok ? kk : 0));
      reg[i] = ok ? v : make_uint4(0, 0, 0, 0);
    } else {
      reg[i] = *reinterpret_cast<const uint4*>(X + (long)row * ld + kk);
    }
  }
}
// K x cols operand (cols contiguous): 64 k-rows x BX/8 chunks.  TAIL: zero-fill rows k >= K.
template <int BX, bool TAIL>
__device__ __forceinline__ void gload_kmajor(const bf16* X, long ld, int Ccols, int K, int c0, int k0, int tid, uint4 (&reg)[BX / 32]) {
  constexpr int CPR = BX / 8;   // chunks per k-row
#pragma unroll
  for (int i = 0; i < BX / 32; ++i) {
    const int c = tid + NTHREADS * i;
    const int kk = k0 + c / CPR, col = min(c0 + ((c % CPR) << 3), Ccols - 8);
    if constexpr (TAIL) {
      const bool ok = kk < K;
      const uint4 v = *reinterpret_cast<const uint4*>(X + (long)(ok ? kk : 0) * ld + col);
      reg[i] = ok ? v : make_uint4(0, 0, 0, 0);
    } else {
      reg[i] = *reinterpret_cast<const uint4*>(X + (long)kk * ld + col);
    }
  }
}
template <int BX>
__device__ __forceinline__ void swrite_rowmajor(char* img, int tid, const uint4 (&reg)[BX / 32]) {
#pragma unroll
  for (int i = 0; i < BX / 32; ++i) {
    const int c = tid + NTHREADS * i;
    *reinterpret_cast<uint4*>(img + img128_off(c >> 3, c & 7)) = reg[i];
  }
}
template <int BX>
__device__ __forceinline__ void swrite_kmajor(char* img, int tid, const uint4 (&reg)[BX / 32]) {
  constexpr int CPR = BX / 8;
#pragma unroll
  for (int i = 0; i < BX / 32; ++i) {
    const int c = tid + NTHREADS * i;
    *reinterpret_cast<uint4*>(img + imgT_off<BX>(c / CPR, c % CPR)) = reg[i];
  }
}

// ---- staging: global -> LDS directly (LDS-DMA) ----------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
__device__ __forceinline__ void glds16(const bf16* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
// One wave-instruction fills 1 KiB of the image = 64 consecutive 16-byte chunk POSITIONS; the chunk a lane
// fetches is the inverse swizzle of its position.  Wave `wid` issues instructions wid, wid+4, ...
// Per-lane source pointers are computed once (init) and advanced by one K tile per iteration.
template <int BX>
__device__ __forceinline__ void dma_init_rowmajor(const bf16* X, long ld, int R, int r0, int wid, int lane, const bf16* (&p)[BX / 32]) {
#pragma unroll
  for (int i = 0; i < BX / 32; ++i) {
    const int I = wid + 4 * i;
    const int row = I * 8 + (lane >> 3), ch = (lane & 7) ^ ((lane >> 3) & 7);      // img128_off inverse
    p[i] = X + (long)min(r0 + row, R - 1) * ld + (ch << 3);
  }
}
template <int BX>
__device__ __forceinline__ void dma_init_kmajor(const bf16* X, long ld, int Ccols, int c0, int wid, int lane, const bf16* (&p)[BX / 32]) {
#pragma unroll
  for (int i = 0; i < BX / 32; ++i) {
    const int I = wid + 4 * i;
    int krow, ch;
    if constexpr (BX == 128) {
      krow = I * 4 + (lane >> 4);
      ch = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));                 // img256_off inverse
    } else {
      krow = I * 8 + (lane >> 3);
      ch = (lane & 7) ^ ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1);    // img128t_off inverse
    }
    p[i] = X + (long)krow * ld + min(c0 + (ch << 3), Ccols - 8);
  }
}
template <int BX>
__device__ __forceinline__ void dma_issue(const bf16* (&p)[BX / 32], long step, char* img, int wid) {
#pragma unroll
  for (int i = 0; i < BX / 32; ++i) {
    glds16(p[i], img + (wid + 4 * i) * 1024);
    p[i] += step;
  }
}

// ---- fragment reads -----------------------------------------------------------------------------
// Operand fragment of v_mfma_f32_16x16x32_bf16: lane (r = lane&15, g = lane>>4) holds the 8 values
// k = 8g .. 8g+7 of row/column r.
template <bool T, int BX>
__device__ __forceinline__ bf16x8 read_frag(const char* img, int rc0, int ks, int lane) {
  const int r = lane & 15, g = lane >> 4;
  if constexpr (!T) {
    return *reinterpret_cast<const bf16x8*>(img + img128_off(rc0 + r, 4 * ks + g));
  } else {
    const int q = r >> 2, p = r & 3;
    const int k0 = 32 * ks + 8 * g + q;
    const int ch = (rc0 >> 3) + (p >> 1), sub = (p & 1) << 3;
    const bf16x4 lo = lds_read_tr(img + imgT_off<BX>(k0, ch) + sub);
    const bf16x4 hi = lds_read_tr(img + imgT_off<BX>(k0 + 4, ch) + sub);
    return cat4(lo, hi);
  }
}

template <int BM, int BN, bool A_T, bool B_T, int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int TM = BM / 2, TN = BN / 2;        // wave tile (2 x 2 waves)
  constexpr int MI = TM / 16, NI = TN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int nk = (g.K + BK - 1) / BK;

  char* sA0 = smem;
  char* sB0 = smem + A_BYTES;
  char* sA1 = smem + A_BYTES + B_BYTES;
  char* sB1 = smem + 2 * A_BYTES + B_BYTES;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS-DMA source pointers (advance by one K tile per issue)
  const int uwid = __builtin_amdgcn_readfirstlane(wid);
  const bf16* pa[BM / 32];
  const bf16* pb[BN / 32];
  if constexpr (A_T) dma_init_kmajor<BM>(g.A, g.lda, g.M, m0, uwid, lane, pa); else dma_init_rowmajor<BM>(g.A, g.lda, g.M, m0, uwid, lane, pa);
  if constexpr (B_T) dma_init_kmajor<BN>(g.B, g.ldb, g.N, n0, uwid, lane, pb); else dma_init_rowmajor<BN>(g.B, g.ldb, g.N, n0, uwid, lane, pb);
  const long stepA = A_T ? (long)BK * g.lda : BK, stepB = B_T ? (long)BK * g.ldb : BK;
  const int nfull = g.K / BK;          // K tiles that need no zero fill

  // ragged K tail (only the weight-gradient GEMMs, K = tokens): staged through registers with zero fill
  uint4 ra[BM / 32], rb[BN / 32];
  auto gload_tail = [&](int kt) {
    const int k0 = kt * BK;
    if constexpr (A_T) gload_kmajor<BM, true>(g.A, g.lda, g.M, g.K, m0, k0, tid, ra);
    else gload_rowmajor<BM, true>(g.A, g.lda, g.M, g.K, m0, k0, tid, ra);
    if constexpr (B_T) gload_kmajor<BN, true>(g.B, g.ldb, g.N, g.K, n0, k0, tid, rb);
    else gload_rowmajor<BN, true>(g.B, g.ldb, g.N, g.K, n0, k0, tid, rb);
  };
  auto swrite = [&](char* a, char* b) {
    if constexpr (A_T) swrite_kmajor<BM>(a, tid, ra); else swrite_rowmajor<BM>(a, tid, ra);
    if constexpr (B_T) swrite_kmajor<BN>(b, tid, rb); else swrite_rowmajor<BN>(b, tid, rb);
  };

  if (nfull > 0) {
    dma_issue<BM>(pa, stepA, sA0, uwid);
    dma_issue<BN>(pb, stepB, sB0, uwid);
  } else {
    gload_tail(0);
    swrite(sA0, sB0);
  }
  __syncthreads();   // hipcc drains the outstanding LDS-DMA (vmcnt(0)) in front of the barrier

  for (int kt = 0; kt < nk; ++kt) {
    const char* a = (kt & 1) ? sA1 : sA0;
    const char* b = (kt & 1) ? sB1 : sB0;
    char* na = (kt & 1) ? sA0 : sA1;
    char* nb = (kt & 1) ? sB0 : sB1;
    const bool next_dma = kt + 1 < nfull, next_tail = (kt + 1 < nk) && !next_dma;
    if (next_dma) {              // the buffer was last read in iteration kt-1; every wave has passed that barrier
      dma_issue<BM>(pa, stepA, na, uwid);
      dma_issue<BN>(pb, stepB, nb, uwid);
    } else if (next_tail) {
      gload_tail(kt + 1);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = read_frag<A_T, BM>(a, wm * TM + 16 * i, ks, lane);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = read_frag<B_T, BN>(b, wn * TN + 16 * j, ks, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (next_tail) swrite(na, nb);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3] ---------------------
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * TM + 16 * i + lr;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * TN + 16 * j + 4 * lg;
      if (n >= g.N) continue;
      f32x4 v = acc[i][j];
      if constexpr (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n);
        v += bv;
      }
      if constexpr (EPI == EPI_STORE_BF16) {
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

// ---- host side ---------------------------------------------------------------------------------------
static int g_tile_override = 0;   // 0 = heuristic; else BM*1000 + BN (tuning aid, tools/gemm_bench.py)
extern "C" int nv_gemm_set_tile(int bm, int bn) {
  g_tile_override = (bm == 0) ? 0 : bm * 1000 + bn;
  return 0;
}

template <int BM, int BN, bool A_T, bool B_T, int EPI>
static int launch_tile(const GemmArgs& a, hipStream_t s) {
  constexpr int LDS = 2 * (BM + BN) * BK * 2;
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  auto kern = gemm_bf16_kernel<BM, BN, A_T, B_T, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int slot = nv_prof_begin((A_T ? 2 : (B_T ? 1 : 0)), 2.0 * a.M * a.N * a.K, s);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHREADS), LDS, s, a);
  nv_prof_end(slot, s);
  NV_CHECK_LAUNCH("nv_gemm_bf16");
  return NV_OK;
}

template <bool A_T, bool B_T, int EPI>
static int launch(const GemmArgs& a, hipStream_t s) {
  int sel = g_tile_override;
  if (!sel) {
    // enough workgroups to give every one of the 256 CUs several co-resident blocks; prefer the larger tile
    // (less LDS traffic per MFMA) when the problem is big enough.
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const long t64x128 = (long)((a.M + 63) / 64) * ((a.N + 127) / 128);
    sel = (t128 >= 1024) ? 128128 : (t64x128 >= 768 ? 64128 : 64064);
  }
  switch (sel) {
    case 128128: return launch_tile<128, 128, A_T, B_T, EPI>(a, s);
    case 64128: return launch_tile<64, 128, A_T, B_T, EPI>(a, s);
    case 128064: return launch_tile<128, 64, A_T, B_T, EPI>(a, s);
    default: return launch_tile<64, 64, A_T, B_T, EPI>(a, s);
  }
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
