// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip: small-tile and warp-specialised kernels; gemm_pp.hip: the
// eight-wave ping-pong kernel for large tiles): argument block, LDS images, fragment reads, fused epilogues.
#pragma once
#include "common.h"

enum {
  EPI_STORE_BF16 = 0,   // C(bf16) = acc
  EPI_STORE_F32 = 1,    // C(f32)  = acc (+ C if accumulate); aux_out(bf16), when given, = the same values rounded
  EPI_BIAS_F32 = 2,     // C(f32)  = acc + bias[n]
  EPI_BIAS_GELU = 3,    // aux_out(bf16) = u = acc + bias[n];  C(bf16) = gelu(u)
  EPI_BIAS_RESID = 4,   // C(f32)  = aux_in(f32)[m,n] + acc + bias[n]
  EPI_DGELU = 5,        // C(bf16) = acc * gelu'(aux_in(bf16)[m,n])
  EPI_BIAS_GELU_F8 = 7, // C(fp8 e4m3) = sat(gelu(acc * colscale[n] + bias[n]) * alpha): FC1 of the fp8 inference path, feeding FC2 directly
  EPI_BIAS_GELU_F8T = 8, // the same for a TRAINING forward on fp8 operands: C(fp8) as 7, and what the bf16 backward pass reads - aux_out(bf16) = the
                        // pre-activation u, aux_out2(bf16) = gelu(u) - out of the same epilogue (one pass over the tile instead of a quantise pass)
  EPI_DGELU_COLSUM = 6, // EPI_DGELU + aux_out(f32)[tile_row, n] = column sums of the stored bf16 values over the tile's rows: the bias
                        // gradient of the Linear in front of the GELU, produced where the tile already is (large-tile kernels only)
  // ---- LayerNorm folded into the GEMMs around it (inference forwards): LN(x) W^T = rstd (x W_g^T) - rstd mu colsum(W_g) + (W beta + b) with
  // W_g = W diag(gamma).  The producer of the residual stream also writes x in the operand format and per-(row, 128-column tile) statistics; the consumer
  // contracts the UN-normalised rows with W_g and applies the row statistics in its epilogue: no LayerNorm launch, no normalised copy.  LDS-epilogue kernels only.
  EPI_BIAS_RESID_LN = 10,  // EPI_BIAS_RESID (no dropout) + aux_out(16-bit) = the stored row + aux_out2(f32)[tile column][M][2] = partial statistics of the
                           // tile's <= 128 columns of each row, as (sum, sum of squares) - converted per tile and merged pairwise by the consumer (Chan's update)
  EPI_LNFOLD_STORE = 11,   // C(16-bit) = rstd[m] (acc - mu[m] ln_cs[n]) + bias[n]; (mu, rstd) of row m from ln_stats (ln_tiles partials over K = the LayerNorm width)
  EPI_LNFOLD_GELU = 12,    // C(16-bit) = gelu(the same)
  EPI_ADAMW = 9,        // weight-gradient GEMM that applies torch.optim.AdamW to the weight it differentiates: acc is the gradient of
                        // opt.p[m, n] (same leading dimension as C); p, m, v are updated in place, p16 = bf16(p); C is only written when
                        // opt.keep_grad.  Saves the 4-byte store and the 4-byte re-read of every gradient (8 of 34 B/param per step)
};

struct OptFuse {        // EPI_ADAMW: element (m, n) of the problem is p[m * ldc + n] (the arenas share element offsets with the gradient arena)
  float* p = nullptr; float* m = nullptr; float* v = nullptr; r16* p16 = nullptr;
  AdamArgs a = {};
  int keep_grad = 0;
};

struct GemmArgs {
  const r16* A;
  const r16* B;
  void* C;
  const float* bias;
  const void* aux_in;
  void* aux_out;
  void* aux_out2;   // EPI_BIAS_GELU_F8T only
  long lda, ldb, ldc, ld_aux_in, ld_aux_out, ld_aux_out2;
  int M, N, K;
  int accumulate;
  DropCfg drop;    // EPI_BIAS_RESID: on (acc + bias); EPI_BIAS_GELU: on gelu(u); EPI_DGELU: on acc (the incoming dH)
  int col_order;   // 1: consecutive workgroups walk DOWN a tile column (keeps the B panel in the XCD's L2), 0: along a tile row
  float alpha;
  const float* colscale;   // fp8 operands: acc *= colscale[n] (1 / (activation scale * weight-row scale)) before the epilogue; null otherwise
  OptFuse opt;             // EPI_ADAMW only
  const float* ln_stats = nullptr;   // EPI_LNFOLD_*: [ln_tiles][M][2] partial row statistics written by an EPI_BIAS_RESID_LN launch over the same rows
  const float* ln_cs = nullptr;      //               colsum over k of W_g[n, k] (of the ROUNDED 16-bit values the MFMA reads)
  int ln_tiles = 0;
  float ln_eps = 0.f;
};

// algorithmic bytes of one GEMM launch: both operands read once, every output written once, epilogue inputs read once
static inline double gemm_algo_bytes(const GemmArgs& a, int epi, int operand_bytes) {
  const double mn = (double)a.M * a.N;
  double b = ((double)a.M * a.K + (double)a.N * a.K) * operand_bytes;
  switch (epi) {
    case EPI_STORE_BF16: b += mn * 2; break;
    case EPI_STORE_F32: b += mn * 4 * (a.accumulate ? 2 : 1) + (a.aux_out ? mn * 2 : 0); break;
    case EPI_BIAS_F32: b += mn * 4; break;
    case EPI_BIAS_GELU: b += mn * 2 + (a.aux_out ? mn * 2 : 0); break;
    case EPI_BIAS_RESID: b += mn * 8; break;
    case EPI_DGELU: case EPI_DGELU_COLSUM: b += mn * 4; break;
    case EPI_BIAS_GELU_F8: b += mn; break;
    case EPI_BIAS_GELU_F8T: b += mn * 5; break;
    case EPI_ADAMW: b += mn * (26 + (a.opt.keep_grad ? 4 : 0)); break;      // p, m, v read and written, bf16 shadow written
    case EPI_BIAS_RESID_LN: b += mn * 10; break;
    case EPI_LNFOLD_STORE: case EPI_LNFOLD_GELU: b += mn * 2 + (double)a.M * a.ln_tiles * 8; break;
    default: break;
  }
  return b;
}

constexpr int BK = 64;
constexpr int NTHREADS = 256;
typedef __attribute__((address_space(3))) void lds_void_t;

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
__device__ __forceinline__ void gload_rowmajor(const r16* X, long ld, int R, int K, int r0, int k0, int tid, uint4 (&reg)[BX / 32]) {
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
__device__ __forceinline__ void gload_kmajor(const r16* X, long ld, int Ccols, int K, int c0, int k0, int tid, uint4 (&reg)[BX / 32]) {
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
typedef __attribute__((address_space(1))) const void gbl_void_t;
__device__ __forceinline__ void glds16(const r16* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)lds_wave_base, 16, 0, 0);
}
// One wave-instruction fills 1 KiB of the image = 64 consecutive 16-byte chunk POSITIONS; the chunk a lane
// fetches is the inverse swizzle of its position.  Wave `wid` issues instructions wid, wid+4, ...
// Per-lane source pointers are computed once (init) and advanced by one K tile per iteration.
template <int BX>
__device__ __forceinline__ void dma_init_rowmajor(const r16* X, long ld, int R, int r0, int wid, int lane, const r16* (&p)[BX / 32]) {
#pragma unroll
  for (int i = 0; i < BX / 32; ++i) {
    const int I = wid + 4 * i;
    const int row = I * 8 + (lane >> 3), ch = (lane & 7) ^ ((lane >> 3) & 7);      // img128_off inverse
    p[i] = X + (long)min(r0 + row, R - 1) * ld + (ch << 3);
  }
}
template <int BX>
__device__ __forceinline__ void dma_init_kmajor(const r16* X, long ld, int Ccols, int c0, int wid, int lane, const r16* (&p)[BX / 32]) {
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
__device__ __forceinline__ void dma_issue(const r16* (&p)[BX / 32], long step, char* img, int wid) {
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
__device__ __forceinline__ r16x8 read_frag(const char* img, int rc0, int ks, int lane) {
  const int r = lane & 15, g = lane >> 4;
  if constexpr (!T) {
    return *reinterpret_cast<const r16x8*>(img + img128_off(rc0 + r, 4 * ks + g));
  } else {
    const int q = r >> 2, p = r & 3;
    const int k0 = 32 * ks + 8 * g + q;
    const int ch = (rc0 >> 3) + (p >> 1), sub = (p & 1) << 3;
    const r16x4 lo = lds_read_tr(img + imgT_off<BX>(k0, ch) + sub);
    const r16x4 hi = lds_read_tr(img + imgT_off<BX>(k0 + 4, ch) + sub);
    return cat4(lo, hi);
  }
}

// ---- epilogue core: four consecutive output columns (m, n .. n+3) ----------------------------------------------------
template <int EPI, typename T>
__device__ __forceinline__ f32x4 epilogue4(f32x4 v, const GemmArgs& g, int m, int n) {
  if (g.colscale) v *= *reinterpret_cast<const f32x4*>(g.colscale + n);
  if constexpr (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_GELU_F8 || EPI == EPI_BIAS_GELU_F8T || EPI == EPI_BIAS_RESID_LN)
    v += *reinterpret_cast<const f32x4*>(g.bias + n);
  f32x4 keep = f32x4{1.f, 1.f, 1.f, 1.f};
  if constexpr (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID || EPI == EPI_DGELU || EPI == EPI_DGELU_COLSUM || EPI == EPI_BIAS_GELU_F8T) {
    if (g.drop.thresh) {
      keep = drop_factor4(g.drop, (unsigned long long)m * g.N + n);      // N % 8 == 0, n % 4 == 0
    }
  }
  if constexpr (EPI == EPI_STORE_BF16) {
    *reinterpret_cast<r16x4*>((r16*)g.C + (long)m * g.ldc + n) = cvt4<T>(v[0], v[1], v[2], v[3]);
  } else if constexpr (EPI == EPI_STORE_F32) {
    float* c = (float*)g.C + (long)m * g.ldc + n;
    if (g.accumulate) v += *reinterpret_cast<const f32x4*>(c);
    *reinterpret_cast<f32x4*>(c) = v;
    // optional bf16 mirror of what was stored (data-parallel gradient messages: the weight gradient leaves its GEMM already in the
    // wire format, no cast pass over the arena afterwards)
    if (g.aux_out) *reinterpret_cast<r16x4*>((r16*)g.aux_out + (long)m * g.ld_aux_out + n) = cvt4<T>(v[0], v[1], v[2], v[3]);
  } else if constexpr (EPI == EPI_ADAMW) {
    // the arithmetic of adamw_kernel (optim.hip) on the gradient the fp32-store epilogue would have written: same bits
    const long off = (long)m * g.ldc + n;
    if (g.opt.keep_grad) *reinterpret_cast<f32x4*>((float*)g.C + off) = v;
    f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.opt.p + off));
    f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.opt.m + off));
    f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.opt.v + off));
    adamw_update4(pv, v * g.opt.a.grad_scale, mv, vv, g.opt.a);
    __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(g.opt.p + off));
    __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(g.opt.m + off));
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(g.opt.v + off));
    *reinterpret_cast<r16x4*>(g.opt.p16 + off) = cvt4<T>(pv[0], pv[1], pv[2], pv[3]);
  } else if constexpr (EPI == EPI_BIAS_F32) {
    *reinterpret_cast<f32x4*>((float*)g.C + (long)m * g.ldc + n) = v;
  } else if constexpr (EPI == EPI_BIAS_GELU) {
    // the pre-activation is only read again in the backward pass, milliseconds later: non-temporal store
    // (aux_out = null: inference, nobody reads it - half of this epilogue's HBM bytes saved)
    if (g.aux_out) __builtin_nontemporal_store(cvt4<T>(v[0], v[1], v[2], v[3]), reinterpret_cast<r16x4*>((r16*)g.aux_out + (long)m * g.ld_aux_out + n));
    *reinterpret_cast<r16x4*>((r16*)g.C + (long)m * g.ldc + n) =
        cvt4<T>(gelu_f(v[0]) * keep[0], gelu_f(v[1]) * keep[1], gelu_f(v[2]) * keep[2], gelu_f(v[3]) * keep[3]);
  } else if constexpr (EPI == EPI_BIAS_GELU_F8) {
    const f32x4 h = f32x4{gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])} * g.alpha;
    *reinterpret_cast<unsigned*>((char*)g.C + (long)m * g.ldc + n) = pack_fp8x4(h);
  } else if constexpr (EPI == EPI_BIAS_GELU_F8T) {
    const f32x4 h = f32x4{gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])} * keep;      // nn.Dropout behind the GELU (vit_3d.py:21): h16 and h8 carry the same mask
    if (g.aux_out) __builtin_nontemporal_store(cvt4<T>(v[0], v[1], v[2], v[3]), reinterpret_cast<r16x4*>((r16*)g.aux_out + (long)m * g.ld_aux_out + n));
    *reinterpret_cast<r16x4*>((r16*)g.aux_out2 + (long)m * g.ld_aux_out2 + n) = cvt4<T>(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<unsigned*>((char*)g.C + (long)m * g.ldc + n) = pack_fp8x4(h * g.alpha);
  } else if constexpr (EPI == EPI_BIAS_RESID) {
    v = v * keep + *reinterpret_cast<const f32x4*>((const float*)g.aux_in + (long)m * g.ld_aux_in + n);
    *reinterpret_cast<f32x4*>((float*)g.C + (long)m * g.ldc + n) = v;
  } else if constexpr (EPI == EPI_BIAS_RESID_LN) {
    v += *reinterpret_cast<const f32x4*>((const float*)g.aux_in + (long)m * g.ld_aux_in + n);
    *reinterpret_cast<f32x4*>((float*)g.C + (long)m * g.ldc + n) = v;
    *reinterpret_cast<r16x4*>((r16*)g.aux_out + (long)m * g.ld_aux_out + n) = cvt4<T>(v[0], v[1], v[2], v[3]);      // (the row statistics: epilogue_lds)
  } else if constexpr (EPI == EPI_DGELU || EPI == EPI_DGELU_COLSUM) {
    const f32x4 u = dec4<T>(*reinterpret_cast<const r16x4*>((const r16*)g.aux_in + (long)m * g.ld_aux_in + n));
    const r16x4 o = cvt4<T>(v[0] * keep[0] * gelu_grad_f(u[0]), v[1] * keep[1] * gelu_grad_f(u[1]),
                            v[2] * keep[2] * gelu_grad_f(u[2]), v[3] * keep[3] * gelu_grad_f(u[3]));
    *reinterpret_cast<r16x4*>((r16*)g.C + (long)m * g.ldc + n) = o;
    v = dec4<T>(o);     // what the weight-gradient GEMM will read: summed as stored
  }
  return v;
}

// ---- register epilogue: lane holds C[m = mb + 16i + (lane&15)][n = nb + 16j + 4*(lane>>4) + 0..3] -----------------
template <int EPI, typename T, int MI, int NI>
__device__ __forceinline__ void epilogue(const f32x4 (&acc)[MI][NI], const GemmArgs& g, int mb, int nb, int lane) {
  static_assert(EPI != EPI_DGELU_COLSUM, "fused column sums need the LDS epilogue");
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mb + 16 * i + lr;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = nb + 16 * j + 4 * lg;
      if (n >= g.N) continue;
      epilogue4<EPI, T>(acc[i][j], g, m, n);
    }
  }
}

// ---- LDS epilogue (warp-specialised kernel): the consumers park their accumulators in LDS as a [BM][BN] fp32 tile, then
// ALL EIGHT waves (the loaders are idle by now) run the epilogue row-wise: twice the VALU issue capacity for the GELU
// epilogues, and every wave-instruction reads / writes whole 256-512 byte row segments instead of 16 rows x 32 bytes.
template <int BN> constexpr int cpitch() { return BN * 4 + 16; }   // bytes; +16 keeps the 16-row accumulator writes conflict free
template <int MI, int NI, int BN>
__device__ __forceinline__ void park_acc(const f32x4 (&acc)[MI][NI], char* ctile, int rb, int cb, int lane) {
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      *reinterpret_cast<f32x4*>(ctile + (rb + 16 * i + lr) * cpitch<BN>() + (cb + 16 * j + 4 * lg) * 4) = acc[i][j];
}
template <int BM, int BN, int NT> constexpr int colsum_scratch_bytes() { return (NT / 64) * BN * 4; }   // behind the parked C tile
// EPI_ADAMW: the update reads three arrays and writes four per element, and the compiler cannot move the loads of one chunk above the
// stores of the one before (the arenas are not `restrict` to it): issued chunk by chunk every wave would sit out one full memory
// latency per chunk.  Loads of ADAM_U chunks are issued together, as the streaming kernel does (optim.hip).
constexpr int ADAM_U = 4;
template <typename T, int BM, int BN, int NT>
__device__ __forceinline__ void epilogue_lds_adamw(const char* ctile, const GemmArgs& g, int m0, int n0, int tid) {
  constexpr int CPR = BN / 4, TOT = BM * CPR;
  for (int c0 = tid; c0 < TOT; c0 += NT * ADAM_U) {
    f32x4 pv[ADAM_U], mv[ADAM_U], vv[ADAM_U];
    long off[ADAM_U];
    bool ok[ADAM_U];
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const int c = c0 + u * NT, row = c / CPR, col = (c % CPR) * 4;
      ok[u] = c < TOT && m0 + row < g.M && n0 + col < g.N;
      off[u] = (long)(m0 + row) * g.ldc + n0 + col;
      if (ok[u]) {
        pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.opt.p + off[u]));
        mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.opt.m + off[u]));
        vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.opt.v + off[u]));
      }
    }
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      if (ok[u]) {
        const int c = c0 + u * NT, row = c / CPR, col = (c % CPR) * 4;
        const f32x4 gr = *reinterpret_cast<const f32x4*>(ctile + row * cpitch<BN>() + col * 4);
        if (g.opt.keep_grad) *reinterpret_cast<f32x4*>((float*)g.C + off[u]) = gr;
        adamw_update4(pv[u], gr * g.opt.a.grad_scale, mv[u], vv[u], g.opt.a);
        __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(g.opt.p + off[u]));
        __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(g.opt.m + off[u]));
        __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(g.opt.v + off[u]));
        *reinterpret_cast<r16x4*>(g.opt.p16 + off[u]) = cvt4<T>(pv[u][0], pv[u][1], pv[u][2], pv[u][3]);
      }
    }
  }
}

// (mu, rstd) of row m from the partial sums (sum, sum of squares) of its 128-column tiles: per tile (count, mean, M2 = ss - s^2 / count), merged pairwise in
// tile order (Chan et al.).  fp32 single-pass sums over <= 128 values lose ~1e-7 (mean / std)^2 of the variance: nothing at the ratios (< 16) at which
// rounding x to 16 bits ahead of the subtraction - the fold itself - still works.
constexpr int LN_TILES_INFLIGHT = 8;       // widths up to 1024: every tile's partial pair is requested before the first is used
__device__ __forceinline__ void ln_row_stats(const GemmArgs& g, int m, float& mu, float& rstd) {
  float mean = 0.f, m2 = 0.f, n = 0.f;
  auto merge = [&](int t, float st, float sst) {
    const float nt = (float)min(128, g.K - 128 * t);
    const float mt = st / nt, qt = fmaxf(sst - st * mt, 0.f);
    const float tot = n + nt, delta = mt - mean;
    mean += delta * (nt / tot);
    m2 += qt + delta * delta * (n * nt / tot);
    n = tot;
  };
  if (g.ln_tiles <= LN_TILES_INFLIGHT) {
    // one memory latency instead of one per tile: as a runtime loop the compiler waited for each pair before it asked for the next (six dependent
    // round trips at d = 768, at the head of every consumer workgroup: +1.8 us on the qkv GEMM)
    f32x2 part[LN_TILES_INFLIGHT];
#pragma unroll
    for (int t = 0; t < LN_TILES_INFLIGHT; ++t)
      part[t] = (t < g.ln_tiles) ? *reinterpret_cast<const f32x2*>(g.ln_stats + ((long)t * g.M + m) * 2) : f32x2{0.f, 0.f};
#pragma unroll
    for (int t = 0; t < LN_TILES_INFLIGHT; ++t)
      if (t < g.ln_tiles) merge(t, part[t][0], part[t][1]);
  } else {
    for (int t = 0; t < g.ln_tiles; ++t) merge(t, g.ln_stats[((long)t * g.M + m) * 2], g.ln_stats[((long)t * g.M + m) * 2 + 1]);
  }
  mu = mean;
  rstd = 1.0f / sqrtf(m2 / (float)g.K + g.ln_eps);
}
// EPI_LNFOLD_*: the (mu, rstd) pairs of the workgroup's BM rows, computed by `nthreads` threads (t = 0 .. nthreads-1) into LDS at the START of the kernel, while
// the first operand tiles are still on their way - at the tail, in front of the epilogue, the dependent global loads were 2-3 us of exposed latency per launch
__device__ __forceinline__ void ln_rows_to_lds(const GemmArgs& g, int m0, int BM, float* dst, int t, int nthreads) {
  for (int r = t; r < BM; r += nthreads) {
    float mu = 0.f, rstd = 0.f;
    if (m0 + r < g.M) ln_row_stats(g, m0 + r, mu, rstd);
    dst[2 * r] = mu; dst[2 * r + 1] = rstd;
  }
}
// sum over the 32 lanes of a half-wave on the VALU: DPP inside a row of 16 (quad xor 1, quad xor 2, row_half_mirror, row_mirror), then the two rows of the
// half-wave with v_permlane16_swap (gfx950) - five ds_bpermute round trips per value through __shfl_xor otherwise
__device__ __forceinline__ float dpp_fold(float v, const int ctrl_sel) {
  int o;
  if (ctrl_sel == 0) o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);         // quad_perm [1, 0, 3, 2]
  else if (ctrl_sel == 1) o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true);    // quad_perm [2, 3, 0, 1]
  else if (ctrl_sel == 2) o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true);   // row_half_mirror
  else o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true);                      // row_mirror
  return v + __builtin_bit_cast(float, o);
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v = dpp_fold(v, 0); v = dpp_fold(v, 1); v = dpp_fold(v, 2); v = dpp_fold(v, 3);
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));      // (wait states on both sides: the asm is outside the compiler's hazard bookkeeping) (a, b) = the values of lanes l & ~16 and l | 16 (attention.hip::lane_pair16)
  return a + b;
}

template <int EPI> constexpr bool epi_is_fold() { return EPI == EPI_LNFOLD_STORE || EPI == EPI_LNFOLD_GELU; }
// LDS behind the ring, alive for the whole kernel: (mu, rstd) of the BM rows, then colsum and folded bias of the tile's BN columns
template <int EPI, int BM, int BN = 128> constexpr int ln_rows_bytes() { return epi_is_fold<EPI>() ? BM * 8 + BN * 8 : 0; }
// the tile's BN entries of ln_cs and of the folded bias into LDS behind the row statistics (threads t = 0 .. BN / 2 - 1 of the caller: one f32x4 each)
template <int BM, int BN>
__device__ __forceinline__ void ln_cols_to_lds(const GemmArgs& g, int n0, float* lnrow, int t) {
  if (t < BN / 2) {
    const int which = t / (BN / 4), n = n0 + (t % (BN / 4)) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n < g.N) v = *reinterpret_cast<const f32x4*>((which ? g.bias : g.ln_cs) + n);
    *reinterpret_cast<f32x4*>(lnrow + 2 * BM + which * BN + (t % (BN / 4)) * 4) = v;
  }
}

template <int EPI, typename T, int BM, int BN, int NT>
__device__ __forceinline__ void epilogue_lds(char* ctile, const GemmArgs& g, int m0, int n0, int tid, const float* lnrow = nullptr) {
  if constexpr (EPI == EPI_ADAMW) { epilogue_lds_adamw<T, BM, BN, NT>(ctile, g, m0, n0, tid); return; }
  constexpr int CPR = BN / 4;                 // 16-byte chunks per row
  static_assert(NT % CPR == 0 && 64 % CPR == 0, "every thread keeps one column group");
  constexpr bool FOLD = epi_is_fold<EPI>();
  float* scr = reinterpret_cast<float*>(ctile + BM * cpitch<BN>());       // scratch behind the parked tile (column sums)
  f32x4 csum = f32x4{0.f, 0.f, 0.f, 0.f};
  static_assert(EPI != EPI_BIAS_RESID_LN || ((BM * CPR) % 64 == 0 && NT % 64 == 0 && CPR == 32 && BN == 128), "whole waves leave the loop together; 32 lanes share a row");
#pragma unroll 4
  for (int c = tid; c < BM * CPR; c += NT) {
    const int row = c / CPR, col = (c % CPR) * 4;
    const int m = m0 + row, n = n0 + col;
    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m < g.M && n < g.N) {
      const f32x4 cell = *reinterpret_cast<const f32x4*>(ctile + row * cpitch<BN>() + col * 4);
      if constexpr (FOLD) {
        const float mu = lnrow[2 * row], rstd = lnrow[2 * row + 1];
        const f32x4 y = (cell - mu * *reinterpret_cast<const f32x4*>(lnrow + 2 * BM + col)) * rstd + *reinterpret_cast<const f32x4*>(lnrow + 2 * BM + BN + col);
        if constexpr (EPI == EPI_LNFOLD_GELU) *reinterpret_cast<r16x4*>((r16*)g.C + (long)m * g.ldc + n) = cvt4<T>(gelu_f(y[0]), gelu_f(y[1]), gelu_f(y[2]), gelu_f(y[3]));
        else *reinterpret_cast<r16x4*>((r16*)g.C + (long)m * g.ldc + n) = cvt4<T>(y[0], y[1], y[2], y[3]);
      } else {
        r = epilogue4<EPI, T>(cell, g, m, n);
        if constexpr (EPI == EPI_DGELU_COLSUM) csum += r;
      }
    }
    if constexpr (EPI == EPI_BIAS_RESID_LN) {
      // sum and sum of squares of this tile's columns of the row: the 32 lanes of a half-wave hold one row (4 finished values each; zeros beyond N / M)
      float s1 = (r[0] + r[1]) + (r[2] + r[3]), s2 = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
      s1 = half_wave_sum(s1); s2 = half_wave_sum(s2);
      if ((tid & 31) == 0 && m < g.M) {
        float* out = (float*)g.aux_out2 + ((long)(n0 / BN) * g.M + m) * 2;
        out[0] = s1; out[1] = s2;
      }
    }
  }
  if constexpr (EPI == EPI_DGELU_COLSUM) {
    // deterministic column sums of this tile: lanes of a wave that share a column group, then the waves in a fixed order
    // through a small LDS array behind the parked tile, one partial row per tile row of the grid
#pragma unroll
    for (int o = 32; o >= CPR; o >>= 1) {
      csum[0] += __shfl_xor(csum[0], o, 64); csum[1] += __shfl_xor(csum[1], o, 64);
      csum[2] += __shfl_xor(csum[2], o, 64); csum[3] += __shfl_xor(csum[3], o, 64);
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < CPR) *reinterpret_cast<f32x4*>(scr + wave * BN + lane * 4) = csum;
    __syncthreads();
    if (tid < BN && n0 + tid < g.N) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) s += scr[w * BN + tid];
      ((float*)g.aux_out)[(long)(m0 / BM) * g.ld_aux_out + n0 + tid] = s;
    }
  }
}


// Grouped launch: up to four independent problems of one layout / epilogue in ONE grid.  The four weight-gradient GEMMs of a
// transformer layer have 72-288 tiles each: launched one by one each leaves half of the CUs' two workgroup slots empty and
// its workgroups run latency-bound; together (864 tiles) every CU holds two co-resident workgroups that hide each other's
// LDS / DMA latency, and three launch ramps disappear.
constexpr int GROUP_MAX = 4;
struct GemmGroup {
  GemmArgs p[GROUP_MAX];
  int tile_end[GROUP_MAX];   // exclusive prefix sums of the problems' tile counts
  int count;
};
