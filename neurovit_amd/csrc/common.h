// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of neurovit_amd.
// Wave size is 64; MFMA shape used throughout is v_mfma_f32_16x16x32_bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

// 16-bit MFMA operand formats.  Buffers, LDS images and register fragments carry RAW 16-bit words (r16 ...): loads, LDS-DMA, swizzles and
// transposed reads are format-agnostic.  The element format T - bf16_t (default) or fp16_t (the reference's own autocast arithmetic,
// src/Trainer.py:29,68) - enters only where a value is produced or interpreted: cvt* / dec* and the MFMA instruction (mfma16<T>), which
// issue at the same rate for both on gfx950.  Kernels are templates over T; launchers pick the instantiation from the process-wide
// nv_set_operand_format (api.cpp).  r16 is an integer type on purpose: a stray (float)x on raw words is wrong in BOTH formats.
typedef __bf16 bf16_t;
typedef _Float16 fp16_t;
typedef short r16;
typedef short r16x2 __attribute__((ext_vector_type(2)));
typedef short r16x4 __attribute__((ext_vector_type(4)));
typedef short r16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) r16x4 lds_r16x4;
template <typename T> struct vec16 {
  typedef T x2 __attribute__((ext_vector_type(2)));
  typedef T x4 __attribute__((ext_vector_type(4)));
  typedef T x8 __attribute__((ext_vector_type(8)));
};

#include "../../include/neurovit_hip.h"   // every definition is checked against the published C-ABI declarations

extern "C" void nv_set_error(const char* fmt, ...);
extern "C" int nv_prof_begin(int kind, double work, void* stream);   // -1 when profiling is off
extern "C" void nv_prof_bytes(int slot, double bytes);                // algorithmic bytes of that launch (optional)
extern "C" void nv_prof_end(int slot, void* stream);
extern "C" int nv_stream_sync(void* from, void* to);   // `to` waits for everything enqueued on `from` (pooled events)
unsigned nv_sync_event_flags();                            // hipEventCreateWithFlags flags of those events (api.cpp)

#define NV_CHECK_ARG(cond, ...)                                                                   \
  do {                                                                                            \
    if (!(cond)) {                                                                                \
      nv_set_error(__VA_ARGS__);                                                                  \
      return NV_ERR_ARG;                                                                          \
    }                                                                                             \
  } while (0)

#define NV_CHECK_LAUNCH(name)                                                                     \
  do {                                                                                            \
    hipError_t e__ = hipGetLastError();                                                           \
    if (e__ != hipSuccess) {                                                                      \
      nv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));                        \
      return NV_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

static inline bool nv_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ------------------------------------------------------------------------------------------------
// LDS images.  All byte offsets are relative to a 16-byte aligned tile base.
//
// IMG128: [rows][64 bf16] tile, 128-byte rows, 8 x 16-byte chunks per row, chunk index XOR (row & 7).
//   * ds_read_b128 "row reads" (lane = row r, chunk 4*ks + g) are conflict free,
//   * ds_read_b64_tr_b16 "transposed reads" whose 16-lane group g reads rows k0 + q with
//     k0 = 4*g (+16) are conflict free too, so ONE image serves both MFMA operand orientations.
__device__ __forceinline__ int img128_off(int row, int chunk) {
  return row * 128 + ((chunk ^ (row & 7)) << 4);
}
// IMG256: [64 k-rows][128 bf16] tile, 256-byte rows, 16 chunks per row (transposed-read GEMM operands;
// layout (b) of the CDNA4 guide: conflict free for tr reads whose group g reads rows 8g+q / 8g+4+q).
__device__ __forceinline__ int img256_off(int row, int chunk) {
  return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

__device__ __forceinline__ r16x4 lds_read_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_r16x4*)(p));
}

__device__ __forceinline__ r16x8 cat4(r16x4 a, r16x4 b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// float -> operand format T (round to nearest even: v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32), as raw words; and back
template <typename T>
__device__ __forceinline__ r16 cvt1(float a) {
  const T v = (T)a;
  return __builtin_bit_cast(r16, v);
}
template <typename T>
__device__ __forceinline__ r16x4 cvt4(float a, float b, float c, float d) {
  const typename vec16<T>::x4 r = {(T)a, (T)b, (T)c, (T)d};
  return __builtin_bit_cast(r16x4, r);
}
template <typename T>
__device__ __forceinline__ r16x8 cvt8(f32x4 a, f32x4 b) {
  const typename vec16<T>::x8 r = {(T)a[0], (T)a[1], (T)a[2], (T)a[3], (T)b[0], (T)b[1], (T)b[2], (T)b[3]};
  return __builtin_bit_cast(r16x8, r);
}
template <typename T>
__device__ __forceinline__ float dec1(r16 w) { return (float)__builtin_bit_cast(T, w); }
template <typename T>
__device__ __forceinline__ f32x4 dec4(r16x4 w) {
  const typename vec16<T>::x4 v = __builtin_bit_cast(typename vec16<T>::x4, w);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
// D = A . B + C on the matrix pipe: v_mfma_f32_16x16x32_{bf16,f16} (8 passes) / v_mfma_f32_32x32x16_{bf16,f16}
template <typename T>
__device__ __forceinline__ f32x4 mfma16(r16x8 a, r16x8 b, f32x4 c) {
  typedef typename vec16<T>::x8 V;
  if constexpr (__is_same(T, fp16_t)) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ f32x16 mfma32(r16x8 a, r16x8 b, f32x16 c) {
  typedef typename vec16<T>::x8 V;
  if constexpr (__is_same(T, fp16_t)) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
}

// Host side: the operand format of this process (nv_set_operand_format, api.cpp) and the dispatch of a templated launch on it.
extern "C" int nv_operand_format(void);
#define NV_DISPATCH_OPERAND(T, ...)                  \
  do {                                               \
    if (nv_operand_format() == NV_OPERAND_FP16) {    \
      typedef fp16_t T;                              \
      __VA_ARGS__;                                   \
    } else {                                         \
      typedef bf16_t T;                              \
      __VA_ARGS__;                                   \
    }                                                \
  } while (0)

// exact-erf GELU (nn.GELU() default, vit_3d.py:20) and its derivative.  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 absolute - three orders below the bf16 resolution of the values these feed), ~16 VALU ops
// instead of the ~40 of libm's erff: the GELU epilogues run on the four consumer waves only.
// Both return through one shared exp(-u^2/2).
__device__ __forceinline__ void erf_parts(float u, float& erf_v, float& gauss) {
  const float x = fabsf(u) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);   // v_rcp_f32 (1 ulp): __frcp_rn is the ten-instruction IEEE division - a third of this function's VALU time, for a
                                                                    // correctly rounded t whose last bit is 1e-7 of an erf that is itself good to 1.5e-7 and feeds bf16 values
  gauss = __expf(-x * x);                                   // = exp(-u^2 / 2)
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  erf_v = copysignf(1.0f - poly * gauss, u);
}
__device__ __forceinline__ float gelu_f(float u) {
  float e, g;
  erf_parts(u, e, g);
  return 0.5f * u * (1.0f + e);
}
__device__ __forceinline__ float gelu_grad_f(float u) {
  float e, g;
  erf_parts(u, e, g);
  return 0.5f * (1.0f + e) + u * g * 0.39894228040143267794f;
}

// OCP e4m3 (gfx950's fp8: v_cvt_pk_fp8_f32), saturating: four floats -> four bytes
__device__ __forceinline__ unsigned pack_fp8x4(f32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = fminf(fmaxf(v[i], -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
  return (unsigned)w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Counter-based dropout RNG: one 64-bit hash (splitmix64 finaliser over a seed/group mix) decides FOUR consecutive elements,
// one 16-bit field each: element idx keeps iff field[idx & 3] of hash(seed, idx >> 2) >= thresh (= p * 2^16), and kept values
// are scaled by 1/(1-p).  Everything on the path handles four consecutive elements per lane, so a mask costs a quarter of a
// hash per element; it is recomputed in backward from (seed, idx) - nothing is stored.  oracle/ref_cpu.py::drop_mask restates it.
__device__ __forceinline__ uint64_t nv_hash64(uint64_t seed, uint64_t grp) {
  uint64_t x = (grp + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull ^ seed;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// Dropout configuration of one site.  thresh == 0 means "no dropout" (kernels skip the hash).
struct DropCfg {
  unsigned long long seed;
  unsigned thresh;      // p * 65536, in [0, 65536]
  float scale;
};
static inline DropCfg make_drop(unsigned long long seed, float p) {
  DropCfg d;
  d.seed = seed;
  if (p <= 0.f) { d.thresh = 0; d.scale = 1.f; }
  else if (p >= 1.f) { d.thresh = 0x10000u; d.scale = 0.f; }
  else { d.thresh = (unsigned)((double)p * 65536.0); d.scale = 1.0f / (1.0f - p); }
  return d;
}
// factors of elements idx4 .. idx4 + 3 (idx4 % 4 == 0)
__device__ __forceinline__ f32x4 drop_factor4(const DropCfg& d, unsigned long long idx4) {
  const uint64_t h = nv_hash64(d.seed, idx4 >> 2);
  const unsigned lo = (unsigned)h, hi = (unsigned)(h >> 32);
  f32x4 f;
  f[0] = ((lo & 0xFFFFu) >= d.thresh) ? d.scale : 0.f;
  f[1] = ((lo >> 16) >= d.thresh) ? d.scale : 0.f;
  f[2] = ((hi & 0xFFFFu) >= d.thresh) ? d.scale : 0.f;
  f[3] = ((hi >> 16) >= d.thresh) ? d.scale : 0.f;
  return f;
}
__device__ __forceinline__ float drop_factor(const DropCfg& d, unsigned long long idx) {
  const uint64_t h = nv_hash64(d.seed, idx >> 2);
  const unsigned field = (unsigned)(h >> (16 * (unsigned)(idx & 3))) & 0xFFFFu;
  return (field >= d.thresh) ? d.scale : 0.f;
}

// ---- nn.CrossEntropyLoss of ONE row (256 threads, every thread returns the row's loss term): shared by ce_loss_kernel (optim.hip) and the
// fused head step (norm.hip) so that both produce the same bits.  red: 4 floats of LDS.  dl (optional): d(loss)/d(logits) of the row.
// torch raises a device-side assert for a label outside [0, C); here nothing may read out of bounds or abort the stream: the term is
// NaN (visible at the first .item()) and the row's gradient is that of no target.
__device__ __forceinline__ float ce_row_term(const float* __restrict__ row, long tl, int C, float gscale_over_B, float* red, float* __restrict__ dl) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float mx = -INFINITY;
  for (int c = tid; c < C; c += 256) mx = fmaxf(mx, row[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int c = tid; c < C; c += 256) se += expf(row[c] - mx);
  se = wave_sum(se);
  if (lane == 0) red[wid] = se;
  __syncthreads();
  se = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  const bool t_ok = tl >= 0 && tl < (long)C;
  const int t = t_ok ? (int)tl : -1;
  const float term = t_ok ? (mx + logf(se)) - row[t] : __builtin_nanf("");
  if (dl)
    for (int c = tid; c < C; c += 256) dl[c] = (expf(row[c] - mx) / se - (c == t ? 1.f : 0.f)) * gscale_over_B;
  return term;
}

// ---- dynamic loss scale: indices into the device state block of nv_loss_scale_* (optim.hip; NV_LOSS_SCALE_FLOATS floats)
enum { LS_SCALE = 0,       // what the NEXT loss gradient is multiplied by
       LS_UNSCALE = 1,     // 1 / (scale of the gradients now in the arena): AdamW's extra grad factor
       LS_FOUND_INF = 2,   // a gradient of the current step was inf / NaN (nv_loss_scale_check)
       LS_SKIP = 3,        // the current optimizer update is skipped (set by nv_loss_scale_update)
       LS_TRACKER = 4,     // clean steps since the scale last changed
       LS_STEPS = 5,       // optimizer updates applied so far (AdamW's t; skipped steps do not count - as with GradScaler + torch.optim)
       LS_STEP_SIZE = 6, LS_BC2_SQRT = 7,      // lr / (1 - beta1^t), sqrt(1 - beta2^t) of the current update
       LS_GROWTH = 8, LS_BACKOFF = 9, LS_INTERVAL = 10, LS_SKIPPED = 11 };

// ---- AdamW (torch.optim.AdamW, Trainer.py:31,75), shared by the streaming kernel (optim.hip) and the weight-gradient GEMM epilogue
// that applies the update in place (gemm_common.h EPI_ADAMW): ONE definition, so both forms produce the same bits.
// Every derived constant is formed in double on the host, as torch does, then rounded once.
struct AdamArgs {
  float decay, one_minus_b1, beta2, one_minus_b2, eps, step_size, bc2_sqrt, grad_scale;
};
static inline AdamArgs make_adam_args(int step, double lr, double beta1, double beta2, double eps, double weight_decay, float grad_scale) {
  AdamArgs a;
  a.decay = (float)(1.0 - lr * weight_decay);
  a.one_minus_b1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.one_minus_b2 = (float)(1.0 - beta2);
  a.eps = (float)eps; a.grad_scale = grad_scale;
  a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
  a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  return a;
}
// Same operation order as torch's single-tensor AdamW (param.mul_; exp_avg.lerp_; exp_avg_sq.mul_.addcmul_;
// denom = sqrt(v)/bc2_sqrt + eps; param.addcdiv_) so results track the reference optimizer to fp32 rounding.  gv = grad * grad_scale.
__device__ __forceinline__ void adamw_update4(f32x4& pv, const f32x4 gv, f32x4& mv, f32x4& vv, const AdamArgs& a) {
  pv *= a.decay;
  mv += (gv - mv) * a.one_minus_b1;
  vv = vv * a.beta2 + (gv * a.one_minus_b2) * gv;
#pragma unroll
  for (int j = 0; j < 4; ++j) pv[j] -= a.step_size * (mv[j] / (sqrtf(vv[j]) / a.bc2_sqrt + a.eps));
}

// LDS-DMA issued by the waves that also read the tiles.  Two things the builtin form costs here:
//   * descriptor: the bases come out of 64-bit VALU address arithmetic on blockIdx, the compiler treats them as divergent and wraps
//     EVERY buffer_load ... lds in a waterfall loop (four v_readfirstlane + compares + saveexec + branch per DMA instruction) -
//     the words go through v_readfirstlane once instead;
//   * waits: the compiler knows a builtin DMA writes LDS and, unable to tell the ring stages apart, puts s_waitcnt vmcnt(0) in
//     front of the next transposed LDS read of the same wave - the transfer of tile k+1, just issued, is drained before tile k is
//     read, and every tile costs a full L2 / HBM round trip.  Issued from inline asm the DMA is invisible to that bookkeeping;
//     completion is tracked by hand (counted s_waitcnt vmcnt + workgroup barrier, as the kernels already did).
typedef int dma_desc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_desc uniform_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  return dma_desc{(int)__builtin_amdgcn_readfirstlane((unsigned)a), (int)(__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffff),
                  (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000};
}
// one wave-instruction: lane l copies BYTES (16 or 4) from base + voff(l) + soff to LDS at dst + l * BYTES (dst, soff wave-uniform).
// M0 (the LDS base of the transfer) is written inside the statement and not declared as clobbered - the compiler rejects reserved
// registers there; it keeps nothing in M0 itself on gfx950 unless the kernel also uses the builtin LDS-DMA / GWS / s_movrel forms:
// do not mix the builtin and this helper in one kernel.
template <int BYTES>
__device__ __forceinline__ void lds_dma(dma_desc rsrc, const char* dst, int voff, int soff) {
  typedef __attribute__((address_space(3))) const char lds_cchar;
  const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_cchar*)dst);
  const int so = __builtin_amdgcn_readfirstlane(soff);
  if constexpr (BYTES == 16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(a), "v"(voff), "s"(rsrc), "s"(so) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(a), "v"(voff), "s"(rsrc), "s"(so) : "memory");
}

// Lane id the compiler cannot hoist: per-lane DMA offsets computed from it INSIDE a tile loop are not kept in (or spilled from)
// registers across the loop - a spilled offset comes back through scratch_load + s_waitcnt vmcnt(0), which drains the DMA in flight.
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// Zero rows [0, rows) of a dense [rows, row_bytes] matrix EXCEPT the rows whose index is a multiple of `keep_every` (another part of the
// same launch writes those: the cls rows of a [B, n, d] tensor), as slice `part` of `nparts` workgroup-sized slices; row_bytes % 16 == 0.
// Folded into the kernel that writes the kept rows, so that "everything else is zero" costs no launch (and no hipMemsetAsync node).
__device__ __forceinline__ void zero_rows_except(char* base, long rows, long row_bytes, int keep_every, int part, int nparts) {
  const long chunks_per_row = row_bytes >> 4, total = rows * chunks_per_row;
  const long per = (total + nparts - 1) / nparts, lo = (long)part * per, hi = (lo + per < total) ? lo + per : total;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (long c = lo + threadIdx.x; c < hi; c += blockDim.x) {
    const long row = c / chunks_per_row;
    if (keep_every > 0 && row % keep_every == 0) continue;
    *reinterpret_cast<uint4*>(base + (c << 4)) = z;
  }
}

// XCD-aware bijective remap of a 1-D block id: blocks b, b+8, ... share an XCD (round-robin dispatch),
// so give each XCD a contiguous range of logical tiles (L2 locality of shared operand panels).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// A logical 2-D grid (nbx fast, nby slow) launched as ONE dimension of nbx * nby workgroups: xcd_remap gives every XCD a contiguous
// run of logical ids, so the nbx workgroups that share an operand (the row blocks of one attention head re-reading its K / V) sit
// on ONE XCD and its L2 fetches that operand once.  With a plain 2-D launch consecutive block ids - the row blocks of one head -
// are dealt round-robin over the eight XCDs and every L2 fetches every head (measured on the LDS-resident attention kernels at
// n = 513: forward 13.9 -> 11.5 us, dQ 14.9 -> 13.4, dK/dV 18.9 -> 16.4; profiles/r03_attn_xcd_grid_ab.log).
struct Grid2 { int bx, by; };
__device__ __forceinline__ Grid2 grid2d_xcd(int nbx) {
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  Grid2 g;
  g.by = lid / nbx;
  g.bx = lid - g.by * nbx;
  return g;
}
