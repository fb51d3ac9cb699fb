// Optimizer / loss / dtype plumbing kernels, gfx950.  All HBM-bound streaming kernels.
//
//   adamw_step     fused AdamW over a flat fp32 parameter arena (Trainer.py:31,75; torch.optim.AdamW
//                  defaults, decoupled weight decay on every parameter) that also refreshes the bf16
//                  shadow copy the MFMA kernels read - 16 B/param read + 14 B/param written per step.
//   cast_bf16_2d   fp32 -> bf16 with optional zero column padding (patch-embed weight at P % 8 != 0)
//   ce_loss        nn.CrossEntropyLoss() (mean) forward + d(loss)/d(logits) in one tiny kernel
#include "common.h"

// Streaming kernel: every byte is touched once per step, so loads and stores carry the non-temporal hint (5.8 vs 5.4 TB/s
// back to back on MI355X); the bf16 shadow is stored normally - the next forward pass reads it.
// The update itself is adamw_update4 (common.h).
// G16: the gradient is read from a bf16 buffer (the data-parallel all-reduce ran on bf16 messages; reading them here saves the
// cast back to fp32 - a full extra pass over the arena - and 2 of the 16 bytes this kernel reads per parameter)
#ifndef NV_ADAMW_UNROLL
#define NV_ADAMW_UNROLL 2          // float4 groups per thread per pass, all loads issued before the first use: 88.6 M parameters back to back
                                   // 471 us (1) -> 402-407 (2) -> 417-420 (4) on one box = 5.64 -> 6.6 TB/s
#endif
// dyn (may be null): the device block of a dynamic loss scale (nv_loss_scale_*, below) - the launch does nothing when the step is
// skipped, and takes the bias-correction constants and the un-scaling factor from it (adam_dyn)
__device__ __forceinline__ bool adam_dyn(AdamArgs& a, const float* __restrict__ dyn) {
  if (!dyn) return true;
  if (dyn[LS_SKIP] != 0.f) return false;
  a.step_size = dyn[LS_STEP_SIZE]; a.bc2_sqrt = dyn[LS_BC2_SQRT]; a.grad_scale *= dyn[LS_UNSCALE];
  return true;
}
template <bool G16, typename T>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const void* __restrict__ grad, float* __restrict__ m,
                                                    float* __restrict__ v, r16* __restrict__ p16, long n4, AdamArgs a, const float* __restrict__ dyn) {
  if (!adam_dyn(a, dyn)) return;
  // grid-stride: a full-size grid runs one iteration per thread; a capped grid (max_blocks) streams the range with a
  // fraction of the chip's wave slots so that it can run beside compute-bound kernels of another stream
  constexpr int U = NV_ADAMW_UNROLL;
  for (long i0 = (long)blockIdx.x * blockDim.x * U + threadIdx.x; i0 < n4; i0 += (long)gridDim.x * blockDim.x * U) {
    f32x4 pv[U], gv[U], mv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + (long)u * blockDim.x;
      if (i < n4) {
        pv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i);
        if constexpr (G16) {
          const r16x4 g4 = reinterpret_cast<const r16x4*>(grad)[i];
          gv[u] = dec4<T>(g4) * a.grad_scale;
        } else {
          gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grad) + i) * a.grad_scale;
        }
        mv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
        vv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + (long)u * blockDim.x;
      if (i < n4) {
        adamw_update4(pv[u], gv[u], mv[u], vv[u], a);
        __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(p) + i);
        __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(m) + i);
        __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v) + i);
        if (p16) reinterpret_cast<r16x4*>(p16)[i] = cvt4<T>(pv[u][0], pv[u][1], pv[u][2], pv[u][3]);
      }
    }
  }
}

// count must be a multiple of 4 (arena segments are padded); step >= 1.
extern "C" int nv_adamw_step(float* p, const void* grad, int grad_bf16, float* m, float* v, void* p16, long count, int step, double lr,
                             double beta1, double beta2, double eps, double weight_decay, float grad_scale, int max_blocks, void* stream) {
  return nv_adamw_step_scaled(p, grad, grad_bf16, m, v, p16, count, step, lr, beta1, beta2, eps, weight_decay, grad_scale, max_blocks, nullptr, stream);
}

extern "C" int nv_adamw_step_scaled(float* p, const void* grad, int grad_bf16, float* m, float* v, void* p16, long count, int step, double lr,
                                    double beta1, double beta2, double eps, double weight_decay, float grad_scale, int max_blocks,
                                    const float* scale_state, void* stream) {
  NV_CHECK_ARG(count > 0 && (count % 4) == 0 && step >= 1, "nv_adamw_step: count=%ld must be a positive multiple of 4", count);
  NV_CHECK_ARG(nv_aligned16(p) && (grad_bf16 ? ((uintptr_t)grad & 7) == 0 : nv_aligned16(grad)) && nv_aligned16(m) && nv_aligned16(v) &&
                   (!p16 || ((uintptr_t)p16 & 7) == 0),
               "nv_adamw_step: alignment");
  const AdamArgs a = make_adam_args(step, lr, beta1, beta2, eps, weight_decay, grad_scale);
  const long n4 = count / 4;
  long blocks = (n4 + 256L * NV_ADAMW_UNROLL - 1) / (256L * NV_ADAMW_UNROLL);
  if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
  NV_DISPATCH_OPERAND(T,
    if (grad_bf16) hipLaunchKernelGGL((adamw_kernel<true, T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, (r16*)p16, n4, a, scale_state);
    else hipLaunchKernelGGL((adamw_kernel<false, T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, (r16*)p16, n4, a, scale_state));
  NV_CHECK_LAUNCH("nv_adamw_step");
  return NV_OK;
}

// The same update over a list of element ranges in ONE launch: what is left of the arena when the Linear weights were updated by
// their weight-gradient GEMMs (EPI_ADAMW) - biases, LayerNorm affine parameters, embeddings, head: ~45 ranges of a few hundred
// elements and the patch-embedding weight.  A workgroup owns one 2048-element chunk of one range (chunk_end = running chunk counts).
constexpr int ADAM_MAX_RANGES = 64, ADAM_CHUNK = 2048;
struct AdamRanges { long begin[ADAM_MAX_RANGES]; long len[ADAM_MAX_RANGES]; int chunk_end[ADAM_MAX_RANGES]; int count; };
template <typename T>
__global__ __launch_bounds__(256) void adamw_ranges_kernel(float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ m,
                                                           float* __restrict__ v, r16* __restrict__ p16, const AdamRanges R, AdamArgs a) {
  const int chunks = R.chunk_end[ADAM_MAX_RANGES - 1];
  int r = 0;
  for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {      // (a capped grid walks the chunks: few CUs for an HBM-bound pass)
    while (r + 1 < R.count && chunk >= R.chunk_end[r]) ++r;                // workgroup-uniform
    const long c0 = (long)(chunk - (r ? R.chunk_end[r - 1] : 0)) * ADAM_CHUNK;
    const long base = R.begin[r] + c0, n4 = min((long)ADAM_CHUNK, R.len[r] - c0) / 4;
    constexpr int U = ADAM_CHUNK / (256 * 4);
    f32x4 pv[U], gv[U], mv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = threadIdx.x + 256 * u;
      if (i < n4) {
        pv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p + base) + i);
        gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grad + base) + i) * a.grad_scale;
        mv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m + base) + i);
        vv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v + base) + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = threadIdx.x + 256 * u;
      if (i < n4) {
        adamw_update4(pv[u], gv[u], mv[u], vv[u], a);
        __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(p + base) + i);
        __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(m + base) + i);
        __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v + base) + i);
        reinterpret_cast<r16x4*>(p16 + base)[i] = cvt4<T>(pv[u][0], pv[u][1], pv[u][2], pv[u][3]);
      }
    }
  }
}

int g_adamw_ranges_cap = 0;      // tuning aid (nv_gemm_set_tile(13, n)): at most n workgroups per nv_adamw_ranges launch, walking the chunks (0 = one per chunk).
                                 // ViT3D-base batch 4, per-layer launches of fuse_update = 3: 16 -> 502 volumes/s, 32 -> 754, 64 -> 1000, 128 -> 1110,
                                 // 256 ... 2048 and uncapped (3456) 1140 ... 1170: the update must not become the auxiliary stream's critical path
extern "C" int nv_adamw_ranges(const nv_adamw_arena* opt, const long* begins, const long* lens, int count, void* stream) {
  const int max_blocks = g_adamw_ranges_cap;
  NV_CHECK_ARG(opt && opt->struct_size == (int)sizeof(nv_adamw_arena), "nv_adamw_ranges: nv_adamw_arena.struct_size = %d, this library expects %d (ABI revision %d)",
               opt ? opt->struct_size : -1, (int)sizeof(nv_adamw_arena), NV_ABI_VERSION);
  NV_CHECK_ARG(opt->params && opt->grads && opt->adam_m && opt->adam_v && opt->params16 && opt->step >= 1 && count >= 0 && (count == 0 || (begins && lens)),
               "nv_adamw_ranges: null arena / range list or step < 1");
  NV_CHECK_ARG(nv_aligned16(opt->params) && nv_aligned16(opt->grads) && nv_aligned16(opt->adam_m) && nv_aligned16(opt->adam_v) && nv_aligned16(opt->params16),
               "nv_adamw_ranges: arenas must be 16-byte aligned");
  const AdamArgs a = make_adam_args(opt->step, opt->lr, opt->beta1, opt->beta2, opt->eps, opt->weight_decay, opt->grad_scale);
  for (int first = 0; first < count; first += ADAM_MAX_RANGES) {
    AdamRanges R;
    int n = 0, chunks = 0;
    for (int i = first; i < count && i < first + ADAM_MAX_RANGES; ++i) {
      NV_CHECK_ARG(begins[i] >= 0 && lens[i] >= 0 && (begins[i] % 4) == 0 && (lens[i] % 4) == 0, "nv_adamw_ranges: range %d = [%ld, +%ld) must be non-negative multiples of 4", i, begins[i], lens[i]);
      if (lens[i] == 0) continue;
      R.begin[n] = begins[i]; R.len[n] = lens[i];
      chunks += (int)((lens[i] + ADAM_CHUNK - 1) / ADAM_CHUNK);
      R.chunk_end[n++] = chunks;
    }
    if (!n) continue;
    R.count = n;
    for (int i = n; i < ADAM_MAX_RANGES; ++i) { R.begin[i] = 0; R.len[i] = 0; R.chunk_end[i] = chunks; }
    const int blocks = (max_blocks > 0 && max_blocks < chunks) ? max_blocks : chunks;
    NV_DISPATCH_OPERAND(T, hipLaunchKernelGGL(adamw_ranges_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, opt->params, opt->grads, opt->adam_m,
                                              opt->adam_v, (r16*)opt->params16, R, a));
    NV_CHECK_LAUNCH("nv_adamw_ranges");
  }
  return NV_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void cast_bf16_2d_kernel(const float* __restrict__ src, long lds_, int rows, int cols, r16* __restrict__ dst,
                                                           long ldd) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per 4 output columns
  const long per_row = ldd / 4;
  if (idx >= (long)rows * per_row) return;
  const long r = idx / per_row;
  const int c = (int)(idx - r * per_row) * 4;
  const float* s = src + r * lds_ + c;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (c + j < cols) ? s[j] : 0.f;
  *reinterpret_cast<r16x4*>(dst + r * ldd + c) = cvt4<T>(v[0], v[1], v[2], v[3]);
}

// dst[r, c] = bf16(src[r, c]) for c < cols, 0 for cols <= c < ld_dst.  ld_dst % 4 == 0.
extern "C" int nv_cast_bf16_2d(const float* src, long ld_src, int rows, int cols, void* dst, long ld_dst, void* stream) {
  NV_CHECK_ARG(rows > 0 && cols > 0 && ld_dst >= cols && (ld_dst % 4) == 0 && ld_src >= cols && ((uintptr_t)dst & 7) == 0,
               "nv_cast_bf16_2d: bad dims");
  const long tot = (long)rows * (ld_dst / 4);
  NV_DISPATCH_OPERAND(T, hipLaunchKernelGGL(cast_bf16_2d_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, rows, cols,
                                            (r16*)dst, ld_dst));
  NV_CHECK_LAUNCH("nv_cast_bf16_2d");
  return NV_OK;
}

// dst[b .. b + len) = bf16(src[b .. b + len)) for up to CAST_MAX_RANGES element ranges per launch (arguments by value): the small
// parameter gradients (biases, LayerNorm affine, embeddings) of a data-parallel bucket whose Linear weight gradients were already
// written in bf16 by their GEMMs.  One workgroup walks one range at a time (ranges are a few hundred to a few hundred thousand
// elements); ranges start 8-element aligned in the arena, so the body moves 16 bytes in / 8 bytes out per lane and the tail is scalar.
constexpr int CAST_MAX_RANGES = 48;
struct CastRanges { long begin[CAST_MAX_RANGES]; long len[CAST_MAX_RANGES]; int count; };
template <typename T>
__global__ __launch_bounds__(256) void cast_ranges_kernel(const float* __restrict__ src, r16* __restrict__ dst, const CastRanges R) {
  for (int r = blockIdx.y; r < R.count; r += gridDim.y) {
    const long b = R.begin[r], n = R.len[r];
    const bool vec = ((b & 3) == 0);
    const long n4 = vec ? (n >> 2) : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + b + 4 * i);
      *reinterpret_cast<r16x4*>(dst + b + 4 * i) = cvt4<T>(v[0], v[1], v[2], v[3]);
    }
    for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[b + i] = cvt1<T>(src[b + i]);
  }
}

extern "C" int nv_cast_ranges_bf16(const float* src, void* dst, const long* begins, const long* lens, int count, void* stream) {
  NV_CHECK_ARG(src && dst && (count == 0 || (begins && lens)) && count >= 0 && nv_aligned16(src) && ((uintptr_t)dst & 7) == 0, "nv_cast_ranges_bf16: bad arguments");
  for (int first = 0; first < count; first += CAST_MAX_RANGES) {
    CastRanges R;
    R.count = (count - first < CAST_MAX_RANGES) ? count - first : CAST_MAX_RANGES;
    long longest = 0;
    for (int i = 0; i < R.count; ++i) {
      NV_CHECK_ARG(begins[first + i] >= 0 && lens[first + i] >= 0, "nv_cast_ranges_bf16: negative range");
      R.begin[i] = begins[first + i]; R.len[i] = lens[first + i];
      if (R.len[i] > longest) longest = R.len[i];
    }
    if (longest == 0) continue;
    long gx = (longest / 4 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    NV_DISPATCH_OPERAND(T, hipLaunchKernelGGL(cast_ranges_kernel<T>, dim3((unsigned)gx, (unsigned)R.count), dim3(256), 0, (hipStream_t)stream, src, (r16*)dst, R));
    NV_CHECK_LAUNCH("nv_cast_ranges_bf16");
  }
  return NV_OK;
}

// out16[m, n] = bf16(x[m, n] * mask(m, n)), out32 likewise (either may be null): the nn.Dropout mask of one site (same
// counter-based mask the GEMM epilogues apply, element index m*N + n) - used by the standalone Attention / FeedForward
// modules, whose last Dropout (vit_3d.py:23,45) has no residual epilogue to ride in.
template <typename T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, long ldx, int M, int N, DropCfg dc, r16* __restrict__ out16,
                                                            long ld16, float* __restrict__ out32, long ld32) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per 4 columns
  const long per_row = N / 4;
  if (idx >= (long)M * per_row) return;
  const long m = idx / per_row;
  const int n = (int)(idx - m * per_row) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ldx + n);
  if (dc.thresh) v = v * drop_factor4(dc, (unsigned long long)m * N + n);
  if (out16) *reinterpret_cast<r16x4*>(out16 + m * ld16 + n) = cvt4<T>(v[0], v[1], v[2], v[3]);
  if (out32) *reinterpret_cast<f32x4*>(out32 + m * ld32 + n) = v;
}

extern "C" int nv_dropout_apply(const float* x, long ldx, int M, int N, unsigned long drop_seed, float drop_p, void* out16, long ld16,
                                float* out32, long ld32, void* stream) {
  NV_CHECK_ARG(x && M > 0 && N > 0 && (N % 4) == 0 && (ldx % 4) == 0 && nv_aligned16(x), "nv_dropout_apply: N and ldx must be multiples of 4");
  NV_CHECK_ARG((!out16 || ((ld16 % 4) == 0 && ((uintptr_t)out16 & 7) == 0)) && (!out32 || ((ld32 % 4) == 0 && nv_aligned16(out32))), "nv_dropout_apply: output alignment");
  const long tot = (long)M * (N / 4);
  NV_DISPATCH_OPERAND(T, hipLaunchKernelGGL(dropout_apply_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, M, N,
                                            make_drop(drop_seed, drop_p), (r16*)out16, ld16, out32, ld32));
  NV_CHECK_LAUNCH("nv_dropout_apply");
  return NV_OK;
}

// loss = mean_b( logsumexp(logits[b]) - logits[b, target[b]] );  dlogits = (softmax - onehot) * grad_scale / B
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits, const long* __restrict__ target, int B, int C,
                                                      float grad_scale, const float* __restrict__ scale_state, float* __restrict__ loss,
                                                      float* __restrict__ dlogits) {
  __shared__ float red[4];
  if (scale_state) grad_scale *= scale_state[LS_SCALE];      // dynamic loss scale: the gradients carry it, the reported loss does not
  float total = 0.f;
  for (int b = 0; b < B; ++b)
    total += ce_row_term(logits + (long)b * C, target[b], C, grad_scale / (float)B, red, dlogits ? dlogits + (long)b * C : nullptr);
  if (threadIdx.x == 0) loss[0] = total / (float)B;
}

extern "C" int nv_ce_loss(const float* logits, const long* target, int B, int C, float grad_scale, float* loss, float* dlogits, void* stream) {
  return nv_ce_loss_scaled(logits, target, B, C, grad_scale, nullptr, loss, dlogits, stream);
}

extern "C" int nv_ce_loss_scaled(const float* logits, const long* target, int B, int C, float grad_scale, const float* scale_state, float* loss,
                                 float* dlogits, void* stream) {
  NV_CHECK_ARG(B > 0 && C > 0 && logits && target && loss, "nv_ce_loss: bad args");
  hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, B, C, grad_scale, scale_state, loss, dlogits);
  NV_CHECK_LAUNCH("nv_ce_loss");
  return NV_OK;
}

// dst[r, c] (=|+=) src[r, c] for c < cols: strips the column padding of a [rows, ld_src] fp32 scratch matrix
// (weight gradient of the patch embedding when patch_dim is not a multiple of 8, e.g. the reference default p = 9).
__global__ __launch_bounds__(256) void copy_2d_f32_kernel(const float* __restrict__ src, long ld_src, int rows, int cols, float* __restrict__ dst,
                                                          long ld_dst, int accumulate) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cols) return;
  const long r = idx / cols, c = idx - r * cols;
  const float v = src[r * ld_src + c];
  float* o = dst + r * ld_dst + c;
  *o = accumulate ? *o + v : v;
}

extern "C" int nv_copy_2d_f32(const float* src, long ld_src, int rows, int cols, float* dst, long ld_dst, int accumulate, void* stream) {
  NV_CHECK_ARG(rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols && src && dst, "nv_copy_2d_f32: bad dims");
  const long tot = (long)rows * cols;
  hipLaunchKernelGGL(copy_2d_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, rows, cols, dst,
                     ld_dst, accumulate);
  NV_CHECK_LAUNCH("nv_copy_2d_f32");
  return NV_OK;
}

// ---- LayerNorm folded into the Linear behind it (inference forwards): Wg16[n, k] = T(W[n, k] gamma[k]); colsum[n] = sum_k of the ROUNDED Wg16[n, k] (what the MFMA
// will really multiply the row mean with); fbias[n] = sum_k W[n, k] beta[k] (+ bias[n]).  One wave per output row.
template <typename T>
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(const float* __restrict__ W, long ldw, int N, int K, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias, r16* __restrict__ Wg, long ldg,
                                                             float* __restrict__ colsum, float* __restrict__ fbias) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float cs = 0.f, fb = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)n * ldw + k);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k), b = *reinterpret_cast<const f32x4*>(beta + k);
    const r16x4 q = cvt4<T>(w[0] * g[0], w[1] * g[1], w[2] * g[2], w[3] * g[3]);
    *reinterpret_cast<r16x4*>(Wg + (long)n * ldg + k) = q;
    const f32x4 qf = dec4<T>(q);
    cs += (qf[0] + qf[1]) + (qf[2] + qf[3]);
    fb += (w[0] * b[0] + w[1] * b[1]) + (w[2] * b[2] + w[3] * b[3]);
  }
  cs = wave_sum(cs); fb = wave_sum(fb);
  if (lane == 0) { colsum[n] = cs; fbias[n] = fb + (bias ? bias[n] : 0.f); }
}
extern "C" int nv_ln_fold_weight(const float* W, long ldw, int N, int K, const float* gamma, const float* beta, const float* bias, void* Wg16, long ldg,
                                 float* colsum, float* fbias, void* stream) {
  NV_CHECK_ARG(W && gamma && beta && Wg16 && colsum && fbias && N > 0 && K > 0 && (K % 4) == 0 && (ldw % 4) == 0 && (ldg % 4) == 0 && ldw >= K && ldg >= K,
               "nv_ln_fold_weight: K, ldw, ldg must be multiples of 4");
  NV_CHECK_ARG(nv_aligned16(W) && nv_aligned16(gamma) && nv_aligned16(beta) && ((uintptr_t)Wg16 & 7) == 0, "nv_ln_fold_weight: alignment");
  NV_DISPATCH_OPERAND(T, hipLaunchKernelGGL(ln_fold_weight_kernel<T>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, gamma, beta, bias, (r16*)Wg16, ldg,
                                            colsum, fbias));
  NV_CHECK_LAUNCH("nv_ln_fold_weight");
  return NV_OK;
}

// ---- dynamic loss scale: torch.amp.GradScaler (src/Trainer.py:29,74-76: scaler.scale(loss).backward(); scaler.step(optimizer);
// scaler.update()) for training on fp16 operands, kept ENTIRELY on the device - the reference's scaler.step() reads found_inf back
// to the host every step.  State block: NV_LOSS_SCALE_FLOATS floats (indices LS_*, common.h).  Per optimizer step:
//   forward / loss:       dlogits *= state[LS_SCALE]                               (nv_ce_loss_scaled / nv_head_step_scaled)
//   nv_loss_scale_check:  state[LS_FOUND_INF] = 1 if any gradient is inf / NaN      (every gradient, as GradScaler.unscale_ does)
//   nv_loss_scale_update: found_inf -> skip this update, scale *= backoff, growth tracker = 0; otherwise applied steps t += 1,
//                         AdamW's bias-correction constants for step t (double arithmetic, as torch forms them on the host),
//                         tracker += 1 and scale *= growth every `growth_interval` clean steps; LS_UNSCALE = 1 / (the scale the
//                         gradients in the arena carry); found_inf cleared
//   nv_adamw_step_scaled: returns at once when LS_SKIP is set; reads LS_UNSCALE, LS_STEP_SIZE, LS_BC2_SQRT
__global__ void loss_scale_init_kernel(float* st, float init_scale, float growth, float backoff, int interval, int start_step) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < NV_LOSS_SCALE_FLOATS; ++i) st[i] = 0.f;
  st[LS_SCALE] = init_scale; st[LS_UNSCALE] = 1.f / init_scale; st[LS_GROWTH] = growth; st[LS_BACKOFF] = backoff;
  st[LS_INTERVAL] = (float)interval; st[LS_STEPS] = (float)start_step;
}
extern "C" int nv_loss_scale_init(float* state, float init_scale, float growth_factor, float backoff_factor, int growth_interval, int start_step, void* stream) {
  NV_CHECK_ARG(state && init_scale > 0.f && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1 && start_step >= 0,
               "nv_loss_scale_init: scale > 0, growth >= 1, 0 < backoff <= 1, interval >= 1, start_step >= 0");
  hipLaunchKernelGGL(loss_scale_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, init_scale, growth_factor, backoff_factor, growth_interval, start_step);
  NV_CHECK_LAUNCH("nv_loss_scale_init");
  return NV_OK;
}

// any |x| that is not < inf (inf or NaN): exponent bits all ones.  HBM-bound read of the arena (354 MB for ViT3D-base: ~60 us).
// The words are read as integers (no float value is ever formed: nothing for value-based reasoning to fold).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned nonfinite_bits(unsigned w) { return ((w & 0x7f800000u) == 0x7f800000u) ? 1u : 0u; }
__global__ __launch_bounds__(256) void grad_check_kernel(const unsigned* __restrict__ g, long count, float* __restrict__ st) {
  unsigned bad = 0;
  const long nv = (((unsigned long)g & 15) == 0) ? count / 4 : 0;      // 16-byte pieces, then the scalar tail
  const u32x4* gv = reinterpret_cast<const u32x4*>(g);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const u32x4 v = __builtin_nontemporal_load(gv + i);
    bad |= nonfinite_bits(v[0]) | nonfinite_bits(v[1]) | nonfinite_bits(v[2]) | nonfinite_bits(v[3]);
  }
  for (long i = 4 * nv + (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) bad |= nonfinite_bits(g[i]);
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && (threadIdx.x & 63) == 0) st[LS_FOUND_INF] = 1.f;      // racing stores of the same value
}
extern "C" int nv_loss_scale_check(const float* grads, long count, float* state, void* stream) {
  NV_CHECK_ARG(grads && state && count > 0, "nv_loss_scale_check: null pointer / empty range");
  long blocks = (count / 4 + 256 * 8 - 1) / (256 * 8);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(grad_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned*>(grads), count, state);
  NV_CHECK_LAUNCH("nv_loss_scale_check");
  return NV_OK;
}

__global__ void loss_scale_update_kernel(float* st, double lr, double beta1, double beta2) {
  if (threadIdx.x != 0) return;
  const float scale = st[LS_SCALE];
  st[LS_UNSCALE] = 1.f / scale;                 // of the gradients that are in the arena now
  if (st[LS_FOUND_INF] != 0.f) {
    st[LS_SKIP] = 1.f;
    st[LS_SCALE] = scale * st[LS_BACKOFF];
    st[LS_TRACKER] = 0.f;
    st[LS_SKIPPED] += 1.f;
  } else {
    st[LS_SKIP] = 0.f;
    const double t = (double)st[LS_STEPS] + 1.0;
    st[LS_STEPS] = (float)t;
    st[LS_STEP_SIZE] = (float)(lr / (1.0 - pow(beta1, t)));
    st[LS_BC2_SQRT] = (float)sqrt(1.0 - pow(beta2, t));
    const float tr = st[LS_TRACKER] + 1.f;
    if (tr >= st[LS_INTERVAL]) { st[LS_SCALE] = scale * st[LS_GROWTH]; st[LS_TRACKER] = 0.f; }
    else st[LS_TRACKER] = tr;
  }
  st[LS_FOUND_INF] = 0.f;
}
extern "C" int nv_loss_scale_update(float* state, double lr, double beta1, double beta2, void* stream) {
  NV_CHECK_ARG(state, "nv_loss_scale_update: null state");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr, beta1, beta2);
  NV_CHECK_LAUNCH("nv_loss_scale_update");
  return NV_OK;
}
