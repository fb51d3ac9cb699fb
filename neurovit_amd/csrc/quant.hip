// fp8 (OCP e4m3) quantisation for the fp8 inference path of ViT3D-large (BASELINE.json configs[4]): weights per output row,
// activations per tensor with a calibrated static scale, straight out of the LayerNorm that produces them.
//   w8[n, k] = sat(W[n, k] * sw[n]),  sw[n] = 448 / max_k |W[n, k]|;   x8 = sat(x * sa)
//   y[m, n]  = (sum_k x8[m, k] w8[n, k]) * colscale[n],  colscale[n] = 1 / (sa * sw[n])   (applied in the GEMM epilogue)
#include "gemm_common.h"

namespace {
constexpr int QW = 4;   // waves per workgroup, one row per wave

__global__ __launch_bounds__(64 * QW) void quant_rows_f8_kernel(const float* __restrict__ W, long ldw, int rows, int cols, unsigned char* __restrict__ out,
                                                                long ldo, float act_scale, float* __restrict__ colscale) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * QW + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* w = W + (long)row * ldw;
  float amax = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(w + c);
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const float sw = amax > 0.f ? 448.f / amax : 1.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(w + c) * sw;
    *reinterpret_cast<unsigned*>(out + (long)row * ldo + c) = pack_fp8x4(v);
  }
  if (lane == 0) colscale[row] = 1.0f / (act_scale * sw);
}

// LayerNorm(d) -> fp8 with a static scale (inference: no statistics saved)
// y16 / mean_out / rstd_out (all optional): the bf16 output and the statistics nv_ln_fwd would have produced for the same row - bit for
// bit (same summation order, same formula) - for a TRAINING forward on fp8 operands, whose bf16 backward pass reads them
__global__ __launch_bounds__(64 * QW) void ln_fwd_f8_kernel(const float* __restrict__ x, long ldx, int M, int d, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, float out_scale, unsigned char* __restrict__ y, long ldy,
                                                            r16* __restrict__ y16, long ldy16, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * QW + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long)row * ldx;
  constexpr int NV = 8;                        // d <= 2048
  f32x4 xv[NV];
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (lane + 64 * v) * 4;
    xv[v] = (c < d) ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (xv[v][0] + xv[v][1]) + (xv[v][2] + xv[v][3]);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (lane + 64 * v) * 4;
    if (c < d) {
      const f32x4 t = xv[v] - mean;
      q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (lane + 64 * v) * 4;
    if (c < d) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
      const f32x4 o = (xv[v] - mean) * rstd * gm + bt;
      if (y16) *reinterpret_cast<r16x4*>(y16 + (long)row * ldy16 + c) = cvt4<bf16_t>(o[0], o[1], o[2], o[3]);      // (the fp8 path sits beside bf16 operands: nv_set_operand_format)
      *reinterpret_cast<unsigned*>(y + (long)row * ldy + c) = pack_fp8x4(o * out_scale);
    }
  }
  if (lane == 0 && mean_out) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}
}  // namespace

extern "C" int nv_quant_rows_f8(const float* W, long ldw, int rows, int cols, void* out8, long ld8, float act_scale, float* colscale, void* stream) {
  NV_CHECK_ARG(W && out8 && colscale && rows > 0 && cols > 0 && (cols % 4) == 0 && (ldw % 4) == 0 && (ld8 % 4) == 0 && ldw >= cols && ld8 >= cols && act_scale > 0.f,
               "nv_quant_rows_f8: cols, ldw, ld8 must be multiples of 4; act_scale > 0");
  NV_CHECK_ARG(nv_aligned16(W) && ((uintptr_t)out8 & 3) == 0, "nv_quant_rows_f8: alignment");
  hipLaunchKernelGGL(quant_rows_f8_kernel, dim3((rows + QW - 1) / QW), dim3(64 * QW), 0, (hipStream_t)stream, W, ldw, rows, cols, (unsigned char*)out8, ld8,
                     act_scale, colscale);
  NV_CHECK_LAUNCH("nv_quant_rows_f8");
  return NV_OK;
}

extern "C" int nv_ln_fwd_f8(const float* x, long ldx, int M, int d, const float* gamma, const float* beta, float eps, float out_scale, void* y8, long ldy,
                            void* stream) {
  NV_CHECK_ARG(x && gamma && beta && y8 && M > 0 && d > 0 && (d % 4) == 0 && d <= 2048 && (ldx % 4) == 0 && (ldy % 4) == 0, "nv_ln_fwd_f8: d must be a multiple of 4 and <= 2048");
  NV_CHECK_ARG(nv_aligned16(x) && nv_aligned16(gamma) && nv_aligned16(beta) && ((uintptr_t)y8 & 3) == 0, "nv_ln_fwd_f8: alignment");
  hipLaunchKernelGGL(ln_fwd_f8_kernel, dim3((M + QW - 1) / QW), dim3(64 * QW), 0, (hipStream_t)stream, x, ldx, M, d, gamma, beta, eps, out_scale,
                     (unsigned char*)y8, ldy, (r16*)nullptr, 0L, (float*)nullptr, (float*)nullptr);
  NV_CHECK_LAUNCH("nv_ln_fwd_f8");
  return NV_OK;
}

// LayerNorm(d) of a training forward on fp8 operands: e4m3 (static scale) for the fp8 GEMM that follows, and the bf16 output + row
// statistics of nv_ln_fwd for the bf16 backward pass, in one pass over x
extern "C" int nv_ln_fwd_f8_train(const float* x, long ldx, int M, int d, const float* gamma, const float* beta, float eps, float out_scale, void* y8, long ldy8,
                                  void* y16, long ldy16, float* mean, float* rstd, void* stream) {
  NV_CHECK_ARG(nv_operand_format() == NV_OPERAND_BF16, "nv_ln_fwd_f8_train: the fp8 path is built beside bf16 operands (nv_set_operand_format(NV_OPERAND_BF16))");
  NV_CHECK_ARG(x && gamma && beta && y8 && y16 && mean && rstd && M > 0 && d > 0 && (d % 4) == 0 && d <= 2048 && (ldx % 4) == 0 && (ldy8 % 4) == 0 && (ldy16 % 4) == 0,
               "nv_ln_fwd_f8_train: d must be a multiple of 4 and <= 2048");
  NV_CHECK_ARG(nv_aligned16(x) && nv_aligned16(gamma) && nv_aligned16(beta) && ((uintptr_t)y8 & 3) == 0 && ((uintptr_t)y16 & 7) == 0, "nv_ln_fwd_f8_train: alignment");
  hipLaunchKernelGGL(ln_fwd_f8_kernel, dim3((M + QW - 1) / QW), dim3(64 * QW), 0, (hipStream_t)stream, x, ldx, M, d, gamma, beta, eps, out_scale,
                     (unsigned char*)y8, ldy8, (r16*)y16, ldy16, mean, rstd);
  NV_CHECK_LAUNCH("nv_ln_fwd_f8_train");
  return NV_OK;
}
