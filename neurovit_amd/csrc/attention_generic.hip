// Attention for head dims other than 64 (vit_3d.py:29 takes any dim_head; NeuroEncoder.py never passes one, so 64 is the hot path and
// owns the MFMA kernels of attention.hip).  One wave per query row (forward, dQ) or per key row (dK / dV), lanes over the other index,
// fp32 FMAs: O(n^2 dh) scalar work - about 1 / 20 of the MFMA kernels' rate, for correctness of the whole ViT surface, not for speed.
// Same cast points as the fast kernels and oracle/ref_cpu.py::_AttnEmu (P and dS rounded to bf16 where they enter a product, bf16
// outputs, dropout on P.V and dP with the counter-based mask of common.h); the row maximum is the exact one (two passes), not the
// running maximum of 64-key tiles.
#include "common.h"

namespace {

constexpr int GW = 4;   // waves (rows) per workgroup

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// dot product of an fp32 row in LDS with a bf16 row in memory (dh % 8 == 0)
template <typename T>
__device__ __forceinline__ float dot_row(const float* __restrict__ a, const r16* __restrict__ x, int dh) {
  float s = 0.f;
  for (int c = 0; c < dh; c += 8) {
    const r16x8 v = *reinterpret_cast<const r16x8*>(x + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) s = __builtin_fmaf(a[c + e], dec1<T>(v[e]), s);
  }
  return s;
}

// acc[0 .. cnt) += w * x[d0 .. d0 + cnt)   (cnt % 8 == 0, compile-time bound ACC on the register array)
template <int ACC, typename T>
__device__ __forceinline__ void axpy_row(float (&acc)[ACC], float w, const r16* __restrict__ x, int cnt) {
#pragma unroll
  for (int c = 0; c < ACC; c += 8) {
    if (c < cnt) {
      const r16x8 v = *reinterpret_cast<const r16x8*>(x + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[c + e] = __builtin_fmaf(w, dec1<T>(v[e]), acc[c + e]);
    }
  }
}

// wave totals of acc[0 .. cnt), written as bf16(total * mul) to dst[0 .. cnt): lane (d & 63) keeps element d
template <int ACC, typename T>
__device__ __forceinline__ void reduce_store(float (&acc)[ACC], int cnt, float mul, r16* __restrict__ dst, int lane) {
#pragma unroll
  for (int blk = 0; blk < ACC; blk += 64) {
    float mine = 0.f;
#pragma unroll
    for (int d = blk; d < blk + 64 && d < ACC; ++d) {
      if (d < cnt) {
        const float t = wave_sum(acc[d]);
        if (lane == (d & 63)) mine = t;
      }
    }
    if (blk + lane < cnt) dst[blk + lane] = cvt1<T>(mine * mul);
  }
}

template <int DHM, typename T>
__global__ __launch_bounds__(64 * GW) void attn_gen_fwd_kernel(const r16* __restrict__ qkv, long ld, int n, int heads, int dh, float scale_log2e,
                                                               r16* __restrict__ out, long ldo, float* __restrict__ lse, DropCfg drop) {
  __shared__ float sq[GW][DHM];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = blockIdx.x * GW + wid, bh = blockIdx.y, b = bh / heads, h = bh - b * heads, inner = heads * dh;
  const r16* base = qkv + (long)b * n * ld + h * dh;
  if (i < n)
    for (int d = lane; d < dh; d += 64) sq[wid][d] = dec1<T>(base[(long)i * ld + d]);
  __syncthreads();
  if (i >= n) return;
  const float* q = sq[wid];
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, dot_row<T>(q, base + (long)j * ld + inner, dh));
  mx = wave_max_f(mx) * scale_log2e;               // scale > 0: the maximum commutes with it
  float l = 0.f, o[DHM];
#pragma unroll
  for (int d = 0; d < DHM; ++d) o[d] = 0.f;
  const unsigned long long row = ((unsigned long long)bh * n + i) * ((n + 3) & ~3);
  for (int j = lane; j < n; j += 64) {
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(dot_row<T>(q, base + (long)j * ld + inner, dh), scale_log2e, -mx));
    l += p;
    const float keep = drop.thresh ? drop_factor(drop, row + j) : 1.f;     // dropout hits P.V, not the normaliser (vit_3d.py:56)
    axpy_row<DHM, T>(o, dec1<T>(cvt1<T>(p * keep)), base + (long)j * ld + 2 * inner, dh);
  }
  l = wave_sum(l);
  reduce_store<DHM, T>(o, dh, 1.0f / l, out + ((long)b * n + i) * ldo + h * dh, lane);
  if (lane == 0 && lse) lse[(long)bh * n + i] = (mx + log2f(l)) * 0.69314718055994530942f;
}

template <int DHM, typename T>
__global__ __launch_bounds__(64 * GW) void attn_gen_dq_kernel(const r16* __restrict__ qkv, long ld, const r16* __restrict__ out,
                                                              const r16* __restrict__ dout, long ldo, const float* __restrict__ lse, int n,
                                                              int heads, int dh, float scale, float* __restrict__ delta,
                                                              r16* __restrict__ dqkv, long ldd, DropCfg drop) {
  __shared__ float sq[GW][DHM], sdo[GW][DHM];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = blockIdx.x * GW + wid, bh = blockIdx.y, b = bh / heads, h = bh - b * heads, inner = heads * dh;
  const r16* base = qkv + (long)b * n * ld + h * dh;
  float dl = 0.f;
  if (i < n)
    for (int d = lane; d < dh; d += 64) {
      const float g = dec1<T>(dout[((long)b * n + i) * ldo + h * dh + d]);
      sq[wid][d] = dec1<T>(base[(long)i * ld + d]);
      sdo[wid][d] = g;
      dl = __builtin_fmaf(g, dec1<T>(out[((long)b * n + i) * ldo + h * dh + d]), dl);
    }
  __syncthreads();
  if (i >= n) return;
  dl = wave_sum(dl);                                // delta = rowsum(dO * O)
  if (lane == 0) delta[(long)bh * n + i] = dl;
  const float lse2 = lse[(long)bh * n + i] * 1.44269504088896340736f, scale_log2e = scale * 1.44269504088896340736f;
  float dq[DHM];
#pragma unroll
  for (int d = 0; d < DHM; ++d) dq[d] = 0.f;
  const unsigned long long row = ((unsigned long long)bh * n + i) * ((n + 3) & ~3);
  for (int j = lane; j < n; j += 64) {
    const r16* kj = base + (long)j * ld + inner;
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(dot_row<T>(sq[wid], kj, dh), scale_log2e, -lse2));
    const float keep = drop.thresh ? drop_factor(drop, row + j) : 1.f;
    const float dp = dot_row<T>(sdo[wid], kj + inner, dh) * keep;
    axpy_row<DHM, T>(dq, dec1<T>(cvt1<T>(p * (dp - dl))), kj, dh);
  }
  reduce_store<DHM, T>(dq, dh, scale, dqkv + ((long)b * n + i) * ldd + h * dh, lane);
}

// one wave per key row j and per slice [d0, d0 + ACC) of the head dim (accumulators of dK and dV: 2 * ACC registers)
template <int DHM, int ACC, typename T>
__global__ __launch_bounds__(64 * GW) void attn_gen_dkv_kernel(const r16* __restrict__ qkv, long ld, const r16* __restrict__ dout, long ldo,
                                                               const float* __restrict__ lse, const float* __restrict__ delta, int n, int heads,
                                                               int dh, float scale, r16* __restrict__ dqkv, long ldd, DropCfg drop) {
  __shared__ float sk[GW][DHM], sv[GW][DHM];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int j = blockIdx.x * GW + wid, bh = blockIdx.y, b = bh / heads, h = bh - b * heads, inner = heads * dh;
  const int d0 = blockIdx.z * ACC, cnt = (dh - d0 < ACC) ? dh - d0 : ACC;
  const r16* base = qkv + (long)b * n * ld + h * dh;
  if (j < n)
    for (int d = lane; d < dh; d += 64) {
      sk[wid][d] = dec1<T>(base[(long)j * ld + inner + d]);
      sv[wid][d] = dec1<T>(base[(long)j * ld + 2 * inner + d]);
    }
  __syncthreads();
  if (j >= n || cnt <= 0) return;
  const float scale_log2e = scale * 1.44269504088896340736f;
  float dk[ACC], dv[ACC];
#pragma unroll
  for (int d = 0; d < ACC; ++d) dk[d] = dv[d] = 0.f;
  for (int i = lane; i < n; i += 64) {
    const r16* qi = base + (long)i * ld;
    const r16* gi = dout + ((long)b * n + i) * ldo + h * dh;
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(dot_row<T>(sk[wid], qi, dh), scale_log2e, -lse[(long)bh * n + i] * 1.44269504088896340736f));
    const float keep = drop.thresh ? drop_factor(drop, ((unsigned long long)bh * n + i) * ((n + 3) & ~3) + j) : 1.f;
    const float dp = dot_row<T>(sv[wid], gi, dh) * keep;
    axpy_row<ACC, T>(dv, dec1<T>(cvt1<T>(p * keep)), gi + d0, cnt);
    axpy_row<ACC, T>(dk, dec1<T>(cvt1<T>(p * (dp - delta[(long)bh * n + i]))), qi + d0, cnt);
  }
  r16* drow = dqkv + ((long)b * n + j) * ldd + h * dh + d0;
  reduce_store<ACC, T>(dk, cnt, scale, drow + inner, lane);
  reduce_store<ACC, T>(dv, cnt, 1.f, drow + 2 * inner, lane);
}

}  // namespace

bool attn_generic_supported(int dim_head) { return dim_head >= 8 && dim_head <= 128 && (dim_head % 8) == 0; }

int launch_attn_generic_fwd(const void* qkv, long ld, int B, int n, int heads, int dh, float scale, void* out, long ldo, float* lse, DropCfg drop,
                            hipStream_t s) {
  const dim3 grid((n + GW - 1) / GW, B * heads), block(64 * GW);
  const float sl = scale * 1.44269504088896340736f;
#define GEN_FWD(M) hipLaunchKernelGGL((attn_gen_fwd_kernel<M, T>), grid, block, 0, s, (const r16*)qkv, ld, n, heads, dh, sl, (r16*)out, ldo, lse, drop)
  NV_DISPATCH_OPERAND(T, if (dh <= 32) GEN_FWD(32); else if (dh <= 64) GEN_FWD(64); else GEN_FWD(128));
#undef GEN_FWD
  NV_CHECK_LAUNCH("nv_attn_fwd/generic");
  return NV_OK;
}

int launch_attn_generic_bwd(const void* qkv, long ld, const void* out, const void* dout, long ldo, const float* lse, int B, int n, int heads, int dh,
                            float scale, float* delta, void* dqkv, long ldd, DropCfg drop, hipStream_t s) {
  const dim3 grid((n + GW - 1) / GW, B * heads), block(64 * GW);
#define GEN_DQ(M) hipLaunchKernelGGL((attn_gen_dq_kernel<M, T>), grid, block, 0, s, (const r16*)qkv, ld, (const r16*)out, (const r16*)dout, ldo, lse, n, heads, dh, scale, delta, (r16*)dqkv, ldd, drop)
  NV_DISPATCH_OPERAND(T, if (dh <= 32) GEN_DQ(32); else if (dh <= 64) GEN_DQ(64); else GEN_DQ(128));
#undef GEN_DQ
  NV_CHECK_LAUNCH("nv_attn_bwd/generic dq");
#define GEN_DKV(M, A) hipLaunchKernelGGL((attn_gen_dkv_kernel<M, A, T>), dim3(grid.x, grid.y, (dh + A - 1) / A), block, 0, s, (const r16*)qkv, ld, (const r16*)dout, ldo, lse, delta, n, heads, dh, scale, (r16*)dqkv, ldd, drop)
  NV_DISPATCH_OPERAND(T, if (dh <= 32) GEN_DKV(32, 32); else if (dh <= 64) GEN_DKV(64, 64); else GEN_DKV(128, 64));
#undef GEN_DKV
  NV_CHECK_LAUNCH("nv_attn_bwd/generic dkv");
  return NV_OK;
}
