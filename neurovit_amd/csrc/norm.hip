// Row-wise (LayerNorm-family) kernels of the ViT3D hot path, gfx950.  All HBM-bound, one 64-lane
// wave per row, rows cached in registers, 16-byte accesses, two-pass (mean, then centred variance)
// statistics in fp32 exactly as nn.LayerNorm (eps inside the sqrt, biased variance).
//
//   patch_ln_fwd      A1+A2  gather p^3 voxels of a token straight from the [B,H,W,D] volume (any strides),
//                            LayerNorm(patch_dim), write bf16 GEMM operand            (vit_3d.py:92-93, NeuroEncoder.py:200-202)
//   embed_finish_fwd  A4+A5  LayerNorm(dim) + pos-embedding add + cls row             (vit_3d.py:95,116-118)
//   ln_fwd            LN of the residual stream -> bf16 GEMM operand                  (vit_3d.py:18,37)
//   head_fwd          A9     cls-row LayerNorm + Linear(dim, C)                       (vit_3d.py:107-110,123-126)
//   *_bwd             the matching backward passes; parameter gradients are produced as deterministic
//                     per-workgroup partial sums + reduce_partials (no float atomics).
#include "common.h"

#define WAVES_PER_BLOCK 4

#include <type_traits>
// Output element type of the forward row kernels: bf16_t / fp16_t (MFMA operand of the 16-bit paths, by nv_set_operand_format) or float
// (the fp32 inference path, precise.hip: the reference validates in fp32, Trainer.py:101-118).
template <typename OT>
__device__ __forceinline__ void store4(OT* p, const f32x4& o) {
  if constexpr (std::is_same<OT, float>::value) *reinterpret_cast<f32x4*>(p) = o;
  else *reinterpret_cast<r16x4*>(p) = cvt4<OT>(o[0], o[1], o[2], o[3]);
}
// scalar store of one 16-bit operand in the format chosen at run time (the head kernels: a few hundred values per launch)
__device__ __forceinline__ r16 cvt1_rt(float v, int fp16) { return fp16 ? cvt1<fp16_t>(v) : cvt1<bf16_t>(v); }

// --------------------------------------------------------------------------------------- helpers
// Row of d floats (d % 4 == 0, d <= 256*NV) spread over a wave: lane holds float4 #(lane + 64 v).
template <int NV>
__device__ __forceinline__ void row_load(const float* row, int d, int lane, f32x4 (&x)[NV]) {
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (lane + 64 * v) * 4;
    x[v] = (c < d) ? *reinterpret_cast<const f32x4*>(row + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int NV>
__device__ __forceinline__ void row_stats(const f32x4 (&x)[NV], int d, int lane, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) s += (x[v][0] + x[v][1]) + (x[v][2] + x[v][3]);
  mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (lane + 64 * v) * 4;
    if (c < d) {
      const f32x4 t = x[v] - mean;
      q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
    }
  }
  rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
}

// --------------------------------------------------------------------------------------- ln_fwd
template <int NV, typename OT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, int M, int d, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, OT* __restrict__ y, long ldy,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  f32x4 xv[NV];
  row_load<NV>(x + (long)row * ldx, d, lane, xv);
  float mean, rstd;
  row_stats<NV>(xv, d, lane, eps, mean, rstd);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (lane + 64 * v) * 4;
    if (c < d) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
      const f32x4 o = (xv[v] - mean) * rstd * gm + bt;
      store4<OT>(y + (long)row * ldy + c, o);
    }
  }
  if (lane == 0 && mean_out) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

extern "C" int nv_ln_fwd(const float* x, long ldx, int M, int d, const float* gamma, const float* beta, float eps, void* y,
                         long ldy, float* mean, float* rstd, void* stream) {
  NV_CHECK_ARG(M > 0 && d > 0 && (d % 4) == 0 && d <= 2048, "nv_ln_fwd: d=%d must be a multiple of 4 and <= 2048", d);
  NV_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0 && nv_aligned16(x) && nv_aligned16(y) && nv_aligned16(gamma) && nv_aligned16(beta),
               "nv_ln_fwd: alignment");
  const dim3 grid((M + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(256);
  hipStream_t s = (hipStream_t)stream;
  NV_DISPATCH_OPERAND(T,
    if (d <= 1024) hipLaunchKernelGGL((ln_fwd_kernel<4, T>), grid, block, 0, s, x, ldx, M, d, gamma, beta, eps, (T*)y, ldy, mean, rstd);
    else hipLaunchKernelGGL((ln_fwd_kernel<8, T>), grid, block, 0, s, x, ldx, M, d, gamma, beta, eps, (T*)y, ldy, mean, rstd));
  NV_CHECK_LAUNCH("nv_ln_fwd");
  return NV_OK;
}

// fp32 output (the fp32 inference path); mean / rstd optional (both or neither)
extern "C" int nv_ln_fwd_f32(const float* x, long ldx, int M, int d, const float* gamma, const float* beta, float eps, float* y,
                             long ldy, float* mean, float* rstd, void* stream) {
  NV_CHECK_ARG(M > 0 && d > 0 && (d % 4) == 0 && d <= 2048, "nv_ln_fwd_f32: d=%d must be a multiple of 4 and <= 2048", d);
  NV_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0 && nv_aligned16(x) && nv_aligned16(y) && nv_aligned16(gamma) && nv_aligned16(beta) && (!mean == !rstd),
               "nv_ln_fwd_f32: alignment");
  const dim3 grid((M + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (d <= 1024) hipLaunchKernelGGL((ln_fwd_kernel<4, float>), grid, block, 0, s, x, ldx, M, d, gamma, beta, eps, y, ldy, mean, rstd);
  else hipLaunchKernelGGL((ln_fwd_kernel<8, float>), grid, block, 0, s, x, ldx, M, d, gamma, beta, eps, y, ldy, mean, rstd);
  NV_CHECK_LAUNCH("nv_ln_fwd_f32");
  return NV_OK;
}

// --------------------------------------------------------------------------------------- ln_bwd
// g_out = g_in + LN'(dy);  g16 = bf16(g_out * dropmask);  partials[blk][0] = sum dy*xhat, [1] = sum dy, [2] = sum g_out * dropmask.
// ROWS rows per wave are in flight together (x, dy and the incoming residual gradient are all requested before the first
// use): the kernel is one dependent memory round trip, not 2 * ROWS of them.
template <int NV, int ROWS, typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, int M, int d, const float* g_in, float* g_out,
                                                     long ldg, r16* __restrict__ g16, long ldg16, float* __restrict__ partials,
                                                     DropCfg drop, int seg_rows, int seg_skip) {
  // seg_rows > 0: dy is segmented - after every seg_rows rows seg_skip rows are skipped (token rows of a [B, 1+N, d] tensor)
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][d]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = gridDim.x * WAVES_PER_BLOCK;
  f32x4 gm[NV], a_g[NV], a_b[NV], a_c[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (lane + 64 * v) * 4;
    gm[v] = (c < d) ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    a_g[v] = a_b[v] = a_c[v] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int base = blockIdx.x * WAVES_PER_BLOCK + wid; base < M; base += ROWS * nw) {
    f32x4 xv[ROWS][NV], dv[ROWS][NV], gi[ROWS][NV];
    float mean[ROWS], rstd[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int row = base + r * nw;
      const int rr = row < M ? row : base;       // out-of-range slots re-read the first row; their results are dropped
      row_load<NV>(x + (long)rr * ldx, d, lane, xv[r]);
      const long dyr = seg_rows ? (long)rr + (long)(rr / seg_rows) * seg_skip : (long)rr;
      row_load<NV>(dy + dyr * lddy, d, lane, dv[r]);
      if (g_in) row_load<NV>(g_in + (long)rr * ldg, d, lane, gi[r]);
      mean[r] = mean_in[rr]; rstd[r] = rstd_in[rr];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int row = base + r * nw;
      if (row < M) {                              // wave-uniform
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          xv[r][v] = (xv[r][v] - mean[r]) * rstd[r];   // xhat (zero-padded lanes hold -mean*rstd but dv = 0 there)
          const f32x4 dyh = dv[r][v] * gm[v];
          s1 += (dyh[0] + dyh[1]) + (dyh[2] + dyh[3]);
          const f32x4 t = dyh * xv[r][v];
          s2 += (t[0] + t[1]) + (t[2] + t[3]);
          a_g[v] += dv[r][v] * xv[r][v];
          a_b[v] += dv[r][v];
        }
        const float c1 = wave_sum(s1) / (float)d, c2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int c = (lane + 64 * v) * 4;
          if (c < d) {
            f32x4 o = (dv[r][v] * gm[v] - c1 - xv[r][v] * c2) * rstd[r];
            if (g_in) o += gi[r][v];
            *reinterpret_cast<f32x4*>(g_out + (long)row * ldg + c) = o;
            // g16 / colsum feed the backward of the Linear whose (dropped-out) output was added to this residual stream
            if (drop.thresh) {
              o *= drop_factor4(drop, (unsigned long long)row * ldg + c);      // element index of the DENSE [M, d] tensor g is a (row-strided) view of: ldg = d x row spacing
            }
            if (g16) *reinterpret_cast<r16x4*>(g16 + (long)row * ldg16 + c) = cvt4<T>(o[0], o[1], o[2], o[3]);
            a_c[v] += o;
          }
        }
      }
    }
  }
  // deterministic in-block reduction of the three column accumulators, one array at a time through a [4 waves][d] buffer:
  // 12 KiB at d = 768 instead of 36, so that these workgroups fit on a CU beside two 72 KiB GEMM workgroups of the other stream
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (lane + 64 * v) * 4;
      if (c < d) *reinterpret_cast<f32x4*>(red + wid * d + c) = (a == 0) ? a_g[v] : (a == 1 ? a_b[v] : a_c[v]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += 256)
      partials[(long)blockIdx.x * 3 * d + a * d + i] = (red[i] + red[d + i]) + (red[2 * d + i] + red[3 * d + i]);
    __syncthreads();
  }
}

// out[c] (=|+=) sum_r part[r][c] for up to three concatenated segments of width d each.
// Block = 32 columns x 8 row groups: every thread streams R/8 independent loads (all in flight), the 8 row groups
// are combined through LDS in a fixed order -> deterministic, and wide enough (nseg*d/32 blocks) to be latency-free.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int R, int d, int nseg, float* o0, float* o1,
                                                              float* o2, int accumulate) {
  __shared__ float sh[8][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + cl;
  const long stride = (long)nseg * d;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < nseg * d) {
    int r = rg;
    for (; r + 24 < R; r += 32) {
      s0 += part[(long)r * stride + i];
      s1 += part[(long)(r + 8) * stride + i];
      s2 += part[(long)(r + 16) * stride + i];
      s3 += part[(long)(r + 24) * stride + i];
    }
    for (; r < R; r += 8) s0 += part[(long)r * stride + i];
  }
  sh[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && i < nseg * d) {
    const float s = ((sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl])) + ((sh[4][cl] + sh[5][cl]) + (sh[6][cl] + sh[7][cl]));
    const int seg = i / d, c = i - seg * d;
    float* o = seg == 0 ? o0 : (seg == 1 ? o1 : o2);
    if (o) o[c] = accumulate ? o[c] + s : s;
  }
}

// Several partial-sum reductions in ONE launch (a transformer layer's bias / LayerNorm-affine gradients all wait for the same
// point of the backward pass): job j owns blocks [block_end[j-1], block_end[j]), each block = 32 columns x 8 row groups as above.
constexpr int REDUCE_MAX_JOBS = 8;
struct ReduceJobs {
  nv_reduce_job job[REDUCE_MAX_JOBS];
  int block_end[REDUCE_MAX_JOBS];
  int count;
};
// VEC4 (every job: 16-byte aligned partials, nseg * width a multiple of 4): a block is still 32 columns, but as 8 float4 column
// groups x 32 row groups - a thread's R / 32 loads (8 at ViT3D-base's 257 partial rows) are all in flight together instead of eight
// dependent rounds of four, and every wave-instruction reads 128-byte row pieces as 16-byte lanes (10.9 -> ~4 us per launch).
template <bool VEC4>
__global__ __launch_bounds__(256) void reduce_multi_kernel(const ReduceJobs J) {
  __shared__ float sh[32][33];
  int j = 0;
  while (j + 1 < J.count && (int)blockIdx.x >= J.block_end[j]) ++j;          // workgroup-uniform
  const nv_reduce_job& q = J.job[j];
  const int blk = blockIdx.x - (j ? J.block_end[j - 1] : 0);
  const int R = q.rows, tot = q.nseg * q.width;
  const long stride = tot;
  const float* __restrict__ part = q.partials;
  constexpr int NRG = VEC4 ? 32 : 8;          // row groups
  if constexpr (VEC4) {
    const int c4 = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const int i = blk * 32 + c4 * 4;
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (i < tot) {
      int r = rg;
      for (; r + 96 < R; r += 128) {
        a0 += *reinterpret_cast<const f32x4*>(part + (long)r * stride + i);
        a1 += *reinterpret_cast<const f32x4*>(part + (long)(r + 32) * stride + i);
        a2 += *reinterpret_cast<const f32x4*>(part + (long)(r + 64) * stride + i);
        a3 += *reinterpret_cast<const f32x4*>(part + (long)(r + 96) * stride + i);
      }
      for (; r < R; r += 32) a0 += *reinterpret_cast<const f32x4*>(part + (long)r * stride + i);
    }
    const f32x4 a = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[rg][c4 * 4 + e] = a[e];
  } else {
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int i = blk * 32 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < tot) {
      int r = rg;
      for (; r + 24 < R; r += 32) {
        s0 += part[(long)r * stride + i];
        s1 += part[(long)(r + 8) * stride + i];
        s2 += part[(long)(r + 16) * stride + i];
        s3 += part[(long)(r + 24) * stride + i];
      }
      for (; r < R; r += 8) s0 += part[(long)r * stride + i];
    }
    sh[rg][cl] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  const int cl = threadIdx.x, i = blk * 32 + cl;
  if (cl < 32 && i < tot) {
    float s = 0.f;
#pragma unroll
    for (int g8 = 0; g8 < NRG; g8 += 8)       // fixed order: deterministic
      s += ((sh[g8][cl] + sh[g8 + 1][cl]) + (sh[g8 + 2][cl] + sh[g8 + 3][cl])) + ((sh[g8 + 4][cl] + sh[g8 + 5][cl]) + (sh[g8 + 6][cl] + sh[g8 + 7][cl]));
    const int seg = i / q.width, c = i - seg * q.width;
    float* o = q.out[seg];
    if (o) o[c] = q.accumulate ? o[c] + s : s;
  }
}

extern "C" int nv_reduce_multi(const nv_reduce_job* jobs, int count, void* stream) {
  NV_CHECK_ARG(jobs && count >= 1 && count <= REDUCE_MAX_JOBS, "nv_reduce_multi: 1..%d jobs", REDUCE_MAX_JOBS);
  ReduceJobs J;
  int blocks = 0;
  for (int j = 0; j < count; ++j) {
    const nv_reduce_job& q = jobs[j];
    NV_CHECK_ARG(q.partials && q.rows > 0 && q.width > 0 && q.nseg >= 1 && q.nseg <= 3, "nv_reduce_multi: job %d: bad shape", j);
    J.job[j] = q;
    blocks += (q.nseg * q.width + 31) / 32;
    J.block_end[j] = blocks;
  }
  for (int j = count; j < REDUCE_MAX_JOBS; ++j) { J.job[j] = jobs[0]; J.block_end[j] = blocks; }
  J.count = count;
  bool vec4 = true;
  for (int j = 0; j < count; ++j) vec4 = vec4 && nv_aligned16(jobs[j].partials) && ((jobs[j].nseg * jobs[j].width) % 4) == 0;
  if (vec4) hipLaunchKernelGGL(reduce_multi_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J);
  else hipLaunchKernelGGL(reduce_multi_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J);
  NV_CHECK_LAUNCH("nv_reduce_multi");
  return NV_OK;
}

// Workgroups (= partial rows) of ln_bwd_kernel: every wave takes two rows per pass, so ceil(M / 8) workgroups cover M rows in ONE
// pass (a grid capped at the 256 CUs sent the first waves round the loop again for the last 4 of ViT3D-base's 2052 rows).  Worth
// little by itself - 8.8 -> 8.5 us alone; inside the step the kernel shares HBM with the weight-gradient GEMMs of the auxiliary
// stream and averages 17-19 us either way.  Above 4096 rows the grid stops growing (the partial rows are reduced afterwards:
// 512 x 3 x d floats) and the waves loop.
static int ln_bwd_blocks(int M) {
  int b = (M + 2 * WAVES_PER_BLOCK - 1) / (2 * WAVES_PER_BLOCK);
  return b < 512 ? b : 512;
}
extern "C" int nv_ln_bwd_partial_rows(int M) { return ln_bwd_blocks(M); }

extern "C" long nv_ln_bwd_workspace_bytes(int M, int d) { return (long)ln_bwd_blocks(M) * 3 * d * sizeof(float); }

// dgamma/dbeta/dcolsum may be null (skipped).  g_in may be null (g_out = dx) or equal g_out (in place).
static int ln_bwd_launch(const float* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd, const float* gamma,
                         int M, int d, const float* g_in, float* g_out, long ldg, void* g16, long ldg16, float* dgamma, float* dbeta,
                         float* dcolsum, int accumulate, void* workspace, long ws_bytes, unsigned long drop_seed, float drop_p,
                         void* stream, void* reduce_stream, int seg_rows, int seg_skip) {
  NV_CHECK_ARG(M > 0 && d > 0 && (d % 4) == 0 && d <= 2048, "nv_ln_bwd: d=%d must be a multiple of 4 and <= 2048", d);
  NV_CHECK_ARG(drop_p == 0.f || (ldg % d) == 0, "nv_ln_bwd: with dropout g must be a (row-strided) view of a dense [*, d] tensor (ldg a multiple of d)");
  const DropCfg drop = make_drop(drop_seed, drop_p);
  NV_CHECK_ARG(ws_bytes >= nv_ln_bwd_workspace_bytes(M, d), "nv_ln_bwd: workspace too small");
  NV_CHECK_ARG((lddy % 4) == 0 && (ldx % 4) == 0 && (ldg % 4) == 0 && (ldg16 % 4) == 0, "nv_ln_bwd: leading dims must be multiples of 4");
  const int nb = ln_bwd_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)WAVES_PER_BLOCK * d * sizeof(float);
  NV_DISPATCH_OPERAND(T,
    if (d <= 1024)
      hipLaunchKernelGGL((ln_bwd_kernel<4, 2, T>), dim3(nb), dim3(256), lds, s, dy, lddy, x, ldx, mean, rstd, gamma, M, d, g_in, g_out, ldg,
                         (r16*)g16, ldg16, (float*)workspace, drop, seg_rows, seg_skip);
    else
      hipLaunchKernelGGL((ln_bwd_kernel<8, 1, T>), dim3(nb), dim3(256), lds, s, dy, lddy, x, ldx, mean, rstd, gamma, M, d, g_in, g_out, ldg,
                         (r16*)g16, ldg16, (float*)workspace, drop, seg_rows, seg_skip));
  NV_CHECK_LAUNCH("nv_ln_bwd");
  // the parameter-gradient reduction is off the data path: it may run on another stream (the caller then owns `workspace`
  // until that stream has passed this point), or be left to a later nv_ln_bwd_reduce call (reduce_stream = NV_LN_NO_REDUCE)
  if (reduce_stream == NV_LN_NO_REDUCE) return NV_OK;
  if (reduce_stream && reduce_stream != stream) {
    if (nv_stream_sync(stream, reduce_stream) != NV_OK) return NV_ERR_HIP;
    s = (hipStream_t)reduce_stream;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((3 * d + 31) / 32), dim3(256), 0, s, (const float*)workspace, nb, d, 3, dgamma,
                     dbeta, dcolsum, accumulate);
  NV_CHECK_LAUNCH("nv_ln_bwd/reduce");
  return NV_OK;
}

// second half of nv_ln_bwd(..., reduce_stream = NV_LN_NO_REDUCE): the caller orders `stream` after the main kernel
extern "C" int nv_ln_bwd_reduce(const void* workspace, int M, int d, float* dgamma, float* dbeta, float* dcolsum, int accumulate, void* stream) {
  NV_CHECK_ARG(workspace && M > 0 && d > 0 && (d % 4) == 0 && d <= 2048, "nv_ln_bwd_reduce: bad arguments");
  // through nv_reduce_multi: the same kernel, hence the same summation order, as when the caller folds these partials into a
  // multi-job launch - a staged (data-parallel) backward must reproduce the single-call one bit for bit
  const nv_reduce_job job = {(const float*)workspace, ln_bwd_blocks(M), d, 3, {dgamma, dbeta, dcolsum}, accumulate};
  return nv_reduce_multi(&job, 1, stream);
}

extern "C" int nv_ln_bwd(const float* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd,
                         const float* gamma, int M, int d, const float* g_in, float* g_out, long ldg, void* g16, long ldg16,
                         float* dgamma, float* dbeta, float* dcolsum, int accumulate, void* workspace, long ws_bytes,
                         unsigned long drop_seed, float drop_p, void* stream, void* reduce_stream) {
  return ln_bwd_launch(dy, lddy, x, ldx, mean, rstd, gamma, M, d, g_in, g_out, ldg, g16, ldg16, dgamma, dbeta, dcolsum, accumulate, workspace,
                       ws_bytes, drop_seed, drop_p, stream, reduce_stream, 0, 0);
}

// --------------------------------------------------------------------------------------- patch gather + LN(patch_dim)
struct PatchGeom {
  long sb, sc, sf, sh, sw;     // element strides of video[B,C,F,H,W]
  int B, C, F, H, W, p1, p2, pf;
  int gf, gh, gw;              // patch grid
  int P, N;                    // patch_dim, patches per volume
};

// feature k = ((i1*p2 + i2)*pf + ifr)*C + c  ('p1 p2 pf c');  token n = (ft*gh + ht)*gw + wt  ('f h w')
// The offset separates into a part of the token and a part of the feature: the kernels compute each once where it is invariant
// (the six integer divisions of the full form, once per element, were most of the gather kernels' instructions).
__device__ __forceinline__ long patch_tok_offset(const PatchGeom& g, int b, int n) {
  const int wt = n % g.gw; int u = n / g.gw;
  const int ht = u % g.gh, ft = u / g.gh;
  return (long)b * g.sb + (long)(ft * g.pf) * g.sf + (long)(ht * g.p1) * g.sh + (long)(wt * g.p2) * g.sw;
}
__device__ __forceinline__ long patch_feat_offset(const PatchGeom& g, int k) {
  const int c = k % g.C; int t = k / g.C;
  const int ifr = t % g.pf; t /= g.pf;
  const int i2 = t % g.p2, i1 = t / g.p2;
  return (long)c * g.sc + (long)ifr * g.sf + (long)i1 * g.sh + (long)i2 * g.sw;
}
__device__ __forceinline__ long patch_elem_offset(const PatchGeom& g, int b, int n, int k) {
  return patch_tok_offset(g, b, n) + patch_feat_offset(g, k);
}

#ifndef NV_PATCH_NT
#define NV_PATCH_NT 1
#endif
#if NV_PATCH_NT
#define NV_PATCH_LOAD(p) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p))
#else
#define NV_PATCH_LOAD(p) (*reinterpret_cast<const f32x4*>(p))
#endif
// VEC: C == 1, sf == 1, pf % 4 == 0, 16-byte aligned runs, P <= 4096: float4 gathers, row cached in registers.
template <bool VEC, typename OT>
__global__ __launch_bounds__(256) void patch_ln_fwd_kernel(const float* __restrict__ video, PatchGeom g, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, OT* __restrict__ out, long ldo,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           const float* __restrict__ vol_sigma) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (tok >= g.B * g.N) return;
  const int b = tok / g.N, n = tok - b * g.N;
  // RAW (un-normalised) volumes: LayerNorm over a patch of z = (x - mu) / sigma equals LayerNorm over the patch of x with
  // eps * sigma^2 in place of eps (mu drops out, sigma scales numerator and denominator): the dataset's per-volume z-score
  // (DatasetADNI.py:213) costs one multiply here instead of a pass over the volume
  if (vol_sigma) { const float sg = vol_sigma[b]; eps *= sg * sg; }
  OT* orow = out + (long)tok * ldo;
  float mean, rstd;
  if constexpr (VEC) {
    constexpr int NV = 16;
    f32x4 xv[NV];
    const float* tokp = video + patch_tok_offset(g, b, n);
    const int inner = g.pf * g.p2;                          // C == 1 here: features per patch row i1
    if ((256 % inner) == 0) {
      // lane's features 4 lane + 256 v: adding 256 only advances the patch row i1 (by 256 / inner): one decomposition per lane
      const float* lp = tokp + patch_feat_offset(g, 4 * lane);
      const long vstep = (long)(256 / inner) * g.sh;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        xv[v] = ((lane + 64 * v) * 4 < g.P) ? NV_PATCH_LOAD(lp + v * vstep) : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int k = (lane + 64 * v) * 4;
        xv[v] = (k < g.P) ? NV_PATCH_LOAD(tokp + patch_feat_offset(g, k)) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    row_stats<NV>(xv, g.P, lane, eps, mean, rstd);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int k = (lane + 64 * v) * 4;
      if (k < g.P) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + k), bt = *reinterpret_cast<const f32x4*>(beta + k);
        const f32x4 o = (xv[v] - mean) * rstd * gm + bt;
        store4<OT>(orow + k, o);
      }
    }
  } else if (g.P <= 64 * 16) {
    // any geometry whose patch fits sixteen elements per lane (the reference's shipped config: 9^3 = 729 features, not a multiple of four): ONE gather
    // into registers instead of three sweeps over the volume (mean, variance, normalise) - the same sums in the same order as the loop form below
    float xr[16];
    const long tbase = patch_tok_offset(g, b, n);
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int k = lane + 64 * v;
      xr[v] = (k < g.P) ? __builtin_nontemporal_load(video + tbase + patch_feat_offset(g, k)) : 0.f;
      s += xr[v];
    }
    mean = wave_sum(s) / (float)g.P;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float t = xr[v] - mean;
      if (lane + 64 * v < g.P) q += t * t;
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)g.P + eps);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int k = lane + 64 * v;
      if (k < ldo) orow[k] = (k < g.P) ? (OT)((xr[v] - mean) * rstd * gamma[k] + beta[k]) : (OT)0.f;
    }
    for (int k = lane + 1024; k < ldo; k += 64) orow[k] = (OT)0.f;
  } else {
    float s = 0.f;
    for (int k = lane; k < g.P; k += 64) s += video[patch_elem_offset(g, b, n, k)];
    mean = wave_sum(s) / (float)g.P;
    float q = 0.f;
    for (int k = lane; k < g.P; k += 64) {
      const float t = video[patch_elem_offset(g, b, n, k)] - mean;
      q += t * t;
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)g.P + eps);
    for (int k = lane; k < ldo; k += 64)
      orow[k] = (k < g.P) ? (OT)((video[patch_elem_offset(g, b, n, k)] - mean) * rstd * gamma[k] + beta[k]) : (OT)0.f;
  }
  if (lane == 0) {
    mean_out[tok] = mean;
    rstd_out[tok] = rstd;
  }
}

// LDS-staged (F, H, W) slabs (north_star: "LDS-staged (D,H,W) tiles and coalesced HBM reads") - the SELECTABLE form, see the measurement below.  One workgroup owns a patch COLUMN - the gf tokens that share
// (b, ht, wt) - and streams its p1 * p2 rows of F contiguous floats into LDS with full-line, fully coalesced 16-byte loads (for [B, H, W, D] volumes the rows of one
// i1 are one contiguous run of p2 * F floats); the per-token gather of patch_ln_fwd_kernel reads 64-byte half lines, every line twice (two tokens share it).  Each
// wave then takes tokens ft = wave, wave + 4, ...: its lanes pick the SAME features as in the per-token kernel (k = 4 lane + 256 v) out of the slab and run the same
// row_stats / normalise / store code - bit-identical output.  Row pitch F + 16 floats: the four rows a 16-lane phase of a ds_read_b128 touches sit on disjoint banks.
template <typename OT>
__global__ __launch_bounds__(256) void patch_ln_fwd_slab_kernel(const float* __restrict__ video, PatchGeom g, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, OT* __restrict__ out, long ldo,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                const float* __restrict__ vol_sigma) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cols = g.gh * g.gw;
  const int b = blockIdx.x / cols, hw = blockIdx.x - b * cols;
  const int ht = hw / g.gw, wt = hw - ht * g.gw;
  const int pitch = g.F + 16, rows = g.p1 * g.p2, f4 = g.F >> 2, items = rows * f4;
  const float* base = video + (long)b * g.sb + (long)(ht * g.p1) * g.sh + (long)(wt * g.p2) * g.sw;
  constexpr int U = 8;                                     // loads in flight per thread
  for (int q0 = tid; q0 < items; q0 += 256 * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + 256 * u;
      if (q < items) {
        const int r = q / f4, c4 = q - r * f4, i1 = r / g.p2, i2 = r - i1 * g.p2;
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (long)i1 * g.sh + (long)i2 * g.sw + 4 * c4));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + 256 * u;
      if (q < items) {
        const int r = q / f4, c4 = q - r * f4;
        *reinterpret_cast<f32x4*>(slab + r * pitch + 4 * c4) = v[u];
      }
    }
  }
  __syncthreads();
  if (vol_sigma) { const float sg = vol_sigma[b]; eps *= sg * sg; }      // (see patch_ln_fwd_kernel)
  constexpr int NV = 16;
  for (int ft = wave; ft < g.gf; ft += WAVES_PER_BLOCK) {
    const int tok = b * g.N + (ft * g.gh + ht) * g.gw + wt;
    OT* orow = out + (long)tok * ldo;
    f32x4 xv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int k = (lane + 64 * v) * 4;                   // feature ((i1 p2 + i2) pf + ifr): slab row k / pf, column ft pf + k % pf
      const int run = k / g.pf, ifr = k - run * g.pf;
      xv[v] = (k < g.P) ? *reinterpret_cast<const f32x4*>(slab + run * pitch + ft * g.pf + ifr) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float mean, rstd;
    row_stats<NV>(xv, g.P, lane, eps, mean, rstd);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int k = (lane + 64 * v) * 4;
      if (k < g.P) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + k), bt = *reinterpret_cast<const f32x4*>(beta + k);
        const f32x4 o = (xv[v] - mean) * rstd * gm + bt;
        store4<OT>(orow + k, o);
      }
    }
    if (lane == 0) {
      mean_out[tok] = mean;
      rstd_out[tok] = rstd;
    }
  }
}
// MEASURED (round 5, ViT3D-base batch 4, same box): 21.2 - 22.7 us against 14.4 - 14.9 for the per-token gather - a workgroup loads its 128 KiB, THEN computes, one
// workgroup per CU, so HBM idles during the arithmetic; the gather's thousands of independent waves overlap the two.  The gather therefore stays the default and the
// slabs are the selectable form (nv_patch_set_mode(2); tests hold the two bit-identical).  What the comparison did find: the gather's loads were missing the
// non-temporal hint - the 33.5 MB volume batch displaced that much of the weights from the Infinity Cache on every forward (NV_PATCH_LOAD: forward 0.998 -> 0.982 ms).
static int g_patch_mode = 0;      // 0 = per-token gather (default), 2 = LDS-staged slabs where they apply
extern "C" int nv_patch_set_mode(int mode) { g_patch_mode = mode == 2 ? 2 : 0; return 0; }
static long patch_slab_lds(const PatchGeom& g) { return (long)g.p1 * g.p2 * (g.F + 16) * sizeof(float); }
static bool patch_slab_ok(const PatchGeom& g) {
  return g_patch_mode == 2 && (g.F % 4) == 0 && g.gf >= 2 && patch_slab_lds(g) <= 156 * 1024;
}

static bool patch_vec_ok(const float* video, const PatchGeom& g, long ldo) {
  return g.C == 1 && g.sf == 1 && (g.pf % 4) == 0 && g.P <= 4096 && nv_aligned16(video) && (g.sb % 4) == 0 && (g.sh % 4) == 0 &&
         (g.sw % 4) == 0 && ldo == g.P;
}

static int make_geom(PatchGeom& g, const long* strides, int B, int C, int F, int H, int W, int p1, int p2, int pf) {
  NV_CHECK_ARG(B > 0 && C > 0 && p1 > 0 && p2 > 0 && pf > 0 && H % p1 == 0 && W % p2 == 0 && F % pf == 0,
               "patch geometry: image dims must be divisible by the patch size");
  g.sb = strides[0]; g.sc = strides[1]; g.sf = strides[2]; g.sh = strides[3]; g.sw = strides[4];
  g.B = B; g.C = C; g.F = F; g.H = H; g.W = W; g.p1 = p1; g.p2 = p2; g.pf = pf;
  g.gf = F / pf; g.gh = H / p1; g.gw = W / p2;
  g.P = C * p1 * p2 * pf; g.N = g.gf * g.gh * g.gw;
  return NV_OK;
}

// out: [B*N, ldo] bf16 with ldo >= P (columns P..ldo-1 are written as zero), mean/rstd: [B*N].
template <typename OT>
static int patch_ln_fwd_launch(const float* video, const long* strides5, int B, int C, int F, int H, int W, int p1, int p2, int pf,
                               const float* gamma, const float* beta, float eps, OT* out, long ldo, float* mean, float* rstd,
                               const float* vol_sigma, void* stream) {
  PatchGeom g;
  int rc = make_geom(g, strides5, B, C, F, H, W, p1, p2, pf);
  if (rc) return rc;
  if constexpr (std::is_same<OT, float>::value) NV_CHECK_ARG(ldo >= g.P, "nv_patch_ln_fwd_f32: ldo=%ld must be >= patch_dim=%d", ldo, g.P);
  else NV_CHECK_ARG(ldo >= g.P && (ldo % 8) == 0, "nv_patch_ln_fwd: ldo=%ld must be >= patch_dim=%d and a multiple of 8", ldo, g.P);
  const dim3 grid((g.B * g.N + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = patch_vec_ok(video, g, ldo) && nv_aligned16(gamma) && nv_aligned16(beta) && nv_aligned16(out);
  if (vec && patch_slab_ok(g)) {
    const int lds = (int)patch_slab_lds(g);
    static int lds_set = 0;
    if (lds > lds_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(patch_ln_fwd_slab_kernel<OT>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
      lds_set = 156 * 1024;
    }
    hipLaunchKernelGGL((patch_ln_fwd_slab_kernel<OT>), dim3(g.B * g.gh * g.gw), block, lds, s, video, g, gamma, beta, eps, out, ldo, mean, rstd, vol_sigma);
  } else if (vec)
    hipLaunchKernelGGL((patch_ln_fwd_kernel<true, OT>), grid, block, 0, s, video, g, gamma, beta, eps, out, ldo, mean, rstd, vol_sigma);
  else
    hipLaunchKernelGGL((patch_ln_fwd_kernel<false, OT>), grid, block, 0, s, video, g, gamma, beta, eps, out, ldo, mean, rstd, vol_sigma);
  NV_CHECK_LAUNCH("nv_patch_ln_fwd");
  return NV_OK;
}

extern "C" int nv_patch_ln_fwd(const float* video, const long* strides5, int B, int C, int F, int H, int W, int p1, int p2, int pf,
                               const float* gamma, const float* beta, float eps, void* out, long ldo, float* mean, float* rstd,
                               const float* vol_sigma, void* stream) {
  NV_DISPATCH_OPERAND(T, return patch_ln_fwd_launch<T>(video, strides5, B, C, F, H, W, p1, p2, pf, gamma, beta, eps, (T*)out, ldo, mean, rstd, vol_sigma, stream));
}
// fp32 tokens [B*N, ldo] (ldo >= patch_dim; no padding columns are needed: the fp32 GEMM takes any K)
extern "C" int nv_patch_ln_fwd_f32(const float* video, const long* strides5, int B, int C, int F, int H, int W, int p1, int p2, int pf,
                                   const float* gamma, const float* beta, float eps, float* out, long ldo, float* mean, float* rstd,
                                   const float* vol_sigma, void* stream) {
  return patch_ln_fwd_launch<float>(video, strides5, B, C, F, H, W, p1, p2, pf, gamma, beta, eps, out, ldo, mean, rstd, vol_sigma, stream);
}

// ---- 4D samples: patch gather + LayerNorm(patch_dim) of ALL timepoints of one patch position, straight from [Bo, H, W, D, T]
// (T innermost, as DatasetADNI_4D returns it).  The reference regroups the sample into T volumes with a strided copy
// (NeuroEncoder.py:54-56: 168 MB per sample); a T-strided gather per volume would touch every cache line of the sample once per
// timepoint.  Here one workgroup owns a patch position for all T: its p1*p2 runs of pf*T contiguous floats are read coalesced
// (float4 = four timepoints of one voxel), the per-timepoint statistics are reduced deterministically through LDS, a second
// sweep (L2 / Infinity-Cache hits) normalises, and an LDS transpose turns the [voxel][t] order into T token rows written in
// 8-byte pieces.  Token (bo*T + t)*N + n is bit-for-bit what a volume-by-volume call would address.  Needs C = 1, T % 4 == 0.
struct PatchTGeom {
  int Bo, H, W, D, T, p1, p2, pf, gf, gh, gw, N, P;
};
template <typename OT>
__global__ __launch_bounds__(256) void patch_ln_fwd_t_kernel(const float* __restrict__ x, PatchTGeom g, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, OT* __restrict__ out, long ldo,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             const float* __restrict__ vol_sigma, int nact) {
  extern __shared__ __attribute__((aligned(16))) char tsm[];
  const int TG = g.T >> 2;                               // float4 groups per voxel
  const int KR = nact / TG;                              // voxels (features) per round
  float* red = reinterpret_cast<float*>(tsm);            // [nact][8] partial sums, then [T] mean | [T] rstd
  OT* tile = reinterpret_cast<OT*>(tsm + (size_t)nact * 8 * sizeof(float));   // [T][KR] transposed round
  const int tid = threadIdx.x;
  const int bo = blockIdx.x / g.N, n = blockIdx.x - bo * g.N;
  const int wt = n % g.gw, ht = (n / g.gw) % g.gh, ft = n / (g.gw * g.gh);
  const bool act = tid < nact;
  const int tg = tid % TG, kk = tid / TG;                // this thread's timepoint group and voxel slot inside a round
  const int run_len = g.pf * g.T;                        // contiguous floats of one (i1, i2) run
  auto item_ptr = [&](int k) -> const float* {           // first of the four timepoints of voxel k (feature index k) for this tg
    const int run = k / g.pf, ifr = k - run * g.pf;
    const int i1 = run / g.p2, i2 = run - i1 * g.p2;
    const long base = ((((long)bo * g.H + ht * g.p1 + i1) * g.W + wt * g.p2 + i2) * g.D + (long)ft * g.pf) * g.T;
    return x + base + (long)ifr * g.T + 4 * tg;
  };
  // ---- sweep 1: shifted sums (pivot = voxel 0) per timepoint
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 pv = {0.f, 0.f, 0.f, 0.f};
  if (act) {
    pv = *reinterpret_cast<const f32x4*>(item_ptr(0));
    for (int k = kk; k < g.P; k += KR) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(item_ptr(k)) - pv;
      s1 += v;
      s2 += v * v;
    }
    *reinterpret_cast<f32x4*>(red + tid * 8) = s1;
    *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = s2;
  }
  __syncthreads();
  float mean_t = 0.f, rstd_t = 0.f;
  if (tid < g.T) {                                       // fixed summation order: deterministic
    const int mytg = tid >> 2, c = tid & 3;
    float a = 0.f, q = 0.f;
    for (int j = 0; j < KR; ++j) {
      a += red[(j * TG + mytg) * 8 + c];
      q += red[(j * TG + mytg) * 8 + 4 + c];
    }
    const float pivot = x[(item_ptr(0) - 4 * tg - x) + tid];            // voxel 0, timepoint tid
    const float ms = a / (float)g.P;
    float var = q / (float)g.P - ms * ms;
    var = var < 0.f ? 0.f : var;
    float e = eps;
    if (vol_sigma) { const float sg = vol_sigma[bo]; e *= sg * sg; }
    mean_t = pivot + ms;
    rstd_t = 1.0f / sqrtf(var + e);
  }
  __syncthreads();                                       // partial sums consumed
  if (tid < g.T) {
    red[tid] = mean_t;
    red[g.T + tid] = rstd_t;
    const long tok = ((long)bo * g.T + tid) * g.N + n;
    mean_out[tok] = mean_t;
    rstd_out[tok] = rstd_t;
  }
  __syncthreads();
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
  if (act) {
    mu = *reinterpret_cast<const f32x4*>(red + 4 * tg);
    rs = *reinterpret_cast<const f32x4*>(red + g.T + 4 * tg);
  }
  // ---- sweep 2: normalise, transpose [voxel][t] -> [t][voxel] through LDS, write 8-byte pieces of the T token rows
  const int pieces = (KR + 3) >> 2;                      // 4-feature pieces per timepoint per round
  for (int k0 = 0; k0 < g.P; k0 += KR) {
    const int k = k0 + kk;
    if (act && k < g.P) {
      const f32x4 v = NV_PATCH_LOAD(item_ptr(k));            // second and last sweep over the sample: non-temporal
      const float gm = gamma[k], bt = beta[k];
      const f32x4 y = (v - mu) * rs * gm + bt;
#pragma unroll
      for (int c = 0; c < 4; ++c) tile[(4 * tg + c) * KR + kk] = (OT)y[c];
    }
    __syncthreads();
    for (int e = tid; e < g.T * pieces; e += 256) {
      const int t = e / pieces, pc = e - t * pieces;
      const int kf = k0 + 4 * pc;
      if (kf < g.P) {                                    // P % 4 == 0 and KR % 4 == 0: whole pieces
        const long tok = ((long)bo * g.T + t) * g.N + n;
        typedef typename std::conditional<std::is_same<OT, float>::value, f32x4, r16x4>::type V4;
        *reinterpret_cast<V4*>(out + tok * ldo + kf) = *reinterpret_cast<const V4*>(tile + t * KR + 4 * pc);
      }
    }
    __syncthreads();
  }
}

// x: contiguous [Bo, H, W, D, T] float32; tokens of volume (bo, t) are rows (bo*T + t)*N + n of `out` (bf16 [Bo*T*N, ldo]).
template <typename OT>
static int patch_ln_fwd_4d_launch(const float* x, int Bo, int H, int W, int D, int T, int p1, int p2, int pf, const float* gamma,
                                  const float* beta, float eps, OT* out, long ldo, float* mean, float* rstd, const float* vol_sigma,
                                  void* stream) {
  NV_CHECK_ARG(x && gamma && beta && out && mean && rstd, "nv_patch_ln_fwd_4d: null pointer");
  NV_CHECK_ARG(Bo > 0 && T > 0 && (T % 4) == 0 && T <= 64 && p1 > 0 && p2 > 0 && pf > 0 && H % p1 == 0 && W % p2 == 0 && D % pf == 0,
               "nv_patch_ln_fwd_4d: T=%d must be a multiple of 4 (<= 64) and the extents divisible by the patch", T);
  PatchTGeom g;
  g.Bo = Bo; g.H = H; g.W = W; g.D = D; g.T = T; g.p1 = p1; g.p2 = p2; g.pf = pf;
  g.gf = D / pf; g.gh = H / p1; g.gw = W / p2; g.N = g.gf * g.gh * g.gw; g.P = p1 * p2 * pf;
  NV_CHECK_ARG((g.P % 4) == 0 && ldo >= g.P && (ldo % 4) == 0 && nv_aligned16(x) && nv_aligned16(out), "nv_patch_ln_fwd_4d: patch_dim %% 4, ldo, alignment");
  const int TG = T / 4;
  int nact = (256 / TG) * TG;                            // active threads: a multiple of the float4 groups per voxel ...
  nact -= (nact / TG % 4) * TG;                          // ... with a multiple of four voxels per round (whole 8-byte output pieces)
  NV_CHECK_ARG(nact >= TG * 4, "nv_patch_ln_fwd_4d: T too large for one workgroup");
  const size_t lds = (size_t)nact * 8 * sizeof(float) + (size_t)T * (nact / TG) * sizeof(OT);
  hipLaunchKernelGGL(patch_ln_fwd_t_kernel<OT>, dim3((unsigned)(Bo * g.N)), dim3(256), lds, (hipStream_t)stream, x, g, gamma, beta, eps, out, ldo,
                     mean, rstd, vol_sigma, nact);
  NV_CHECK_LAUNCH("nv_patch_ln_fwd_4d");
  return NV_OK;
}

extern "C" int nv_patch_ln_fwd_4d(const float* x, int Bo, int H, int W, int D, int T, int p1, int p2, int pf, const float* gamma,
                                  const float* beta, float eps, void* out, long ldo, float* mean, float* rstd, const float* vol_sigma,
                                  void* stream) {
  NV_DISPATCH_OPERAND(OT, return patch_ln_fwd_4d_launch<OT>(x, Bo, H, W, D, T, p1, p2, pf, gamma, beta, eps, (OT*)out, ldo, mean, rstd, vol_sigma, stream));
}
extern "C" int nv_patch_ln_fwd_4d_f32(const float* x, int Bo, int H, int W, int D, int T, int p1, int p2, int pf, const float* gamma,
                                      const float* beta, float eps, float* out, long ldo, float* mean, float* rstd, const float* vol_sigma,
                                      void* stream) {
  return patch_ln_fwd_4d_launch<float>(x, Bo, H, W, D, T, p1, p2, pf, gamma, beta, eps, out, ldo, mean, rstd, vol_sigma, stream);
}

// Backward of LayerNorm(patch_dim) w.r.t. its affine parameters only (the volume needs no gradient):
// dgamma[k] = sum_tok dxp[tok,k] * xhat[tok,k], dbeta[k] = sum_tok dxp[tok,k]; xhat is re-gathered.
// VEC (the conditions of the forward gather: one channel, frames contiguous, 16-byte aligned runs): four consecutive features per
// thread - a float4 of the volume and one of dxp instead of four 4-byte gathers (53 -> 22 us at ViT3D-base).
template <bool VEC>
__global__ __launch_bounds__(256) void patch_ln_bwd_kernel(const float* __restrict__ video, PatchGeom g, const float* __restrict__ dxp,
                                                           long ldd, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           int tok_per_block, float* __restrict__ partials) {
  const int T = g.B * g.N;
  const int t0 = blockIdx.x * tok_per_block, t1 = min(T, t0 + tok_per_block);
  if constexpr (VEC) {
    for (int k = threadIdx.x * 4; k < g.P; k += 1024) {
      f32x4 ag = f32x4{0.f, 0.f, 0.f, 0.f}, ab = ag;
      const float* fp = video + patch_feat_offset(g, k);   // the feature part once per thread, the token part is wave-uniform
      for (int tok = t0; tok < t1; ++tok) {
        const int b = tok / g.N, n = tok - b * g.N;
        // (both streams are read for the last time in this step: non-temporal, so that they do not displace the weights from the Infinity Cache)
        const f32x4 xh = (NV_PATCH_LOAD(fp + patch_tok_offset(g, b, n)) - mean_in[tok]) * rstd_in[tok];
        const f32x4 dv = NV_PATCH_LOAD(dxp + (long)tok * ldd + k);
        ag += dv * xh;
        ab += dv;
      }
      *reinterpret_cast<f32x4*>(partials + (long)blockIdx.x * 2 * g.P + k) = ag;
      *reinterpret_cast<f32x4*>(partials + (long)blockIdx.x * 2 * g.P + g.P + k) = ab;
    }
  } else {
    for (int k = threadIdx.x; k < g.P; k += 256) {
      float ag = 0.f, ab = 0.f;
      const float* fp = video + patch_feat_offset(g, k);
      for (int tok = t0; tok < t1; ++tok) {
        const int b = tok / g.N, n = tok - b * g.N;
        const float xh = (fp[patch_tok_offset(g, b, n)] - mean_in[tok]) * rstd_in[tok];
        const float dv = dxp[(long)tok * ldd + k];
        ag += dv * xh;
        ab += dv;
      }
      partials[(long)blockIdx.x * 2 * g.P + k] = ag;
      partials[(long)blockIdx.x * 2 * g.P + g.P + k] = ab;
    }
  }
}

static int patch_bwd_blocks(int T, int* tpb) {
  int t = (T + 511) / 512;
  if (t < 1) t = 1;
  *tpb = t;
  return (T + t - 1) / t;
}

extern "C" long nv_patch_ln_bwd_workspace_bytes(int tokens, int P) {
  int tpb;
  return (long)patch_bwd_blocks(tokens, &tpb) * 2 * P * sizeof(float);
}

extern "C" int nv_patch_ln_bwd(const float* video, const long* strides5, int B, int C, int F, int H, int W, int p1, int p2, int pf,
                               const float* dxp, long ldd, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                               int accumulate, void* workspace, long ws_bytes, void* stream) {
  PatchGeom g;
  int rc = make_geom(g, strides5, B, C, F, H, W, p1, p2, pf);
  if (rc) return rc;
  const int T = g.B * g.N;
  int tpb;
  const int nb = patch_bwd_blocks(T, &tpb);
  NV_CHECK_ARG(ws_bytes >= (long)nb * 2 * g.P * (long)sizeof(float), "nv_patch_ln_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (patch_vec_ok(video, g, ldd) && nv_aligned16(dxp) && (ldd % 4) == 0 && (g.P % 4) == 0 && nv_aligned16(workspace))
    hipLaunchKernelGGL(patch_ln_bwd_kernel<true>, dim3(nb), dim3(256), 0, s, video, g, dxp, ldd, mean, rstd, tpb, (float*)workspace);
  else
    hipLaunchKernelGGL(patch_ln_bwd_kernel<false>, dim3(nb), dim3(256), 0, s, video, g, dxp, ldd, mean, rstd, tpb, (float*)workspace);
  NV_CHECK_LAUNCH("nv_patch_ln_bwd");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * g.P + 31) / 32), dim3(256), 0, s, (const float*)workspace, nb, g.P, 2, dgamma,
                     dbeta, (float*)nullptr, accumulate);
  NV_CHECK_LAUNCH("nv_patch_ln_bwd/reduce");
  return NV_OK;
}

// --------------------------------------------------------------------------------------- embed finish (A4 + A5)
// x[b, 0, :] = cls + pos[0];  x[b, 1+i, :] = LN(t[b*N+i]) * gamma + beta + pos[1+i]
template <int NV>
__global__ __launch_bounds__(256) void embed_finish_fwd_kernel(const float* __restrict__ t, long ldt, int B, int N, int d,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                               const float* __restrict__ pos, const float* __restrict__ cls,
                                                               float* __restrict__ x, long ldx, float* __restrict__ mean_out,
                                                               float* __restrict__ rstd_out, DropCfg drop) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int n = N + 1;
  if (r >= B * n) return;
  const int b = r / n, i = r - b * n;
  float* xrow = x + (long)r * ldx;
  if (i == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (lane + 64 * v) * 4;
      if (c < d) {
        f32x4 o = *reinterpret_cast<const f32x4*>(cls + c) + *reinterpret_cast<const f32x4*>(pos + c);
        if (drop.thresh) {
          o *= drop_factor4(drop, (unsigned long long)r * d + c);
        }
        *reinterpret_cast<f32x4*>(xrow + c) = o;
      }
    }
    return;
  }
  const int tok = b * N + (i - 1);
  f32x4 xv[NV];
  row_load<NV>(t + (long)tok * ldt, d, lane, xv);
  float mean, rstd;
  row_stats<NV>(xv, d, lane, eps, mean, rstd);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (lane + 64 * v) * 4;
    if (c < d) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
      const f32x4 pe = *reinterpret_cast<const f32x4*>(pos + (long)i * d + c);
      // same association as the reference: (LN output) + pos   (vit_3d.py:118 `x += pos_embedding`), then emb dropout (:119)
      f32x4 o = ((xv[v] - mean) * rstd * gm + bt) + pe;
      if (drop.thresh) {
          o *= drop_factor4(drop, (unsigned long long)r * d + c);
      }
      *reinterpret_cast<f32x4*>(xrow + c) = o;
    }
  }
  if (lane == 0) {
    mean_out[tok] = mean;
    rstd_out[tok] = rstd;
  }
}

extern "C" int nv_embed_finish_fwd(const float* t, long ldt, int B, int N, int d, const float* gamma, const float* beta, float eps,
                                   const float* pos, const float* cls, float* x, long ldx, float* mean, float* rstd,
                                   unsigned long drop_seed, float drop_p, void* stream) {
  const DropCfg drop = make_drop(drop_seed, drop_p);
  NV_CHECK_ARG(B > 0 && N > 0 && (d % 4) == 0 && d <= 2048, "nv_embed_finish_fwd: d=%d must be a multiple of 4 and <= 2048", d);
  NV_CHECK_ARG((ldt % 4) == 0 && (ldx % 4) == 0, "nv_embed_finish_fwd: leading dims must be multiples of 4");
  const int rows = B * (N + 1);
  const dim3 grid((rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (d <= 1024) hipLaunchKernelGGL(embed_finish_fwd_kernel<4>, grid, block, 0, s, t, ldt, B, N, d, gamma, beta, eps, pos, cls, x, ldx, mean, rstd, drop);
  else hipLaunchKernelGGL(embed_finish_fwd_kernel<8>, grid, block, 0, s, t, ldt, B, N, d, gamma, beta, eps, pos, cls, x, ldx, mean, rstd, drop);
  NV_CHECK_LAUNCH("nv_embed_finish_fwd");
  return NV_OK;
}

// dpos[i, :] = sum_b g[b, i, :];  dcls = dpos[0]
__global__ void batch_sum_kernel(const float* __restrict__ g, long ldg, int B, int n, int d, float* dpos, float* dcls, int accumulate) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * d) return;
  const int i = (int)(idx / d), c = (int)(idx - (long)i * d);
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += g[((long)b * n + i) * ldg + c];
  if (dpos) dpos[idx] = accumulate ? dpos[idx] + s : s;
  if (i == 0 && dcls) dcls[c] = accumulate ? dcls[c] + s : s;
}

__global__ void apply_drop_kernel(float* __restrict__ g, long count, DropCfg drop) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= count) return;
  f32x4 v = *reinterpret_cast<f32x4*>(g + i);
  v *= drop_factor4(drop, (unsigned long long)i);
  *reinterpret_cast<f32x4*>(g + i) = v;
}

// Backward of A4/A5: g [B, n, d] -> dt [B*N, d] (fp32 + bf16 copy), dgamma3/dbeta3, dbias_pe = colsum(dt), dpos, dcls.
// With embedding dropout the incoming g is first multiplied by the forward mask IN PLACE (g is dead afterwards).
extern "C" long nv_embed_finish_bwd_workspace_bytes(int B, int N, int d) { return nv_ln_bwd_workspace_bytes(B * N, d); }

extern "C" int nv_embed_finish_bwd(const float* g, long ldg, const float* t, long ldt, const float* mean, const float* rstd,
                                   const float* gamma, int B, int N, int d, float* dt, long lddt, void* dt16, long lddt16,
                                   float* dgamma, float* dbeta, float* dbias_pe, float* dpos, float* dcls, int accumulate,
                                   void* workspace, long ws_bytes, unsigned long drop_seed, float drop_p, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const DropCfg drop = make_drop(drop_seed, drop_p);
  if (drop.thresh) {
    NV_CHECK_ARG(ldg == d, "nv_embed_finish_bwd: dropout needs a dense g");
    const long count = (long)B * (N + 1) * d;
    hipLaunchKernelGGL(apply_drop_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, s, (float*)g, count, drop);
    NV_CHECK_LAUNCH("nv_embed_finish_bwd/drop");
  }
  NV_CHECK_ARG(B > 0 && N > 0 && (d % 4) == 0 && d <= 2048, "nv_embed_finish_bwd: bad dims");
  const int n = N + 1;
  const long need = nv_embed_finish_bwd_workspace_bytes(B, N, d);
  NV_CHECK_ARG(ws_bytes >= need, "nv_embed_finish_bwd: workspace too small");
  // g's token rows (row b*n + 1 + i) form one segmented [B*N] row set: a single LN backward over all volumes
  {
    const int rc = ln_bwd_launch(g + ldg, ldg, t, ldt, mean, rstd, gamma, B * N, d, nullptr, dt, lddt, dt16, lddt16, dgamma, dbeta, dbias_pe,
                                 accumulate, workspace, ws_bytes, 0, 0.f, stream, nullptr, N, 1);
    if (rc) return rc;
  }
  const long tot = (long)n * d;
  hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, g, ldg, B, n, d, dpos, dcls, accumulate);
  NV_CHECK_LAUNCH("nv_embed_finish_bwd/batch_sum");
  return NV_OK;
}

// --------------------------------------------------------------------------------------- classification head (A9)
// One workgroup per volume: xh = LN(x[b, 0, :]); logits[b, c] = xh . W[c, :] + bias[c].  All fp32 (VALU).
// (body shared with head_step_kernel below: one definition, the same bits)
__device__ __forceinline__ void head_fwd_row(int b, float* sh, const float* __restrict__ x, long row_stride, int d, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float eps, const float* __restrict__ Wt,
                                             const float* __restrict__ bias, int C, float* __restrict__ xh_out,
                                             float* __restrict__ stats_out, float* __restrict__ logits) {
  float* xs = sh;              // d floats + 8 scratch
  float* scratch = sh + d;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* row = x + (long)b * row_stride;
  float s = 0.f;
  for (int c = tid; c < d; c += 256) { xs[c] = row[c]; s += xs[c]; }
  s = wave_sum(s);
  if (lane == 0) scratch[wid] = s;
  __syncthreads();
  const float mean = ((scratch[0] + scratch[1]) + (scratch[2] + scratch[3])) / (float)d;
  float q = 0.f;
  for (int c = tid; c < d; c += 256) { const float t = xs[c] - mean; q += t * t; }
  q = wave_sum(q);
  if (lane == 0) scratch[4 + wid] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((scratch[4] + scratch[5]) + (scratch[6] + scratch[7])) / (float)d + eps);
  for (int c = tid; c < d; c += 256) {
    const float v = (xs[c] - mean) * rstd * gamma[c] + beta[c];
    xs[c] = v;
    if (xh_out) xh_out[(long)b * d + c] = v;
  }
  if (tid == 0 && stats_out) { stats_out[2 * b] = mean; stats_out[2 * b + 1] = rstd; }
  __syncthreads();
  for (int c = wid; c < C; c += WAVES_PER_BLOCK) {
    float a = 0.f;
    for (int k = lane; k < d; k += 64) a += xs[k] * Wt[(long)c * d + k];
    a = wave_sum(a);
    if (lane == 0) logits[(long)b * C + c] = a + bias[c];
  }
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, long row_stride, int d, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, const float* __restrict__ Wt,
                                                       const float* __restrict__ bias, int C, float* __restrict__ xh_out,
                                                       float* __restrict__ stats_out, float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) float sh[];   // d floats + 8 scratch
  head_fwd_row(blockIdx.x, sh, x, row_stride, d, gamma, beta, eps, Wt, bias, C, xh_out, stats_out, logits);
}

extern "C" int nv_head_fwd(const float* x, long row_stride, int B, int d, const float* gamma, const float* beta, float eps,
                           const float* W, const float* bias, int C, float* xh, float* stats, float* logits, void* stream) {
  NV_CHECK_ARG(B > 0 && d > 0 && C > 0, "nv_head_fwd: bad dims");
  hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(256), (d + 8) * sizeof(float), (hipStream_t)stream, x, row_stride, d, gamma, beta,
                     eps, W, bias, C, xh, stats, logits);
  NV_CHECK_LAUNCH("nv_head_fwd");
  return NV_OK;
}

// Backward of the head for volume b (one workgroup): dxh = dlogits[b] . W; LN backward on the pooled row (cls row, or the
// token mean when pool_mean: then x is the [B, d] pooled matrix and every row of g gets dx / n);
// writes g[b, 0, :] = dx (fp32 + bf16; the other rows were zeroed by a memset node in front) and per-volume
// partials [b][3][d] = (dgamma, dbeta, dx) reduced afterwards.
__device__ __forceinline__ void head_bwd_x_row(int b, float* sh, const float* __restrict__ dlogits, int C, const float* __restrict__ Wt,
                                               const float* __restrict__ x, long row_stride, const float* __restrict__ stats,
                                               const float* __restrict__ gamma, int d, int n, float* __restrict__ g, long ldg,
                                               r16* __restrict__ g16, long ldg16, float* __restrict__ partials, const DropCfg& drop, int pool_mean, int fp16) {
  float* dys = sh;             // dyh[d] + 8 scratch
  float* scratch = sh + d;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  const float* row = x + (long)b * row_stride;
  float s1 = 0.f, s2 = 0.f;
  for (int c = tid; c < d; c += 256) {
    float dy = 0.f;
    for (int k = 0; k < C; ++k) dy += dlogits[(long)b * C + k] * Wt[(long)k * d + c];
    const float xh = (row[c] - mean) * rstd;
    partials[((long)b * 3 + 0) * d + c] = dy * xh;
    partials[((long)b * 3 + 1) * d + c] = dy;
    const float dyh = dy * gamma[c];
    dys[c] = dyh;
    s1 += dyh;
    s2 += dyh * xh;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { scratch[wid] = s1; scratch[4 + wid] = s2; }
  __syncthreads();
  const float c1 = ((scratch[0] + scratch[1]) + (scratch[2] + scratch[3])) / (float)d;
  const float c2 = ((scratch[4] + scratch[5]) + (scratch[6] + scratch[7])) / (float)d;
  for (int c = tid; c < d; c += 256) {
    const float xh = (row[c] - mean) * rstd;
    float dx = (dys[c] - c1 - xh * c2) * rstd;
    if (pool_mean) {   // pool = 'mean' (vit_3d.py:127): every token row receives dx / n
      dx /= (float)n;
      float cs = 0.f;
      for (int t = 0; t < n; ++t) {
        const long r = (long)b * n + t;
        g[r * ldg + c] = dx;
        const float dxm = drop.thresh ? dx * drop_factor(drop, (unsigned long long)r * d + c) : dx;
        if (g16) g16[r * ldg16 + c] = cvt1_rt(dxm, fp16);
        cs += dxm;
      }
      partials[((long)b * 3 + 2) * d + c] = cs;
      continue;
    }
    g[(long)b * n * ldg + c] = dx;
    const float dxm = drop.thresh ? dx * drop_factor(drop, (unsigned long long)b * n * d + c) : dx;   // last block's FF output dropout
    if (g16) g16[(long)b * n * ldg16 + c] = cvt1_rt(dxm, fp16);
    partials[((long)b * 3 + 2) * d + c] = dxm;
  }
}

__global__ __launch_bounds__(256) void head_bwd_x_kernel(const float* __restrict__ dlogits, int C, const float* __restrict__ Wt,
                                                         const float* __restrict__ x, long row_stride, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, int d, int n, float* __restrict__ g, long ldg,
                                                         r16* __restrict__ g16, long ldg16, float* __restrict__ partials, DropCfg drop,
                                                         int pool_mean, int nvol, int fp16) {
  extern __shared__ __attribute__((aligned(16))) float sh[];   // dyh[d] + 8 scratch
  if ((int)blockIdx.x >= nvol) {
    // pool = 'cls': the residual gradient is zero except for the cls rows the first nvol workgroups write - the rest of this grid
    // clears the other rows of g (fp32) and g16 (bf16) in the same launch (were two hipMemsetAsync nodes in front of it)
    const int part = blockIdx.x - nvol, nparts = gridDim.x - nvol;
    zero_rows_except(reinterpret_cast<char*>(g), (long)nvol * n, (long)d * 4, n, part, nparts);
    if (g16) zero_rows_except(reinterpret_cast<char*>(g16), (long)nvol * n, (long)d * 2, n, part, nparts);
    return;
  }
  head_bwd_x_row(blockIdx.x, sh, dlogits, C, Wt, x, row_stride, stats, gamma, d, n, g, ldg, g16, ldg16, partials, drop, pool_mean, fp16);
}

// dW[c, k] = sum_b dlogits[b, c] * xh[b, k];  dbias[c] = sum_b dlogits[b, c]
__global__ void head_bwd_w_kernel(const float* __restrict__ dlogits, const float* __restrict__ xh, int B, int C, int d, float* dW,
                                  float* dbias, int accumulate) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long)C * d) {
    const int c = (int)(idx / d), k = (int)(idx - (long)c * d);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogits[(long)b * C + c] * xh[(long)b * d + k];
    dW[idx] = accumulate ? dW[idx] + s : s;
  }
  if (idx < C) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogits[(long)b * C + idx];
    dbias[idx] = accumulate ? dbias[idx] + s : s;
  }
}

// pool = 'mean' (vit_3d.py:127): out[b, c] = mean over the n token rows of x[b, :, c]. 64 columns x 4 row groups per workgroup.
__global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ x, int n, int d, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  float s = 0.f;
  if (c < d)
    for (int t = rg; t < n; t += 4) s += x[((long)b * n + t) * d + c];
  part[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < d) out[(long)b * d + c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) / (float)n;
}

extern "C" int nv_token_mean(const float* x, int B, int n, int d, float* out, void* stream) {
  NV_CHECK_ARG(B > 0 && n > 0 && d > 0, "nv_token_mean: bad shape B=%d n=%d d=%d", B, n, d);
  hipLaunchKernelGGL(token_mean_kernel, dim3((d + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, n, d, out);
  NV_CHECK_LAUNCH("nv_token_mean");
  return NV_OK;
}

extern "C" long nv_head_bwd_workspace_bytes(int B, int d) { return (long)B * 3 * d * sizeof(float); }

extern "C" int nv_head_bwd(const float* dlogits, int B, int C, const float* W, const float* x, long row_stride, const float* stats,
                           const float* xh, const float* gamma, int d, int n, float* g, long ldg, void* g16, long ldg16,
                           float* dgamma, float* dbeta, float* dW, float* dbias, float* dcolsum, int accumulate, void* workspace,
                           long ws_bytes, unsigned long drop_seed, float drop_p, int pool_mean, void* stream) {
  NV_CHECK_ARG(ws_bytes >= nv_head_bwd_workspace_bytes(B, d), "nv_head_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  NV_CHECK_ARG(ldg == d && (!g16 || ldg16 == d), "nv_head_bwd: g / g16 must be dense [B*n, d]");
  // the residual gradient is zero except for the cls rows (pool = 'cls'): B workgroups write those, the others clear the rest
  NV_CHECK_ARG(pool_mean || ((d % 8) == 0 && nv_aligned16(g) && (!g16 || nv_aligned16(g16))), "nv_head_bwd: g / g16 must be 16-byte aligned, d a multiple of 8");
  const long fill_bytes = pool_mean ? 0 : (long)B * n * d * (g16 ? 6 : 4);
  int fill_blocks = (int)((fill_bytes + (1 << 16) - 1) >> 16);            // ~64 KiB per workgroup
  if (fill_blocks > 1024) fill_blocks = 1024;
  hipLaunchKernelGGL(head_bwd_x_kernel, dim3(B + fill_blocks), dim3(256), (d + 8) * sizeof(float), s, dlogits, C, W, x, row_stride, stats, gamma, d, n,
                     g, ldg, (r16*)g16, ldg16, (float*)workspace, make_drop(drop_seed, drop_p), pool_mean, B, nv_operand_format() == NV_OPERAND_FP16);
  NV_CHECK_LAUNCH("nv_head_bwd/x");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((3 * d + 31) / 32), dim3(256), 0, s, (const float*)workspace, B, d, 3, dgamma, dbeta,
                     dcolsum, accumulate);
  NV_CHECK_LAUNCH("nv_head_bwd/reduce");
  const long tot = (long)C * d;
  hipLaunchKernelGGL(head_bwd_w_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, dlogits, xh, B, C, d, dW, dbias, accumulate);
  NV_CHECK_LAUNCH("nv_head_bwd/w");
  return NV_OK;
}

// ---- the head's forward, nn.CrossEntropyLoss and the head's backward in TWO launches instead of five (the train step as one call,
// Trainer.py:69-74: dependent launches of a few microseconds of work each).  Launch 1: one workgroup per volume runs head_fwd_row,
// ce_row_term and head_bwd_x_row - the bodies of head_fwd_kernel, ce_loss_kernel and head_bwd_x_kernel - while the other workgroups
// clear the rows of g / g16 outside the cls rows; launch 2: everything that sums over the volumes, in the order of ce_loss_kernel,
// reduce_partials_kernel and head_bwd_w_kernel.  Every output bit-identical to the five launches.  pool = 'cls' only.
// (One launch with a single workgroup walking the volumes was measured first: 1.1 % SLOWER than the five launches - 4 x ~10 us in a row.)
struct HeadStep {
  const float* x; long row_stride; int B, d, C, n; const float* gamma; const float* beta; float eps; const float* W; const float* bias;
  const long* labels; float grad_scale; float* xh; float* stats; float* logits; float* loss; float* dlogits;
  float* g; long ldg; r16* g16; long ldg16; float* dgamma; float* dbeta; float* dW; float* dbias; float* dcolsum; int accumulate;
  float* partials; float* terms; DropCfg drop;
  int fp16;                     // format of g16 (nv_operand_format)
  const float* scale_state;     // dynamic loss scale (nv_loss_scale_*): dlogits are multiplied by scale_state[0]; null = grad_scale alone
};
// sum over the R rows of column i of part[R][stride], in reduce_partials_kernel's order (8 row groups, four interleaved accumulators each)
__device__ __forceinline__ float partial_column_sum(const float* part, int R, long stride, int i) {
  float grp[8];
#pragma unroll
  for (int rg = 0; rg < 8; ++rg) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rg;
    for (; r + 24 < R; r += 32) {
      s0 += part[(long)r * stride + i];
      s1 += part[(long)(r + 8) * stride + i];
      s2 += part[(long)(r + 16) * stride + i];
      s3 += part[(long)(r + 24) * stride + i];
    }
    for (; r < R; r += 8) s0 += part[(long)r * stride + i];
    grp[rg] = (s0 + s1) + (s2 + s3);
  }
  return ((grp[0] + grp[1]) + (grp[2] + grp[3])) + ((grp[4] + grp[5]) + (grp[6] + grp[7]));
}
// launch 1: workgroup b = volume b (forward, loss term, backward to the cls row of g), the others clear g / g16 outside the cls rows
__global__ __launch_bounds__(256) void head_rows_kernel(const HeadStep a) {
  extern __shared__ __attribute__((aligned(16))) float sh[];   // d floats + 8 scratch + 4 (loss reductions)
  if ((int)blockIdx.x >= a.B) {
    const int part = blockIdx.x - a.B, nparts = gridDim.x - a.B;
    zero_rows_except(reinterpret_cast<char*>(a.g), (long)a.B * a.n, (long)a.d * 4, a.n, part, nparts);
    if (a.g16) zero_rows_except(reinterpret_cast<char*>(a.g16), (long)a.B * a.n, (long)a.d * 2, a.n, part, nparts);
    return;
  }
  const int b = blockIdx.x;
  head_fwd_row(b, sh, a.x, a.row_stride, a.d, a.gamma, a.beta, a.eps, a.W, a.bias, a.C, a.xh, a.stats, a.logits);
  __syncthreads();                                   // logits[b], stats[b] (global, written by this workgroup) are visible to all of it
  const float term = ce_row_term(a.logits + (long)b * a.C, a.labels[b], a.C, (a.scale_state ? a.grad_scale * a.scale_state[0] : a.grad_scale) / (float)a.B, sh + a.d + 8, a.dlogits + (long)b * a.C);
  if (threadIdx.x == 0) a.terms[b] = term;
  __syncthreads();
  head_bwd_x_row(b, sh, a.dlogits, a.C, a.W, a.x, a.row_stride, a.stats, a.gamma, a.d, a.n, a.g, a.ldg, a.g16, a.ldg16, a.partials, a.drop, 0, a.fp16);
}
// launch 2: everything that sums over the volumes - the loss, (dgamma, dbeta, column sum of the cls-row gradient), dW, dbias
__global__ __launch_bounds__(256) void head_sums_kernel(const HeadStep a) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx == 0) {
    float total = 0.f;
    for (int b = 0; b < a.B; ++b) total += a.terms[b];
    a.loss[0] = total / (float)a.B;
  }
  if (idx < 3L * a.d) {
    const int i = (int)idx;
    const float s = partial_column_sum(a.partials, a.B, 3L * a.d, i);
    const int seg = i / a.d, c = i - seg * a.d;
    float* o = seg == 0 ? a.dgamma : (seg == 1 ? a.dbeta : a.dcolsum);
    if (o) o[c] = a.accumulate ? o[c] + s : s;
    return;
  }
  const long j = idx - 3L * a.d;
  if (j < (long)a.C * a.d) {
    const int c = (int)(j / a.d), k = (int)(j - (long)c * a.d);
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += a.dlogits[(long)b * a.C + c] * a.xh[(long)b * a.d + k];
    a.dW[j] = a.accumulate ? a.dW[j] + s : s;
    if (j < a.C) {
      float t = 0.f;
      for (int b = 0; b < a.B; ++b) t += a.dlogits[(long)b * a.C + j];
      a.dbias[j] = a.accumulate ? a.dbias[j] + t : t;
    }
  }
}

extern "C" long nv_head_step_workspace_bytes(int B, int d) { return nv_head_bwd_workspace_bytes(B, d) + (long)((B + 3) / 4 * 4) * sizeof(float); }

extern "C" int nv_head_step(const float* x, long row_stride, int B, int d, const float* gamma, const float* beta, float eps, const float* W,
                            const float* bias, int C, const long* labels, float grad_scale, float* xh, float* stats, float* logits, float* loss,
                            float* dlogits, int n, float* g, long ldg, void* g16, long ldg16, float* dgamma, float* dbeta, float* dW, float* dbias,
                            float* dcolsum, int accumulate, void* workspace, long ws_bytes, unsigned long drop_seed, float drop_p, void* stream) {
  return nv_head_step_scaled(x, row_stride, B, d, gamma, beta, eps, W, bias, C, labels, grad_scale, nullptr, xh, stats, logits, loss, dlogits, n, g, ldg, g16, ldg16,
                             dgamma, dbeta, dW, dbias, dcolsum, accumulate, workspace, ws_bytes, drop_seed, drop_p, stream);
}

// scale_state (may be NULL): the device block of nv_loss_scale_init - dlogits (and with them every gradient) carry its current scale
extern "C" int nv_head_step_scaled(const float* x, long row_stride, int B, int d, const float* gamma, const float* beta, float eps, const float* W,
                                   const float* bias, int C, const long* labels, float grad_scale, const float* scale_state, float* xh, float* stats,
                                   float* logits, float* loss, float* dlogits, int n, float* g, long ldg, void* g16, long ldg16, float* dgamma,
                                   float* dbeta, float* dW, float* dbias, float* dcolsum, int accumulate, void* workspace, long ws_bytes,
                                   unsigned long drop_seed, float drop_p, void* stream) {
  NV_CHECK_ARG(B > 0 && d > 0 && C > 0 && C <= d && n > 0, "nv_head_step: B = %d, d = %d, C = %d, n = %d", B, d, C, n);
  NV_CHECK_ARG(x && gamma && beta && W && bias && labels && xh && stats && logits && loss && dlogits && g && dgamma && dbeta && dW && dbias && workspace,
               "nv_head_step: null pointer");
  NV_CHECK_ARG(ws_bytes >= nv_head_step_workspace_bytes(B, d), "nv_head_step: workspace too small");
  NV_CHECK_ARG(ldg == d && (!g16 || ldg16 == d) && (d % 8) == 0 && nv_aligned16(g) && (!g16 || nv_aligned16(g16)), "nv_head_step: g / g16 must be dense [B*n, d], 16-byte aligned, d a multiple of 8");
  HeadStep a;
  a.x = x; a.row_stride = row_stride; a.B = B; a.d = d; a.C = C; a.n = n; a.gamma = gamma; a.beta = beta; a.eps = eps; a.W = W; a.bias = bias;
  a.labels = labels; a.grad_scale = grad_scale; a.xh = xh; a.stats = stats; a.logits = logits; a.loss = loss; a.dlogits = dlogits;
  a.g = g; a.ldg = ldg; a.g16 = (r16*)g16; a.ldg16 = ldg16; a.dgamma = dgamma; a.dbeta = dbeta; a.dW = dW; a.dbias = dbias; a.dcolsum = dcolsum;
  a.accumulate = accumulate; a.partials = (float*)workspace; a.terms = (float*)workspace + (long)B * 3 * d; a.drop = make_drop(drop_seed, drop_p);
  a.fp16 = nv_operand_format() == NV_OPERAND_FP16; a.scale_state = scale_state;
  const long fill_bytes = (long)B * n * d * (g16 ? 6 : 4);
  int fill_blocks = (int)((fill_bytes + (1 << 16) - 1) >> 16);            // ~64 KiB per workgroup
  if (fill_blocks > 1024) fill_blocks = 1024;
  hipLaunchKernelGGL(head_rows_kernel, dim3(B + fill_blocks), dim3(256), (d + 12) * sizeof(float), (hipStream_t)stream, a);
  NV_CHECK_LAUNCH("nv_head_step/rows");
  const long outs = 3L * d + (long)C * d;
  hipLaunchKernelGGL(head_sums_kernel, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  NV_CHECK_LAUNCH("nv_head_step");
  return NV_OK;
}

// --------------------------------------------------------------------------------------- column sum of a bf16 matrix (bias grads)
// partial[chunk][c] = sum over the chunk's rows of X[r, c]; 8 columns per lane (16-byte loads).
template <typename T>
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const r16* __restrict__ X, long ld, int M, int N, int rows_per_chunk,
                                                          float* __restrict__ partials) {
  const int col = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (col >= N) return;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = r0; r < r1; ++r) {
    const r16x8 v = *reinterpret_cast<const r16x8*>(X + (long)r * ld + col);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += dec1<T>(v[j]);
  }
  float* p = partials + (long)blockIdx.y * N + col;
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = a[j];
}

static int colsum_chunks(int M) { int c = (M + 31) / 32; return c > 128 ? 128 : c; }
extern "C" long nv_colsum_workspace_bytes(int M, int N) { return (long)colsum_chunks(M) * N * sizeof(float); }

extern "C" int nv_colsum_bf16(const void* X, long ld, int M, int N, float* out, int accumulate, void* workspace, long ws_bytes,
                              void* stream) {
  NV_CHECK_ARG(M > 0 && N > 0 && (N % 8) == 0 && (ld % 8) == 0 && nv_aligned16(X), "nv_colsum_bf16: N, ld must be multiples of 8");
  NV_CHECK_ARG(ws_bytes >= nv_colsum_workspace_bytes(M, N), "nv_colsum_bf16: workspace too small");
  const int chunks = colsum_chunks(M), rpc = (M + chunks - 1) / chunks;
  hipStream_t s = (hipStream_t)stream;
  NV_DISPATCH_OPERAND(T, hipLaunchKernelGGL(colsum_bf16_kernel<T>, dim3((N / 8 + 255) / 256, chunks), dim3(256), 0, s, (const r16*)X, ld, M, N, rpc, (float*)workspace));
  NV_CHECK_LAUNCH("nv_colsum_bf16");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((N + 31) / 32), dim3(256), 0, s, (const float*)workspace, chunks, N, 1, out,
                     (float*)nullptr, (float*)nullptr, accumulate);
  NV_CHECK_LAUNCH("nv_colsum_bf16/reduce");
  return NV_OK;
}
