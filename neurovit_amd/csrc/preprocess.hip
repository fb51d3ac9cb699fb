// Input contract of the hot path: crop of the raw scanner volume and per-volume z-score,
// (x - mean) / (std + 1e-8) with the POPULATION standard deviation (numpy's default ddof = 0) over the cropped volume -
// src/data/DatasetADNI.py:212-213 (3D: one timepoint) and src/data/DatasetADNI_4D.py:86-87 (4D: one statistic over
// all timepoints of the sample).  The reference does this on the CPU in every DataLoader worker; here one pass
// accumulates sum / sum of squares in double (as numpy does for integer / float32 input), a second pass normalises
// and writes the dense [*, Sx, Sy, Sz(, T)] float32 tensor the encoder reads.  HBM-bound: 2 reads + 1 write per voxel.
#include "common.h"

struct CropGeom {
  long s[5];        // element strides of raw[B, X, Y, Z, T]
  int n[5];         // cropped extents B, Sx, Sy, Sz, T
  int o[4];         // crop origin x0, y0, z0, t0
  long per_volume;  // Sx * Sy * Sz * T
};

template <typename T>
__device__ __forceinline__ float crop_load(const T* raw, const CropGeom& g, int b, long i) {
  long r = i;
  const int t = (int)(r % g.n[4]); r /= g.n[4];
  const int z = (int)(r % g.n[3]); r /= g.n[3];
  const int y = (int)(r % g.n[2]); r /= g.n[2];
  const int x = (int)r;
  return (float)raw[(long)b * g.s[0] + (long)(x + g.o[0]) * g.s[1] + (long)(y + g.o[1]) * g.s[2] + (long)(z + g.o[2]) * g.s[3] +
                    (long)(t + g.o[3]) * g.s[4]];
}

// partial[b][blk] = (sum, sum of squares) of the block's slice of cropped volume b, in double
template <typename T>
__global__ __launch_bounds__(256) void zscore_stats_kernel(const T* __restrict__ raw, CropGeom g, double* __restrict__ partial) {
  __shared__ double sh[2][4];
  const int b = blockIdx.y;
  double s = 0.0, q = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < g.per_volume; i += (long)gridDim.x * 256) {
    const double v = (double)crop_load(raw, g, b, i);
    s += v;
    q += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wid] = s; sh[1][wid] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* p = partial + ((long)b * gridDim.x + blockIdx.x) * 2;
    p[0] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    p[1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void zscore_apply_kernel(const T* __restrict__ raw, CropGeom g, const double* __restrict__ partial,
                                                           int nparts, float eps, float* __restrict__ out, float* __restrict__ stats) {
  __shared__ float sh_mean, sh_inv;
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {                     // every block re-reduces the (few hundred) partials in the same fixed order
    double s = 0.0, q = 0.0;
    for (int i = 0; i < nparts; ++i) {
      s += partial[((long)b * nparts + i) * 2];
      q += partial[((long)b * nparts + i) * 2 + 1];
    }
    const double cnt = (double)g.per_volume;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double sd = sqrt(var);
    sh_mean = (float)mean;
    sh_inv = (float)(1.0 / (sd + (double)eps));
    if (blockIdx.x == 0 && stats) { stats[2 * b] = (float)mean; stats[2 * b + 1] = (float)sd; }
  }
  __syncthreads();
  const float mean = sh_mean, inv = sh_inv;
  float* o = out + (long)b * g.per_volume;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < g.per_volume; i += (long)gridDim.x * 256)
    o[i] = (crop_load(raw, g, b, i) - mean) * inv;
}

// sigma[b] = population std of cropped volume b + eps, mean[b] optional: the statistics alone (for the fused raw-volume path:
// nv_vit_forward(..., vol_sigma) folds the z-score into the patch LayerNorm's epsilon, so no normalised copy is written)
__global__ __launch_bounds__(64) void zscore_finish_kernel(const double* __restrict__ partial, int nparts, double cnt, float eps, float* __restrict__ sigma,
                                                          float* __restrict__ mean_out) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < nparts; ++i) {
    s += partial[((long)b * nparts + i) * 2];
    q += partial[((long)b * nparts + i) * 2 + 1];
  }
  const double mean = s / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  sigma[b] = (float)(sqrt(var) + (double)eps);
  if (mean_out) mean_out[b] = (float)mean;
}

extern "C" long nv_zscore_crop_workspace_bytes(int B) { return (long)B * 256 * 2 * sizeof(double); }

// raw: [B, X, Y, Z, T] with element strides `strides5` (T = 1 for 3D input), dtype 0 = float32, 1 = int16.
// out: dense float32 [B, Sx, Sy, Sz, T'] (T' = crop[9]); crop = {x0, y0, z0, t0, Sx, Sy, Sz, T'}.
// stats (optional): [B, 2] mean and population std of each cropped volume.
extern "C" int nv_zscore_crop(const void* raw, int dtype, const long* strides5, int B, const int* crop8, float eps, float* out,
                              float* stats, void* workspace, long ws_bytes, void* stream) {
  NV_CHECK_ARG(raw && out && strides5 && crop8 && B > 0, "nv_zscore_crop: null argument");
  NV_CHECK_ARG(dtype == 0 || dtype == 1, "nv_zscore_crop: dtype %d unsupported (0 = float32, 1 = int16)", dtype);
  NV_CHECK_ARG(ws_bytes >= nv_zscore_crop_workspace_bytes(B), "nv_zscore_crop: workspace too small");
  CropGeom g;
  for (int i = 0; i < 5; ++i) g.s[i] = strides5[i];
  g.n[0] = B;
  for (int i = 0; i < 4; ++i) { g.o[i] = crop8[i]; g.n[1 + i] = crop8[4 + i]; }
  NV_CHECK_ARG(g.n[1] > 0 && g.n[2] > 0 && g.n[3] > 0 && g.n[4] > 0 && g.o[0] >= 0 && g.o[1] >= 0 && g.o[2] >= 0 && g.o[3] >= 0,
               "nv_zscore_crop: bad crop");
  g.per_volume = (long)g.n[1] * g.n[2] * g.n[3] * g.n[4];
  int nparts = (int)((g.per_volume + 256 * 16 - 1) / (256 * 16));
  if (nparts > 256) nparts = 256;
  if (nparts < 1) nparts = 1;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(nparts, B);
  long nblk = (g.per_volume + 256 * 8 - 1) / (256 * 8);
  if (nblk > 2048) nblk = 2048;
  const dim3 grid2((unsigned)nblk, B);
  if (dtype == 0) {
    hipLaunchKernelGGL(zscore_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)raw, g, (double*)workspace);
    hipLaunchKernelGGL(zscore_apply_kernel<float>, grid2, dim3(256), 0, s, (const float*)raw, g, (const double*)workspace, nparts, eps, out, stats);
  } else {
    hipLaunchKernelGGL(zscore_stats_kernel<short>, grid, dim3(256), 0, s, (const short*)raw, g, (double*)workspace);
    hipLaunchKernelGGL(zscore_apply_kernel<short>, grid2, dim3(256), 0, s, (const short*)raw, g, (const double*)workspace, nparts, eps, out, stats);
  }
  NV_CHECK_LAUNCH("nv_zscore_crop");
  return NV_OK;
}

// Statistics only: sigma[b] = std(cropped volume b) + eps (and mean[b] when asked).  Same crop / dtype conventions as nv_zscore_crop.
extern "C" int nv_volume_sigma(const void* raw, int dtype, const long* strides5, int B, const int* crop8, float eps, float* sigma, float* mean,
                               void* workspace, long ws_bytes, void* stream) {
  NV_CHECK_ARG(raw && sigma && strides5 && crop8 && B > 0, "nv_volume_sigma: null argument");
  NV_CHECK_ARG(dtype == 0 || dtype == 1, "nv_volume_sigma: dtype %d unsupported (0 = float32, 1 = int16)", dtype);
  NV_CHECK_ARG(ws_bytes >= nv_zscore_crop_workspace_bytes(B), "nv_volume_sigma: workspace too small");
  CropGeom g;
  for (int i = 0; i < 5; ++i) g.s[i] = strides5[i];
  g.n[0] = B;
  for (int i = 0; i < 4; ++i) { g.o[i] = crop8[i]; g.n[1 + i] = crop8[4 + i]; }
  NV_CHECK_ARG(g.n[1] > 0 && g.n[2] > 0 && g.n[3] > 0 && g.n[4] > 0 && g.o[0] >= 0 && g.o[1] >= 0 && g.o[2] >= 0 && g.o[3] >= 0, "nv_volume_sigma: bad crop");
  g.per_volume = (long)g.n[1] * g.n[2] * g.n[3] * g.n[4];
  int nparts = (int)((g.per_volume + 256 * 16 - 1) / (256 * 16));
  if (nparts > 256) nparts = 256;
  if (nparts < 1) nparts = 1;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(zscore_stats_kernel<float>, dim3(nparts, B), dim3(256), 0, s, (const float*)raw, g, (double*)workspace);
  else hipLaunchKernelGGL(zscore_stats_kernel<short>, dim3(nparts, B), dim3(256), 0, s, (const short*)raw, g, (double*)workspace);
  hipLaunchKernelGGL(zscore_finish_kernel, dim3(B), dim3(64), 0, s, (const double*)workspace, nparts, (double)g.per_volume, eps, sigma, mean);
  NV_CHECK_LAUNCH("nv_volume_sigma");
  return NV_OK;
}
