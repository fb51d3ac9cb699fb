// Grad-CAM reduction of the reference's NeuroEncoder.get_attention_map (src/models/NeuroEncoder.py:101-116) as ONE launch.
//
//   weights[b,t] = mean_d grad[b,t,:]          cam[b,t] = sum_d weights[b,t] * act[b,t,:] = weights[b,t] * sum_d act[b,t,:]
//   cam = relu(cam[:, 1:])  (cls token dropped)  ->  (cam - min) / (max - min + 1e-8)  over the whole map
//
// act = output of the last block's attention LayerNorm (bf16, the engine's xn1 buffer), grad = its gradient (fp32, the
// engine's hookg buffer): 2 x [B, n, d] on the device -> [B, n-1] floats.  HBM-bound (reads 6 B per element once).
// One wave per token row (two row reductions by wave shuffles); the min / max of the map crosses workgroups through
// per-workgroup partials and an arrival ticket: the LAST workgroup to arrive normalises the (tiny) map.  The hand-off is the
// placement-independent agent-scope release / acquire of the CDNA4 guide (Guideline 16): results do not depend on which
// workgroup is last.
#include "common.h"

namespace {
constexpr int GC_THREADS = 256;
constexpr int GC_WAVES = GC_THREADS / 64;

template <typename T>
__global__ __launch_bounds__(GC_THREADS) void gradcam_reduce_kernel(const r16* __restrict__ act, const float* __restrict__ grad, int B, int n,
                                                                    int d, float* cam, float* __restrict__ part, unsigned* ticket,
                                                                    float* minmax) {
  __shared__ float s_min[GC_WAVES], s_max[GC_WAVES];
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int N = n - 1, rows = B * N;
  float lo = INFINITY, hi = 0.f;                         // relu output is >= 0, every workgroup owns >= 0 rows
  for (int r = blockIdx.x * GC_WAVES + wid; r < rows; r += gridDim.x * GC_WAVES) {
    const int b = r / N, t = r - b * N + 1;              // token 0 is the cls token
    const long base = ((long)b * n + t) * d;
    float sg = 0.f, sa = 0.f;
    for (int k = lane * 8; k < d; k += 64 * 8) {         // d % 8 == 0 (engine requirement)
      const r16x8 a = *reinterpret_cast<const r16x8*>(act + base + k);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(grad + base + k), g1 = *reinterpret_cast<const f32x4*>(grad + base + k + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) sa += dec1<T>(a[j]);
      sg += (g0[0] + g0[1]) + (g0[2] + g0[3]) + (g1[0] + g1[1]) + (g1[2] + g1[3]);
    }
    sg = wave_sum(sg); sa = wave_sum(sa);
    const float v = fmaxf((sg / (float)d) * sa, 0.f);
    if (lane == 0) cam[r] = v;
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  if (lane == 0) { s_min[wid] = lo; s_max[wid] = hi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < GC_WAVES; ++w) { lo = fminf(lo, s_min[w]); hi = fmaxf(hi, s_max[w]); }
    part[2 * blockIdx.x] = lo; part[2 * blockIdx.x + 1] = hi;
  }
  // publish: every storing wave drains its stores, the workgroup meets, ONE lane releases at agent scope and takes a ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (tk == gridDim.x - 1) ? 1u : 0u;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!s_last) return;
  // last arriver: min / max of the partials (fixed order), then normalise the whole map
  lo = INFINITY; hi = 0.f;
  for (int i = 0; i < (int)gridDim.x; ++i) { lo = fminf(lo, part[2 * i]); hi = fmaxf(hi, part[2 * i + 1]); }
  const float inv = 1.0f / (hi - lo + 1e-8f);
  for (int i = tid; i < rows; i += GC_THREADS) cam[i] = (cam[i] - lo) * inv;
  if (tid == 0) {
    if (minmax) { minmax[0] = lo; minmax[1] = hi; }
    *ticket = 0;                                          // self-reset for the next call (the caller also zeroes it once)
  }
}
}  // namespace

static int gradcam_blocks(int B, int n) {
  const int rows = B * (n - 1);
  int blocks = (rows + GC_WAVES - 1) / GC_WAVES;
  return blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks);
}

extern "C" long nv_gradcam_workspace_bytes(int B, int n) { return 16 + 8L * gradcam_blocks(B, n); }

extern "C" int nv_gradcam_reduce(const void* act, const float* grad, int B, int n, int d, float* cam, float* minmax, void* workspace,
                                 long ws_bytes, void* stream) {
  NV_CHECK_ARG(act && grad && cam && workspace && B > 0 && n > 1 && d > 0 && (d % 8) == 0, "nv_gradcam_reduce: bad arguments (d %% 8 == 0, n > 1)");
  NV_CHECK_ARG(nv_aligned16(act) && nv_aligned16(grad) && nv_aligned16(workspace), "nv_gradcam_reduce: 16-byte alignment");
  NV_CHECK_ARG(ws_bytes >= nv_gradcam_workspace_bytes(B, n), "nv_gradcam_reduce: workspace too small");
  unsigned* ticket = (unsigned*)workspace;
  float* part = (float*)((char*)workspace + 16);
  if (hipMemsetAsync(ticket, 0, 16, (hipStream_t)stream) != hipSuccess) { nv_set_error("nv_gradcam_reduce: memset failed"); return NV_ERR_HIP; }
  NV_DISPATCH_OPERAND(T, hipLaunchKernelGGL(gradcam_reduce_kernel<T>, dim3(gradcam_blocks(B, n)), dim3(GC_THREADS), 0, (hipStream_t)stream, (const r16*)act, grad, B, n,
                                            d, cam, part, ticket, minmax));
  NV_CHECK_LAUNCH("nv_gradcam_reduce");
  return NV_OK;
}
