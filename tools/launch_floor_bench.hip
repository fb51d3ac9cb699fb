// Micro-benchmark: what does ONE dependent kernel cost on MI355X, whatever it computes?  N kernels are enqueued back to back on
// one stream (each waits for the previous one: in-order queue, barrier bit) and the whole chain is timed with two events.
// Build: hipcc --offload-arch=gfx950 -O3 -o launch_floor_bench launch_floor_bench.hip ; run on the GPU box.  Tuning aid only.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_empty() {}
__global__ void k_touch(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.f; }
__global__ __launch_bounds__(512) void k_lds(float* p) {      // a workgroup that owns a large LDS allocation, like the GEMM / attention kernels
  extern __shared__ float sm[];
  sm[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) p[blockIdx.x] = sm[blockDim.x - 1];
}

template <typename F>
static double chain_us(F launch, int n) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 50; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / n;
}

int main() {
  float* d; hipMalloc(&d, 1 << 24);
  const int n = 2000;
  printf("empty kernel, 1 x 64 threads                      : %6.2f us per dependent launch\n", chain_us([&] { k_empty<<<1, 64>>>(); }, n));
  printf("empty kernel, 256 x 256 threads                   : %6.2f us\n", chain_us([&] { k_empty<<<256, 256>>>(); }, n));
  printf("touch kernel, 1024 x 256 threads (1 MB rmw)       : %6.2f us\n", chain_us([&] { k_touch<<<1024, 256>>>(d); }, n));
  hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  printf("LDS kernel, 256 x 512 threads, 72 KiB LDS each    : %6.2f us\n", chain_us([&] { k_lds<<<256, 512, 72 * 1024>>>(d); }, n));
  printf("LDS kernel, 256 x 512 threads, 144 KiB LDS each   : %6.2f us\n", chain_us([&] { k_lds<<<256, 512, 144 * 1024>>>(d); }, n));
  printf("LDS kernel, 792 x 512 threads, 72 KiB LDS each    : %6.2f us\n", chain_us([&] { k_lds<<<792, 512, 72 * 1024>>>(d); }, n));
  return 0;
}
