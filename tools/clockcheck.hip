// Box sanity check: shader clock under load, launch floor, copy bandwidth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
__global__ void k_clock(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x*3+2] = (unsigned long long)a; }
}
__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}
int main() {
  unsigned long long* d; hipMalloc(&d, 256 * 3 * 8);
  for (int rep = 0; rep < 3; ++rep) {
    k_clock<<<256, 256>>>(d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 3);
    hipMemcpy(h.data(), d, 256 * 3 * 8, hipMemcpyDeviceToHost);
    printf("clock: %.0f MHz (cycles %llu, real %llu)\n", 100.0 * h[0] / h[1], h[0], h[1]);
  }
  // launch floor
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 100; ++i) k_empty<<<256, 256>>>(nullptr);
  hipEventRecord(e0);
  for (int i = 0; i < 1000; ++i) k_empty<<<256, 256>>>(nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("empty kernel back-to-back: %.2f us each\n", ms);
  size_t n = (size_t)1 << 26;  // 1 GiB of float4
  float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
  hipMemset(a, 1, n * 16);
  for (int i = 0; i < 3; ++i) k_copy<<<2048, 256>>>(a, b, n);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) k_copy<<<2048, 256>>>(a, b, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("copy: %.2f TB/s (read+write)\n", 10.0 * 2 * n * 16 / (ms * 1e-3) / 1e12);
  // small copy 15 MB (L2/MALL resident)
  size_t m = 960000;
  for (int i = 0; i < 3; ++i) k_copy<<<2048, 256>>>(a, b, m);
  hipEventRecord(e0);
  for (int i = 0; i < 100; ++i) k_copy<<<2048, 256>>>(a, b, m);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("15MB copy: %.2f us each\n", ms * 10);
  return 0;
}
