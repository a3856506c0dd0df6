#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1;} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, const f16* __restrict__ w_in,
                                         const f16* __restrict__ b_in, f16* __restrict__ h, int n) {
  const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t node = item >> 4;
  const int chunk = (int)(item & 15);
  if (node >= n) return;
  f16x8 out = {1, 2, 3, 4, 5, 6, 7, 8};
  if (MODE >= 1) {
    float s = 0;
    for (int k2 = 0; k2 < 7; ++k2) s += x[node * 7 + k2];
    out[0] = (f16)s;
  }
  if (MODE >= 2) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f16x8 w = *reinterpret_cast<const f16x8*>(w_in + (chunk * 8 + c) * 8);
      out[c] += w[0] + w[3];
    }
  }
  *reinterpret_cast<f16x8*>(h + node * 128 + chunk * 8) = out;
}
template <int MODE>
__global__ __launch_bounds__(256) void k2(const float* __restrict__ x, const f16* __restrict__ w_in,
                                         const f16* __restrict__ b_in, f16* __restrict__ h, int n) {
  const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t node = item >> 4;
  const int chunk = (int)(item & 15);
  if (node >= n) return;
  float xv[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) xv[k] = MODE == 1 ? x[node * 7 + k] : (float)(f16)x[node * 7 + k];
  const f16x8 bias = *reinterpret_cast<const f16x8*>(b_in + chunk * 8);
  f16x8 out;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f16x8 w = *reinterpret_cast<const f16x8*>(w_in + (MODE == 2 ? (c * 16) * 8 : (c * 16 + chunk) * 8));
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc = __builtin_fmaf(xv[k], (float)w[k], acc);
    out[c] = (f16)(acc + (float)bias[c]);
  }
  *reinterpret_cast<f16x8*>(h + node * 128 + chunk * 8) = out;
}
// one thread per node x 2 chunks? no: variant with LDS-staged weights
__global__ __launch_bounds__(256) void k3(const float* __restrict__ x, const f16* __restrict__ w_in,
                                         const f16* __restrict__ b_in, f16* __restrict__ h, int n) {
  __shared__ f16x8 ws[128];
  if (threadIdx.x < 128) ws[threadIdx.x] = reinterpret_cast<const f16x8*>(w_in)[threadIdx.x];
  __syncthreads();
  const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t node = item >> 4;
  const int chunk = (int)(item & 15);
  if (node >= n) return;
  float xv[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) xv[k] = (float)(f16)x[node * 7 + k];
  const f16x8 bias = *reinterpret_cast<const f16x8*>(b_in + chunk * 8);
  f16x8 out;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f16x8 w = ws[c * 16 + chunk];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc = __builtin_fmaf(xv[k], (float)w[k], acc);
    out[c] = (f16)(acc + (float)bias[c]);
  }
  *reinterpret_cast<f16x8*>(h + node * 128 + chunk * 8) = out;
}
// MODE 3: like store-only but through a grid-stride loop with 2048 blocks
__global__ __launch_bounds__(256) void kgs(f16* __restrict__ h, int64_t items) {
  f16x8 out = {1, 2, 3, 4, 5, 6, 7, 8};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<f16x8*>(h)[i] = out;
}
// ---- row-gather micro-benchmark: 7 rows of 256 B per node (self, +-1, +-2, random partner) ----
template <int THREADS>
__global__ __launch_bounds__(THREADS) void kgather(const f16* __restrict__ h, const int* __restrict__ partner,
                                                   f16* __restrict__ z, int n) {
  for (int64_t item = (int64_t)blockIdx.x * THREADS + threadIdx.x; item < (int64_t)n * 16;
       item += (int64_t)gridDim.x * THREADS) {
    const int node = (int)(item >> 4), chunk = (int)(item & 15);
    int nb[6] = {node - 1, node + 1, node - 2, node + 2, partner[node], node};
    f16x8 v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      int s = nb[i] < 0 ? 0 : (nb[i] >= n ? n - 1 : nb[i]);
      v[i] = *reinterpret_cast<const f16x8*>(h + (size_t)s * 128 + chunk * 8);
    }
    f16x8 acc = v[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) acc += v[i];
    *reinterpret_cast<f16x8*>(z + (size_t)node * 128 + chunk * 8) = acc;
  }
}
int main() {
  const int n = 60000;
  float* x; f16 *w, *b, *h;
  CK(hipMalloc(&x, n * 7 * 4)); CK(hipMalloc(&w, 128 * 8 * 2)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&h, (size_t)n * 256));
  CK(hipMemset(x, 0, n * 7 * 4)); CK(hipMemset(w, 0, 2048)); CK(hipMemset(b, 0, 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = (n * 16 + 255) / 256;
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.2f us\n", name, ms * 5);
  };
  timeit("store only", [&] { k<0><<<blocks, 256>>>(x, w, b, h, n); });
  timeit("x loads + store", [&] { k<1><<<blocks, 256>>>(x, w, b, h, n); });
  timeit("x + w loads + store", [&] { k<2><<<blocks, 256>>>(x, w, b, h, n); });
  timeit("full, packed w", [&] { k2<0><<<blocks, 256>>>(x, w, b, h, n); });
  timeit("full, packed w, no x cvt", [&] { k2<1><<<blocks, 256>>>(x, w, b, h, n); });
  timeit("full, uniform w addr", [&] { k2<2><<<blocks, 256>>>(x, w, b, h, n); });
  timeit("full, LDS w", [&] { k3<<<blocks, 256>>>(x, w, b, h, n); });
  timeit("grid-stride store 2048 blk", [&] { kgs<<<2048, 256>>>(h, (int64_t)n * 16); });
  timeit("grid-stride store 512 blk", [&] { kgs<<<512, 256>>>(h, (int64_t)n * 16); });
  // same but to a fresh large buffer region each time (no reuse)
  f16* big; CK(hipMalloc(&big, (size_t)1 << 30));
  int rot = 0;
  timeit("store only, rotating 1GiB", [&] { k<0><<<blocks, 256>>>(x, w, b, big + (size_t)(rot++ % 64) * n * 128, n); });
  {
    const int n2 = 60000;
    f16 *hh, *zz; int* part;
    CK(hipMalloc(&hh, (size_t)n2 * 256)); CK(hipMalloc(&zz, (size_t)n2 * 256)); CK(hipMalloc(&part, n2 * 4));
    std::vector<int> pp(n2); for (int i = 0; i < n2; ++i) pp[i] = (int)(((long long)i * 7919 + 13) % n2);
    CK(hipMemcpy(part, pp.data(), n2 * 4, hipMemcpyHostToDevice)); CK(hipMemset(hh, 0, (size_t)n2 * 256));
    timeit("gather 60k x6 rows, 3750 blk x256", [&] { kgather<256><<<3750, 256>>>(hh, part, zz, n2); });
    timeit("gather 60k x6 rows, 2048 blk x256", [&] { kgather<256><<<2048, 256>>>(hh, part, zz, n2); });
    timeit("gather 60k x6 rows, 256 blk x512", [&] { kgather<512><<<256, 512>>>(hh, part, zz, n2); });
    timeit("gather 60k x6 rows, 512 blk x512", [&] { kgather<512><<<512, 512>>>(hh, part, zz, n2); });
    timeit("gather 60k x6 rows, 1024 blk x512", [&] { kgather<512><<<1024, 512>>>(hh, part, zz, n2); });
  }
  return 0;
}
