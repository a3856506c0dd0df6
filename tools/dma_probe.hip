#include <hip/hip_runtime.h>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const f16* __restrict__ h, const int* __restrict__ idx, f16* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = idx[blockIdx.x * 16 + wave * 4 + (lane >> 4)];
  const char* src = reinterpret_cast<const char*>(h) + (size_t)row * 256 + (lane & 15) * 16;
  char* dst = smem + wave * 1024;   // wave-uniform base; lane*16 is implicit
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)src,
      (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f16x8 v = *reinterpret_cast<const f16x8*>(smem + wave * 1024 + lane * 16);
  *reinterpret_cast<f16x8*>(out + ((size_t)blockIdx.x * 16 + wave * 4 + (lane >> 4)) * 128 + (lane & 15) * 8) = v;
}
int main() {
  const int n = 4096, rows = 1024;
  f16 *h, *out; int* idx;
  hipMalloc(&h, n * 256); hipMalloc(&out, rows * 256); hipMalloc(&idx, rows * 4);
  f16* hh = new f16[n * 128]; for (int i = 0; i < n * 128; ++i) hh[i] = (f16)(float)((i * 7) % 1000);
  int* hi = new int[rows]; for (int i = 0; i < rows; ++i) hi[i] = (i * 37 + 11) % n;
  hipMemcpy(h, hh, n * 256, hipMemcpyHostToDevice); hipMemcpy(idx, hi, rows * 4, hipMemcpyHostToDevice);
  k<<<rows / 16, 256, 4096>>>(h, idx, out);
  f16* ho = new f16[rows * 128]; hipMemcpy(ho, out, rows * 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < rows; ++r) for (int c = 0; c < 128; ++c) if ((float)ho[r * 128 + c] != (float)hh[hi[r] * 128 + c]) ++bad;
  printf("global_load_lds gather: %d mismatches\n", bad);
  return bad != 0;
}
