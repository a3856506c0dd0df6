// How long does a wave spend ISSUING global_load_lds_dwordx4 (LDS-DMA) instructions, and how
// long until the data has landed?  One workgroup of 4 waves per CU (grid 256), each wave
// issues K instructions of 4 random 256-byte rows back to back.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_rate.hip -o tools/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma16(const void* gbase, unsigned goff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(goff), "s"(gbase) : "memory");
}
template <int K>
__global__ __launch_bounds__(256) void k_rate(const char* rows, int n_rows, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  unsigned off[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const unsigned row = ((blockIdx.x * 977u + wave * 131u + k * 37u + (lane >> 4)) * 2654435761u) % (unsigned)n_rows;
    off[k] = row * 256u + (lane & 15) * 16u;
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int k = 0; k < K; ++k) dma16(rows, off[k], lds0 + (wave * K + k) * 1024);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    atomicAdd(&out[0], t1 - t0);
    atomicAdd(&out[1], t2 - t0);
  }
}
int main() {
  const int n_rows = 60000;
  char* rows; hipMalloc(&rows, (size_t)n_rows * 256);
  hipMemset(rows, 1, (size_t)n_rows * 256);
  unsigned long long* out; hipMalloc(&out, 16);
  auto run = [&](auto kernel, int K, const char* what) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(out, 0, 16);
      kernel<<<256, 256, 4 * K * 1024>>>(rows, n_rows, out);
      hipDeviceSynchronize();
      unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("%s K=%d: issue %.0f cycles per instruction, all landed after %.0f cycles\n", what, K,
             (double)h[0] / (256 * 4) / K, (double)h[1] / (256 * 4));
    }
  };
  run(k_rate<1>, 1, "one wg/CU");
  run(k_rate<4>, 4, "one wg/CU");
  run(k_rate<8>, 8, "one wg/CU");
  return 0;
}
