// Can a persistent kernel hand rows from one workgroup to another (different XCD) without a
// kernel boundary, and what does the hand-off cost?  (persistent encode with an
// ordered work queue.)
//
// 512 workgroups x 256 threads, R rounds.  Round r: every workgroup writes its own 32-row
// slice (256 B rows, value = f(round, row, chunk)) of buffer r&1, publishes with
// release + atomicAdd(done[r]), waits (bounded spin) until done[r] == grid, acquires, then
// reads 32 rows scattered over OTHER workgroups' slices — once with LDS-DMA
// (global_load_lds_dwordx4) and once with plain loads — and counts mismatches.  Buffers
// alternate, so a stale L2 line from round r-2 shows up as a mismatch.
//
//   hipcc -O3 --offload-arch=gfx950 tools/coherence_probe.hip -o tools/coherence_probe
//   tools/coherence_probe [mode]   mode 0: plain stores + __threadfence release/acquire
//                                   mode 1: plain stores, NO fences (expected to fail)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gbase, unsigned goff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds), "v"(goff), "s"(gbase) : "memory");
}

__device__ __forceinline__ unsigned pattern(int round, int row, int chunk, int k) {
  return (unsigned)round * 0x9E3779B9u ^ (unsigned)row * 0x85EBCA6Bu ^ (unsigned)(chunk * 4 + k) * 0xC2B2AE35u;
}

__global__ __launch_bounds__(256) void k_probe(unsigned* buf0, unsigned* buf1, int* done,
                                               int rounds, int mode, unsigned long long* stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wg = blockIdx.x, grid = gridDim.x;
  const int rows = grid * 32;
  unsigned long long bad_dma = 0, bad_ld = 0, spins = 0, t_wait = 0, t_fence = 0;
  for (int r = 0; r < rounds; ++r) {
    unsigned* buf = (r & 1) ? buf1 : buf0;
    // write own slice: 32 rows x 16 chunks x 16 B, thread -> (row = t >> 3 .. ), 2 chunks each
    for (int i = t; i < 32 * 16; i += 256) {
      const int row = wg * 32 + (i >> 4), chunk = i & 15;
      u32x4 v = {pattern(r, row, chunk, 0), pattern(r, row, chunk, 1), pattern(r, row, chunk, 2),
                 pattern(r, row, chunk, 3)};
      *reinterpret_cast<u32x4*>(buf + (size_t)row * 64 + chunk * 4) = v;
    }
    const unsigned long long f0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (t == 0) {
      if (mode == 0) __atomic_fetch_add(&done[r], 1, __ATOMIC_RELEASE);   // agent scope by default in HIP? use builtin below
      else atomicAdd(&done[r], 1);
    }
    const unsigned long long f1 = __builtin_amdgcn_s_memtime();
    // wait for everybody (bounded)
    if (t == 0) {
      int seen = 0;
      unsigned long long n = 0;
      while ((seen = __atomic_load_n(&done[r], mode == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED)) < grid && n < 4000000ull) ++n;
      spins += n;
      if (seen < grid) stats[4] = 1;   // gave up
    }
    __syncthreads();
    if (mode == 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);   // every wave: L1 (and L2) invalidate
    const unsigned long long f2 = __builtin_amdgcn_s_memtime();
    t_fence += f1 - f0;
    t_wait += f2 - f1;
    // read 32 rows of other workgroups: row j of workgroup (wg + 1 + 37 j) % grid
    for (int q = 0; q < 2; ++q) {
      const int g = wave * 2 + q;                 // 4 rows per DMA instruction
      const int j = 4 * g + (lane >> 4);
      const int src = ((wg + 1 + 37 * j) % grid) * 32 + j;
      dma16(buf, (unsigned)src * 256u + (unsigned)(lane & 15) * 16u, lds0 + g * 1024);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = t; i < 32 * 16; i += 256) {
      const int j = i >> 4, chunk = i & 15;
      const int src = ((wg + 1 + 37 * j) % grid) * 32 + j;
      const u32x4 a = *reinterpret_cast<const u32x4*>(smem + j * 256 + chunk * 16);
      const u32x4 b = *reinterpret_cast<const u32x4*>(buf + (size_t)src * 64 + chunk * 4);
      for (int k = 0; k < 4; ++k) {
        bad_dma += a[k] != pattern(r, src, chunk, k);
        bad_ld += b[k] != pattern(r, src, chunk, k);
      }
    }
    __syncthreads();
  }
  atomicAdd(&stats[0], bad_dma);
  atomicAdd(&stats[1], bad_ld);
  if (t == 0) {
    atomicAdd(&stats[2], t_fence);
    atomicAdd(&stats[3], t_wait);
    atomicAdd(&stats[5], spins);
  }
  (void)rows;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int grid = 512, rounds = 40;
  unsigned *b0, *b1;
  int* done;
  unsigned long long* stats;
  hipMalloc(&b0, (size_t)grid * 32 * 256);
  hipMalloc(&b1, (size_t)grid * 32 * 256);
  hipMalloc(&done, rounds * sizeof(int));
  hipMalloc(&stats, 8 * sizeof(unsigned long long));
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(done, 0, rounds * sizeof(int));
    hipMemset(stats, 0, 8 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k_probe<<<grid, 256, 8192>>>(b0, b1, done, rounds, mode, stats);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, stats, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d: %d rounds in %.1f us (%.2f us/round); mismatched words: dma %llu, loads %llu; "
           "per workgroup and round: publish %.0f cycles, wait+acquire %.0f cycles, spins %.0f%s\n",
           mode, rounds, ms * 1e3, ms * 1e3 / rounds, h[0], h[1],
           (double)h[2] / grid / rounds, (double)h[3] / grid / rounds, (double)h[5] / grid / rounds,
           h[4] ? "  [GAVE UP WAITING]" : "");
  }
  return 0;
}
