// What dense fp16 MFMA rate does this MI355X sustain?  Pure v_mfma_f32_32x32x16_f16 loops,
// no memory: 4 independent accumulator chains per wave, W waves per SIMD.  Prints TFLOP/s and
// the shader clock seen by the kernel (s_memtime is a fixed 100 MHz counter).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// kRandom: operands with random mantissas (unit-vector-like magnitudes) instead of a few small
// integers — the toggle rate of the datapath decides how far the part can hold its clock.
template <int kChains, bool kRandom = false>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, unsigned long long* clocks) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  if (kRandom) {
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 8; ++j) {
      h = h * 1664525u + 1013904223u; a[j] = (_Float16)(((int)(h >> 8) % 2001 - 1000) * 1e-4f);
      h = h * 1664525u + 1013904223u; b[j] = (_Float16)(((int)(h >> 8) % 2001 - 1000) * 1e-4f);
    }
  }
  f32x16 acc[kChains];
  for (int c = 0; c < kChains; ++c) for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < kChains; ++c)
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
  }
  const unsigned long long c1 = clock64();
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < kChains; ++c) for (int q = 0; q < 16; ++q) s += acc[c][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = c1 - c0; }
}

// The same FLOPs per wave as 16x16x32 instructions (4 x kChains independent accumulators of four
// registers): the part may hold a different clock on this shape.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int kChains, bool kRandom = false>
__global__ __launch_bounds__(256) void k_mfma16(float* out, int iters, unsigned long long* clocks) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  if (kRandom) {
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 8; ++j) {
      h = h * 1664525u + 1013904223u; a[j] = (_Float16)(((int)(h >> 8) % 2001 - 1000) * 1e-4f);
      h = h * 1664525u + 1013904223u; b[j] = (_Float16)(((int)(h >> 8) % 2001 - 1000) * 1e-4f);
    }
  }
  f32x4 acc[4 * kChains];
  for (int c = 0; c < 4 * kChains; ++c) for (int q = 0; q < 4; ++q) acc[c][q] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 4 * kChains; ++c)   // 4 x (16x16x32) = the MACs of one 32x32x16... x2: see flop below
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));   // in place: the
      // builtin's accumulators were allocated as a rotating, overlapping register window
  }
  const unsigned long long c1 = clock64();
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < 4 * kChains; ++c) for (int q = 0; q < 4; ++q) s += acc[c][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = c1 - c0; }
}
template <int kChains, bool kRandom = false>
void run16(int blocks, int iters, float* out, unsigned long long* clocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_mfma16<kChains, kRandom><<<blocks, 256>>>(out, iters / 10, clocks);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_mfma16<kChains, kRandom><<<blocks, 256>>>(out, iters, clocks);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost);
  const double flop = (double)blocks * 4 * iters * (4 * kChains) * 16384.0;
  printf("16x16x32 %s chains %d blocks %d (waves/SIMD %.1f): %.3f ms  %.0f TFLOP/s   memtime ticks %llu\n",
         kRandom ? "random " : "integer", 4 * kChains, blocks, blocks / 256.0, ms, flop / ms * 1e-9, h[0]);
}

template <int kChains, bool kRandom = false>
void run(int blocks, int iters, float* out, unsigned long long* clocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_mfma<kChains, kRandom><<<blocks, 256>>>(out, iters / 10, clocks);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_mfma<kChains, kRandom><<<blocks, 256>>>(out, iters, clocks);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost);
  const double flop = (double)blocks * 4 * iters * kChains * 32768.0;
  printf("%s chains %d blocks %d (waves/SIMD %.1f): %.3f ms  %.0f TFLOP/s   memtime ticks %llu  clock64 %llu\n",
         kRandom ? "random " : "integer", kChains, blocks, blocks / 256.0, ms, flop / ms * 1e-9, h[0], h[1]);
}

int main() {
  float* out; unsigned long long* clocks;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clocks, 16);
  const int iters = 20000;
  run<4>(256, iters, out, clocks);
  run<4>(512, iters, out, clocks);
  run<2>(512, iters, out, clocks);
  run<4>(1024, iters, out, clocks);
  run<1>(1024, iters, out, clocks);
  run<4>(512, iters * 10, out, clocks);
  run<4, true>(512, iters, out, clocks);
  run<4, true>(512, iters * 10, out, clocks);
  run<4, true>(512, iters * 40, out, clocks);
  run16<2>(512, iters * 10, out, clocks);
  run16<2, true>(512, iters, out, clocks);
  run16<2, true>(512, iters * 10, out, clocks);
  run16<2, true>(512, iters * 40, out, clocks);
  run16<2, true>(256, iters * 10, out, clocks);
  run<4, true>(256, iters * 10, out, clocks);
  return 0;
}
