// Do vector instructions of one wave issue under the matrix instructions of the OTHER wave of the
// same SIMD?  A workgroup of 8 waves (2 per SIMD): waves 0..3 run a chain-free MFMA stream, waves
// 4..7 a stream of independent v_fma_f32; each alone, then together; then ONE wave per SIMD with
// k vector instructions between two MFMAs.  Shader cycles per wave (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bit 0: waves 0..3 do MFMAs; bit 1: waves 4..7 do VALU; kMix: every wave does MFMA + kMix VALU
template <int kMix>
__global__ __launch_bounds__(512) void k_overlap(float* out, int iters, int mode,
                                                 unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(1.0f + j * 0.01f); }
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j;
  const float one = 1.0f + blockIdx.x * 1e-9f;
  const bool mfma = kMix >= 0 ? true : (wave < 4 && (mode & 1));
  const bool valu = kMix >= 0 ? false : (wave >= 4 && (mode & 2));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
        if constexpr (kMix > 0) {
#pragma unroll
          for (int j = 0; j < kMix; ++j)
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j & 7]) : "v"(one));
        }
      }
    }
  }
  if (valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j]) : "v"(one));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < 4; ++c) for (int q = 0; q < 16; ++q) s += acc[c][q];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  float* out; unsigned long long* dcyc;
  const int blocks = 256, iters = 2000;
  CHECK(hipMalloc(&out, blocks * 512 * 4)); CHECK(hipMalloc(&dcyc, blocks * 8 * 8));
  unsigned long long cyc[256 * 8];
  auto report = [&](const char* name) {
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(cyc, dcyc, sizeof cyc, hipMemcpyDeviceToHost));
    double lo = 0, hi = 0;
    for (int i = 0; i < blocks; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += cyc[i * 8 + w];
    printf("%-46s waves 0-3: %8.0f cycles   waves 4-7: %8.0f cycles\n", name, lo / blocks / 4, hi / blocks / 4);
  };
  printf("per wave: %d MFMAs (32x32x16 f16, 4 chains) = %d cycles of one SIMD's matrix pipe; %d v_fma_f32\n",
         iters * 4, iters * 4 * 32, iters * 32);
  for (int rep = 0; rep < 2; ++rep) {
    k_overlap<-1><<<blocks, 512>>>(out, iters, 1, dcyc); report("MFMA stream alone (waves 0-3)");
    k_overlap<-1><<<blocks, 512>>>(out, iters, 2, dcyc); report("VALU stream alone (waves 4-7)");
    k_overlap<-1><<<blocks, 512>>>(out, iters, 3, dcyc); report("both, one wave of each per SIMD");
  }
  k_overlap<0><<<blocks, 512>>>(out, iters, 0, dcyc); report("all 8 waves: MFMA only");
  k_overlap<2><<<blocks, 512>>>(out, iters, 0, dcyc); report("all 8 waves: MFMA + 2 v_fma between");
  k_overlap<4><<<blocks, 512>>>(out, iters, 0, dcyc); report("all 8 waves: MFMA + 4 v_fma between");
  k_overlap<6><<<blocks, 512>>>(out, iters, 0, dcyc); report("all 8 waves: MFMA + 6 v_fma between");
  k_overlap<8><<<blocks, 512>>>(out, iters, 0, dcyc); report("all 8 waves: MFMA + 8 v_fma between");
  k_overlap<0><<<blocks, 256>>>(out, iters, 0, dcyc); report("4 waves (1 per SIMD): MFMA only");
  k_overlap<4><<<blocks, 256>>>(out, iters, 0, dcyc); report("4 waves: MFMA + 4 v_fma between");
  k_overlap<8><<<blocks, 256>>>(out, iters, 0, dcyc); report("4 waves: MFMA + 8 v_fma between");
  return 0;
}
