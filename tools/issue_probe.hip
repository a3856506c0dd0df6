// How many instructions does one SIMD issue per cycle, by the number of waves it holds and by the
// mix?  The layer kernels land at the same ~3.3-3.6 k cycles per 32-node tile and CU with two
// waves per SIMD (k_gine_layer_q, _w) and with three (k_gine_layer_x): this probe asks whether that
// is an issue-slot bound.  Every wave runs the same unrolled stream `iters` times; 1, 2, 3, 4 waves
// per SIMD (256 .. 1,024 threads, one workgroup per CU).  Shader cycles (s_memtime) of the slowest
// wave / instructions of ONE SIMD (all its waves) = cycles per issued instruction.
//
//   hipcc -O3 --offload-arch=gfx950 tools/issue_probe.hip -o tools/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define FMA(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[j]) : "v"(one));
#define PKADD(j) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(b[j]) : "v"(hone));
#define MIX(j) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a[j]) : "v"(b[j]));
#define NOP asm volatile("s_nop 0");
#define SALU asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg) : : "scc");
#define LDSR(j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j]) : "v"(lds_addr), "n"((j) * 1024));
#define WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(n));

// kind: what one iteration of the stream is; `count` = instructions of one iteration (host side)
template <int kKind>
__global__ __launch_bounds__(1024) void k_issue(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  float a[8];
  unsigned b[8];
  u32x4 ld[4];
  f16x8 fa, fb;
  f32x16 acc[2];
  for (int j = 0; j < 8; ++j) {
    a[j] = threadIdx.x * 0.001f + j, b[j] = 0x3C003C00u + threadIdx.x + j;
    fa[j] = (_Float16)(threadIdx.x * 0.001f + j), fb[j] = (_Float16)(1.0f + j * 0.01f);
  }
  for (int c = 0; c < 2; ++c) for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
  for (int j = 0; j < 4; ++j) ld[j] = u32x4{0, 0, 0, 0};
  reinterpret_cast<u32x4*>(lds)[threadIdx.x] = u32x4{threadIdx.x, 1, 2, 3};
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16;
  const float one = 1.0f + blockIdx.x * 1e-9f;
  const unsigned hone = 0x3C003C00u;
  unsigned sreg = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (kKind == 0) {          // 16 independent v_fma_f32
      FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7)
      FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7)
    } else if constexpr (kKind == 1) {   // 16 packed-half / mix instructions (the kernels' own)
      PKADD(0) MIX(0) PKADD(1) MIX(1) PKADD(2) MIX(2) PKADD(3) MIX(3)
      PKADD(4) MIX(4) PKADD(5) MIX(5) PKADD(6) MIX(6) PKADD(7) MIX(7)
    } else if constexpr (kKind == 2) {   // 12 vector + 4 s_nop 0
      FMA(0) FMA(1) FMA(2) NOP FMA(3) FMA(4) FMA(5) NOP FMA(6) FMA(7) FMA(0) NOP FMA(1) FMA(2) FMA(3) NOP
    } else if constexpr (kKind == 3) {   // 12 vector + 4 scalar ALU
      FMA(0) FMA(1) FMA(2) SALU FMA(3) FMA(4) FMA(5) SALU FMA(6) FMA(7) FMA(0) SALU FMA(1) FMA(2) FMA(3) SALU
    } else if constexpr (kKind == 4) {   // 12 vector + 4 ds_read_b128 + one wait
      FMA(0) FMA(1) FMA(2) LDSR(0) FMA(3) FMA(4) FMA(5) LDSR(1) FMA(6) FMA(7) FMA(0) LDSR(2) FMA(1) FMA(2) FMA(3) LDSR(3)
      WAIT(0)
    } else if constexpr (kKind == 5) {   // a pair-pipeline step: 2 MFMAs (two chains) + 10 vector + 2 LDS reads + wait
      LDSR(0) LDSR(1) WAIT(2)
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[1], 0, 0, 0);
      PKADD(0) MIX(0) PKADD(1) MIX(1) PKADD(2) MIX(2) PKADD(3) MIX(3) PKADD(4) MIX(4)
    } else if constexpr (kKind == 6) {   // a block-pipeline step: 1 MFMA (one chain) + 5 vector + 1 LDS read + wait
      LDSR(0) WAIT(1)
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[0], 0, 0, 0);
      PKADD(0) MIX(0) PKADD(1) MIX(1) PKADD(2)
    } else if constexpr (kKind == 7) {   // MFMAs only, two chains
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[1], 0, 0, 0);
    } else if constexpr (kKind == 8) {   // gather-like: per slot 2 LDS reads, 8 packed-half, 1 MFMA (one chain)
      LDSR(0) LDSR(1) WAIT(2)
      PKADD(0) PKADD(1) PKADD(2) PKADD(3) PKADD(4) PKADD(5) PKADD(6) PKADD(7)
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[0], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)sreg;
  for (int c = 0; c < 2; ++c) for (int q = 0; q < 16; ++q) s += acc[c][q];
  for (int j = 0; j < 8; ++j) s += a[j] + (float)b[j];
  for (int j = 0; j < 4; ++j) s += (float)ld[j][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int kKind>
void run(const char* name, int count, int mfmas, float* out, unsigned long long* dcyc) {
  const int blocks = 256, iters = 4000;
  for (int waves = 4; waves <= 16; waves += 4) {
    double best = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
      k_issue<kKind><<<blocks, 64 * waves>>>(out, iters, dcyc);
      CHECK(hipDeviceSynchronize());
      static unsigned long long cyc[256 * 16];
      CHECK(hipMemcpy(cyc, dcyc, sizeof cyc, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int i = 0; i < blocks; ++i)
        for (int w = 0; w < waves; ++w) worst = cyc[i * 16 + w] > worst ? (double)cyc[i * 16 + w] : worst;
      best = worst < best ? worst : best;
    }
    const double per_simd = (double)iters * (waves / 4);   // iterations one SIMD runs
    printf("%-58s %d wave(s)/SIMD: %6.1f cycles per iteration and SIMD, %5.2f per instruction", name,
           waves / 4, best / per_simd, best / per_simd / count);
    if (mfmas) printf(", matrix pipe %4.1f %%", 100.0 * mfmas * 32 / (best / per_simd));
    printf("\n");
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  float* out; unsigned long long* dcyc;
  CHECK(hipMalloc(&out, 256 * 1024 * 4)); CHECK(hipMalloc(&dcyc, 256 * 16 * 8));
  if (only < 0 || only == 0) run<0>("16 v_fma_f32", 16, 0, out, dcyc);
  if (only < 0 || only == 1) run<1>("8 v_pk_add_f16 + 8 v_fma_mix_f32", 16, 0, out, dcyc);
  if (only < 0 || only == 2) run<2>("12 v_fma_f32 + 4 s_nop 0", 16, 0, out, dcyc);
  if (only < 0 || only == 3) run<3>("12 v_fma_f32 + 4 s_add_u32", 16, 0, out, dcyc);
  if (only < 0 || only == 4) run<4>("12 v_fma_f32 + 4 ds_read_b128 + wait", 17, 0, out, dcyc);
  if (only < 0 || only == 7) run<7>("2 MFMA (two chains)", 2, 2, out, dcyc);
  if (only < 0 || only == 5) run<5>("pair step: 2 MFMA + 10 vector + 2 ds_read_b128 + wait", 15, 2, out, dcyc);
  if (only < 0 || only == 6) run<6>("block step: 1 MFMA + 5 vector + 1 ds_read_b128 + wait", 8, 1, out, dcyc);
  if (only < 0 || only == 8) run<8>("gather slot: 1 MFMA + 8 packed-half + 2 ds_read_b128 + wait", 12, 1, out, dcyc);
  return 0;
}
