// VALU issue rates that decide the shape of the gather: cycles per wave-instruction for a
// stream of independent instructions, one and two waves per SIMD (MI355X_MICROARCH.md gives
// v_fma_f32 = 2 cycles per wave64 on a SIMD-32 with two waves interleaved, 4 from one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

#define BODY8(INS)                                                            \
  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

template <int K>
__global__ __launch_bounds__(512) void k_rate(float* out, int iters, unsigned long long* cyc) {
  float a[8];
  unsigned b[8];
  for (int j = 0; j < 8; ++j) { a[j] = threadIdx.x * 0.001f + j; b[j] = 0x3C003C00u + threadIdx.x + j; }
  float one = 1.0f + blockIdx.x * 1e-9f;
  unsigned hone = 0x3C003C00u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define MIX(j) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a[j]) : "v"(b[j]));
#define PKADD(j) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(b[j]) : "v"(hone));
#define PKMAX(j) asm volatile("v_pk_max_f16 %0, %0, 0" : "+v"(b[j]));
#define FMA(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[j]) : "v"(one));
#define ADD(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(one));
#define CVT(j) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[j]) : "v"(b[j]));
#define CVTPK(j) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(b[j]) : "v"(a[j]), "v"(one));
#define DOT2C(j) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[j]) : "v"(b[j]), "v"(hone));
#define DOT2(j) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b[j]), "v"(hone));
#define PKMUL(j) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(b[j]) : "v"(hone));
#define PKFMA16(j) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(b[j]) : "v"(hone));
#define MIXLO(j) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %1" : "+v"(b[j]) : "v"(a[j]), "v"(one));
#define ADDSDWA(j) asm volatile("v_add_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[j]) : "v"(b[j]));
    if constexpr (K == 0) { BODY8(MIX) BODY8(MIX) }
    if constexpr (K == 1) { BODY8(PKADD) BODY8(PKADD) }
    if constexpr (K == 2) { BODY8(PKMAX) BODY8(PKMAX) }
    if constexpr (K == 3) { BODY8(FMA) BODY8(FMA) }
    if constexpr (K == 4) { BODY8(ADD) BODY8(ADD) }
    if constexpr (K == 5) { BODY8(CVT) BODY8(CVT) }
    if constexpr (K == 6) { BODY8(CVTPK) BODY8(CVTPK) }
    if constexpr (K == 7) { BODY8(DOT2C) BODY8(DOT2C) }
    if constexpr (K == 8) { BODY8(DOT2) BODY8(DOT2) }
    if constexpr (K == 9) { BODY8(PKMUL) BODY8(PKMUL) }
    if constexpr (K == 10) { BODY8(PKFMA16) BODY8(PKFMA16) }
    if constexpr (K == 11) { BODY8(MIXLO) BODY8(MIXLO) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int j = 0; j < 8; ++j) s += a[j] + (float)b[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K>
void run(const char* name, float* out, unsigned long long* dcyc) {
  for (int threads = 256; threads <= 512; threads += 256) {
    k_rate<K><<<256, threads>>>(out, 2000, dcyc);
    CHECK(hipDeviceSynchronize());
    unsigned long long cyc[256];
    CHECK(hipMemcpy(cyc, dcyc, sizeof cyc, hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < 256; ++i) m += cyc[i];
    printf("%-18s %d wave(s)/SIMD: %.2f cycles per instruction per wave\n", name, threads / 256,
           m / 256 / 2000 / 16);
  }
}

int main() {
  float* out; unsigned long long* dcyc;
  CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&dcyc, 256 * 8));
  run<3>("v_fma_f32", out, dcyc);
  run<4>("v_add_f32", out, dcyc);
  run<0>("v_fma_mix_f32", out, dcyc);
  run<1>("v_pk_add_f16", out, dcyc);
  run<2>("v_pk_max_f16", out, dcyc);
  run<9>("v_pk_mul_f16", out, dcyc);
  run<10>("v_pk_fma_f16", out, dcyc);
  run<5>("v_cvt_f32_f16", out, dcyc);
  run<6>("v_cvt_pk_f16_f32", out, dcyc);
  run<7>("v_dot2c_f32_f16", out, dcyc);
  run<8>("v_dot2_f32_f16", out, dcyc);
  run<11>("v_fma_mixlo_f16", out, dcyc);
  return 0;
}
