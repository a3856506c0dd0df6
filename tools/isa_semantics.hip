// Probes behind the third-generation layer kernel (gine_layer3.inc).  Answers, on the box:
//   1. v_fma_mixlo_f16: is the fp16 result the fp32 fma result rounded again (what
//      "R(fp32 op)" needs), or a single rounding of the exact value?
//   2. v_cvt_pk_f16_f32 == two v_cvt_f16_f32 (RNE)?
//   3. ds_bpermute_b32 cost per wave-instruction beside VALU work, 8 waves per CU.
//   4. MFMA with the A operand read from LDS per instruction (ds_read_b128 + mfma), 2
//      waves per SIMD: cycles per MFMA.
//   5. a 512-thread workgroup with 154 KB of dynamic LDS launches.
// Build: tools/build_tools.sh.  Run: tools/isa_semantics
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                              \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ __forceinline__ uint32_t mixlo_f32(float a, float b, float c) {   // fma(a,b,c) -> f16 lo
  uint32_t d = 0;
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(d) : "v"(a), "v"(b), "v"(c));
  return d & 0xFFFFu;
}
__device__ __forceinline__ uint32_t mixlo_h(uint32_t a16, float b, float c) {   // a is f16 (lo)
  uint32_t d = 0;
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]"
               : "+v"(d)
               : "v"(a16), "v"(b), "v"(c));
  return d & 0xFFFFu;
}
__device__ __forceinline__ uint32_t bits16(f16 v) {
  return (uint32_t) __builtin_bit_cast(unsigned short, v);
}

__global__ void k_mix(const float* a, const float* b, const float* c, int n, int* counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // (1a) R(a + b)
  const float s = a[i] + b[i];
  const uint32_t two_step = bits16((f16)s);
  const uint32_t mix = mixlo_f32(a[i], 1.0f, b[i]);
  if (two_step != mix) atomicAdd(&counts[0], 1);
  // (1b) R(fma(u16, alpha, shift))
  const f16 u = (f16)a[i];
  const float y = __builtin_fmaf((float)u, b[i], c[i]);
  const uint32_t two_step2 = bits16((f16)y);
  const uint32_t mix2 = mixlo_h(bits16(u), b[i], c[i]);
  if (two_step2 != mix2) atomicAdd(&counts[1], 1);
  // (2) packed convert
  uint32_t pk;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a[i]), "v"(b[i]));
  if ((pk & 0xFFFFu) != bits16((f16)a[i]) || (pk >> 16) != bits16((f16)b[i]))
    atomicAdd(&counts[2], 1);
}

// (3) bpermute beside VALU: per iteration 32 ds_bpermute + 64 v_pk ops, like one gather slot
__global__ __launch_bounds__(512) void k_bperm(uint32_t* out, int iters, unsigned long long* cyc,
                                                int with_perm) {
  uint32_t v[32];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 32; ++j) v[j] = 0x3C003C00u + lane * 7 + j;
  const int addr = ((lane + 1) & 63) * 4;
  uint32_t acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    uint32_t g[32];
#pragma unroll
    for (int j = 0; j < 32; ++j)
      g[j] = with_perm ? (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[j]) : v[j] + it;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(g[j]) : "v"(v[(j + 1) & 31]));
      asm volatile("v_pk_max_f16 %0, %0, 0" : "+v"(g[j]));
      asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(acc[j]) : "v"(g[j]));
      asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(acc[(j + 7) & 31]) : "v"(g[j]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < 32; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// (4) A operand from LDS per MFMA
__global__ __launch_bounds__(512) void k_mfma_lds(const f16* w, float* out, int iters,
                                                   unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // 64 KB of "weights": 64 fragments of 1 KB
  for (int i = threadIdx.x; i < 4096; i += blockDim.x)
    reinterpret_cast<f16x8*>(smem)[i] = reinterpret_cast<const f16x8*>(w)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f16x8 z[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) z[k][j] = (f16)(0.01f * ((lane + k + j) % 17));
  f32x16 total = {0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      f32x16 acc = {0};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const f16x8 a = reinterpret_cast<const f16x8*>(smem)[(b * 8 + ks) * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, z[ks], acc, 0, 0, 0);
      }
      total += acc;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += total[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// (4b) the same with the reads of block b+1 issued one per MFMA of block b (pinned order)
__global__ __launch_bounds__(512) void k_mfma_lds_pipelined(const f16* w, float* out, int iters,
                                                             unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x)
    reinterpret_cast<f16x8*>(smem)[i] = reinterpret_cast<const f16x8*>(w)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const f16x8* frag = reinterpret_cast<const f16x8*>(smem) + lane;
  f16x8 z[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) z[k][j] = (f16)(0.01f * ((lane + k + j) % 17));
  f32x16 total = {0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    f16x8 cur[8], nxt[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) cur[ks] = frag[ks * 64];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      f32x16 acc = {0};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (b < 7) nxt[ks] = frag[((b + 1) * 8 + ks) * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[ks], z[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) cur[ks] = nxt[ks];
      total += acc;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += total[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  // ---- 1, 2
  const int n = 1 << 24;
  std::vector<float> a(n), b(n), c(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const int kind = i & 3;
    if (kind == 0) {   // fp16 tie + dust: the double-rounding witness
      const int e = rand() % 20 - 10;
      const int m = rand() & 1023;
      a[i] = std::ldexp(1.0f + m / 1024.0f + 1.0f / 2048.0f, e);
      b[i] = std::ldexp(1.0f, e - 30) * ((rand() & 1) ? 1.f : -1.f);
      c[i] = b[i];
    } else {
      a[i] = std::ldexp((float)rand() / RAND_MAX - 0.5f, rand() % 12 - 6);
      b[i] = std::ldexp((float)rand() / RAND_MAX - 0.5f, rand() % 12 - 6);
      c[i] = std::ldexp((float)rand() / RAND_MAX - 0.5f, rand() % 12 - 6);
    }
  }
  float *da, *db, *dc;
  int* dcount;
  CHECK(hipMalloc(&da, n * 4));
  CHECK(hipMalloc(&db, n * 4));
  CHECK(hipMalloc(&dc, n * 4));
  CHECK(hipMalloc(&dcount, 16));
  CHECK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(dcount, 0, 16));
  k_mix<<<n / 256, 256>>>(da, db, dc, n, dcount);
  int counts[4];
  CHECK(hipMemcpy(counts, dcount, 16, hipMemcpyDeviceToHost));
  printf("mixlo(a,1,b) != R16(R32(a+b)): %d of %d\n", counts[0], n);
  printf("mixlo(u16,al,sh) != R16(R32(fma)): %d of %d\n", counts[1], n);
  printf("cvt_pk_f16_f32 != cvt_f16_f32 x2: %d of %d\n", counts[2], n);

  // ---- 3
  uint32_t* dout;
  unsigned long long* dcyc;
  CHECK(hipMalloc(&dout, 256 * 512 * 4));
  CHECK(hipMalloc(&dcyc, 256 * 8));
  for (int with = 0; with < 2; ++with) {
    k_bperm<<<256, 512>>>(dout, 200, dcyc, with);
    CHECK(hipDeviceSynchronize());
    unsigned long long cyc[256];
    CHECK(hipMemcpy(cyc, dcyc, sizeof cyc, hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < 256; ++i) m += cyc[i];
    printf("gather-slot loop (8 waves/CU, 128 VALU%s): %.0f cycles per iteration\n",
           with ? " + 32 ds_bpermute" : "", m / 256 / 200);
  }

  // ---- 4, 5
  f16* dw;
  CHECK(hipMalloc(&dw, 65536));
  std::vector<f16> hw(32768);
  for (auto& x : hw) x = (f16)(((rand() % 2001) - 1000) * 1e-3f);
  CHECK(hipMemcpy(dw, hw.data(), 65536, hipMemcpyHostToDevice));
  float* fo;
  CHECK(hipMalloc(&fo, 256 * 512 * 4));
  const int lds = 154 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mfma_lds),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  k_mfma_lds<<<256, 512, lds>>>(dw, fo, 50, dcyc);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  {
    unsigned long long cyc[256];
    CHECK(hipMemcpy(cyc, dcyc, sizeof cyc, hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < 256; ++i) m += cyc[i];
    printf("MFMA with A from LDS (512 threads, 154 KB LDS, 2 waves/SIMD): %.1f cycles per MFMA per wave\n",
           m / 256 / 50 / 64);
  }
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mfma_lds_pipelined),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int threads = 256; threads <= 512; threads += 256) {
    k_mfma_lds_pipelined<<<256, threads, lds>>>(dw, fo, 50, dcyc);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    unsigned long long cyc[256];
    CHECK(hipMemcpy(cyc, dcyc, sizeof cyc, hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < 256; ++i) m += cyc[i];
    printf("  reads pipelined one block ahead, %d waves per SIMD: %.1f cycles per MFMA per wave\n",
           threads / 256, m / 256 / 50 / 64);
  }
  return 0;
}
