// What does a third (or fourth) wave per SIMD buy the update MLP of the fused layer?
//
// The layer kernels' own device code (tools/mlp_probe.hip includes gine_f16.hip with
// GFY_PROBE_BUILD: the pipelines, epilogue slices and LayerNorm are the product's, not copies) on
// synthetic operands, weights RESIDENT in LDS, no gather, no weight streaming, no barrier inside
// the loop: every wave walks `iters` 32-node tiles through
//     products (bias MFMAs, round, BatchNorm fma_mix, round, relu, second product, round)
//     -> LayerNorm + residual -> store,
// the next tile's input being this tile's output.  Shapes:
//   pair     mlp_pipeline_on: two accumulator chains, the previous block PAIR drained beside them
//            (~250 registers: 8 waves per CU = 2 per SIMD) — what k_gine_layer_f16/_q/_w run
//   block    block_mlp_pipeline: one chain, the previous BLOCK drained in its gaps (<= 168
//            registers), with 4, 8 and 12 waves per CU (1, 2, 3 per SIMD)
//   tile16   16-node tiles on v_mfma_f32_16x16x32_f16 (<= 128 registers), 8, 12 and 16 waves:
//            the same arithmetic per node (synthetic weights in that shape's fragment order)
// Reported per shape: shader cycles per 32 nodes and SIMD (s_memtime, slowest wave of the chip),
// and the share of those cycles the SIMD's matrix pipe is busy (MFMAs x 32 or 16 cycles).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/mlp_probe.hip -o tools/mlp_probe
#define GFY_PROBE_BUILD
#include "../ginfinity_amd/csrc/gine_f16.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace gfy {
void set_error(const char*, ...) {}
void clear_error() {}
}  // namespace gfy
using namespace gfy;

#define CHECK(x)                                                         \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

// [W0 | W1] and the constant image into LDS, once; every wave's first tile into registers
template <int kWaves>
__device__ __forceinline__ void probe_prologue(char* smem, uint32_t lds0, const f16* w_image,
                                               const void* cimage, int wave, int lane) {
  for (int p = wave; p < 128; p += kWaves)
    dma16(w_image, (uint32_t)p * 1024u + (uint32_t)lane * 16u, lds0 + (uint32_t)p * 1024u);
  for (int i = threadIdx.x; i < kLdsImageBytes / 16; i += 64 * kWaves)
    reinterpret_cast<u32x4*>(smem + kLdsImage)[i] = reinterpret_cast<const u32x4*>(cimage)[i];
  dma_wait_all();
  lds_barrier();
}

template <int kWaves, bool kBlock>
__global__ __launch_bounds__(64 * kWaves) void k_probe(const f16* w_image, const void* cimage,
                                                       const f16* hin, f16* hout, int iters,
                                                       unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  probe_prologue<kWaves>(smem, lds0, w_image, cimage, wave, lane);
  const int tile = blockIdx.x * kWaves + wave;
  const char* own = reinterpret_cast<const char*>(hin) +
                    ((size_t)(tile * 32 + (lane & 31)) * 256 + (lane >> 5) * 16);
  char* outp = reinterpret_cast<char*>(hout) +
               ((size_t)(tile * 32 + (lane & 31)) * 256 + (lane >> 5) * 16);
  f16x8 z[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) z[ks] = *reinterpret_cast<const f16x8*>(own + 32 * ks);
  HeadF16 no_head{};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    f16x8 wrow[8], hn[8], h[8];
    if constexpr (kBlock) {
      const int lane_c = opaque(lane);
      f16x8 v[16];
      NoBlockHooks hooks;
      block_mlp_pipeline<ResidentLayout>(lds0, lane_c, lane_c >> 5, z, v, wrow, hooks);
      // the residual rows come back from memory (three waves per SIMD have no registers to keep
      // them across the products)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) h[ks] = *reinterpret_cast<const f16x8*>(own + 32 * ks);
    } else {
      const int lane_c = opaque(lane);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) h[ks] = z[ks];
      mlp_pipeline<false>(smem, lds0, wave, lane_c, lane_c >> 5, z, wrow, no_head);
    }
    const int lane_d = opaque(lane);
    layer_norm_residual<true>(wrow, h, hn, smem + kLdsGamma, smem + kLdsBeta, lane_d >> 5);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) *reinterpret_cast<f16x8*>(outp + 32 * ks) = hn[ks];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) z[ks] = hn[ks] * (f16)0.25f;   // keeps the walk bounded
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * kWaves + wave] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------
// merged: the NEXT tile's gather-sum (synthetic stage: 5 slots x 8 chunks = 40 units of two
// ds_read_b128, add + relu in packed fp16 and one selection MFMA; per chunk the rounding tail)
// issued inside the block pipeline's steps of the CURRENT tile (two units per seven steps) instead
// of as a phase of its own in front of them.  Same instructions either way.
// ---------------------------------------------------------------------------------------------
struct GatherState {
  const char* stage;      // fake stage + edge table: 16 KB between the weights and the image
  uint32_t saddr[5], toff[5], own;
  f16x8 sel;
  f32x16 acc;
  f16x8 znext[8];
  f16 scale;
};
template <int kUnit>
__device__ __forceinline__ void gather_unit(GatherState& g) {
  constexpr int chunk = kUnit / 5, slot = kUnit % 5;
  const f16x8 hv = *reinterpret_cast<const f16x8*>(g.stage + (g.saddr[slot] ^ (32u * chunk)));
  const f16x8 ev = *reinterpret_cast<const f16x8*>(g.stage + 9216 + g.toff[slot] + 32 * chunk);
  const f32x16 zero = {0};
  g.acc = mfma(g.sel, __builtin_elementwise_max(hv + ev, zero8()), slot == 0 ? zero : g.acc);
  if constexpr (slot == 4) {
    const f16x8 hk = *reinterpret_cast<const f16x8*>(g.stage + (g.own ^ (32u * chunk)));
    f16x8 agg;
#pragma unroll
    for (int j = 0; j < 8; ++j) agg[j] = (f16)g.acc[j];
    g.znext[chunk] = hk * g.scale + agg;
  }
}
template <int... kU>
__device__ __forceinline__ void gather_units(std::integer_sequence<int, kU...>, GatherState& g) {
  (gather_unit<kU>(g), ...);
}
struct MergeHooks {   // units 2 i, 2 i + 1 at the steps 7 i, 7 i + 3 (i = 0..19)
  GatherState& g;
  template <int kT>
  __device__ __forceinline__ void before_mlp_step() const {
    if constexpr (kT < 140 && (kT % 7 == 0 || kT % 7 == 3)) gather_unit<2 * (kT / 7) + (kT % 7 == 3)>(g);
  }
  template <int kT> __device__ __forceinline__ void before_head_step() const {}
};

template <int kWaves, bool kMerged>
__global__ __launch_bounds__(64 * kWaves) void k_probe_merge(const f16* w_image, const void* cimage,
                                                             const f16* hin, f16* hout, int iters,
                                                             unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  probe_prologue<kWaves>(smem, lds0, w_image, cimage, wave, lane);
  for (int i = t; i < 1024; i += 64 * kWaves)   // the fake stage: any finite halves
    reinterpret_cast<u32x4*>(smem + kLdsWeightBytes)[i] = reinterpret_cast<const u32x4*>(hin)[i];
  const int tile = blockIdx.x * kWaves + wave;
  const char* own = reinterpret_cast<const char*>(hin) +
                    ((size_t)(tile * 32 + (lane & 31)) * 256 + (lane >> 5) * 16);
  char* outp = reinterpret_cast<char*>(hout) +
               ((size_t)(tile * 32 + (lane & 31)) * 256 + (lane >> 5) * 16);
  f16x8 z[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) z[ks] = *reinterpret_cast<const f16x8*>(own + 32 * ks);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    f16x8 wrow[8], hn[8], h[8];
    GatherState g;
    {
      const int lane_g = opaque(lane), r = lane_g & 31, hq = lane_g >> 5;
      g.stage = smem + kLdsWeightBytes;
      g.scale = (f16)1.0f;
      g.own = (uint32_t)r * 128u + (((((uint32_t)r >> 1) & 7u) ^ (uint32_t)hq) << 4);
#pragma unroll
      for (int sl = 0; sl < 5; ++sl) {   // rows r - 2 .. r + 2 of a 72-row half stage, types 0..4
        const uint32_t q = (uint32_t)((r + 70 + sl) % 72);
        g.saddr[sl] = q * 128u + ((((q >> 1) & 7u) ^ (uint32_t)hq) << 4);
        g.toff[sl] = (uint32_t)sl * 256u + (uint32_t)hq * 16u;
      }
      g.sel = zero8();
      const int want_half = (r >> 2) & 1, want_j = (r & 3) + 4 * (r >> 3);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        if (r < 16 && hq == want_half && jj == want_j) g.sel[jj] = (f16)1.0f;
    }
    {
      const int lane_c = opaque(lane);
      f16x8 v[16];
      if constexpr (kMerged) {
        MergeHooks hooks{g};
        block_mlp_pipeline<ResidentLayout>(lds0, lane_c, lane_c >> 5, z, v, wrow, hooks);
      } else {
        gather_units(std::make_integer_sequence<int, 40>{}, g);
        NoBlockHooks hooks;
        block_mlp_pipeline<ResidentLayout>(lds0, lane_c, lane_c >> 5, z, v, wrow, hooks);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) h[ks] = *reinterpret_cast<const f16x8*>(own + 32 * ks);
    const int lane_d = opaque(lane);
    layer_norm_residual<true>(wrow, h, hn, smem + kLdsGamma, smem + kLdsBeta, lane_d >> 5);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) *reinterpret_cast<f16x8*>(outp + 32 * ks) = hn[ks];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)   // the next tile's input: the walk's output, and the gathered sums
      z[ks] = hn[ks] * (f16)0.25f + g.znext[ks] * (f16)0.001f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * kWaves + wave] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------
// tile16: the same update MLP on 16-node tiles with v_mfma_f32_16x16x32_f16.
//   A (weights) 16 rows x 32 k: lane l holds A[l & 15][8 (l >> 4) + j]; B (activations) 32 k x 16
//   nodes: lane l holds B[8 (l >> 4) + j][l & 15]; D 16 x 16: lane l holds D[4 (l >> 4) + reg][l & 15].
// Two result blocks of 16 channels are one B operand of 32 (chained k order: element j of lane
// group g is channel 16 (j >> 2) + 4 g + (j & 3) of the pair) — weights here are synthetic, so the
// k permutation needs no packing.  Per tile: U 16 blocks x (4 k-steps + bias), W 8 blocks x
// (8 k-steps + bias); per block one epilogue slice (the same ten / two instructions per four
// values as the 32-node shapes); LayerNorm row sums by a ones MFMA, the four lane groups'
// squares by one f32 MFMA.
// ---------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16(f16x8 a, f16x8 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
constexpr int kT16Ring = 4;
struct T16Step {
  bool mfma, bias;
  int product, block, kstep, piece;
  int slice, sl_block;   // 1 BatchNorm of U block, 2 round of W block
};
// steps: U block b: 5 b + k (k = 4: bias), 80 steps; W block o: 80 + 9 o + k (k = 8: bias), 72
// steps; the slice of the previous block at step 1 of a block; the last W block's at the tail step
constexpr int kT16Steps = 80 + 72 + 1;
__host__ __device__ constexpr T16Step t16_step(int t) {
  T16Step d{};
  d.piece = -1;
  if (t < 80) {
    d.mfma = true, d.product = 0, d.block = t / 5, d.kstep = t % 5, d.bias = d.kstep == 4;
    if (!d.bias) d.piece = 4 * d.block + d.kstep;
    if (t >= 5 && d.kstep == 1) d.slice = 1, d.sl_block = d.block - 1;
  } else if (t < 152) {
    const int u = t - 80;
    d.mfma = true, d.product = 1, d.block = u / 9, d.kstep = u % 9, d.bias = d.kstep == 8;
    if (!d.bias) d.piece = 64 + 8 * d.block + d.kstep;
    if (u == 1) d.slice = 1, d.sl_block = 15;
    else if (u >= 9 && d.kstep == 1) d.slice = 2, d.sl_block = d.block - 1;
  } else {
    d.slice = 2, d.sl_block = 7;
  }
  return d;
}
struct T16Pipe {
  f32x4v acc[2];
  f16x8 ring[kT16Ring];
  f32x4 al, sh;
  uint32_t frag_base, alpha_base, b0_base, one_at_k0, bias_row;
};
template <int kT>
__device__ __forceinline__ void t16_pipe_step(T16Pipe& m, const f16x8 (&z)[4], f16x8 (&v)[8],
                                              f16x8 (&wrow)[4]) {
  constexpr T16Step d = t16_step(kT);
  constexpr bool kHasNext = kT + 1 < kT16Steps;
  constexpr T16Step nx = t16_step(kHasNext ? kT + 1 : kT);
  constexpr int kAhead = d.piece >= 0 ? d.piece + kT16Ring - 1 : -1;
  constexpr bool kReadFrag = kAhead >= 0 && kAhead < 128;
  constexpr bool kReadConst = kHasNext && nx.slice == 1;
  constexpr bool kReadBias = kHasNext && nx.mfma && nx.bias;
  constexpr int kReads = (kReadFrag ? 1 : 0) + (kReadConst ? 2 : 0) + (kReadBias ? 1 : 0);
  if constexpr (kReadFrag)
    lds_read_b128<((kReadFrag ? kAhead : 0) & 63) * 1024>(
        m.ring[kAhead % kT16Ring], m.frag_base + (uint32_t)(kAhead >> 6) * 65536u);
  if constexpr (kReadConst)
    lds_read2_b128<nx.sl_block * 64, 1024 + nx.sl_block * 64>(m.al, m.sh, m.alpha_base);
  if constexpr (kReadBias)
    lds_read_u16<(nx.product ? 512 : 0) + nx.block * 32>(m.bias_row, m.b0_base);
  lds_wait_keep<kReads>();
  if constexpr (d.piece >= 0) landed(m.ring[(d.piece >= 0 ? d.piece : 0) % kT16Ring]);
  if constexpr (d.slice == 1) landed(m.al), landed(m.sh);
  if constexpr (d.bias) landed(m.bias_row);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (d.mfma && !d.bias) {
    const f32x4v zero = {0, 0, 0, 0};
    f16x8 operand;
    if constexpr (d.product) operand = v[d.kstep];
    else operand = z[d.kstep];
    f32x4v& acc = m.acc[d.block & 1];
    acc = mfma16(m.ring[d.piece % kT16Ring], operand, d.kstep == 0 ? zero : acc);
  } else if constexpr (d.bias) {
    constexpr int spent = d.product ? 64 + 8 * d.block + 7 : 4 * d.block + 3;
    u32x4 ones = {m.one_at_k0, opaque_zero(), opaque_zero(), opaque_zero()};
    u32x4 a = __builtin_bit_cast(u32x4, m.ring[spent % kT16Ring]);
    a[0] = m.bias_row;
    f32x4v& acc = m.acc[d.block & 1];
    acc = mfma16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, ones), acc);
  }
  if constexpr (d.slice == 1) {   // four values: round, fma_mix, round, relu -> half a B operand
    const f32x4v& acc = m.acc[d.sl_block & 1];
    const uint32_t u01 = cvt_pk_f16(acc[0], acc[1]), u23 = cvt_pk_f16(acc[2], acc[3]);
    const float y0 = fma_half<0>(u01, m.al[0], m.sh[0]), y1 = fma_half<1>(u01, m.al[1], m.sh[1]);
    const float y2 = fma_half<0>(u23, m.al[2], m.sh[2]), y3 = fma_half<1>(u23, m.al[3], m.sh[3]);
    u32x4& dst = reinterpret_cast<u32x4&>(v[d.sl_block >> 1]);
    dst[2 * (d.sl_block & 1)] = pk_relu(cvt_pk_f16(y0, y1));
    dst[2 * (d.sl_block & 1) + 1] = pk_relu(cvt_pk_f16(y2, y3));
  } else if constexpr (d.slice == 2) {
    const f32x4v& acc = m.acc[d.sl_block & 1];
    u32x4& dst = reinterpret_cast<u32x4&>(wrow[d.sl_block >> 1]);
    dst[2 * (d.sl_block & 1)] = cvt_pk_f16(acc[0], acc[1]);
    dst[2 * (d.sl_block & 1) + 1] = cvt_pk_f16(acc[2], acc[3]);
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int... kT>
__device__ __forceinline__ void t16_pipe_steps(std::integer_sequence<int, kT...>, T16Pipe& m,
                                               const f16x8 (&z)[4], f16x8 (&v)[8],
                                               f16x8 (&wrow)[4]) {
  (t16_pipe_step<kT>(m, z, v, wrow), ...);
}

template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) void k_probe16(const f16* w_image, const void* cimage,
                                                         const f16* hin, f16* hout, int iters,
                                                         unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  probe_prologue<kWaves>(smem, lds0, w_image, cimage, wave, lane);
  const int tile = blockIdx.x * kWaves + wave;   // 16 nodes: row = node, 64-byte piece per lane group
  const size_t at = (size_t)(tile * 16 + (lane & 15)) * 256 + (lane >> 4) * 16;
  const char* own = reinterpret_cast<const char*>(hin) + at;
  char* outp = reinterpret_cast<char*>(hout) + at;
  f16x8 z[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) z[ks] = *reinterpret_cast<const f16x8*>(own + 64 * ks);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    f16x8 wrow[4], hn[4], h[4];
    {
      const int lane_c = opaque(lane), g = lane_c >> 4;
      T16Pipe m;
      m.frag_base = lds0 + (uint32_t)lane_c * 16u;
      m.alpha_base = lds0 + (uint32_t)kLdsAlpha + (uint32_t)g * 16u;
      m.b0_base = lds0 + (uint32_t)kLdsB0 + ((uint32_t)lane_c & 15u) * 2u;
      m.one_at_k0 = g ? 0u : 0x3C00u;
      lds_read_b128<0>(m.ring[0], m.frag_base);
      lds_read_b128<1024>(m.ring[1], m.frag_base);
      lds_read_b128<2048>(m.ring[2], m.frag_base);
      f16x8 v[8];
      t16_pipe_steps(std::make_integer_sequence<int, kT16Steps>{}, m, z, v, wrow);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) h[ks] = *reinterpret_cast<const f16x8*>(own + 64 * ks);
    {   // LayerNorm + residual, 32 values per lane; moments over the node's four lane groups
      const int lane_d = opaque(lane), g = lane_d >> 4;
      f32x4v sums = {0, 0, 0, 0};
      const f16 one = (f16)opaque_one();
      const f16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sums = mfma16(ones, wrow[ks], sums);
      float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 w4 = __builtin_bit_cast(u32x4, wrow[ks]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sq[q] = fma_hh_acc<0>(w4[q], sq[q]);
          sq[q] = fma_hh_acc<1>(w4[q], sq[q]);
        }
      }
      f32x4v sqsum = {0, 0, 0, 0};
      sqsum = __builtin_amdgcn_mfma_f32_16x16x4f32(opaque_one(), (sq[0] + sq[1]) + (sq[2] + sq[3]),
                                                   sqsum, 0, 0, 0);
      const float mean = sums[0] * (1.0f / kHidden);
      const float var = __builtin_fmaf(-mean, mean, sqsum[0] * (1.0f / kHidden));
      const float rstd = fast_rsqrt(var + 1e-5f);
      const float offset = -rstd * mean;
      const char* gamma = smem + kLdsGamma + g * 64, * beta = smem + kLdsBeta + g * 64;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 w4 = __builtin_bit_cast(u32x4, wrow[ks]);
        const u32x4 gm = *reinterpret_cast<const u32x4*>(gamma + 16 * ks);
        const u32x4 bt = *reinterpret_cast<const u32x4*>(beta + 16 * ks);
        u32x4 y;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a0 = fma_half<0>(w4[q], rstd, offset), a1 = fma_half<1>(w4[q], rstd, offset);
          y[q] = cvt_pk_f16(fma_f32_hh<0>(a0, gm[q], bt[q]), fma_f32_hh<1>(a1, gm[q], bt[q]));
        }
        hn[ks] = h[ks] + __builtin_bit_cast(f16x8, y);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) *reinterpret_cast<f16x8*>(outp + 64 * ks) = hn[ks];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) z[ks] = hn[ks] * (f16)0.25f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * kWaves + wave] = t1 - t0;
}

__global__ void k_lds_only(int* out) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 1;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = smem[1];
}

static float frand(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int blocks = 256, max_waves = 16;
  uint32_t seed = 12345u;
  std::vector<f16> w(64 * 1024), hin((size_t)blocks * max_waves * 32 * 128);
  for (auto& x : w) x = (f16)(frand(seed) * 0.2f);
  for (auto& x : hin) x = (f16)(frand(seed) * 2.0f);
  std::vector<char> image(kLdsImageBytes, 0);
  {
    float* alpha = reinterpret_cast<float*>(image.data() + (kLdsAlpha - kLdsImage));
    float* shift = reinterpret_cast<float*>(image.data() + (kLdsShift - kLdsImage));
    f16* b0 = reinterpret_cast<f16*>(image.data() + (kLdsB0 - kLdsImage));
    f16* b1 = reinterpret_cast<f16*>(image.data() + (kLdsB1 - kLdsImage));
    f16* gm = reinterpret_cast<f16*>(image.data() + (kLdsGamma - kLdsImage));
    f16* bt = reinterpret_cast<f16*>(image.data() + (kLdsBeta - kLdsImage));
    for (int i = 0; i < 256; ++i) alpha[i] = 1.0f + frand(seed), shift[i] = frand(seed), b0[i] = (f16)(frand(seed) * 0.2f);
    for (int i = 0; i < 128; ++i) b1[i] = (f16)(frand(seed) * 0.2f), gm[i] = (f16)(1.0f + 0.2f * frand(seed)), bt[i] = (f16)(0.2f * frand(seed));
  }
  f16 *d_w, *d_hin, *d_hout;
  void* d_image;
  unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_w, w.size() * 2));
  CHECK(hipMalloc(&d_hin, hin.size() * 2));
  CHECK(hipMalloc(&d_hout, hin.size() * 2));
  CHECK(hipMalloc(&d_image, image.size()));
  CHECK(hipMalloc(&d_cyc, blocks * max_waves * 8));
  CHECK(hipMemcpy(d_w, w.data(), w.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_hin, hin.data(), hin.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_image, image.data(), image.size(), hipMemcpyHostToDevice));

  {   // how much LDS may a workgroup ask for and still share a CU with two / three others?
    printf("workgroups of 256 threads per CU by dynamic LDS (occupancy API):");
    for (int bytes : {40960, 53248, 54272, 54528, 54608, 54784, 55296, 65536, 81920}) {
      int n = 0;
      CHECK(hipFuncSetAttribute((const void*)k_lds_only, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_lds_only, 256, bytes));
      printf("  %d B -> %d", bytes, n);
    }
    printf("\n");
  }

  std::vector<unsigned long long> cyc(blocks * max_waves);
  auto report = [&](const char* name, int waves, int nodes_per_tile, int mfma_per_tile,
                    int cycles_per_mfma, float ms) {
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0, sum = 0;
    for (int i = 0; i < blocks * waves; ++i) {
      sum += (double)cyc[i];
      if ((double)cyc[i] > worst) worst = (double)cyc[i];
    }
    const double tiles32_per_simd = (double)iters * waves / 4 * nodes_per_tile / 32.0;
    const double per32 = worst / tiles32_per_simd;
    const double busy = (double)mfma_per_tile * cycles_per_mfma * (32.0 / nodes_per_tile) / per32;
    printf("%-34s %2d waves/CU  %7.0f cycles per 32 nodes and SIMD (mean wave %7.0f)  matrix pipe %4.1f %%"
           "  | %6.3f ms -> %5.2f GHz, %6.1f M nodes/s\n",
           name, waves, per32, sum / (blocks * waves) / tiles32_per_simd, 100.0 * busy, ms,
           worst / (ms * 1e6), (double)blocks * waves * nodes_per_tile * iters / (ms * 1e3));
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
#define RUN(name, kernel, waves, nodes, mfmas, cpm)                                              \
  do {                                                                                          \
    CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                              kLdsBytes));                                                      \
    float best = 1e30f;                                                                         \
    for (int rep = 0; rep < 3; ++rep) {                                                         \
      CHECK(hipEventRecord(e0));                                                                \
      kernel<<<blocks, 64 * waves, kLdsBytes>>>(d_w, d_image, d_hin, d_hout, iters, d_cyc);      \
      CHECK(hipEventRecord(e1));                                                                \
      CHECK(hipDeviceSynchronize());                                                            \
      float ms = 0;                                                                             \
      CHECK(hipEventElapsedTime(&ms, e0, e1));                                                  \
      if (ms < best) best = ms;                                                                 \
    }                                                                                           \
    report(name, waves, nodes, mfmas, cpm, best);                                               \
  } while (0)
  // MFMAs per tile: 128 + 12 bias + 8 LayerNorm sums
  for (int round = 0; round < 2; ++round) {
    RUN("pair pipeline (shipped)", (k_probe<8, false>), 8, 32, 148, 32);
    RUN("pair pipeline, 1 wave per SIMD", (k_probe<4, false>), 4, 32, 148, 32);
    RUN("block pipeline", (k_probe<4, true>), 4, 32, 148, 32);
    RUN("block pipeline", (k_probe<8, true>), 8, 32, 148, 32);
    RUN("block pipeline", (k_probe<12, true>), 12, 32, 148, 32);
    // + 40 selection MFMAs of the gather
    RUN("block pipeline, gather in front", (k_probe_merge<8, false>), 8, 32, 188, 32);
    RUN("block pipeline, gather merged in", (k_probe_merge<8, true>), 8, 32, 188, 32);
    RUN("block pipeline, gather in front", (k_probe_merge<4, false>), 4, 32, 188, 32);
    RUN("block pipeline, gather merged in", (k_probe_merge<4, true>), 4, 32, 188, 32);
    // per 16 nodes: 128 + 24 bias + 4 LayerNorm sums of 16 cycles, and one f32 MFMA (32 cycles)
    RUN("tile16 (16x16x32)", (k_probe16<8>), 8, 16, 158, 16);
    RUN("tile16 (16x16x32)", (k_probe16<12>), 12, 16, 158, 16);
    RUN("tile16 (16x16x32)", (k_probe16<16>), 16, 16, 158, 16);
  }
  return 0;
}
