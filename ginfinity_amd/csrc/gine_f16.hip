// fp16-mode GINE encode for gfx950 (MI355X): input Linear, fused GINE layer,
// head + float64 L2 normalise.
//
// Reference ops replaced (src/ginfinity):
//   _model.py:67      input Linear(7,128)                 -> k_input_linear_f16
//   _model.py:41-46   message / aggregate / (1+eps)x / MLP |
//   _model.py:34-36   Linear-BatchNorm-ReLU-Linear         |-> k_gine_layer_f16
//   _model.py:69-71   LayerNorm + residual                 |
//   _model.py:72      head Linear-ReLU-Linear              |-> k_head_f16
//   api.py:250-259    fp64 normalise, core rows, dtype     |
//
// Numerics contract (oracle/gine_numpy.py): every reference op
// boundary rounds to fp16 in-register; arithmetic inside an op is fp32
// (fp64 for LayerNorm moments and the final normalise).  v_pk_add_f16 /
// v_pk_mul_f16 are used where "fp32 op then round" and the native fp16 op
// agree exactly (sum/product of two fp16 values: 24 >= 2*11+2 bits).
//
// Layer kernel, one 512-thread workgroup (8 waves) per CU, persistent over
// 64-node tiles; tiles are dealt so that each XCD owns a contiguous node range
// (backbone / skip-2 neighbours then hit that XCD's L2):
//   A  gather-sum   16 lanes x 16 B per node row, CSR in-edges in COO order,
//                   fp32 accumulate, z -> LDS (fp16, XOR-swizzled 16-B chunks)
//   B  GEMM1        U^T = W0 . Z^T on v_mfma_f32_32x32x16_f16, W0 fragments live
//                   in registers for the whole launch; epilogue bias, round,
//                   BatchNorm fma, round, ReLU -> LDS
//   C  GEMM2        W^T = W1 . V^T, W1 fragments in registers; bias, round -> LDS
//   D  LayerNorm    fp64 moments over 16 lanes, fma-fma affine, round, residual
//                   add, 16-B coalesced store of the new hidden row
// The node index sits on the MFMA lane (C^T form), so each lane's 4 consecutive
// accumulator registers are 4 consecutive channels of one node: 8-byte LDS
// writes, no transposition.
#include <cstdlib>

#include "gfy_common.h"

namespace gfy {
namespace {

constexpr int kTile = 64;       // nodes per tile
constexpr int kThreads = 512;   // 8 waves, 2 per SIMD

// ---- LDS map of the layer kernel (bytes) -----------------------------------------
constexpr int kLdsZW = 0;                       // 64 x 256 B  z, later w
constexpr int kLdsV = kLdsZW + kTile * 256;     // 64 x 512 B  v
constexpr int kLdsTable = kLdsV + kTile * 512;  // 16 x 128 f16 edge table
constexpr int kLdsB0 = kLdsTable + kMaxEdgeTypes * kHidden * 2;
constexpr int kLdsAlpha = kLdsB0 + kMlp * 4;    // b0 is kept widened to fp32
constexpr int kLdsShift = kLdsAlpha + kMlp * 4;
constexpr int kLdsB1 = kLdsShift + kMlp * 4;
// CSR slice of a tile, double-buffered (tile t is consumed while t+1 is fetched)
constexpr int kMetaCap = 1024;                    // in-edges of one 64-node tile held in LDS
constexpr int kMetaRp = 0;                        // int[80]   row_ptr[base .. base+64]
constexpr int kMetaCol = 320;                     // int[kMetaCap]
constexpr int kMetaTyp = kMetaCol + kMetaCap * 4; // u8[kMetaCap]
constexpr int kMetaBytes = kMetaTyp + kMetaCap;
constexpr int kLdsMeta = kLdsB1 + kHidden * 2;
constexpr int kLdsLayerBytes = kLdsMeta + 2 * kMetaBytes;
constexpr int kGatherSlots = 8;                   // neighbour rows in flight per node row
constexpr int kSlotsA = 6;                        // ... in the single-role kernel (2 rows per thread)

// 16-byte chunk `chunk` of row `row`, XOR-swizzled so that the 16 lanes of one
// ds_read_b128 lane group (16 distinct rows, same chunk) hit 16 different slots.
__device__ __forceinline__ int off256(int row, int chunk) {
  return row * 256 + ((chunk ^ (row & 15)) << 4);
}
__device__ __forceinline__ int off512(int row, int chunk) {
  return row * 512 + ((chunk ^ (row & 15)) << 4);
}

__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f16x8 zero8() {
  f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return z;
}

// Sum over the 16 lanes of a DPP row (the 16 lanes that share one node row), result in
// every lane.  quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror:
// pure VALU cross-lane moves — __shfl_xor would go through LDS (ds_bpermute), four
// dependent round trips per reduction.
template <int kCtrl>
__device__ __forceinline__ double dpp_move(double v) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)bits, kCtrl, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(bits >> 32), kCtrl, 0xF, 0xF, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_move<0xB1>(v);    // lane ^ 1
  v += dpp_move<0x4E>(v);    // lane ^ 2
  v += dpp_move<0x141>(v);   // row_half_mirror: i <-> 7 - i
  v += dpp_move<0x140>(v);   // row_mirror:      i <-> 15 - i
  return v;
}

// 1/sqrt(x): v_rsq_f32 (1 ulp) + one Newton step -> within ~1 ulp of the correctly
// rounded value the oracle uses; ~6 instructions instead of the ~40 of the IEEE
// sqrt + divide sequences, executed redundantly by the 16 lanes of every row.
__device__ __forceinline__ float fast_rsqrt(float x) {
  const float r = __builtin_amdgcn_rsqf(x);
  const float e = __builtin_fmaf(-x * r, r, 1.0f);       // 1 - x r^2
  return __builtin_fmaf(0.5f * r, e, r);
}

template <int kCtrl>
__device__ __forceinline__ float dpp_move32(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum32(float v) {
  v += dpp_move32<0xB1>(v);
  v += dpp_move32<0x4E>(v);
  v += dpp_move32<0x141>(v);
  v += dpp_move32<0x140>(v);
  return v;
}

#ifdef GFY_STAMPS
// Diagnostic build only (python -m ginfinity_amd.build --stamps): per-phase
// shader-clock totals of the layer kernel, one row per workgroup.  Never compiled
// into libgfy.so proper.
__device__ unsigned long long g_stamps[256][16];
__device__ unsigned long long g_real[512][2];   // last launch: 100 MHz begin / end per workgroup
#define STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define STAMP(var)
#endif

// tiles owned by workgroup b: XCD (b & 7) owns tiles [xcd*tpx, (xcd+1)*tpx)
struct TileWalk {
  int tiles_per_xcd, slots_per_xcd, xcd, j;
  __device__ TileWalk(int num_tiles)
      : tiles_per_xcd((num_tiles + 7) >> 3),
        slots_per_xcd(gridDim.x >> 3),
        xcd(blockIdx.x & 7),
        j(blockIdx.x >> 3) {}
  __device__ int tile() const { return xcd * tiles_per_xcd + j; }
  __device__ bool valid(int num_tiles) const {
    return j < tiles_per_xcd && tile() < num_tiles;
  }
  __device__ void next() { j += slots_per_xcd; }
};

// ---------------------------------------------------------------------------------
// input Linear: h0 = R(R(x) . Win^T + b)            (_model.py:67, api.py:237-238)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void input_linear_block(
    const float* __restrict__ x, const f16* __restrict__ w_in /*[8][16][8] packed*/,
    const f16* __restrict__ b_in, f16* __restrict__ h, int n, int block, int blocks) {
  // 16 lanes per node, 8 channels per lane, block-stride over node groups.  The 2-KB
  // weight matrix is staged in LDS once per workgroup, laid out [c][chunk][k]
  // (channel = 8*chunk + c) so a wave's read of one c is 256 contiguous bytes; reading
  // it per-channel-row from memory made every lane hit its own 128-B line (~64 cycles
  // per load instruction, tools/probe.hip).
  __shared__ f16x8 ws[128];
  __shared__ f16x8 bs[16];
  if (threadIdx.x < 128) ws[threadIdx.x] = reinterpret_cast<const f16x8*>(w_in)[threadIdx.x];
  if (threadIdx.x < 16) bs[threadIdx.x] = reinterpret_cast<const f16x8*>(b_in)[threadIdx.x];
  __syncthreads();
  const int chunk = threadIdx.x & 15;
  const f16x8 bias = bs[chunk];
  const int64_t stride = (int64_t)blocks * (blockDim.x >> 4);
  for (int64_t node = (int64_t)block * (blockDim.x >> 4) + (threadIdx.x >> 4); node < n;
       node += stride) {
    float xv[kInDim];
#pragma unroll
    for (int k = 0; k < kInDim; ++k) xv[k] = (float)(f16)x[node * kInDim + k];
    f16x8 out;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f16x8 w = ws[c * 16 + chunk];
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < kInDim; ++k) acc = __builtin_fmaf(xv[k], (float)w[k], acc);
      out[c] = (f16)(acc + (float)bias[c]);
    }
    *reinterpret_cast<f16x8*>(h + node * kHidden + chunk * 8) = out;
  }
}

__global__ __launch_bounds__(256) void k_input_linear_f16(
    const float* __restrict__ x, const f16* __restrict__ w_in, const f16* __restrict__ b_in,
    f16* __restrict__ h, int n) {
  input_linear_block(x, w_in, b_in, h, n, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------
// fused GINE layer
// ---------------------------------------------------------------------------------
template <bool kResidual>
__global__ __launch_bounds__(kThreads, 2) void k_gine_layer_f16(
    const LayerF16 p, const f16* __restrict__ h_in, f16* __restrict__ h_out,
    const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const uint8_t* __restrict__ typ, int n, int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const zw = smem + kLdsZW;
  char* const vt = smem + kLdsV;
  f16* const table = reinterpret_cast<f16*>(smem + kLdsTable);
  float* const b0s = reinterpret_cast<float*>(smem + kLdsB0);
  float* const alphas = reinterpret_cast<float*>(smem + kLdsAlpha);
  float* const shifts = reinterpret_cast<float*>(smem + kLdsShift);
  f16* const b1s = reinterpret_cast<f16*>(smem + kLdsB1);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 31, hq = lane >> 5;
  const int chunk = t & 15, rsub = t >> 4;  // gather / LayerNorm mapping

  // Prologue.  Three dependent round trips are unavoidable before the first tile can
  // be reduced (row_ptr -> col/typ -> neighbour rows, ~1.3 us each under load), so
  // the row_ptr request goes out first and the per-launch constants and the weight
  // fragments travel in its shadow (vmcnt retires in order: nothing queued before
  // row_ptr delays it).
  STAMP(st_begin);
  TileWalk walk(num_tiles);
  int buf = 0;
  const bool any_tile = walk.valid(num_tiles);
  int rp_first = 0;
  if (any_tile && t <= kTile) {
    const int base0 = walk.tile() * kTile;
    rp_first = row_ptr[base0 + t < n ? base0 + t : n];
  }
  // weight fragments -> registers, kept for every tile of this workgroup
  f16x8 w0f[8], w1f[16];
  {
    const f16x8* w0p = reinterpret_cast<const f16x8*>(p.w0_frag) + (wave * 8) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w0f[ks] = w0p[ks * 64];
    const f16x8* w1p =
        reinterpret_cast<const f16x8*>(p.w1_frag) + ((wave >> 1) * 16) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w1f[ks] = w1p[ks * 64];
  }
  const f16x8 gamma8 = reinterpret_cast<const f16x8*>(p.ln_gamma)[chunk];
  const f16x8 beta8 = reinterpret_cast<const f16x8*>(p.ln_beta)[chunk];
  const f16 scale16 = (f16)p.scale;
  if (any_tile && t <= kTile)
    reinterpret_cast<int*>(smem + kLdsMeta + kMetaRp)[t] = rp_first;
  __syncthreads();
  if (any_tile) {   // col/typ of the first tile: second round trip
    char* meta = smem + kLdsMeta;
    const int e0 = reinterpret_cast<const int*>(meta + kMetaRp)[0];
    const int cnt = reinterpret_cast<const int*>(meta + kMetaRp)[kTile] - e0;
    if (cnt <= kMetaCap)
      for (int i = t; i < cnt; i += kThreads) {
        reinterpret_cast<int*>(meta + kMetaCol)[i] = col[e0 + i];
        reinterpret_cast<uint8_t*>(meta + kMetaTyp)[i] = typ[e0 + i];
      }
  }
  // per-launch constants -> LDS
  for (int i = t; i < kMaxEdgeTypes * kHidden / 8; i += kThreads)
    reinterpret_cast<f16x8*>(table)[i] =
        reinterpret_cast<const f16x8*>(p.edge_table)[i];
  if (t < kMlp) b0s[t] = (float)p.b0[t];
  if (t < kMlp / 4) {
    reinterpret_cast<f32x4*>(alphas)[t] = reinterpret_cast<const f32x4*>(p.bn_alpha)[t];
    reinterpret_cast<f32x4*>(shifts)[t] = reinterpret_cast<const f32x4*>(p.bn_shift)[t];
  }
  if (t < kHidden / 8)
    reinterpret_cast<f16x8*>(b1s)[t] = reinterpret_cast<const f16x8*>(p.b1)[t];
  __syncthreads();
#ifdef GFY_STAMPS
  {
    STAMP(st_pro);
    if (t == 0 && blockIdx.x < 256) g_stamps[blockIdx.x][6] += st_pro - st_begin;
  }
#endif

  for (; walk.valid(num_tiles); walk.next(), buf ^= 1) {
    const int base = walk.tile() * kTile;
    const char* meta = smem + kLdsMeta + buf * kMetaBytes;
    char* meta_next = smem + kLdsMeta + (buf ^ 1) * kMetaBytes;
    STAMP(st0);
    TileWalk ahead = walk;
    ahead.next();
    const bool has_next = ahead.valid(num_tiles);
    const int next_base = ahead.tile() * kTile;
    f16x8 hself[2];

    // row_ptr of the NEXT tile: issued now, parked in LDS after the gather
    int rp_next = 0;
    if (has_next && t <= kTile) {
      const int node = next_base + t < n ? next_base + t : n;
      rp_next = row_ptr[node];
    }

    // ---- A: gather-sum -> z --------------------------------------------------
    // Both node rows of a thread are fetched before either is reduced, slot fetches
    // are branch-free (clamped index; a slot beyond the in-degree re-reads the
    // node's own row, already in flight).  Measured alternatives that did NOT help
    // (profiles/README.md): staging the tile + a +-2 halo in LDS and reading the
    // backbone / skip-2 neighbours from there, and a wave-specialised pipeline
    // (gine_layer_ws.inc).
    const int* rp = reinterpret_cast<const int*>(meta + kMetaRp);
    const int e_base = rp[0];
    const bool staged = rp[kTile] - e_base <= kMetaCap;   // tile's edges are in LDS
    const int* col_l = reinterpret_cast<const int*>(meta + kMetaCol);
    const uint8_t* typ_l = reinterpret_cast<const uint8_t*>(meta + kMetaTyp);
    const char* hbytes = reinterpret_cast<const char*>(h_in);
    if (staged) {
      int lo[2], hi[2];
      f16x8 hv[2][kSlotsA];
      int ty[2][kSlotsA];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 32 + rsub;
        const int node = base + row < n ? base + row : n - 1;
        lo[pass] = rp[row] - e_base;
        hi[pass] = base + row < n ? rp[row + 1] - e_base : lo[pass];
        hself[pass] = *reinterpret_cast<const f16x8*>(
            hbytes + ((uint32_t)node * 256u + (uint32_t)chunk * 16u));
#pragma unroll
        for (int i = 0; i < kSlotsA; ++i) {
          const bool valid = lo[pass] + i < hi[pass];
          const int at = valid ? lo[pass] + i : 0;
          const uint32_t s_i = valid ? (uint32_t)col_l[at] : (uint32_t)node;
          ty[pass][i] = typ_l[at];
          hv[pass][i] = *reinterpret_cast<const f16x8*>(
              hbytes + (s_i * 256u + (uint32_t)chunk * 16u));
        }
      }
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 32 + rsub;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int i = 0; i < kSlotsA; ++i) {
          if (lo[pass] + i < hi[pass]) {   // COO order: slot i is the i-th in-edge
            const f16x8 ev = *reinterpret_cast<const f16x8*>(
                table + ty[pass][i] * kHidden + chunk * 8);
            const f16x8 m = __builtin_elementwise_max(hv[pass][i] + ev, zero8());
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)m[j];
          }
        }
        for (int e = lo[pass] + kSlotsA; e < hi[pass]; ++e) {  // in-degree > kSlotsA
          const uint32_t s0 = (uint32_t)col_l[e];
          const f16x8 h0 = *reinterpret_cast<const f16x8*>(
              hbytes + (s0 * 256u + (uint32_t)chunk * 16u));
          const f16x8 ev = *reinterpret_cast<const f16x8*>(table + typ_l[e] * kHidden + chunk * 8);
          const f16x8 m0 = __builtin_elementwise_max(h0 + ev, zero8());
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += (float)m0[j];
        }
        f16x8 agg;
#pragma unroll
        for (int j = 0; j < 8; ++j) agg[j] = (f16)acc[j];   // ONE rounding of the fp32 sum
        f16x8 z = hself[pass] * scale16 + agg;   // R(R(s*h) + a): two fp16 ops
        if (base + row >= n) {
          z = zero8();
          hself[pass] = zero8();
        }
        *reinterpret_cast<f16x8*>(zw + off256(row, chunk)) = z;
      }
    } else {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {   // oversized tile (hubs): CSR from memory
        const int row = pass * 32 + rsub;
        const int node = base + row;
        f16x8 z = zero8();
        hself[pass] = zero8();
        if (node < n) {
          const f16x8 hs = *reinterpret_cast<const f16x8*>(
              h_in + (size_t)node * kHidden + chunk * 8);
          float acc[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = 0.f;
          for (int e = rp[row]; e < rp[row + 1]; ++e) {
            const f16x8 h0 = *reinterpret_cast<const f16x8*>(
                h_in + (size_t)col[e] * kHidden + chunk * 8);
            const f16x8 ev = *reinterpret_cast<const f16x8*>(table + typ[e] * kHidden + chunk * 8);
            const f16x8 m0 = __builtin_elementwise_max(h0 + ev, zero8());
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)m0[j];
          }
          f16x8 agg;
#pragma unroll
          for (int j = 0; j < 8; ++j) agg[j] = (f16)acc[j];
          z = hs * scale16 + agg;
          hself[pass] = hs;
        }
        *reinterpret_cast<f16x8*>(zw + off256(row, chunk)) = z;
      }
    }
    STAMP(st_a0);
    if (has_next && t <= kTile) reinterpret_cast<int*>(meta_next + kMetaRp)[t] = rp_next;
    __syncthreads();
    STAMP(st1);

    // col/typ of the NEXT tile: issued before the GEMMs, parked in LDS after phase D
    int cn[2] = {0, 0};
    int tn[2] = {0, 0};
    int cnt_next = 0;
    if (has_next) {
      const int* rpn = reinterpret_cast<const int*>(meta_next + kMetaRp);
      const int e0n = rpn[0];
      cnt_next = rpn[kTile] - e0n;
      if (cnt_next <= kMetaCap) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = t + k * kThreads;
          if (i < cnt_next) {
            cn[k] = col[e0n + i];
            tn[k] = typ[e0n + i];
          }
        }
      }
    }

    // ---- B: U^T = W0 . Z^T ; v = relu(R(BN(R(u + b0)))) -------------------------
    {
      // chain 0, then chain 1 with the epilogue of chain 0 in its shadow (an MFMA
      // holds the issue port for 8 of its 32 cycles)
      f32x16 acc0 = {0}, acc1 = {0};
      f16x8 z1f[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const f16x8 z0 = *reinterpret_cast<const f16x8*>(zw + off256(r, 2 * ks + hq));
        z1f[ks] = *reinterpret_cast<const f16x8*>(zw + off256(32 + r, 2 * ks + hq));
        acc0 = mfma(w0f[ks], z0, acc0);
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) acc1 = mfma(w0f[ks], z1f[ks], acc1);
      // epilogue on 4-wide vectors so that hipcc emits the packed forms
      // (v_pk_add_f32, v_cvt_pk_f16_f32, v_pk_fma_f32, v_pk_max_f16): the scalar
      // spelling cost ~12 VALU instructions per value and made this phase
      // issue-bound (profiles/README.md)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = wave * 32 + 8 * g + 4 * hq;  // 4 consecutive channels
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(b0s + c0);
          const f32x4 al = *reinterpret_cast<const f32x4*>(alphas + c0);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(shifts + c0);
          f32x4 a4;
#pragma unroll
          for (int i = 0; i < 4; ++i) a4[i] = half == 0 ? acc0[4 * g + i] : acc1[4 * g + i];
          const f16x4 u4 = __builtin_convertvector(a4 + b4, f16x4);               // R(acc + b0)
          const f32x4 y4 = __builtin_elementwise_fma(__builtin_convertvector(u4, f32x4), al, sh);
          const f16x4 zero4 = {0, 0, 0, 0};
          const f16x4 vv = __builtin_elementwise_max(__builtin_convertvector(y4, f16x4), zero4);
          *reinterpret_cast<f16x4*>(vt + off512(32 * half + r, c0 >> 3) + hq * 8) = vv;
        }
      }
    }
    __syncthreads();
    STAMP(st2);

    // ---- C: W^T = W1 . V^T ; w = R(acc + b1) --------------------------------------
    {
      const int nt = wave & 1, ct = wave >> 1;
      f32x16 acc = {0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const f16x8 v8 =
            *reinterpret_cast<const f16x8*>(vt + off512(32 * nt + r, 2 * ks + hq));
        acc = mfma(w1f[ks], v8, acc);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = ct * 32 + 8 * g + 4 * hq;
        const f32x4 b4 = __builtin_convertvector(*reinterpret_cast<const f16x4*>(b1s + c0), f32x4);
        f32x4 a4;
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] = acc[4 * g + i];
        const f16x4 wv = __builtin_convertvector(a4 + b4, f16x4);
        *reinterpret_cast<f16x4*>(zw + off256(32 * nt + r, c0 >> 3) + hq * 8) = wv;
      }
    }
    __syncthreads();
    STAMP(st3);

    // ---- D: LayerNorm, residual, store -----------------------------------------
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = pass * 32 + rsub;
      const int node = base + row;
      const f16x8 w8 = *reinterpret_cast<const f16x8*>(zw + off256(row, chunk));
      float xf[8];
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xf[j] = (float)w8[j];
        sum += xf[j];
      }
      // two-pass fp32 moments over the 16 lanes of the row (DPP, no LDS round trip)
      const float mean = row16_sum32(sum) * (1.0f / kHidden);
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = xf[j] - mean;
        sq = __builtin_fmaf(d, d, sq);
      }
      const float var = row16_sum32(sq) * (1.0f / kHidden);
      const float rstd = fast_rsqrt(var + 1e-5f);
      const float offset = -rstd * mean;
      f16x8 y;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        y[j] = (f16)__builtin_fmaf(__builtin_fmaf(xf[j], rstd, offset),
                                   (float)gamma8[j], (float)beta8[j]);
      const f16x8 hn = kResidual ? (hself[pass] + y) : y;
      if (node < n)
        *reinterpret_cast<f16x8*>(h_out + (size_t)node * kHidden + chunk * 8) = hn;
    }
    if (has_next && cnt_next <= kMetaCap) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = t + k * kThreads;
        if (i < cnt_next) {
          reinterpret_cast<int*>(meta_next + kMetaCol)[i] = cn[k];
          reinterpret_cast<uint8_t*>(meta_next + kMetaTyp)[i] = (uint8_t)tn[k];
        }
      }
    }
    __syncthreads();  // zw is rewritten by the next tile's gather; next meta is complete
#ifdef GFY_STAMPS
    {
      STAMP(st4);
      if (t == 0 && blockIdx.x < 256) {
        g_stamps[blockIdx.x][0] += st_a0 - st0;   // gather, this wave
        g_stamps[blockIdx.x][1] += st1 - st_a0;   // wait at the A->B barrier
        g_stamps[blockIdx.x][2] += st2 - st1;     // GEMM1 + epilogue + barrier
        g_stamps[blockIdx.x][3] += st3 - st2;     // GEMM2 + epilogue + barrier
        g_stamps[blockIdx.x][4] += st4 - st3;     // LayerNorm + store + barrier
        g_stamps[blockIdx.x][5] += 1;             // tiles
      }
    }
#endif
  }
#ifdef GFY_STAMPS
  {
    STAMP(st_end);
    if (t == 0 && blockIdx.x < 256) g_stamps[blockIdx.x][7] += st_end - st_begin;
  }
#endif
}


// ---------------------------------------------------------------------------------
// head + normalise
// ---------------------------------------------------------------------------------
// double -> fp16 with ONE rounding (numpy's astype(float16) from float64):
// round to fp32 toward zero with a sticky bit (round-to-odd), then RNE to fp16.
__device__ __forceinline__ f16 f64_to_f16_rne(double x) {
  float f = (float)x;  // RNE
  const double back = (double)f;
  if (back != x) {
    uint32_t bits = __float_as_uint(f);
    if (__builtin_fabs(back) > __builtin_fabs(x)) bits -= 1u;  // toward zero
    bits |= 1u;                                                  // sticky
    f = __uint_as_float(bits);
  }
  return (f16)f;
}

template <typename OutT>
__device__ __forceinline__ void store8(OutT* dst, const double (&v)[8]);
template <>
__device__ __forceinline__ void store8<f16>(f16* dst, const double (&v)[8]) {
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f64_to_f16_rne(v[j]);
  *reinterpret_cast<f16x8*>(dst) = o;
}
template <>
__device__ __forceinline__ void store8<float>(float* dst, const double (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = (float)v[j];
    b[j] = (float)v[4 + j];
  }
  reinterpret_cast<f32x4*>(dst)[0] = a;
  reinterpret_cast<f32x4*>(dst)[1] = b;
}
template <>
__device__ __forceinline__ void store8<double>(double* dst, const double (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = v[j];
}

template <typename OutT>
__global__ __launch_bounds__(kThreads, 2) void k_head_f16(
    const HeadF16 p, const f16* __restrict__ h, const int32_t* __restrict__ out_rows,
    OutT* __restrict__ out, int n, int num_tiles, int normalise) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ht = smem;                // 64 x 256 B: h, later o
  char* const tt = smem + kTile * 256;  // 64 x 256 B: t

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 31, hq = lane >> 5;
  const int chunk = t & 15, rsub = t >> 4;
  const int nt = wave & 1, ct = wave >> 1;

  f16x8 waf[8], wbf[8];
  {
    const f16x8* wa = reinterpret_cast<const f16x8*>(p.wa_frag) + (ct * 8) * 64 + lane;
    const f16x8* wb = reinterpret_cast<const f16x8*>(p.wb_frag) + (ct * 8) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      waf[ks] = wa[ks * 64];
      wbf[ks] = wb[ks * 64];
    }
  }
  f16x4 bav[4], bbv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c0 = ct * 32 + 8 * g + 4 * hq;
    bav[g] = *reinterpret_cast<const f16x4*>(p.ba + c0);
    bbv[g] = *reinterpret_cast<const f16x4*>(p.bb + c0);
  }

  // rows of the first tile; afterwards every tile's rows are requested one tile
  // ahead (a round trip costs ~1.3 us under load and would otherwise be exposed)
  TileWalk walk(num_tiles);
  f16x8 pre[2] = {zero8(), zero8()};
  if (walk.valid(num_tiles)) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int node = walk.tile() * kTile + pass * 32 + rsub;
      if (node < n)
        pre[pass] = *reinterpret_cast<const f16x8*>(h + (size_t)node * kHidden + chunk * 8);
    }
  }
  for (; walk.valid(num_tiles); walk.next()) {
    const int base = walk.tile() * kTile;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
      *reinterpret_cast<f16x8*>(ht + off256(pass * 32 + rsub, chunk)) = pre[pass];
    {
      TileWalk ahead = walk;
      ahead.next();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int node = ahead.tile() * kTile + pass * 32 + rsub;
        pre[pass] = zero8();
        if (ahead.valid(num_tiles) && node < n)
          pre[pass] = *reinterpret_cast<const f16x8*>(h + (size_t)node * kHidden + chunk * 8);
      }
    }
    __syncthreads();
    {  // t = relu(R(h . Wa^T + ba))                         (_model.py:61-62)
      f32x16 acc = {0};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        acc = mfma(waf[ks],
                   *reinterpret_cast<const f16x8*>(ht + off256(32 * nt + r, 2 * ks + hq)),
                   acc);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = ct * 32 + 8 * g + 4 * hq;
        f16x4 tv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f16 u = (f16)(acc[4 * g + i] + (float)bav[g][i]);
          tv[i] = u > (f16)0 ? u : (f16)0;
        }
        *reinterpret_cast<f16x4*>(tt + off256(32 * nt + r, c0 >> 3) + hq * 8) = tv;
      }
    }
    __syncthreads();
    {  // o = R(t . Wb^T + bb)                               (_model.py:63)
      f32x16 acc = {0};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        acc = mfma(wbf[ks],
                   *reinterpret_cast<const f16x8*>(tt + off256(32 * nt + r, 2 * ks + hq)),
                   acc);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = ct * 32 + 8 * g + 4 * hq;
        f16x4 ov;
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = (f16)(acc[4 * g + i] + (float)bbv[g][i]);
        *reinterpret_cast<f16x4*>(ht + off256(32 * nt + r, c0 >> 3) + hq * 8) = ov;
      }
    }
    __syncthreads();
    // float64 L2 normalise, single rounding to the output dtype (api.py:250-259)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = pass * 32 + rsub;
      const int node = base + row;
      const f16x8 o8 = *reinterpret_cast<const f16x8*>(ht + off256(row, chunk));
      double v[8];
      double ss = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = (double)(float)o8[j];
        ss += v[j] * v[j];
      }
      ss = row16_sum(ss);
      if (normalise) {
        const double nrm = __builtin_sqrt(ss);
        const double den = nrm > 1e-12 ? nrm : 1e-12;
        if constexpr (sizeof(OutT) == 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] / den;      // exact quotient for f64 output
        } else {
          // one division, eight multiplies: differs from the exact quotient by <= 1 ulp of
          // float64, invisible after the single rounding to fp16 / fp32
          const double inv = 1.0 / den;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] * inv;
        }
      }
      if (node < n) {
        const int dest = out_rows ? out_rows[node] : node;
        if (dest >= 0) store8<OutT>(out + (size_t)dest * kOutDim + chunk * 8, v);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_copy_rows_f16(const f16* __restrict__ src,
                                                       f16* __restrict__ dst,
                                                       int64_t chunks) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < chunks; i += stride)
    reinterpret_cast<f16x8*>(dst)[i] = reinterpret_cast<const f16x8*>(src)[i];
}

#include "gine_layer_dma.inc"

int persistent_grid(int num_tiles) {
  int g = num_tiles < 256 ? num_tiles : 256;
  g = (g + 7) & ~7;  // whole XCD rounds (TileWalk divides by 8)
  return g < 8 ? 8 : g;
}

}  // namespace

#ifdef GFY_STAMPS
extern "C" int gfy_debug_real(unsigned long long* host /*[512][2]*/) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_real), sizeof(g_real)) == hipSuccess
             ? GFY_OK
             : GFY_ERR_HIP;
}
extern "C" int gfy_debug_stamps(unsigned long long* host /*[256][16]*/, int reset) {
  if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess)
    return GFY_ERR_HIP;
  if (reset) {
    static unsigned long long zeros[256][16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros)) != hipSuccess)
      return GFY_ERR_HIP;
  }
  return GFY_OK;
}
#endif

// two hidden-state buffers, each padded with spare rows behind the last node (the
// wave-specialised kernel stores unconditionally; rows that do not exist land there)
static size_t h_buffer_bytes(int64_t n) {
  return align_up((size_t)(n + 2 * kTile) * kHidden * sizeof(f16), 256);
}
// ... plus one plan per 32-node tile (gine_layer_dma.inc)
static size_t plan_bytes(int64_t n) {
  return align_up((size_t)((n + kT2 - 1) / kT2) * kPlanBytes, 256);
}
size_t encode_f16_workspace_bytes(int64_t n, int64_t /*e*/) {
  return 2 * h_buffer_bytes(n) + plan_bytes(n);
}

int launch_encode_f16(const gfy_encoder* enc, const float* x,
                      const int32_t* row_ptr, const int32_t* col,
                      const uint8_t* typ, int64_t n, int64_t e,
                      const int32_t* out_rows, void* out, int out_dtype,
                      int normalise, int tap_stage, void* ws, size_t ws_bytes,
                      hipStream_t s) {
  (void)e;
  const size_t need = encode_f16_workspace_bytes(n, e);
  GFY_REQUIRE(ws_bytes >= need, GFY_ERR_WORKSPACE,
              "gfy_encode: workspace %zu < required %zu", ws_bytes, need);
  GFY_REQUIRE(n <= (int64_t)1 << 24, GFY_ERR_UNSUPPORTED,
              "gfy_encode: fp16 path addresses rows with 32-bit byte offsets; "
              "split micro-batches above 16,777,216 nodes (got %lld)", (long long)n);
  f16* ha = (f16*)ws;
  f16* hb = (f16*)((char*)ws + h_buffer_bytes(n));
  char* plans = (char*)ws + 2 * h_buffer_bytes(n);
  const int num_tiles = (int)((n + kTile - 1) / kTile);
  const int grid = persistent_grid(num_tiles);
  const int dma_tiles = (int)((n + kT2 - 1) / kT2);
  int dma_grid = dma_tiles < 512 ? dma_tiles : 512;   // two 256-thread workgroups per CU
  dma_grid = (dma_grid + 7) & ~7;
  if (const char* g = getenv("GFY_DMA_GRID")) dma_grid = atoi(g);   // diagnostic
  // default: the LDS-DMA kernel (gine_layer_dma.inc); GFY_LAYER_KERNEL=v1 selects the
  // first-generation kernel for A/B runs
  static const bool use_dma = [] {
    const char* v = getenv("GFY_LAYER_KERNEL");
    return !v || v[0] == 'd';
  }();

  const int64_t items = n * 16;
  enc->mark(s, 0);
  {
    const int64_t blocks = (items + 255) / 256;
    const int linear_blocks = (int)(blocks > 2048 ? 2048 : blocks);
    if (use_dma && tap_stage != 0)   // + tile plans, once for all layers, in the same launch
      k_encode_setup<<<dma_tiles + linear_blocks, 256, 0, s>>>(
          x, enc->f16.w_in, enc->f16.b_in, ha, (int)n, row_ptr, col, typ, plans, dma_tiles);
    else
      k_input_linear_f16<<<linear_blocks, 256, 0, s>>>(x, enc->f16.w_in, enc->f16.b_in, ha,
                                                       (int)n);
  }
  enc->mark(s, 1);
  const int stop = tap_stage >= 0 ? tap_stage : enc->layers;
  static bool lds_opt_in = false;   // > 64 KB of dynamic LDS needs an explicit opt-in
  if (!lds_opt_in) {
    GFY_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_gine_layer_f16<true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLayerBytes));
    GFY_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_gine_layer_f16<false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLayerBytes));
    GFY_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_gine_layer_dma<true, false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, k2Bytes));
    GFY_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_gine_layer_dma<false, false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, k2Bytes));
    GFY_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_gine_layer_dma<true, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, k2Bytes));
    GFY_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_gine_layer_dma<false, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, k2Bytes));
    lds_opt_in = true;
  }
  // fp16 output of a full encode: the last layer's launch runs the head as well
  const bool fuse_head = use_dma && tap_stage < 0 && out_dtype == GFY_F16 && stop > 0 && n >= kT2 &&
                         !getenv("GFY_SEPARATE_HEAD");
  for (int l = 0; l < stop; ++l) {
    const bool with_head = fuse_head && l == stop - 1;
#define GFY_LAUNCH_DMA(RES, HEAD)                                                        \
  k_gine_layer_dma<RES, HEAD><<<dma_grid, kThreads2, k2Bytes, s>>>(                      \
      enc->f16.layer[l], ha, hb, row_ptr, col, typ, plans, (int)n, dma_tiles, enc->f16.head, \
      out_rows, (f16*)out, normalise)
    if (use_dma && enc->residual && with_head) GFY_LAUNCH_DMA(true, true);
    else if (use_dma && enc->residual) GFY_LAUNCH_DMA(true, false);
    else if (use_dma && with_head) GFY_LAUNCH_DMA(false, true);
    else if (use_dma) GFY_LAUNCH_DMA(false, false);
#undef GFY_LAUNCH_DMA
    else if (enc->residual)
      k_gine_layer_f16<true><<<grid, kThreads, kLdsLayerBytes, s>>>(
          enc->f16.layer[l], ha, hb, row_ptr, col, typ, (int)n, num_tiles);
    else
      k_gine_layer_f16<false><<<grid, kThreads, kLdsLayerBytes, s>>>(
          enc->f16.layer[l], ha, hb, row_ptr, col, typ, (int)n, num_tiles);
    f16* sw = ha;
    ha = hb;
    hb = sw;
    enc->mark(s, 2 + l);
  }
  if (tap_stage >= 0) {
    const int64_t chunks = n * 16;
    int g = (int)((chunks + 255) / 256);
    k_copy_rows_f16<<<g > 2048 ? 2048 : g, 256, 0, s>>>(ha, (f16*)out, chunks);
    GFY_CHECK_HIP(hipGetLastError());
    return GFY_OK;
  }
  if (fuse_head) {
    enc->mark(s, 2 + enc->layers);
    GFY_CHECK_HIP(hipGetLastError());
    return GFY_OK;
  }
  const int head_lds = 2 * kTile * 256;
  switch (out_dtype) {
    case GFY_F16:
      k_head_f16<f16><<<grid, kThreads, head_lds, s>>>(
          enc->f16.head, ha, out_rows, (f16*)out, (int)n, num_tiles, normalise);
      break;
    case GFY_F32:
      k_head_f16<float><<<grid, kThreads, head_lds, s>>>(
          enc->f16.head, ha, out_rows, (float*)out, (int)n, num_tiles, normalise);
      break;
    default:
      k_head_f16<double><<<grid, kThreads, head_lds, s>>>(
          enc->f16.head, ha, out_rows, (double*)out, (int)n, num_tiles, normalise);
  }
  enc->mark(s, 2 + enc->layers);
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

}  // namespace gfy
