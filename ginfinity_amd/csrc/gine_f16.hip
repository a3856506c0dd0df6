// fp16-mode GINE encode for gfx950 (MI355X): per-encode setup (tile plans + input
// Linear), fused GINE layer, head + float64 L2 normalise.
//
// Reference ops replaced (src/ginfinity):
//   _model.py:67      input Linear(7,128)                 -> k_encode_setup / k_input_linear_f16
//   _model.py:41-46   message / aggregate / (1+eps)x / MLP |
//   _model.py:34-36   Linear-BatchNorm-ReLU-Linear         |-> k_gine_layer_f16  (gine_layer.inc)
//   _model.py:69-71   LayerNorm + residual                 |
//   _model.py:72      head Linear-ReLU-Linear              |-> tail of the last layer's launch
//   api.py:250-259    fp64 normalise, core rows, dtype     |   (fp16 out) / k_head_f16
//
// Numerics contract (oracle/gine_numpy.py): every reference op
// boundary rounds to fp16 in-register; arithmetic inside an op is fp32
// (fp64 for the final normalise).  v_pk_add_f16 / v_pk_mul_f16 are used where
// "fp32 op then round" and the native fp16 op agree exactly (sum/product of two
// fp16 values: 24 >= 2*11+2 bits).
//
// One wave owns one 32-node tile from the gather to the store (gine_layer.inc): gather-sum out
// of LDS with the accumulation on the matrix cores, both MLP products with the weights read
// from LDS and the activations in registers, BatchNorm / LayerNorm / residual in place.
// Hidden rows are kept in memory in the order the MFMA lanes hold them (gfy_common.h
// stored_channel); k_copy_rows_f16 restores the natural order for the parity taps.
#include <cstdlib>

#include "gfy_common.h"

namespace gfy {
namespace {

constexpr int kTile = 64;       // nodes per tile of the stand-alone head kernel
constexpr int kThreads = 512;   // ... and its workgroup: 8 waves, 2 per SIMD

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
    const ShardTable& shards, const f16* __restrict__ w_in /*[8][16][8] packed*/,
    const f16* __restrict__ b_in, f16* __restrict__ h, int block, int blocks) {
  // 16 lanes per row, 8 channels per lane, block-stride over row groups of the batch's global
  // (padded) row space; padding rows get zeros.  The 2-KB weight matrix is staged in LDS once
  // per workgroup, laid out [c][chunk][k] (channel = 8*chunk + c) so a wave's read of one c is
  // 256 contiguous bytes; reading it per-channel-row from memory made every lane hit its own
  // 128-B line (~64 cycles per load instruction, tools/probe.hip).
  __shared__ f16x8 ws[128];
  __shared__ f16x8 bs[16];
  if (threadIdx.x < 128) ws[threadIdx.x] = reinterpret_cast<const f16x8*>(w_in)[threadIdx.x];
  if (threadIdx.x < 16) bs[threadIdx.x] = reinterpret_cast<const f16x8*>(b_in)[threadIdx.x];
  __syncthreads();
  const int chunk = threadIdx.x & 15;
  const f16x8 bias = bs[chunk];
  const int64_t rows = shards.total_rows();
  const int64_t stride = (int64_t)blocks * (blockDim.x >> 4);
  for (int64_t row = (int64_t)block * (blockDim.x >> 4) + (threadIdx.x >> 4); row < rows;
       row += stride) {
    const int shard = shards.shard_of_tile((int)(row >> 5));
    const int64_t node = row - (int64_t)shards.tile_base[shard] * 32;   // shard-local
    f16x8 out = zero8();
    if (node < shards.nodes[shard]) {
      const float* __restrict__ x = shards.x[shard];
      float xv[kInDim];
#pragma unroll
      for (int k = 0; k < kInDim; ++k) xv[k] = (float)(f16)x[node * kInDim + k];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f16x8 w = ws[c * 16 + chunk];
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < kInDim; ++k) acc = __builtin_fmaf(xv[k], (float)w[k], acc);
        out[c] = (f16)(acc + (float)bias[c]);
      }
    }
    *reinterpret_cast<f16x8*>(h + row * kHidden + chunk * 8) = out;
  }
}

__global__ __launch_bounds__(256) void k_input_linear_f16(
    const ShardTable shards, const f16* __restrict__ w_in, const f16* __restrict__ b_in,
    f16* __restrict__ h) {
  input_linear_block(shards, w_in, b_in, h, blockIdx.x, gridDim.x);
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

// fp16(o * inv) for eight values with the SAME result as rounding the float64 product once:
// the fp32 product o * (float)inv is within 2 fp32 ulps of the float64 value, so both round
// to the same fp16 unless the fp32 value lies within a few ulps of a rounding tie (the 13
// dropped mantissa bits ~ 0x1000) or in fp16's subnormal range.  Only then (a fraction of
// a percent of the rows) is the float64 path taken; it used to be 40 % of the vector
// instructions of the head pass.
__device__ __forceinline__ f16x8 scale_to_f16(const f16x8 o8, double inv) {
  const float invf = (float)inv;
  f16x8 out;
  bool slow = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float q = (float)o8[j] * invf;
    const uint32_t bits = __float_as_uint(q);
    slow |= ((bits & 0x1FFFu) - 0xFFCu) <= 8u;               // within 4 fp32 ulps of a tie
    slow |= ((bits & 0x7FFFFFFFu) - 1u) < 0x387FFFFFu;       // 0 < |q| < 2^-14
    out[j] = (f16)q;
  }
  if (slow) {
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = f64_to_f16_rne((double)(float)o8[j] * inv);
  }
  return out;
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
          if constexpr (sizeof(OutT) == 2) {
            if (node < n) {
              const int dest = out_rows ? out_rows[node] : node;
              if (dest >= 0)
                *reinterpret_cast<f16x8*>(out + (size_t)dest * kOutDim + chunk * 8) =
                    scale_to_f16(o8, inv);
            }
            continue;
          }
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

// hidden rows in stored order -> natural channel order (parity taps): per 16 channels the
// groups 4-7 and 8-11 trade places, i.e. the inner 8-byte halves of two 16-byte chunks
__global__ __launch_bounds__(256) void k_copy_rows_f16(const f16* __restrict__ src,
                                                       f16* __restrict__ dst,
                                                       int64_t groups /* of 16 channels */) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < groups; i += stride) {
    const u32x4 a = reinterpret_cast<const u32x4*>(src)[2 * i];
    const u32x4 b = reinterpret_cast<const u32x4*>(src)[2 * i + 1];
    const u32x4 lo = {a[0], a[1], b[0], b[1]}, hi = {a[2], a[3], b[2], b[3]};
    reinterpret_cast<u32x4*>(dst)[2 * i] = lo;
    reinterpret_cast<u32x4*>(dst)[2 * i + 1] = hi;
  }
}

#include "csr_finish.inc"
#include "gine_layer.inc"
#include "csr_records.inc"
#include "gine_layer_q.inc"
#include "gine_block_pipe.inc"
#include "gine_layer_w.inc"
#include "gine_layer_x.inc"

#ifdef GFY_PROBE_BUILD   // tools/mlp_probe.hip: the device code above, none of the launch code below
}  // namespace
}  // namespace gfy
#else
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

// two hidden-state buffers over the batch's padded rows, each with spare rows behind the last
static size_t h_buffer_bytes(int64_t rows) {
  return align_up((size_t)(rows + 2 * kTile) * kHidden * sizeof(f16), 256);
}
// ... plus one plan per 32-node tile (gine_layer.inc)
static size_t plan_bytes(int64_t rows) {
  return align_up((size_t)((rows + kLTile - 1) / kLTile) * kPlanBytes, 256);
}
static int64_t padded_rows(int64_t n) { return (n + kLTile - 1) / kLTile * kLTile; }
size_t encode_f16_workspace_bytes(int64_t n, int64_t /*e*/) {
  return 2 * h_buffer_bytes(padded_rows(n)) + plan_bytes(padded_rows(n));
}

// > 64 KB of dynamic LDS needs an opt-in per kernel and per DEVICE (a process may drive
// several GPUs, from several threads)
static PerDeviceOnce g_layer_lds_opt_in;
int prepare_device_f16() {   // gfy_encoder_create, with the encoder's device current
  return g_layer_lds_opt_in.run([]() -> int {
#define GFY_OPT_IN(kernel, bytes)                                                          \
  GFY_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel),                \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes))
    GFY_OPT_IN((k_gine_layer_f16<true, false>), kLdsBytes);
    GFY_OPT_IN((k_gine_layer_f16<false, false>), kLdsBytes);
    GFY_OPT_IN((k_gine_layer_f16<true, true>), kLdsBytes);
    GFY_OPT_IN((k_gine_layer_f16<false, true>), kLdsBytes);
    GFY_OPT_IN((k_gine_layer_q<true, false, false>), kQLdsBytes);
    GFY_OPT_IN((k_gine_layer_q<false, false, false>), kQLdsBytes);
    GFY_OPT_IN((k_gine_layer_q<true, false, true>), kQLdsBytes);
    GFY_OPT_IN((k_gine_layer_q<false, false, true>), kQLdsBytes);
    GFY_OPT_IN((k_gine_layer_q<true, true, false>), kQLdsBytes);
    GFY_OPT_IN(k_head_d, kLdsBytes);
    GFY_OPT_IN((k_gine_layer_w<true, false, false>), kWLdsBytes);
    GFY_OPT_IN((k_gine_layer_w<false, false, false>), kWLdsBytes);
    GFY_OPT_IN((k_gine_layer_w<true, false, true>), kWLdsBytes);
    GFY_OPT_IN((k_gine_layer_w<false, false, true>), kWLdsBytes);
    GFY_OPT_IN((k_gine_layer_w<true, true, false>), kWLdsBytes);
    GFY_OPT_IN((k_gine_layer_x<true, false, false>), kXLdsBytes);
    GFY_OPT_IN((k_gine_layer_x<false, false, false>), kXLdsBytes);
    GFY_OPT_IN((k_gine_layer_x<true, false, true>), kXLdsBytes);
    GFY_OPT_IN((k_gine_layer_x<false, false, true>), kXLdsBytes);
    GFY_OPT_IN((k_gine_layer_x<true, true, false>), kXLdsBytes);
#undef GFY_OPT_IN
    return GFY_OK;
  });
}

// `coo` != nullptr: the CSR is finished inside the setup launch (gfy_encode_coo)
struct CooInput {
  CsrScratch scratch;
  const RecordTable* records;   // or nullptr: the counting kernel has filled the scratch table
  bool scan_free;
  int32_t* row_ptr;
  int32_t* col;
  uint8_t* typ;
};

// One shard or a batch (ShardTable), CSR given (row_ptr / col / typ: one shard only) or finished
// here from the counting kernel's table (coo).
static int encode_f16_on(const gfy_encoder* enc, const ShardTable& shards, const int32_t* row_ptr,
                         const int32_t* col, const uint8_t* typ, const CooInput* coo,
                         int out_dtype, int normalise, int tap_stage, void* ws, size_t ws_bytes,
                         hipStream_t s) {
  const int64_t rows = shards.total_rows();
  const size_t need = 2 * h_buffer_bytes(rows) + plan_bytes(rows);
  GFY_REQUIRE(ws_bytes >= need, GFY_ERR_WORKSPACE,
              "gfy_encode: workspace %zu < required %zu", ws_bytes, need);
  GFY_REQUIRE(rows <= (int64_t)1 << 24, GFY_ERR_UNSUPPORTED,
              "gfy_encode: fp16 path addresses rows with 32-bit byte offsets; "
              "split micro-batches / batches above 16,777,216 nodes (got %lld)", (long long)rows);
  f16* ha = (f16*)ws;
  f16* hb = (f16*)((char*)ws + h_buffer_bytes(rows));
  char* plans = (char*)ws + 2 * h_buffer_bytes(rows);
  const int layer_tiles = shards.total_tiles();
  // one tile per wave, eight per workgroup, XCD x = workgroups x, x + 8, ... (gine_layer.inc)
  const int layer_grid = 8 * ((layer_tiles + 8 * kLWaves - 1) / (8 * kLWaves));

  const int64_t items = rows * 16;
  enc->mark(s, 0);
  {
    const int64_t blocks = (items + 255) / 256;
    const int linear_blocks = (int)(blocks > 2048 ? 2048 : blocks);
    if (tap_stage == 0)
      k_input_linear_f16<<<linear_blocks, 256, 0, s>>>(shards, enc->f16.w_in, enc->f16.b_in, ha);
    else if (coo && coo->records) {   // COO -> plans by record ranges (csr_records.inc): no counting launch
#ifdef GFY_DIAG_SETUP_SPLIT   // diagnostic: GFY_DIAG_SETUP=1 range workgroups only, 2 Linear only (wrong results)
      static const int diag = getenv("GFY_DIAG_SETUP") ? atoi(getenv("GFY_DIAG_SETUP")) : 0;
      const int ranges = diag == 2 ? 0 : coo->records->range_base[shards.shards];
      static const int diag_linear = getenv("GFY_DIAG_LINEAR") ? atoi(getenv("GFY_DIAG_LINEAR")) : 0;
      const int linear = diag == 1 ? 0 : diag_linear > 0 ? diag_linear : linear_blocks;
#else
      const int ranges = coo->records->range_base[shards.shards], linear = linear_blocks;
#endif
      // the Linear's workgroups have the range workgroups' size: the same number of threads
#define GFY_LAUNCH_SETUP_REC(ROWS, THREADS)                                                   \
  k_encode_setup_rec<ROWS, THREADS>                                                           \
      <<<ranges + (linear * 256 + THREADS - 1) / THREADS, THREADS, 0, s>>>(                   \
          shards, *coo->records, enc->f16.w_in, enc->f16.b_in, ha, coo->row_ptr, coo->col,    \
          coo->typ, coo->scratch.perm, plans, ranges, enc->edge_dim)
      if (coo->records->range_rows == kRecRowsLarge) GFY_LAUNCH_SETUP_REC(kRecRowsLarge, kRecThreadsLarge);
      else if (coo->records->range_rows == kRecRowsSmall) GFY_LAUNCH_SETUP_REC(kRecRowsSmall, kRecThreadsSmall);
      else GFY_LAUNCH_SETUP_REC(kRecRowsLone, kRecThreadsLone);
#undef GFY_LAUNCH_SETUP_REC
    }
    else if (coo && coo->scan_free)   // + last CSR stage (row offsets included) + tile plans
      k_encode_setup_coo<false><<<layer_tiles + linear_blocks, 256, 0, s>>>(
          shards, enc->f16.w_in, enc->f16.b_in, ha, coo->scratch, coo->row_ptr, coo->col,
          coo->typ, plans, layer_tiles, enc->edge_dim);
    else if (coo)
      k_encode_setup_coo<true><<<layer_tiles + linear_blocks, 256, 0, s>>>(
          shards, enc->f16.w_in, enc->f16.b_in, ha, coo->scratch, coo->row_ptr, coo->col,
          coo->typ, plans, layer_tiles, enc->edge_dim);
    else            // + tile plans
      k_encode_setup<<<layer_tiles + linear_blocks, 256, 0, s>>>(
          shards, enc->f16.w_in, enc->f16.b_in, ha, row_ptr, col, typ, plans, layer_tiles,
          enc->edge_dim);
  }
  enc->mark(s, 1);
  const int stop = tap_stage >= 0 ? tap_stage : enc->layers;
  if (enc->timing == 3) {   // start = +inf, end = 0 for every layer launch
    static const unsigned long long init[2 * kMaxLayers] = {
        ~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
    GFY_CHECK_HIP(hipMemcpyAsync(enc->device_spans, init, sizeof init, hipMemcpyHostToDevice, s));
  }
  // Which layer kernel: a launch that gives every CU more than one round of eight tiles (a
  // large micro-batch, or a batch of shards) runs the persistent-rounds kernel
  // (gine_layer_q.inc: the fills of round r + 1 under the arithmetic of round r, CUs de-phased
  // by a start stagger) and the stand-alone head behind it; a single-round launch (the 60,000-
  // node shard by itself) runs the round-2 kernel, whose last launch carries the head.
  // GFY_OPT_LAYER_KERNEL forces either (A/B runs, parity tests).
  const int tiles_per_xcd = (layer_tiles + 7) / 8;
  const int wanted = (tiles_per_xcd + kLWaves - 1) / kLWaves, per_xcd = enc->cus / 8;
  const int p_grid = 8 * (wanted < per_xcd ? wanted : per_xcd);
  const int rounds = (wanted + per_xcd - 1) / per_xcd;
  // ... and, where the edge table leaves its last rows free for the plan-head slots, as two
  // 4-wave workgroups per CU with the weights streamed through a window (gine_layer_w.inc)
  const bool window_ok = enc->edge_dim <= kWMaxEdgeTypes;
  // ... or as three 4-wave workgroups per CU of <= 168 registers (gine_layer_x.inc)
  const bool triple = window_ok && enc->layer_kernel == 5;
  const bool windowed =
      !triple && window_ok && (enc->layer_kernel == 4 || (enc->layer_kernel < 0 && rounds > 1));
  const bool persistent =
      !triple && !windowed && (enc->layer_kernel >= 3 || (enc->layer_kernel < 0 && rounds > 1));
  const int w_wanted = (tiles_per_xcd + kWWaves - 1) / kWWaves, w_per_xcd = 2 * per_xcd;
  const int w_grid = 8 * (w_wanted < w_per_xcd ? w_wanted : w_per_xcd);
  const int w_rounds = (w_wanted + w_per_xcd - 1) / w_per_xcd;
  const int x_wanted = (tiles_per_xcd + kXWaves - 1) / kXWaves, x_per_xcd = 3 * per_xcd;
  const int x_grid = 8 * (x_wanted < x_per_xcd ? x_wanted : x_per_xcd);
  const int x_rounds = (x_wanted + x_per_xcd - 1) / x_per_xcd;
  const int x_stagger = enc->stagger >= 0 ? enc->stagger : x_rounds >= 2 ? 250 : 0;
  const int x_priority = enc->priority >= 0 ? enc->priority : 0x24;   // residents 0, 1, 2 at levels 0, 1, 2
  enc->last_layer_kernel = triple ? 5 : windowed ? 4 : persistent ? 3 : 1;
  const int w_stagger = enc->stagger >= 0 ? enc->stagger : w_rounds >= 3 ? 250 : 0;
  // the second workgroup of a CU multiplies at s_setprio 1 (gine_layer_w.inc)
  const int w_priority = enc->priority >= 0 ? enc->priority : 4;
  // start offsets only pay once a workgroup runs several rounds (the offset costs up to one
  // round at the start of the launch)
  const int stagger = enc->stagger >= 0 ? enc->stagger : rounds >= 3 ? 500 : 0;
  // fp16 output of a full encode: the round-2 kernel's last launch runs the head as well
  // (GFY_OPT_SEPARATE_HEAD keeps the stand-alone head kernel: A/B runs and parity tests)
  const bool fuse_head =
      tap_stage < 0 && out_dtype == GFY_F16 && stop > 0 && !enc->separate_head;
  const int32_t* const csr_rows = coo ? coo->row_ptr : row_ptr;
  const int32_t* const csr_col = coo ? coo->col : col;
  const uint8_t* const csr_typ = coo ? coo->typ : typ;
  // rows the layer kernels may look up in row_ptr (the direct path of hub tiles): the caller's
  // CSR has n + 1 entries, the one finished here covers the padded rows as well
  const int csr_limit = coo ? (int)rows : shards.nodes[0];
  HeadOut no_out{};
  const HeadOut head_out{enc->f16.head, shards, normalise};
  for (int l = 0; l < stop; ++l) {
    int32_t* const spent =
        coo && l == 0 && coo->scan_free && !coo->records ? coo->scratch.tile_sum : nullptr;
    unsigned long long* const span = enc->timing == 3 ? enc->device_spans + 2 * l : nullptr;
    const bool with_head = fuse_head && l == stop - 1;
#define GFY_LAUNCH_LAYER(RES, HEAD)                                                          \
  k_gine_layer_f16<RES, HEAD><<<layer_grid, kLThreads, kLdsBytes, s>>>(                      \
      enc->f16.layer[l], ha, hb, csr_rows, csr_col, csr_typ, plans, csr_limit, layer_tiles,  \
      enc->f16.head, shards, normalise, spent, span)
#define GFY_LAUNCH_ROUNDS(RES, HEAD)                                                         \
  k_gine_layer_q<RES, false, HEAD><<<p_grid, kLThreads, kQLdsBytes, s>>>(                    \
      enc->f16.layer[l], ha, hb, csr_rows, csr_col, csr_typ, plans, csr_limit, layer_tiles,  \
      stagger, spent, span, kTapNone, nullptr, HEAD ? head_out : no_out)
#define GFY_LAUNCH_WINDOW(RES, HEAD)                                                         \
  k_gine_layer_w<RES, false, HEAD><<<w_grid, kWThreads, kWLdsBytes, s>>>(                    \
      enc->f16.layer[l], ha, hb, csr_rows, csr_col, csr_typ, plans, csr_limit, layer_tiles,  \
      w_stagger, w_priority, spent, span, kTapNone, nullptr,         \
      HEAD ? head_out : no_out)
#define GFY_LAUNCH_TRIPLE(RES, HEAD)                                                         \
  k_gine_layer_x<RES, false, HEAD><<<x_grid, kXThreads, kXLdsBytes, s>>>(                    \
      enc->f16.layer[l], ha, hb, csr_rows, csr_col, csr_typ, plans, csr_limit, layer_tiles,  \
      x_stagger, x_priority, spent, span, kTapNone, nullptr, HEAD ? head_out : no_out)
    if (triple && enc->residual && with_head) GFY_LAUNCH_TRIPLE(true, true);
    else if (triple && enc->residual) GFY_LAUNCH_TRIPLE(true, false);
    else if (triple && with_head) GFY_LAUNCH_TRIPLE(false, true);
    else if (triple) GFY_LAUNCH_TRIPLE(false, false);
    else if (windowed && enc->residual && with_head) GFY_LAUNCH_WINDOW(true, true);
    else if (windowed && enc->residual) GFY_LAUNCH_WINDOW(true, false);
    else if (windowed && with_head) GFY_LAUNCH_WINDOW(false, true);
    else if (windowed) GFY_LAUNCH_WINDOW(false, false);
    else if (persistent && enc->residual && with_head) GFY_LAUNCH_ROUNDS(true, true);
    else if (persistent && enc->residual) GFY_LAUNCH_ROUNDS(true, false);
    else if (persistent && with_head) GFY_LAUNCH_ROUNDS(false, true);
    else if (persistent) GFY_LAUNCH_ROUNDS(false, false);
    else if (enc->residual && with_head) GFY_LAUNCH_LAYER(true, true);
    else if (enc->residual) GFY_LAUNCH_LAYER(true, false);
    else if (with_head) GFY_LAUNCH_LAYER(false, true);
    else GFY_LAUNCH_LAYER(false, false);
#undef GFY_LAUNCH_LAYER
#undef GFY_LAUNCH_ROUNDS
#undef GFY_LAUNCH_WINDOW
#undef GFY_LAUNCH_TRIPLE
    f16* sw = ha;
    ha = hb;
    hb = sw;
    enc->mark(s, 2 + l);
  }
  if (tap_stage >= 0) {   // one shard: its rows, natural channel order
    const int64_t groups = (int64_t)shards.nodes[0] * 8;
    int g = (int)((groups + 255) / 256);
    k_copy_rows_f16<<<g > 2048 ? 2048 : g, 256, 0, s>>>(ha, (f16*)shards.out[0], groups);
    GFY_CHECK_HIP(hipGetLastError());
    return GFY_OK;
  }
  if (fuse_head) {
    enc->mark(s, 2 + enc->layers);
    GFY_CHECK_HIP(hipGetLastError());
    return GFY_OK;
  }
  if (out_dtype == GFY_F16 && (persistent || windowed || triple)) {
    k_head_d<<<p_grid, kLThreads, kLdsBytes, s>>>(enc->f16.head, ha, shards, layer_tiles,
                                                  normalise);
  } else {   // the tiled stand-alone head, shard by shard (f32 / f64 output, A/B runs)
    const int head_lds = 2 * kTile * 256;
    for (int k = 0; k < shards.shards; ++k) {
      const int n = shards.nodes[k];
      const int num_tiles = (n + kTile - 1) / kTile;
      const int grid = persistent_grid(num_tiles);
      const f16* rows_k = ha + (size_t)shards.tile_base[k] * kLTile * kHidden;
      switch (out_dtype) {
        case GFY_F16:
          k_head_f16<f16><<<grid, kThreads, head_lds, s>>>(
              enc->f16.head, rows_k, shards.out_rows[k], (f16*)shards.out[k], n, num_tiles,
              normalise);
          break;
        case GFY_F32:
          k_head_f16<float><<<grid, kThreads, head_lds, s>>>(
              enc->f16.head, rows_k, shards.out_rows[k], (float*)shards.out[k], n, num_tiles,
              normalise);
          break;
        default:
          k_head_f16<double><<<grid, kThreads, head_lds, s>>>(
              enc->f16.head, rows_k, shards.out_rows[k], (double*)shards.out[k], n, num_tiles,
              normalise);
      }
    }
  }
  enc->mark(s, 2 + enc->layers);
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

int launch_encode_f16(const gfy_encoder* enc, const float* x,
                      const int32_t* row_ptr, const int32_t* col,
                      const uint8_t* typ, int64_t n, int64_t e,
                      const int32_t* out_rows, void* out, int out_dtype,
                      int normalise, int tap_stage, void* ws, size_t ws_bytes,
                      hipStream_t s) {
  const ShardTable shards = single_shard(x, nullptr, nullptr, n, e, out_rows, out);
  return encode_f16_on(enc, shards, row_ptr, col, typ, nullptr, out_dtype, normalise, tap_stage,
                       ws, ws_bytes, s);
}

// workspace of gfy_encode_coo / gfy_encode_coo_batch over `rows` padded rows and `e` edges:
// [CSR scratch, counters first][row_ptr][col][typ][encode]
struct CooWorkspace {
  CsrScratch scratch;
  int32_t* scan_sums;
  int32_t* row_ptr;
  int32_t* col;
  uint8_t* typ;
  void* encode;
  size_t encode_bytes, bytes;
};
static CooWorkspace carve_coo(void* base, int64_t rows, int64_t e) {
  CooWorkspace w;
  size_t off = 0;
  w.scratch = carve_csr(base, rows, e, &w.scan_sums, &off);
  auto take = [&](size_t size) {
    void* p = base ? (char*)base + off : nullptr;
    off += align_up(size, 256);
    return p;
  };
  w.row_ptr = (int32_t*)take((size_t)(rows + 1) * 4);
  w.col = (int32_t*)take((size_t)(e > 0 ? e : 1) * 4);
  w.typ = (uint8_t*)take((size_t)(e > 0 ? e : 1));
  w.encode_bytes = 2 * h_buffer_bytes(rows) + plan_bytes(rows);
  w.encode = take(w.encode_bytes);
  w.bytes = off;
  return w;
}
size_t encode_coo_f16_workspace_bytes(int64_t rows, int64_t e) {
  return carve_coo(nullptr, rows, e).bytes;
}

int launch_encode_coo_f16(const gfy_encoder* enc, const ShardTable& shards,
                          const RecordTable* records, int out_dtype, int normalise, void* ws,
                          size_t ws_bytes, hipStream_t s) {
  // before anything is launched (an error return must leave the workspace's counters zero): the
  // counting kernel's table keeps an edge's source row in 24 bits, 0xFFFFFF = none
  GFY_REQUIRE(shards.total_rows() < kCsrMaxRows, GFY_ERR_UNSUPPORTED,
              "gfy_encode_coo: at most 16,777,215 (padded) nodes per call (got %lld); split "
              "micro-batches / batches at record boundaries", (long long)shards.total_rows());
  const CooWorkspace w = carve_coo(ws, shards.total_rows(), shards.total_edges());
  GFY_REQUIRE(ws_bytes >= w.bytes, GFY_ERR_WORKSPACE,
              "gfy_encode_coo: workspace %zu < required %zu", ws_bytes, w.bytes);
  const bool scan_free = csr_scan_free(largest_shard_nodes(shards));
  // (Counting blocks and the input Linear in ONE launch — two independent jobs — were
  // measured: 41-46 us for the pair at 240,000 nodes against 16-22 + 17 us apart; the launch
  // holds more blocks than the chip, so the Linear's blocks only start when counting blocks
  // retire, and nothing overlaps.  profiles/README.md, round 4.)
  if (!records)
    if (const int rc = launch_csr_count_scan(w.scratch, w.scan_sums, shards, scan_free, w.row_ptr,
                                             shards.total_rows(), s))
      return rc;
  const CooInput coo{w.scratch, records, scan_free, w.row_ptr, w.col, w.typ};
  return encode_f16_on(enc, shards, nullptr, nullptr, nullptr, &coo, out_dtype, normalise, -1,
                       w.encode, w.encode_bytes, s);
}

// ---- parity tap of one layer (gfy_debug_layer) ----------------------------------------------
static size_t tap_bytes(int64_t rows) { return align_up((size_t)rows * kMlp * sizeof(f16), 256); }
size_t debug_layer_f16_workspace_bytes(int64_t n) {
  const int64_t rows = padded_rows(n);
  return 2 * h_buffer_bytes(rows) + plan_bytes(rows) + tap_bytes(rows);
}

int launch_debug_layer_f16(const gfy_encoder* enc, int layer, const void* hidden_in,
                           const int32_t* row_ptr, const int32_t* col, const uint8_t* typ,
                           int64_t n, int64_t e, int tap, void* out, void* ws, size_t ws_bytes,
                           hipStream_t s) {
  GFY_REQUIRE(enc->residual, GFY_ERR_UNSUPPORTED,
              "gfy_debug_layer: only the residual architecture has a tap instantiation");
  const ShardTable shards = single_shard(nullptr, nullptr, nullptr, n, e, nullptr, out);
  const int64_t rows = shards.total_rows();
  GFY_REQUIRE(ws_bytes >= debug_layer_f16_workspace_bytes(n), GFY_ERR_WORKSPACE,
              "gfy_debug_layer: workspace %zu < required %zu", ws_bytes,
              debug_layer_f16_workspace_bytes(n));
  f16* ha = (f16*)ws;
  f16* hb = (f16*)((char*)ws + h_buffer_bytes(rows));
  char* plans = (char*)ws + 2 * h_buffer_bytes(rows);
  f16* taps = (f16*)(plans + plan_bytes(rows));
  const int layer_tiles = shards.total_tiles();
  auto copy_rows = [&](const f16* from, f16* to, int64_t groups) {
    const int g = (int)((groups + 255) / 256);
    k_copy_rows_f16<<<g > 2048 ? 2048 : g, 256, 0, s>>>(from, to, groups);
  };
  GFY_CHECK_HIP(hipMemsetAsync(ha, 0, h_buffer_bytes(rows), s));   // padding rows: zeros
  copy_rows((const f16*)hidden_in, ha, n * 8);                      // natural -> stored order
  k_encode_setup<<<layer_tiles, 256, 0, s>>>(shards, enc->f16.w_in, enc->f16.b_in, ha, row_ptr,
                                             col, typ, plans, layer_tiles,
                                             enc->edge_dim);   // plans only
  // the tap instantiation of the layer kernel the encoder is set to (GFY_OPT_LAYER_KERNEL 4 / 5:
  // windowed / three workgroups per CU, where the edge table leaves room for their plan-head
  // slots; anything else: persistent rounds, whose pipelines the one-round kernel shares)
  const int tiles_per_xcd = (layer_tiles + 7) / 8;
  const int per_xcd = enc->cus / 8;
  const bool slots_ok = enc->edge_dim <= kWMaxEdgeTypes;
  if (slots_ok && enc->layer_kernel == 5) {
    const int wanted = (tiles_per_xcd + kXWaves - 1) / kXWaves;
    const int grid = 8 * (wanted < 3 * per_xcd ? wanted : 3 * per_xcd);
    k_gine_layer_x<true, true, false><<<grid, kXThreads, kXLdsBytes, s>>>(
        enc->f16.layer[layer], ha, hb, row_ptr, col, typ, plans, (int)n, layer_tiles, 0, 0,
        nullptr, nullptr, tap, taps, HeadOut{});
  } else if (slots_ok && enc->layer_kernel == 4) {
    const int wanted = (tiles_per_xcd + kWWaves - 1) / kWWaves;
    const int grid = 8 * (wanted < 2 * per_xcd ? wanted : 2 * per_xcd);
    k_gine_layer_w<true, true, false><<<grid, kWThreads, kWLdsBytes, s>>>(
        enc->f16.layer[layer], ha, hb, row_ptr, col, typ, plans, (int)n, layer_tiles, 0, 0,
        nullptr, nullptr, tap, taps, HeadOut{});
  } else {
    const int wanted = (tiles_per_xcd + kLWaves - 1) / kLWaves;
    const int grid = 8 * (wanted < per_xcd ? wanted : per_xcd);
    k_gine_layer_q<true, true, false><<<grid, kLThreads, kQLdsBytes, s>>>(
        enc->f16.layer[layer], ha, hb, row_ptr, col, typ, plans, (int)n, layer_tiles, 0, nullptr,
        nullptr, tap, taps, HeadOut{});
  }
  if (tap == kTapNone) copy_rows(hb, (f16*)out, n * 8);
  else copy_rows(taps, (f16*)out, n * (tap == kTapV ? 16 : 8));   // stored -> natural order
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

}  // namespace gfy
#endif   // GFY_PROBE_BUILD
