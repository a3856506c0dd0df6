// All-pairs L2 / cosine over 128-d fp16 embeddings on the gfx950 matrix cores.
//
// The reference ships no implementation of this half of the north star (the
// aligner is the external `ginfinity-sw`; src/ginfinity/api.py:47-50 only
// exports its parameters), so the definition is ours:
//   L2      D_ij = sqrt(max(|a_i|^2 + |b_j|^2 - 2 a_i.b_j, 0))
//   cosine  S_ij = a_i.b_j / (max(|a_i|,1e-12) max(|b_j|,1e-12))
// oracle: oracle/gine_numpy.py pairwise_l2 / pairwise_cosine (float64).
//
// One kernel serves both outputs.  A workgroup (4 waves, 2x2) owns 128 a-rows,
// whose MFMA fragments stay in registers, and sweeps its chunk of b-rows in
// 128-row tiles that LDS-DMA (global_load_lds_dwordx4) lands in one of two LDS
// buffers a tile ahead; every lane fetches the 16-byte chunk that belongs in its
// slot of the XOR-swizzled layout, so no register ever holds b-rows in flight.
// The product is taken as (B-tile) x (A-block)^T so the a-row sits on the MFMA
// lane: the running best of an a-row is lane-local state and a lane's four
// consecutive accumulator registers are four consecutive b-rows.
// Both metrics reduce to minimising  key_ij = fma(dot_ij, s_j, t_j):
//   L2      (s_j, t_j) = (-2, |b_j|^2)          value = sqrt(max(|a_i|^2 + key, 0))
//   cosine  (s_j, t_j) = (-1/|b_j|, 0)          value = -key / |a_i|
// `nearest` keeps (min key, arg min) per a-row, ties to the lowest index; with
// the b-rows split over several workgroups the partial results are merged by a
// second small kernel.  Workgroups are numbered chunk-major so the ones running
// together sweep the same b-tiles and share them through L2.
#include "gfy_common.h"

namespace gfy {
namespace {

constexpr int kBlockA = 128;  // a-rows per workgroup
constexpr int kTileB = 128;   // b-rows per LDS tile
constexpr int kThreads = 256;

__device__ __forceinline__ int off256(int row, int chunk) {
  return row * 256 + ((chunk ^ (row & 15)) << 4);
}

// per-row (s, t) on the b side, (na or 1/|a|) on the a side
// s/t are written for `padded` >= count rows: the rows past the end get key = +inf
__global__ __launch_bounds__(256) void k_row_terms(const f16* __restrict__ rows, int64_t count,
                                                   int64_t padded, int metric,
                                                   float* __restrict__ s_out,
                                                   float* __restrict__ t_out,
                                                   float* __restrict__ a_term) {
  const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = item >> 4;
  const int chunk = (int)(item & 15);
  float ss = 0.f;
  if (row < count) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(rows + row * 128 + chunk * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss = __builtin_fmaf((float)v[j], (float)v[j], ss);
  }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) ss += __shfl_xor(ss, m, 64);
  if (row < count && chunk == 0) {
    const float nrm = __builtin_sqrtf(ss);
    const float inv = 1.0f / (nrm > 1e-12f ? nrm : 1e-12f);
    if (s_out) {
      s_out[row] = metric == GFY_L2 ? -2.0f : -inv;
      t_out[row] = metric == GFY_L2 ? ss : 0.0f;
    }
    if (a_term) a_term[row] = metric == GFY_L2 ? ss : inv;
  }
  if (row >= count && row < padded && chunk == 0 && s_out) {
    s_out[row] = 0.f;
    t_out[row] = __builtin_inff();   // never wins
  }
}

struct PairArgs {
  const f16* a;
  const f16* b;
  const float* s;       // [m]
  const float* t;       // [m]
  const float* a_term;  // [n]
  int64_t n, m;
  int metric;
  int64_t exclude_offset;
  int blocks_a, chunks;
  int64_t chunk_rows;   // multiple of kTileB
  float* part_val;      // [chunks][n]   (nearest)
  int32_t* part_idx;    // [chunks][n]
  float* dense;         // [n][m]        (dense)
};

constexpr int kBufBytes = kTileB * 256 + 2 * kTileB * 4;   // b-tile + its (s, t)

template <bool kDense>
__global__ __launch_bounds__(kThreads, 2) void k_pairwise(const PairArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // two buffers
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hq = lane >> 5;
  const int wa = wave & 1, wb = wave >> 1;  // wave owns a-rows [64wa,64wa+64), b-rows [64wb, 64wb+64) of each tile
  const int chunk = blockIdx.x / p.blocks_a;
  const int block_a = blockIdx.x - chunk * p.blocks_a;
  const int64_t a0 = (int64_t)block_a * kBlockA;
  const int64_t j_begin = (int64_t)chunk * p.chunk_rows;
  const int64_t j_end = j_begin + p.chunk_rows < p.m ? j_begin + p.chunk_rows : p.m;

  // one b-tile -> buffer `buf`: 128 rows as 32 DMA instructions (8 per wave, 4 rows each),
  // (s, t) as one more by waves 0 and 1.  Rows past the end re-read the last row; their
  // t is +inf (k_row_terms pads s/t to whole tiles).
  auto request = [&](int64_t j0, int buf) {
    const uint32_t base = lds0 + (uint32_t)buf * kBufBytes;
    const int sub = lane >> 4, slot = lane & 15;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int g = wave * 8 + q;
      const int row = 4 * g + sub;
      int64_t j = j0 + row;
      j = j < p.m ? j : p.m - 1;
      dma16_at(p.b + j * 128 + (slot ^ (row & 15)) * 8, base + (uint32_t)g * 1024u);
    }
    if (wave < 2 && lane < 32)   // 128 floats = 32 lanes x 16 B
      dma16_at((wave == 0 ? p.s : p.t) + j0 + lane * 4,
               base + kTileB * 256 + (uint32_t)wave * (kTileB * 4));
  };

  // stage the a-block through LDS once (coalesced), then keep ALL its fragments in registers
  {
    char* atile = smem + kBufBytes;   // second buffer, not yet in use
    for (int i = t; i < kBlockA * 16; i += kThreads) {
      const int row = i >> 4, ch = i & 15;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (a0 + row < p.n) v = *reinterpret_cast<const f16x8*>(p.a + (a0 + row) * 128 + ch * 8);
      *reinterpret_cast<f16x8*>(atile + off256(row, ch)) = v;
    }
  }
  if (j_begin < j_end) request(j_begin, 0);
  __syncthreads();
  f16x8 af[2][8];
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      af[at][ks] = *reinterpret_cast<const f16x8*>(
          smem + kBufBytes + off256(64 * wa + 32 * at + r, 2 * ks + hq));

  float best[2];
  int bidx[2];
#pragma unroll
  for (int at = 0; at < 2; ++at) {
    best[at] = __builtin_inff();
    bidx[at] = 0x7fffffff;
  }

  int cur = 0;
  for (int64_t j0 = j_begin; j0 < j_end; j0 += kTileB, cur ^= 1) {
    // this wave's share of tile j0 has landed; the barrier publishes everybody's and tells
    // us that the other buffer (tile j0 - 128, or the a-block) is no longer being read
    dma_wait_all();
    __syncthreads();
    if (j0 + kTileB < j_end) request(j0 + kTileB, cur ^ 1);
    const char* tile = smem + cur * kBufBytes;
    const float* s_l = reinterpret_cast<const float*>(tile + kTileB * 256);
    const float* t_l = s_l + kTileB;

    // does this tile contain an excluded (i, i + offset) pair of this block?
    const int64_t ex_lo = a0 + p.exclude_offset, ex_hi = ex_lo + kBlockA;
    const bool may_exclude = p.exclude_offset >= 0 && ex_lo < j0 + kTileB && ex_hi > j0;

    // the wave's 64 x 64 block of this tile: four independent accumulator chains (a
    // 32x32x16 MFMA that reads the previous one's result stalls the issue port; two chains
    // per wave left 46 % of the wave cycles in that stall), b operands read one k-step ahead
    f32x16 acc4[2][2];   // [bt][at]
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
      for (int at = 0; at < 2; ++at)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc4[bt][at][q] = 0.f;
    {
      f16x8 bf[2][2];   // [parity of ks][bt]
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
        bf[0][bt] = *reinterpret_cast<const f16x8*>(tile + off256(64 * wb + 32 * bt + r, hq));
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks < 7) {
#pragma unroll
          for (int bt = 0; bt < 2; ++bt)
            bf[(ks + 1) & 1][bt] = *reinterpret_cast<const f16x8*>(
                tile + off256(64 * wb + 32 * bt + r, 2 * (ks + 1) + hq));
        }
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
          for (int at = 0; at < 2; ++at)
            acc4[bt][at] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks & 1][bt], af[at][ks],
                                                                  acc4[bt][at], 0, 0, 0);
      }
    }
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      f32x16(&accs)[2] = acc4[bt];
      if constexpr (kDense) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int jl = 64 * wb + 32 * bt + 8 * g + 4 * hq;  // 4 consecutive b-rows
          const f32x4 sv = *reinterpret_cast<const f32x4*>(s_l + jl);
          const f32x4 tv = *reinterpret_cast<const f32x4*>(t_l + jl);
#pragma unroll
          for (int at = 0; at < 2; ++at) {
            const int64_t ai = a0 + 64 * wa + 32 * at + r;
            const float aterm = ai < p.n ? p.a_term[ai] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int64_t j = j0 + jl + i;
              const float key = __builtin_fmaf(accs[at][4 * g + i], sv[i], tv[i]);
              if (ai < p.n && j < p.m) {
                float val;
                if (p.metric == GFY_L2) {
                  const float d2 = aterm + key;
                  val = __builtin_sqrtf(d2 > 0.f ? d2 : 0.f);
                } else {
                  val = -key * aterm;
                }
                p.dense[ai * p.m + j] = val;
              }
            }
          }
        }
      } else {
        __builtin_amdgcn_sched_barrier(0);   // keep the next b-tile's operand reads behind us
        // The contraction is only 128 deep, so an epilogue of fma + compare + two selects
        // per element costs more vector cycles than the MFMAs that produced it.  Instead:
        // 16 keys per a-row with packed fmas, their minimum with v_min3 (~1 instruction per
        // element in all), and the position of the minimum is recovered only while some
        // lane of the wave still improves its running best — soon rare.
#pragma unroll
        for (int at = 0; at < 2; ++at) {
          const int64_t ai = a0 + 64 * wa + 32 * at + r;
          f32x4 key[4];   // one a-row's 16 keys at a time (registers); s/t are re-read per a-tile
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int jl = 64 * wb + 32 * bt + 8 * g + 4 * hq;  // 4 consecutive b-rows
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s_l + jl);
            const f32x4 tv = *reinterpret_cast<const f32x4*>(t_l + jl);
            f32x4 a4;
#pragma unroll
            for (int i = 0; i < 4; ++i) a4[i] = accs[at][4 * g + i];
            key[g] = __builtin_elementwise_fma(a4, sv, tv);
          }
          if (may_exclude) {   // block-uniform, at most two tiles per block
            // the one excluded b-row of this a-row, as a position among the lane's 16 keys
            const int64_t off = ai + p.exclude_offset - (j0 + 64 * wb + 32 * bt + 4 * hq);
            const int d = off >= 0 && off < 32 ? (int)off : 4;   // 4: not a position of this lane
            const int slot = (d & 4) ? -1 : (d >> 3) * 4 + (d & 3);
#pragma unroll
            for (int q = 0; q < 16; ++q)
              key[q >> 2][q & 3] = q == slot ? __builtin_inff() : key[q >> 2][q & 3];
          }
          float low = __builtin_fminf(key[0][0], key[0][1]);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = (g == 0 ? 2 : 0); i < 4; i += 2)
              low = __builtin_fminf(__builtin_fminf(low, key[g][i]), key[g][i + 1]);   // v_min3_f32
          const bool better = low < best[at];   // strict: an earlier tile keeps a tie
          if (__ballot(better)) {               // wave-uniform skip once the sweep has settled
            int first = 15;                     // lowest position holding the minimum
#pragma unroll
            for (int q = 14; q >= 0; --q) first = key[q >> 2][q & 3] == low ? q : first;
            const int j = (int)(j0 + 64 * wb + 32 * bt + 8 * (first >> 2) + 4 * hq + (first & 3));
            best[at] = better ? low : best[at];
            bidx[at] = better ? j : bidx[at];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();   // the result merge below reuses the first buffer

  if constexpr (!kDense) {
    // merge the two lane halves (different b-rows, same a-row), then the two
    // waves that share this a-row range (wb = 0/1) through LDS
    float* m_val = reinterpret_cast<float*>(smem);          // [2 wb][128]
    int* m_idx = reinterpret_cast<int*>(smem + 2 * kBlockA * 4);
#pragma unroll
    for (int at = 0; at < 2; ++at) {
      const float ov = __shfl_xor(best[at], 32, 64);
      const int oi = __shfl_xor(bidx[at], 32, 64);
      if (ov < best[at] || (ov == best[at] && oi < bidx[at])) {
        best[at] = ov;
        bidx[at] = oi;
      }
      if (hq == 0) {
        m_val[wb * kBlockA + 64 * wa + 32 * at + r] = best[at];
        m_idx[wb * kBlockA + 64 * wa + 32 * at + r] = bidx[at];
      }
    }
    __syncthreads();
    if (t < kBlockA && a0 + t < p.n) {
      float v0 = m_val[t], v1 = m_val[kBlockA + t];
      int i0 = m_idx[t], i1 = m_idx[kBlockA + t];
      if (v1 < v0 || (v1 == v0 && i1 < i0)) {
        v0 = v1;
        i0 = i1;
      }
      p.part_val[(int64_t)chunk * p.n + a0 + t] = v0;
      p.part_idx[(int64_t)chunk * p.n + a0 + t] = i0;
    }
  }
}

__global__ __launch_bounds__(256) void k_nearest_finish(const float* __restrict__ part_val,
                                                        const int32_t* __restrict__ part_idx,
                                                        const float* __restrict__ a_term,
                                                        int64_t n, int chunks, int metric,
                                                        float* __restrict__ best_val,
                                                        int32_t* __restrict__ best_idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = part_val[i];
  int idx = part_idx[i];
  for (int c = 1; c < chunks; ++c) {
    const float ov = part_val[(int64_t)c * n + i];
    const int oi = part_idx[(int64_t)c * n + i];
    if (ov < v || (ov == v && oi < idx)) {
      v = ov;
      idx = oi;
    }
  }
  const float at = a_term[i];
  float out;
  if (metric == GFY_L2) {
    const float d2 = at + v;
    out = __builtin_sqrtf(d2 > 0.f ? d2 : 0.f);
  } else {
    out = -v * at;
  }
  best_val[i] = out;
  best_idx[i] = idx == 0x7fffffff ? -1 : idx;
}

struct PairWorkspace {
  float *s, *t, *a_term, *part_val;
  int32_t* part_idx;
  int blocks_a, chunks;
  int64_t chunk_rows;
  size_t bytes;
};

PairWorkspace carve(void* base, int64_t n, int64_t m) {
  PairWorkspace w;
  w.blocks_a = (int)((n + kBlockA - 1) / kBlockA);
  const int64_t tiles_b = (m + kTileB - 1) / kTileB;
  int64_t chunks = (1024 + w.blocks_a - 1) / w.blocks_a;
  if (chunks > tiles_b) chunks = tiles_b;
  if (chunks < 1) chunks = 1;
  const int64_t tiles_per_chunk = (tiles_b + chunks - 1) / chunks;
  w.chunk_rows = tiles_per_chunk * kTileB;
  w.chunks = (int)((tiles_b + tiles_per_chunk - 1) / tiles_per_chunk);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void* ptr = base ? (char*)base + off : nullptr;
    off += align_up(bytes, 256);
    return ptr;
  };
  w.s = (float*)take((size_t)tiles_b * kTileB * 4);   // padded to whole tiles
  w.t = (float*)take((size_t)tiles_b * kTileB * 4);
  w.a_term = (float*)take((size_t)n * 4);
  w.part_val = (float*)take((size_t)w.chunks * n * 4);
  w.part_idx = (int32_t*)take((size_t)w.chunks * n * 4);
  w.bytes = off;
  return w;
}

constexpr int kPairLds = 2 * kBufBytes;   // two b-tile buffers (the second stages the a-block first)

}  // namespace

size_t pairwise_workspace_bytes(int64_t n, int64_t m) { return carve(nullptr, n, m).bytes; }

int launch_pairwise_nearest(const void* a, int64_t n, const void* b, int64_t m,
                            int metric, int64_t exclude_offset, float* best_val,
                            int32_t* best_idx, void* ws, size_t ws_bytes,
                            hipStream_t s) {
  const PairWorkspace w = carve(ws, n, m);
  GFY_REQUIRE(ws_bytes >= w.bytes, GFY_ERR_WORKSPACE,
              "gfy_pairwise_nearest: workspace %zu < required %zu", ws_bytes, w.bytes);
  const f16* ap = (const f16*)a;
  const f16* bp = (const f16*)b;
  const int64_t padded_m = (m + kTileB - 1) / kTileB * kTileB;
  k_row_terms<<<(int)((padded_m * 16 + 255) / 256), 256, 0, s>>>(bp, m, padded_m, metric, w.s, w.t, nullptr);
  k_row_terms<<<(int)((n * 16 + 255) / 256), 256, 0, s>>>(ap, n, n, metric, nullptr, nullptr, w.a_term);
  PairArgs p{};
  p.a = ap;
  p.b = bp;
  p.s = w.s;
  p.t = w.t;
  p.a_term = w.a_term;
  p.n = n;
  p.m = m;
  p.metric = metric;
  p.exclude_offset = exclude_offset;
  p.blocks_a = w.blocks_a;
  p.chunks = w.chunks;
  p.chunk_rows = w.chunk_rows;
  p.part_val = w.part_val;
  p.part_idx = w.part_idx;
  static bool lds_opt_in = false;
  if (!lds_opt_in) {
    GFY_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
    GFY_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
    lds_opt_in = true;
  }
  k_pairwise<false><<<w.blocks_a * w.chunks, kThreads, kPairLds, s>>>(p);
  k_nearest_finish<<<(int)((n + 255) / 256), 256, 0, s>>>(
      w.part_val, w.part_idx, w.a_term, n, w.chunks, metric, best_val, best_idx);
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

int launch_pairwise_dense(const void* a, int64_t n, const void* b, int64_t m,
                          int metric, float* out, void* ws, size_t ws_bytes,
                          hipStream_t s) {
  const PairWorkspace w = carve(ws, n, m);
  GFY_REQUIRE(ws_bytes >= w.bytes, GFY_ERR_WORKSPACE,
              "gfy_pairwise_dense: workspace %zu < required %zu", ws_bytes, w.bytes);
  float* sv = w.s;
  float* tv = w.t;
  float* at = w.a_term;
  const f16* ap = (const f16*)a;
  const f16* bp = (const f16*)b;
  const int64_t padded_m = (m + kTileB - 1) / kTileB * kTileB;
  k_row_terms<<<(int)((padded_m * 16 + 255) / 256), 256, 0, s>>>(bp, m, padded_m, metric, sv, tv, nullptr);
  k_row_terms<<<(int)((n * 16 + 255) / 256), 256, 0, s>>>(ap, n, n, metric, nullptr, nullptr, at);
  PairArgs p{};
  p.a = ap;
  p.b = bp;
  p.s = sv;
  p.t = tv;
  p.a_term = at;
  p.n = n;
  p.m = m;
  p.metric = metric;
  p.exclude_offset = -1;
  p.blocks_a = (int)((n + kBlockA - 1) / kBlockA);
  p.chunks = 1;
  p.chunk_rows = (m + kTileB - 1) / kTileB * kTileB;
  p.dense = out;
  GFY_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
  k_pairwise<true><<<p.blocks_a, kThreads, kPairLds, s>>>(p);
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

}  // namespace gfy
