// All-pairs L2 / cosine over 128-d fp16 embeddings on the gfx950 matrix cores.
//
// The reference ships no implementation of this half of the north star (the
// aligner is the external `ginfinity-sw`; src/ginfinity/api.py:47-50 only
// exports its parameters), so the definition is ours:
//   L2      D_ij = sqrt(max(|a_i|^2 + |b_j|^2 - 2 a_i.b_j, 0))
//   cosine  S_ij = a_i.b_j / (max(|a_i|,1e-12) max(|b_j|,1e-12))
// oracle: oracle/gine_numpy.py pairwise_l2 / pairwise_cosine (float64).
//
// One kernel serves both outputs.  A workgroup (8 waves, 4x2, one per CU) owns 256
// a-rows, whose MFMA fragments stay in registers, and sweeps its chunk of b-rows in
// 128-row tiles that LDS-DMA (global_load_lds_dwordx4) lands in a ring of four LDS
// buffers three tiles ahead; every lane fetches the 16-byte chunk that belongs in its
// slot of the XOR-swizzled layout, so no register ever holds b-rows in flight.
// (Every workgroup streams all of B: 256 a-rows per workgroup instead of 128 halves
// that traffic — 1M x 1M: 2 TB -> 1 TB through L2 — and is what the ring's LDS buys.)
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

constexpr int kBlockA = 256;  // a-rows per workgroup
constexpr int kTileB = 128;   // b-rows per LDS tile
constexpr int kThreads = 512;
constexpr int kBuffers = 4;   // b-tile ring: tile i is consumed while i+1 .. i+3 are in flight
#ifndef GFY_PAIRWISE_REQUEST_AFTER_MULTIPLY
#define GFY_PAIRWISE_REQUEST_AFTER_MULTIPLY 1
#endif
#ifndef GFY_PAIRWISE_TILES_PER_BARRIER
#define GFY_PAIRWISE_TILES_PER_BARRIER 2   // nearest: 2 = a barrier per pair of tiles, 1 = per tile
#endif

template <class T>
__device__ __forceinline__ const T* uniform_pointer(const T* pointer) {
  const uint64_t bits = (uint64_t)(uintptr_t)pointer;
  const uint32_t low = __builtin_amdgcn_readfirstlane((uint32_t)bits);
  const uint32_t high = __builtin_amdgcn_readfirstlane((uint32_t)(bits >> 32));
  return reinterpret_cast<const T*>(((uint64_t)high << 32) | low);
}

__device__ __forceinline__ int off256(int row, int chunk) {
  return row * 256 + ((chunk ^ (row & 15)) << 4);
}

// per-row (s, t) on the b side, (na or 1/|a|) on the a side
// s/t are written for `padded` >= count rows: the rows past the end get key = +inf
// fold (L2 nearest): t = -|b_j|^2 / 2 — the accumulators of k_pairwise<false, true> START from
// it, so that what comes out of the MFMAs is already  g_ij = a_i.b_j - |b_j|^2 / 2  =  -key_ij / 2
// (maximised; -2 g is exact); padding rows get -inf
__global__ __launch_bounds__(256) void k_row_terms(const f16* __restrict__ rows, int64_t count,
                                                   int64_t padded, int metric, int fold,
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
      t_out[row] = metric == GFY_L2 ? (fold ? -0.5f * ss : ss) : 0.0f;
    }
    if (a_term) a_term[row] = metric == GFY_L2 ? ss : inv;
  }
  if (row >= count && row < padded && chunk == 0 && s_out) {
    s_out[row] = 0.f;
    t_out[row] = fold ? -__builtin_inff() : __builtin_inff();   // never wins
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
  int64_t exclude_offset;   // with exclude_on: the pair (i, i + exclude_offset) is skipped (any sign)
  int exclude_on;
  int blocks_a, chunks;
  int64_t chunk_rows;   // multiple of kTileB
  float* part_val;      // [chunks][n]   (nearest)
  int32_t* part_idx;    // [chunks][n]
  float* dense;         // [n][m]        (dense)
};

constexpr int kRowBytes = kTileB * 256;        // one b-tile of rows
constexpr int kTermBytes = 2 * kTileB * 4;     // its (s, t)
constexpr int kTermSlots = 4;                  // (s, t) ring, like the rows

// kFold (nearest, L2): see k_row_terms — the epilogue is a running maximum of the accumulators
// themselves: no per-element fma, no (s, t) operand reads per a-tile.
template <bool kDense, bool kFold = false>
__global__ __launch_bounds__(kThreads, 1) void k_pairwise(const PairArgs p) {
  static_assert(!(kDense && kFold), "the folded form is the nearest-row epilogue's");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // kBuffers buffers
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hq = lane >> 5;
  // wave owns a-rows [128wa, 128wa+128) and b-rows [32wb, 32wb+32) of each tile: one b operand
  // read from LDS feeds FOUR MFMAs (with 64 x 64 per wave it fed two, and the kernel was bound
  // by ds_read_b128 traffic: 58 % of the MFMA peak with DMA, barrier and epilogue all removed)
  const int wa = wave & 1, wb = wave >> 1;
  const int chunk = blockIdx.x / p.blocks_a;
  const int block_a = blockIdx.x - chunk * p.blocks_a;
  const int64_t a0 = (int64_t)block_a * kBlockA;
  const int64_t j_begin = (int64_t)chunk * p.chunk_rows;
  const int64_t j_end = j_begin + p.chunk_rows < p.m ? j_begin + p.chunk_rows : p.m;

  // one b-tile -> buffer `buf`: 128 rows as 32 DMA instructions (4 per wave, 4 rows each),
  // (s, t) as one more by waves 0 and 1.  Rows past the end re-read the last row; their
  // t is +inf (k_row_terms pads s/t to whole tiles).
  // Per-lane byte offset of its 16-byte piece q inside a tile: row 16 wave + 4 q + sub, slot
  // (lane & 15) ^ (row & 15)  =  (home ^ (q << 6)) + 1024 q  with ONE loop-invariant register
  // (`home`); the tile's base travels in SGPRs.  (64-bit per-lane pointers, or the four
  // offsets kept in registers, spilled — and a scratch reload is a vmcnt(0) wait that drains
  // the DMA look-ahead.  The asm keeps hipcc from hoisting them out of the loop again.)
  // `home` is rebuilt from threadIdx.x per request (six VALU operations): any loop-invariant
  // register here is one that hipcc spills.
  auto request = [&](int k) __attribute__((always_inline)) {   // tile k of this workgroup's sweep
    const int64_t j0 = j_begin + (int64_t)k * kTileB;
    const uint32_t base = lds0 + (uint32_t)(k & (kBuffers - 1)) * kRowBytes;
    // wave-uniform, and said so: with the carried reduce below in the loop hipcc's divergence
    // analysis puts j0 in vector registers, which the DMA's scalar base operand cannot take
    const f16* rows = uniform_pointer(p.b + j0 * 128);
    uint32_t me = threadIdx.x;
    asm volatile("" : "+v"(me));
    const uint32_t sub = (me >> 4) & 3u, slot = me & 15u;
    const uint32_t at_home = ((uint32_t)(16 * wave) + sub) * 256u + ((slot ^ sub) << 4);
    if (j0 + kTileB <= p.m) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dma16(rows, (at_home ^ (uint32_t)(q << 6)) + 1024u * q,
              base + (uint32_t)(wave * 4 + q) * 1024u);
    } else {   // ragged last tile: rows past the end re-read the last row (their t is +inf)
      const int last = (int)(p.m - 1 - j0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t full = (at_home ^ (uint32_t)(q << 6)) + 1024u * q;
        const int row = (int)(full >> 8);   // 16 wave + 4 q + sub
        const int from = row < last ? row : last;
        dma16(rows, (uint32_t)from * 256u + (full & 255u),
              base + (uint32_t)(wave * 4 + q) * 1024u);
      }
    }
    if (wave < (kFold ? 1 : 2) && (me & 32u) == 0)   // 128 floats = 32 lanes x 16 B
      dma16(uniform_pointer((wave == 0 && !kFold ? p.s : p.t) + j0), (me & 31u) * 16u,
            lds0 + kBuffers * kRowBytes + (uint32_t)(k & (kTermSlots - 1)) * kTermBytes
                + (uint32_t)(kFold ? 1 : wave) * (kTileB * 4));
  };

  // stage the a-block through LDS once (coalesced), then keep ALL its fragments in registers
  {
    char* atile = smem + kRowBytes;   // buffers 1 and 2 (64 KB), not yet in use
    for (int i = t; i < kBlockA * 16; i += kThreads) {
      const int row = i >> 4, ch = i & 15;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (a0 + row < p.n) v = *reinterpret_cast<const f16x8*>(p.a + (a0 + row) * 128 + ch * 8);
      *reinterpret_cast<f16x8*>(atile + off256(row, ch)) = v;
    }
  }
  if (j_begin < j_end) request(0);
  __syncthreads();
  f16x8 af[4][8];
#pragma unroll
  for (int at = 0; at < 4; ++at)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      af[at][ks] = *reinterpret_cast<const f16x8*>(
          smem + kRowBytes + off256(128 * wa + 32 * at + r, 2 * ks + hq));

  float best[4];   // running minimum of key (kFold: running maximum of g = -key / 2)
  int bidx[4];
#pragma unroll
  for (int at = 0; at < 4; ++at) {
    best[at] = kFold ? -__builtin_inff() : __builtin_inff();
    bidx[at] = 0x7fffffff;
  }

  // the a-block has left buffers 1 and 2: two more tiles go in flight
  __syncthreads();
  const int tiles = j_begin < j_end ? (int)((j_end - j_begin + kTileB - 1) / kTileB) : 0;
  if (tiles > 1) request(1);
  if (tiles > 2 && (kDense || GFY_PAIRWISE_TILES_PER_BARRIER != 2)) request(2);

  // the wave's 32 x 128 block of tile k: four independent accumulator chains (a 32x32x16
  // MFMA that reads the previous one's result stalls the issue port), the b operand read
  // kAheadK k-steps ahead
  f32x16 acc[4];   // [at]
  const int jw = 32 * wb + 4 * hq;   // first of this lane's b-rows inside a tile
  auto multiply = [&](int k) __attribute__((always_inline)) {
    const char* tile = smem + (k & (kBuffers - 1)) * kRowBytes;
    f32x16 start;    // what every chain starts from: 0, or (kFold) -|b_j|^2 / 2 of the lane's 16 b-rows
    if constexpr (kFold) {
      const float* u_l = reinterpret_cast<const float*>(
          smem + kBuffers * kRowBytes + (k & (kTermSlots - 1)) * kTermBytes) + kTileB;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 uv = *reinterpret_cast<const f32x4*>(u_l + jw + 8 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) start[4 * g + i] = uv[i];
      }
    } else {
#pragma unroll
      for (int at = 0; at < 4; ++at)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[at][q] = 0.f;
    }
    constexpr int kAheadK = 2, kRing = kAheadK + 1;
    f16x8 bf[kRing];   // [ks % kRing]
#pragma unroll
    for (int ks = 0; ks < kAheadK; ++ks)
      bf[ks] = *reinterpret_cast<const f16x8*>(tile + off256(32 * wb + r, 2 * ks + hq));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + kAheadK < 8)
        bf[(ks + kAheadK) % kRing] = *reinterpret_cast<const f16x8*>(
            tile + off256(32 * wb + r, 2 * (ks + kAheadK) + hq));
      // hipcc otherwise sinks every operand read down to its MFMAs (one register quad,
      // read -> lgkmcnt(0) -> MFMAs: the LDS latency exposed eight times a tile)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int at = 0; at < 4; ++at)
        acc[at] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks % kRing], af[at][ks],
                                                         kFold && ks == 0 ? start : acc[at],
                                                         0, 0, 0);
    }
  };

  // what happens to the products of tile k (still in acc)
  auto reduce = [&](int k) __attribute__((always_inline)) {
    const int64_t j0 = j_begin + (int64_t)k * kTileB;
    const float* s_l = reinterpret_cast<const float*>(
        smem + kBuffers * kRowBytes + (k & (kTermSlots - 1)) * kTermBytes);
    const float* t_l = s_l + kTileB;
    if constexpr (kDense) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int jl = jw + 8 * g;  // 4 consecutive b-rows
        const f32x4 sv = *reinterpret_cast<const f32x4*>(s_l + jl);
        const f32x4 tv = *reinterpret_cast<const f32x4*>(t_l + jl);
#pragma unroll
        for (int at = 0; at < 4; ++at) {
          const int64_t ai = a0 + 128 * wa + 32 * at + r;
          const float aterm = ai < p.n ? p.a_term[ai] : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int64_t j = j0 + jl + i;
            const float key = __builtin_fmaf(acc[at][4 * g + i], sv[i], tv[i]);
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
    } else if constexpr (kFold) {
      const int64_t ex_lo = a0 + p.exclude_offset, ex_hi = ex_lo + kBlockA;
      const bool may_exclude = p.exclude_on && ex_lo < j0 + kTileB && ex_hi > j0;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int at = 0; at < 4; ++at) {
        if (may_exclude) {   // block-uniform, at most three tiles per block
          const int64_t off = (a0 + p.exclude_offset - j0) + (128 * wa + 32 * at + r - jw);
          const int d = off >= 0 && off < 32 ? (int)off : 4;   // 4: not a position of this lane
          const int slot = (d & 4) ? -1 : (d >> 3) * 4 + (d & 3);
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[at][q] = q == slot ? -__builtin_inff() : acc[at][q];
        }
        float high = __builtin_fmaxf(acc[at][0], acc[at][1]);
#pragma unroll
        for (int q = 2; q < 16; q += 2)
          high = __builtin_fmaxf(__builtin_fmaxf(high, acc[at][q]), acc[at][q + 1]);   // v_max3_f32
        const bool better = high > best[at];   // strict: an earlier tile keeps a tie
        if (__ballot(better)) {                // wave-uniform skip once the sweep has settled
          int first = 15;                      // lowest position holding the maximum
#pragma unroll
          for (int q = 14; q >= 0; --q) first = acc[at][q] == high ? q : first;
          const int j = (int)(j0 + jw + 8 * (first >> 2) + (first & 3));
          best[at] = better ? high : best[at];
          bidx[at] = better ? j : bidx[at];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // does this tile contain an excluded (i, i + offset) pair of this block?
      const int64_t ex_lo = a0 + p.exclude_offset, ex_hi = ex_lo + kBlockA;
      const bool may_exclude = p.exclude_on && ex_lo < j0 + kTileB && ex_hi > j0;
      __builtin_amdgcn_sched_barrier(0);
      // The contraction is only 128 deep, so an epilogue of fma + compare + two selects
      // per element costs more vector cycles than the MFMAs that produced it.  Instead:
      // 16 keys per a-row with packed fmas, their minimum with v_min3 (~1 instruction per
      // element in all), and the position of the minimum is recovered only while some
      // lane of the wave still improves its running best — soon rare.
#pragma unroll
      for (int at = 0; at < 4; ++at) {
        f32x4 key[4];   // one a-row's 16 keys at a time (registers); s/t are re-read per a-tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int jl = jw + 8 * g;  // 4 consecutive b-rows
          const f32x4 sv = *reinterpret_cast<const f32x4*>(s_l + jl);
          const f32x4 tv = *reinterpret_cast<const f32x4*>(t_l + jl);
          f32x4 a4;
#pragma unroll
          for (int i = 0; i < 4; ++i) a4[i] = acc[at][4 * g + i];
          key[g] = __builtin_elementwise_fma(a4, sv, tv);
        }
        if (may_exclude) {   // block-uniform, at most three tiles per block
          // the one excluded b-row of this a-row, as a position among the lane's 16 keys
          const int64_t off = (a0 + p.exclude_offset - j0) + (128 * wa + 32 * at + r - jw);
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
          const int j = (int)(j0 + jw + 8 * (first >> 2) + (first & 3));
          best[at] = better ? low : best[at];
          bidx[at] = better ? j : bidx[at];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Barrier k: wait for this wave's share of tile k only — the DMA instructions of the (up
  // to two) younger requests stay in flight, 4 per request, 5 on the waves that fetch
  // (s, t) — then the barrier, which publishes everybody's share and says that the rows of
  // tile k - 1 are no longer being read, so tile k + 3 can be requested into their buffer.
  auto sync = [&](int k) __attribute__((always_inline)) {
    const int younger = tiles - 1 - k < 2 ? tiles - 1 - k : 2;
    const bool fetches_terms = wave < (kFold ? 1 : 2);
    if (younger == 2) {
      if (fetches_terms) __builtin_amdgcn_s_waitcnt(0x0F7A); else __builtin_amdgcn_s_waitcnt(0x0F78);
    } else if (younger == 1) {
      if (fetches_terms) __builtin_amdgcn_s_waitcnt(0x0F75); else __builtin_amdgcn_s_waitcnt(0x0F74);
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    if (k + 3 < tiles) request(k + 3);
  };

  // (Running the two waves of a SIMD half a tile out of phase — one multiplies while the other
  // reduces — was measured: no change, see profiles/README.md.)
  if constexpr (GFY_PAIRWISE_TILES_PER_BARRIER == 2 && !kDense) {
    // Two tiles per barrier: the ring holds the pair being consumed and the pair in flight.
    // Between two barriers each wave runs multiply, reduce, multiply, reduce on its own, so
    // the two waves of a SIMD interleave (one reduces under the other's MFMAs) and the matrix
    // core only idles for one reduce per PAIR of tiles.
    // (-DGFY_PW_NO_*: timing experiments, results are garbage — profiles/README.md)
    auto reduce_x = [&](int k) __attribute__((always_inline)) {
#ifdef GFY_PW_NO_REDUCE
#pragma unroll
      for (int at = 0; at < 4; ++at) best[at] = __builtin_fmaxf(best[at], acc[at][0]);
#else
      reduce(k);
#endif
    };
    // kFold: waves 4..7 (the second wave of every SIMD) carry the reduce of a pair's second
    // tile over the barrier — it reads registers only — so that the two waves of a SIMD leave
    // the barrier one reduce apart and stay that way: one reduces while the other multiplies.
    // (Not for the (s, t) form: its reduce reads the term ring, whose slot the requests after
    // the barrier refill.)
    const bool late = kFold && wave >= 4;
    int carried = -1;
    for (int ti = 0; ti < tiles; ti += 2) {
      __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this wave's share of the pair
      asm volatile("" ::: "memory");
      __syncthreads();                             // everybody's share; the previous pair is spent
#if GFY_PAIRWISE_REQUEST_AFTER_MULTIPLY == 0
      if (ti + 2 < tiles) request(ti + 2);
      if (ti + 3 < tiles) request(ti + 3);
#endif
      if (carried >= 0) reduce_x(carried);
      carried = -1;
      multiply(ti);
#if GFY_PAIRWISE_REQUEST_AFTER_MULTIPLY
      // the next pair is requested BEHIND the first multiply: issuing a tile's eight DMA
      // pieces costs a wave several hundred cycles, and right behind the barrier all eight
      // waves would pay them at once with the matrix cores idle; here waves 0..3 and 4..7 come
      // by one reduce apart, each under the other's MFMAs
      if (ti + 2 < tiles) request(ti + 2);
#if GFY_PAIRWISE_REQUEST_AFTER_MULTIPLY == 1
      if (ti + 3 < tiles) request(ti + 3);
#endif
#endif
      reduce_x(ti);
#if GFY_PAIRWISE_REQUEST_AFTER_MULTIPLY == 2
      if (ti + 3 < tiles) request(ti + 3);
#endif
      if (ti + 1 < tiles) {
        multiply(ti + 1);
        if (late) carried = ti + 1; else reduce_x(ti + 1);
      }
    }
    if (carried >= 0) reduce_x(carried);
  } else {
    for (int ti = 0; ti < tiles; ++ti) {
      sync(ti);
      multiply(ti);
      reduce(ti);
    }
  }
  __syncthreads();   // the result merge below reuses the first buffer

  if constexpr (!kDense) {
    if constexpr (kFold) {   // back to keys: -2 g is exact, order and ties carry over
#pragma unroll
      for (int at = 0; at < 4; ++at) best[at] *= -2.0f;
    }
    // merge the two lane halves (different b-rows, same a-row), then the two
    // waves that share this a-row range (wb = 0/1) through LDS
    float* m_val = reinterpret_cast<float*>(smem);          // [4 wb][kBlockA]
    int* m_idx = reinterpret_cast<int*>(smem + 4 * kBlockA * 4);
#pragma unroll
    for (int at = 0; at < 4; ++at) {
      const float ov = __shfl_xor(best[at], 32, 64);
      const int oi = __shfl_xor(bidx[at], 32, 64);
      if (ov < best[at] || (ov == best[at] && oi < bidx[at])) {
        best[at] = ov;
        bidx[at] = oi;
      }
      if (hq == 0) {
        m_val[wb * kBlockA + 128 * wa + 32 * at + r] = best[at];
        m_idx[wb * kBlockA + 128 * wa + 32 * at + r] = bidx[at];
      }
    }
    __syncthreads();
    if (t < kBlockA && a0 + t < p.n) {
      float v0 = m_val[t];
      int i0 = m_idx[t];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float v1 = m_val[w * kBlockA + t];
        const int i1 = m_idx[w * kBlockA + t];
        if (v1 < v0 || (v1 == v0 && i1 < i0)) {
          v0 = v1;
          i0 = i1;
        }
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
  {
    // One workgroup per CU at a time, all of one length: a grid of blocks_a x chunks workgroups
    // ends after ceil(grid / CUs) of them, each 1 / chunks of a sweep long.  1,000,000 rows are
    // 3,907 a-blocks = 15.26 per CU: one chunk ends after 16 sweeps, three after 46 / 3 = 15.33.
    // A few more chunks than the minimum cost a merge entry per a-row and chunk (k_nearest_finish).
    constexpr int64_t kCus = 256;   // MI355X; another part only loses the fit
    const int64_t least = chunks;
    double best = 1e300;
    for (int64_t c = least; c < least + 6; ++c) {
      const double sweeps = (double)((w.blocks_a * c + kCus - 1) / kCus) / (double)c;
      if (sweeps < best * 0.99) best = sweeps, chunks = c;   // a later count only for a real gain
    }
  }
#ifdef GFY_DIAG_PAIRWISE_CHUNKS   // diagnostic builds: the sweep behind the choice above (profiles/README.md)
  if (const char* forced = getenv("GFY_PAIRWISE_CHUNKS")) chunks = atoll(forced);
#endif
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

constexpr int kPairLds = kBuffers * kRowBytes + kTermSlots * kTermBytes;   // row ring (buffers 1, 2 stage the a-block first) + (s, t) ring

}  // namespace

size_t pairwise_workspace_bytes(int64_t n, int64_t m) { return carve(nullptr, n, m).bytes; }

// > 64 KB of dynamic LDS: opt in once per device, thread-safe (gfy_common.h)
static PerDeviceOnce g_pairwise_lds_opt_in;
static int opt_in_pairwise_lds() {
  return g_pairwise_lds_opt_in.run([]() -> int {
    GFY_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
    GFY_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
    GFY_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise<false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
    return GFY_OK;
  });
}

int launch_pairwise_nearest(const void* a, int64_t n, const void* b, int64_t m,
                            int metric, int64_t exclude_offset, int exclude_on, float* best_val,
                            int32_t* best_idx, void* ws, size_t ws_bytes,
                            hipStream_t s) {
  const PairWorkspace w = carve(ws, n, m);
  GFY_REQUIRE(ws_bytes >= w.bytes, GFY_ERR_WORKSPACE,
              "gfy_pairwise_nearest: workspace %zu < required %zu", ws_bytes, w.bytes);
  const f16* ap = (const f16*)a;
  const f16* bp = (const f16*)b;
  const int64_t padded_m = (m + kTileB - 1) / kTileB * kTileB;
  const int fold = metric == GFY_L2;
  k_row_terms<<<(int)((padded_m * 16 + 255) / 256), 256, 0, s>>>(bp, m, padded_m, metric, fold, w.s, w.t, nullptr);
  k_row_terms<<<(int)((n * 16 + 255) / 256), 256, 0, s>>>(ap, n, n, metric, 0, nullptr, nullptr, w.a_term);
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
  p.exclude_on = exclude_on;
  p.blocks_a = w.blocks_a;
  p.chunks = w.chunks;
  p.chunk_rows = w.chunk_rows;
  p.part_val = w.part_val;
  p.part_idx = w.part_idx;
  if (const int rc = opt_in_pairwise_lds()) return rc;
  if (fold) k_pairwise<false, true><<<w.blocks_a * w.chunks, kThreads, kPairLds, s>>>(p);
  else k_pairwise<false><<<w.blocks_a * w.chunks, kThreads, kPairLds, s>>>(p);
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
  k_row_terms<<<(int)((padded_m * 16 + 255) / 256), 256, 0, s>>>(bp, m, padded_m, metric, 0, sv, tv, nullptr);
  k_row_terms<<<(int)((n * 16 + 255) / 256), 256, 0, s>>>(ap, n, n, metric, 0, nullptr, nullptr, at);
  PairArgs p{};
  p.a = ap;
  p.b = bp;
  p.s = sv;
  p.t = tv;
  p.a_term = at;
  p.n = n;
  p.m = m;
  p.metric = metric;
  p.exclude_offset = 0;
  p.exclude_on = 0;
  p.blocks_a = (int)((n + kBlockA - 1) / kBlockA);
  p.chunks = 1;
  p.chunk_rows = (m + kTileB - 1) / kTileB * kTileB;
  p.dense = out;
  if (const int rc = opt_in_pairwise_lds()) return rc;
  k_pairwise<true><<<p.blocks_a, kThreads, kPairLds, s>>>(p);
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

}  // namespace gfy
