// COO -> destination-major CSR, edges of one destination in COO order.
//
// Replaces the addressing half of the reference's message passing
// (src/ginfinity/_model.py:41-45: index_select by source, index_add_ by
// destination, on int64 copies made at api.py:239-242).  Integer work: the
// result is bit-exact against a stable argsort by destination
// (oracle/gine_numpy.py:build_csr) and identical from run to run.
//
// Pipeline (all on one stream, no host sync):
//   1 count    slot = atomicAdd(count[dst[e]], 1); the first kCsrSlots ids of a row go to
//              table[dst][slot], later ones to one overflow list       (arbitrary order)
//   2 scan     row_ptr = exclusive_scan(count)              (1 kernel up to 131,072 nodes)
//   3 finish   per row: rank of every id among the row's ids = its CSR position, col/typ
//              gathered through it; hub rows by the whole block (csr_finish.inc).  The
//              counters are left zero again.
// RNA graphs have in-degree <= 6, so every row is finished from its eight table
// entries; the overflow path keeps arbitrary interchange shards (hubs, degree in the
// thousands) correct.  Round 1 ran five launches (zero, histogram, scan, scatter, sort); a
// dependent launch costs ~4.5 us of stream time whatever it does, and the fused per-encode
// setup (gine_f16.hip) runs stage 3 together with the tile plans and the input Linear.
#include "gfy_common.h"

namespace gfy {
#include "csr_finish.inc"
#include "csr_count.inc"
namespace {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;  // per thread
constexpr int kScanTile = kScanBlock * kScanItems;
constexpr int kDirectScanTiles = 64;   // up to here every scan block sums its own prefix

constexpr int kCountEdgesPerBlock = kCsrCountEdgesPerBlock;   // gfy_common.h

template <bool kTileSums>
__global__ __launch_bounds__(256) void k_csr_count(
    const ShardTable shards, int32_t* __restrict__ count, int2* __restrict__ table,
    int32_t* __restrict__ overflow, int32_t* __restrict__ overflow_count,
    int32_t* __restrict__ tile_sum, int bins, int total_blocks) {
  extern __shared__ int s_bins[];
  const int block = xcd_range_block((int)gridDim.x);
  if (block >= total_blocks) return;                                // block-uniform
  csr_count_block<kTileSums>(shards, count, table, overflow, overflow_count, tile_sum, bins,
                             block, s_bins);
}

// block-wide exclusive scan of kScanTile ints held kScanItems per thread
__device__ __forceinline__ int block_exclusive_scan(int (&v)[kScanItems],
                                                    int* lds_wave_sums,
                                                    int* block_total) {
  int run = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const int t = v[i];
    v[i] = run;
    run += t;
  }
  // inclusive scan of per-thread totals across the wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(inc, off, 64);
    if (lane >= off) inc += up;
  }
  if (lane == 63) lds_wave_sums[wave] = inc;
  __syncthreads();
  int wave_base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / 64; ++w) {
    const int s = lds_wave_sums[w];
    if (w < wave) wave_base += s;
    total += s;
  }
  const int thread_base = wave_base + inc - run;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) v[i] += thread_base;
  *block_total = total;
  return thread_base;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_partial(
    const int32_t* __restrict__ count, int64_t n, int32_t* __restrict__ sums) {
  __shared__ int wave_sums[kScanBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  int local = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) local += count[base + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kScanBlock / 64; ++w) t += wave_sums[w];
    sums[blockIdx.x] = t;
  }
}

// one workgroup: exclusive scan of the per-tile sums, any length
__global__ __launch_bounds__(kScanBlock) void k_scan_sums(
    int32_t* __restrict__ sums, int64_t tiles) {
  __shared__ int wave_sums[kScanBlock / 64];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < tiles; base += kScanTile) {
    int v[kScanItems];
    const int64_t at = base + threadIdx.x * kScanItems;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) v[i] = (at + i < tiles) ? sums[at + i] : 0;
    int total;
    block_exclusive_scan(v, wave_sums, &total);
    const int carry = carry_s;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
      if (at + i < tiles) sums[at + i] = v[i] + carry;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
}

// sums == nullptr: the block adds up the counters of all tiles before its own itself
// (coalesced re-reads out of L2; <= kDirectScanTiles tiles), which makes the scan ONE
// launch with no hand-off between blocks — each launch costs 3-4 us of stream time.
__global__ __launch_bounds__(kScanBlock) void k_scan_final(
    const int32_t* __restrict__ count, int32_t* __restrict__ row_ptr, int64_t n,
    const int32_t* __restrict__ sums) {
  __shared__ int wave_sums[kScanBlock / 64];
  __shared__ int prefix_s;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  int v[kScanItems];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) v[i] = (base + i < n) ? count[base + i] : 0;
  int offset;
  if (sums) {
    offset = sums[blockIdx.x];
  } else {
    int local = 0;   // tiles before this one are full: no bounds to check
    const int4* before = reinterpret_cast<const int4*>(count);
#pragma unroll 8
    for (int64_t i = threadIdx.x; i < (int64_t)blockIdx.x * (kScanTile / 4); i += kScanBlock) {
      const int4 x = before[i];
      local += x.x + x.y + x.z + x.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < kScanBlock / 64; ++w) t += wave_sums[w];
      prefix_s = t;
    }
    __syncthreads();
    offset = prefix_s;
    __syncthreads();
  }
  int total;
  block_exclusive_scan(v, wave_sums, &total);
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) row_ptr[base + i] = v[i] + offset;
  // row_ptr[n] = total edge count: written by the thread that owns slot n
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) row_ptr[n] = offset + total;
}

// zero the degree counters (hipMemsetAsync costs a fill kernel as well)
__global__ __launch_bounds__(256) void k_zero(int32_t* __restrict__ count, int64_t items) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < items; i += stride) count[i] = 0;
}

// stage 3 on its own (gfy_build_csr); the fused setup of gine_f16.hip has the same body
template <bool kScanned>
__global__ __launch_bounds__(256) void k_csr_finish(
    const CsrScratch w, const ShardTable shards, int32_t* __restrict__ row_ptr,
    int32_t* __restrict__ col, uint8_t* __restrict__ typ, int row_limit) {
  csr_finish_tile<kScanned>(w, shards, row_ptr, blockIdx.x, col, typ, row_limit);
  csr_finish_release(w, gridDim.x);
}

int grid_for(int64_t items, int block, int cap = 2048) {
  int64_t g = (items + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

// The counters come first: csr_clear_bytes() bytes at the start of the workspace are what has
// to be zero when a build starts (and is zero again when it ends).
static int64_t finish_tiles(int64_t n) { return (n + kCsrTileRows - 1) / kCsrTileRows; }
size_t csr_clear_bytes(int64_t n) {
  return align_up((size_t)(n + 1) * 4, 256) + 256 + align_up((size_t)finish_tiles(n) * 4, 256);
}
bool csr_scan_free(int64_t n) { return finish_tiles(n) <= kCsrLocalScanTiles; }

CsrScratch carve_csr(void* base, int64_t n, int64_t e, int32_t** sums, size_t* bytes) {
  const int64_t tiles = (n + kScanTile - 1) / kScanTile + 1;
  size_t off = 0;
  auto take = [&](size_t size) {
    void* p = base ? (char*)base + off : nullptr;
    off += align_up(size, 256);
    return p;
  };
  CsrScratch w;
  w.count = (int32_t*)take((size_t)(n + 1) * 4);
  w.overflow_count = (int32_t*)take(256);
  w.tile_sum = (int32_t*)take((size_t)finish_tiles(n) * 4);
  w.table = (int2*)take((size_t)n * kCsrSlots * sizeof(int2));
  w.overflow = (int32_t*)take((size_t)e * 4);
  w.perm = (int32_t*)take((size_t)e * 4);
  int32_t* scan_sums = (int32_t*)take((size_t)tiles * 4);
  if (sums) *sums = scan_sums;
  if (bytes) *bytes = off;
  return w;
}

size_t csr_workspace_bytes(int64_t n, int64_t e) {
  size_t bytes = 0;
  carve_csr(nullptr, finish_tiles(n) * kCsrTileRows, e, nullptr, &bytes);
  return bytes;
}

int launch_csr_clear(void* ws, int64_t n, hipStream_t s) {
  const int64_t items = (int64_t)(csr_clear_bytes(n) / 4);
  k_zero<<<grid_for(items, 256, 1024), 256, 0, s>>>((int32_t*)ws, items);
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

ShardTable single_shard(const float* x, const int32_t* edge_index, const uint8_t* edge_types,
                        int64_t n, int64_t e, const int32_t* out_rows, void* out) {
  ShardTable t{};
  t.shards = 1;
  t.tile_base[1] = (int)finish_tiles(n);
  t.edge_base[1] = (int)e;
  t.count_block_base[1] = (int)((e + kCountEdgesPerBlock - 1) / kCountEdgesPerBlock);
  t.nodes[0] = (int)n;
  t.edges[0] = (int)e;
  t.x[0] = x;
  t.edge_index[0] = edge_index;
  t.edge_types[0] = edge_types;
  t.out_rows[0] = out_rows;
  t.out[0] = out;
  return t;
}

int64_t largest_shard_nodes(const ShardTable& shards) {
  int64_t most = 1;
  for (int s = 0; s < shards.shards; ++s) most = shards.nodes[s] > most ? shards.nodes[s] : most;
  return most;
}

// stages 1 and 2 (the counters must be zero).  While every shard has at most 131,072 nodes no
// scan launch is needed: the finish stage sums the per-tile edge counts in front of its tile
// (within its shard) itself.
int launch_csr_count_scan(const CsrScratch& w, int32_t* sums, const ShardTable& shards,
                          bool scan_free, int32_t* row_ptr, int64_t n /* rows of row_ptr */,
                          hipStream_t s) {
  const int blocks = shards.count_block_base[shards.shards];
  if (blocks > 0 && scan_free) {
    const int bins = (int)finish_tiles(largest_shard_nodes(shards));
    k_csr_count<true><<<(blocks + 7) & ~7, 256, (size_t)bins * sizeof(int), s>>>(
        shards, w.count, w.table, w.overflow, w.overflow_count, w.tile_sum, bins, blocks);
  } else if (blocks > 0) {
    k_csr_count<false><<<(blocks + 7) & ~7, 256, 0, s>>>(shards, w.count, w.table, w.overflow,
                                                         w.overflow_count, nullptr, 0, blocks);
  }
  if (!scan_free) {
    const int tiles = (int)((n + kScanTile - 1) / kScanTile);
    if (tiles <= kDirectScanTiles) {
      k_scan_final<<<tiles, kScanBlock, 0, s>>>(w.count, row_ptr, n, nullptr);
    } else {
      // (a single-pass chained scan was measured slower: 17 us, profiles/README.md)
      k_scan_partial<<<tiles, kScanBlock, 0, s>>>(w.count, n, sums);
      k_scan_sums<<<1, kScanBlock, 0, s>>>(sums, tiles);
      k_scan_final<<<tiles, kScanBlock, 0, s>>>(w.count, row_ptr, n, sums);
    }
  }
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

int launch_build_csr(const int32_t* edge_index, const uint8_t* edge_types,
                     int64_t n, int64_t e, int32_t* row_ptr, int32_t* col,
                     uint8_t* typ, void* ws, size_t ws_bytes, hipStream_t s) {
  GFY_REQUIRE(n > 0 && n < INT32_MAX - 64 && e >= 0 && e < INT32_MAX, GFY_ERR_INVALID,
              "gfy_build_csr: node/edge counts must fit int32 (n=%lld e=%lld)",
              (long long)n, (long long)e);
  const ShardTable shards = single_shard(nullptr, edge_index, edge_types, n, e, nullptr, nullptr);
  const int64_t rows = shards.total_rows();   // n rounded up to whole 32-row tiles
  // the counting kernel's table keeps an edge's source row in 24 bits (0xFFFFFF = none)
  GFY_REQUIRE(rows < kCsrMaxRows, GFY_ERR_UNSUPPORTED,
              "gfy_build_csr: at most 16,777,215 (padded) nodes per call (got %lld); split the "
              "graph at record boundaries", (long long)rows);
  int32_t* sums = nullptr;
  size_t need = 0;
  const CsrScratch w = carve_csr(ws, rows, e, &sums, &need);
  GFY_REQUIRE(ws_bytes >= need, GFY_ERR_WORKSPACE,
              "gfy_build_csr: workspace %zu < required %zu", ws_bytes, need);
  // this entry point takes any scratch memory, so it clears the counters itself: four
  // launches, three without a scan (gfy_encode_coo on a cleared workspace: two, stage 3 fused
  // with the setup).  The caller's row_ptr has n + 1 entries: rows of the last tile beyond n
  // are not written (csr_finish.inc).
  if (const int rc = launch_csr_clear(ws, rows, s)) return rc;
  const bool scan_free = csr_scan_free(n);
  if (const int rc = launch_csr_count_scan(w, sums, shards, scan_free, row_ptr, n, s)) return rc;
  const int tiles = shards.total_tiles();
  if (scan_free)
    k_csr_finish<false><<<tiles, 256, 0, s>>>(w, shards, row_ptr, col, typ, (int)n);
  else
    k_csr_finish<true><<<tiles, 256, 0, s>>>(w, shards, row_ptr, col, typ, (int)n);
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

}  // namespace gfy
