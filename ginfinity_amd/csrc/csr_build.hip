// COO -> destination-major CSR, edges of one destination in COO order.
//
// Replaces the addressing half of the reference's message passing
// (src/ginfinity/_model.py:41-45: index_select by source, index_add_ by
// destination, on int64 copies made at api.py:239-242).  Integer work: the
// result is bit-exact against a stable argsort by destination
// (oracle/gine_numpy.py:build_csr) and identical from run to run.
//
// Pipeline (all on one stream, no host sync):
//   1 histogram   slot[e] = atomicAdd(count[dst[e]], 1)      (arbitrary rank)
//   2 scan        row_ptr = exclusive_scan(count)            (1 kernel up to 131,072 nodes)
//   3 scatter     perm[row_ptr[dst[e]] + slot[e]] = e
//   4 sort rows   each row's edge ids ascending -> COO order restored, then
//                 col/typ gathered through perm.  Rows longer than kSmallRow are
//                 handed to the whole workgroup (rank sort) in the same launch.
// RNA graphs have in-degree <= 5, so step 4 is a 5-element
// insertion sort per thread; the worklist path keeps arbitrary interchange
// shards (hubs, degree in the thousands) correct.
#include "gfy_common.h"

namespace gfy {
namespace {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;  // per thread
constexpr int kScanTile = kScanBlock * kScanItems;
constexpr int kSmallRow = 32;
constexpr int kDirectScanTiles = 64;   // up to here every scan block sums its own prefix

__global__ __launch_bounds__(256) void k_histogram(
    const int32_t* __restrict__ dst, int64_t e_count, int64_t n,
    int32_t* __restrict__ count, int32_t* __restrict__ slot) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < e_count; e += stride) {
    const int32_t d = dst[e];
    slot[e] = ((uint32_t)d < (uint64_t)n) ? atomicAdd(&count[d], 1) : -1;
  }
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

__global__ __launch_bounds__(256) void k_scatter(
    const int32_t* __restrict__ dst, const int32_t* __restrict__ slot,
    const int32_t* __restrict__ row_ptr, int64_t e_count,
    int32_t* __restrict__ perm) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < e_count; e += stride) {
    const int32_t s = slot[e];
    if (s >= 0) perm[row_ptr[dst[e]] + s] = (int32_t)e;
  }
}

__global__ __launch_bounds__(256) void k_sort_rows(
    const int32_t* __restrict__ row_ptr, int64_t n, int32_t* __restrict__ perm,
    const int32_t* __restrict__ src, const uint8_t* __restrict__ types,
    int32_t* __restrict__ col, uint8_t* __restrict__ typ) {
  __shared__ int big_rows[256];
  __shared__ int big_count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t first = (int64_t)blockIdx.x * blockDim.x; first < n; first += stride) {
    if (threadIdx.x == 0) big_count = 0;
    __syncthreads();
    const int64_t row = first + threadIdx.x;
    if (row < n) {
      const int32_t lo = row_ptr[row], hi = row_ptr[row + 1];
      if (hi - lo > kSmallRow) {
        big_rows[atomicAdd(&big_count, 1)] = (int32_t)row;
      } else {
        // insertion sort of the edge ids (tiny, L1/L2 resident)
        for (int32_t i = lo + 1; i < hi; ++i) {
          const int32_t key = perm[i];
          int32_t j = i - 1;
          while (j >= lo) {
            const int32_t p = perm[j];
            if (p <= key) break;
            perm[j + 1] = p;
            --j;
          }
          perm[j + 1] = key;
        }
        for (int32_t i = lo; i < hi; ++i) {
          const int32_t edge = perm[i];
          col[i] = src[edge];
          typ[i] = types[edge];
        }
      }
    }
    __syncthreads();
    // rows above kSmallRow: the whole workgroup, rank sort (edge ids are distinct)
    const int rows = big_count;
    for (int w = 0; w < rows; ++w) {
      const int32_t big = big_rows[w];
      const int32_t lo = row_ptr[big], deg = row_ptr[big + 1] - lo;
      for (int32_t i = threadIdx.x; i < deg; i += blockDim.x) {
        const int32_t key = perm[lo + i];
        int32_t rank = 0;
        for (int32_t j = 0; j < deg; ++j) rank += perm[lo + j] < key;
        col[lo + rank] = src[key];
        typ[lo + rank] = types[key];
      }
    }
    __syncthreads();
  }
}

struct CsrWorkspace {
  int32_t* count;     // [N + 1] in-degree counters (16-byte aligned, read as int4)
  int32_t* slot;      // [E]
  int32_t* perm;      // [E]
  int32_t* sums;      // [tiles]
  size_t bytes;
};

CsrWorkspace carve(void* base, int64_t n, int64_t e) {
  const int64_t tiles = (n + kScanTile - 1) / kScanTile + 1;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void* p = base ? (char*)base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  CsrWorkspace w;
  w.count = (int32_t*)take((size_t)(n + 1) * 4);
  w.slot = (int32_t*)take((size_t)e * 4);
  w.perm = (int32_t*)take((size_t)e * 4);
  w.sums = (int32_t*)take((size_t)tiles * 4);
  w.bytes = off;
  return w;
}

int grid_for(int64_t items, int block, int cap = 2048) {
  int64_t g = (items + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

size_t csr_workspace_bytes(int64_t n, int64_t e) { return carve(nullptr, n, e).bytes; }

int launch_build_csr(const int32_t* edge_index, const uint8_t* edge_types,
                     int64_t n, int64_t e, int32_t* row_ptr, int32_t* col,
                     uint8_t* typ, void* ws, size_t ws_bytes, hipStream_t s) {
  GFY_REQUIRE(n > 0 && n < INT32_MAX && e >= 0 && e < INT32_MAX, GFY_ERR_INVALID,
              "gfy_build_csr: node/edge counts must fit int32 (n=%lld e=%lld)",
              (long long)n, (long long)e);
  const CsrWorkspace w = carve(ws, n, e);
  GFY_REQUIRE(ws_bytes >= w.bytes, GFY_ERR_WORKSPACE,
              "gfy_build_csr: workspace %zu < required %zu", ws_bytes, w.bytes);
  const int32_t* src = edge_index;
  const int32_t* dst = edge_index + e;
  const int tiles = (int)((n + kScanTile - 1) / kScanTile);

  // five launches: zero, histogram, scan, scatter, sort
  k_zero<<<grid_for(n + 1, 256, 1024), 256, 0, s>>>(w.count, n + 1);
  if (e > 0)
    k_histogram<<<grid_for(e, 256), 256, 0, s>>>(dst, e, n, w.count, w.slot);
  if (tiles <= kDirectScanTiles) {
    k_scan_final<<<tiles, kScanBlock, 0, s>>>(w.count, row_ptr, n, nullptr);
  } else {
    // (a single-pass chained scan was measured slower: 17 us, profiles/README.md)
    k_scan_partial<<<tiles, kScanBlock, 0, s>>>(w.count, n, w.sums);
    k_scan_sums<<<1, kScanBlock, 0, s>>>(w.sums, tiles);
    k_scan_final<<<tiles, kScanBlock, 0, s>>>(w.count, row_ptr, n, w.sums);
  }
  if (e > 0) {
    k_scatter<<<grid_for(e, 256), 256, 0, s>>>(dst, w.slot, row_ptr, e, w.perm);
    k_sort_rows<<<grid_for(n, 256), 256, 0, s>>>(row_ptr, n, w.perm, src, edge_types, col, typ);
  }
  GFY_CHECK_HIP(hipGetLastError());
  return GFY_OK;
}

}  // namespace gfy
