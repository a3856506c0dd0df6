// Shared host/device helpers of libgfy (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <utility>

#include "../../include/gfy.h"

namespace gfy {

// ---- per-thread error channel -------------------------------------------------
void set_error(const char* fmt, ...);
void clear_error();

#define GFY_CHECK_HIP(expr)                                                    \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      ::gfy::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                    \
      return GFY_ERR_HIP;                                                      \
    }                                                                          \
  } while (0)

#define GFY_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      ::gfy::set_error(__VA_ARGS__);  \
      return (code);                  \
    }                                 \
  } while (0)

// ---- model geometry compiled into the kernels -----------------------------------
constexpr int kHidden = 128;   // data/model.json:12
constexpr int kMlp = 256;      // 2 * hidden       (_model.py:34)
constexpr int kInDim = 7;      // node features    (graph.py:91-93)
constexpr int kOutDim = 128;
constexpr int kMaxEdgeTypes = 16;
constexpr int kMaxLayers = 8;

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Runs `configure` once per HIP device (the calling thread's current device), thread-safe:
// hipFuncSetAttribute is per device, and one process may drive several GPUs from several
// threads (Ginfinity.load accepts "cuda:<i>").
class PerDeviceOnce {
 public:
  template <typename F>
  int run(F&& configure) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 256) return GFY_ERR_HIP;
    std::lock_guard<std::mutex> guard(mutex_);
    if ((done_[device >> 6] >> (device & 63)) & 1u) return GFY_OK;
    const int rc = configure();
    if (rc == GFY_OK) done_[device >> 6] |= 1ull << (device & 63);
    return rc;
  }

 private:
  std::mutex mutex_;
  unsigned long long done_[4] = {};
};

// ---- fp16 layer parameters on the device (layouts: gine_layer.inc, gfy_api.hip) ---------
// Hidden rows are stored with the channel groups 4-7 and 8-11 of every 16 swapped: 16-byte
// chunk 2 s + hq of a row is then the eight values lane half hq holds for k-step s in the
// MFMA C/D layout, so activations move between memory, LDS and MFMA operands as whole chunks.
inline int stored_channel(int position) {   // an involution: bits 2 and 3 trade places
  return (position & ~12) | ((position & 4) << 1) | ((position & 8) >> 1);
}
inline int gemm_result_channel(int block, int half, int reg) {   // 32x32 C/D layout
  return 32 * block + (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

struct LayerF16 {
  float scale;            // fp16 value R(1 + R(eps)) widened to fp32
  int edge_types;         // edge_dim: rows of the edge table; an edge of another type is ignored
  const f16* w01_image;   // 128 KB: mlp.0.weight | mlp.4.weight fragments (pack_k_chained)
  const void* image;      // 7,680-byte LDS image: edge table R(W_edge[:,t] + b_edge) in stored
                          // order + the -inf row of idle slots; BatchNorm alpha = invstd*gamma,
                          // shift = beta - mean*alpha (fp32, from fp16-rounded buffers) and
                          // LayerNorm gamma, beta as [block][lane half][register] of an MFMA
                          // result; b0, b1 in natural channel order (added on the matrix cores)
};

struct HeadF16 {
  const f16* wa_frag;  // head.0.weight fragments, pack_k_chained (the stand-alone head kernel)
  const f16* ba;       // [128]
  const f16* wb_frag;  // head.2.weight fragments, pack_b_fragments
  const f16* bb;       // [128]
  const f16* w_image;  // 64 KB: head.0 (pack_k_chained) | head.2 (pack_chain_fragments)
  const void* image;   // 512 B: ba as [block][half][register] | bb as natural 16-byte chunks
};

struct ModelF16 {
  const f16* w_in;  // [8][16][8]: [p & 7][p >> 3][k] for stored position p, k padded 7 -> 8
  const f16* b_in;  // [128] in stored order
  LayerF16 layer[kMaxLayers];
  HeadF16 head;
};

// ---- fp32 (full_precision) parameters, derived on the host in fp32 ----------------
struct LayerF32 {
  const float* table;   // [kMaxEdgeTypes][128]  fl32(W_edge[:,t] + b_edge)
  float one_plus_eps;   // fl32(1 + eps)
  const float* w0t;     // [128][256]  mlp.0.weight transposed ([K][N])
  const float* b0;      // [256]
  const float* alpha;   // [256]  invstd * gamma
  const float* shift;   // [256]  beta - mean * alpha
  const float* w1t;     // [256][128]  mlp.4.weight transposed
  const float* b1;      // [128]
  const float *ln_g, *ln_b;
};

struct ModelF32 {
  const float* w_in_t;  // [7][128]
  const float* b_in;
  LayerF32 layer[kMaxLayers];
  const float *ha_wt, *ha_b, *hb_wt, *hb_b;  // head, weights transposed
};

}  // namespace gfy

struct gfy_encoder {
  int device = 0;
  int model_dtype = GFY_F16;
  int layers = 0;
  int edge_dim = 0;
  int residual = 1;
  void* device_blob = nullptr;  // one allocation holding every derived tensor
  size_t device_blob_bytes = 0;
  gfy::ModelF16 f16{};
  gfy::ModelF32 f32{};
  // optional per-kernel timing (gfy_encoder_set_timing)
  // slot 0 before the setup launch, 1 after it, 1 + l after layer launch l, layers + 2
  // after the stand-alone head.  Mode 2 leaves out the events between layer launches
  // 1 .. layers-1: an event between two dependent kernels costs ~2.5 us of stream time that
  // rocprof's kernel durations do not contain.
  int separate_head = 0;      // GFY_OPT_SEPARATE_HEAD
  int layer_kernel = -1;      // GFY_OPT_LAYER_KERNEL: -1 by the launch's rounds, 1 round-2 kernel, 3 persistent rounds, 4 windowed
  int cus = 256;              // compute units of the device (persistent grid)
  int stagger = -1;           // GFY_OPT_STAGGER: start offset between the workgroups of an XCD in
                              // shader cycles (persistent rounds); -1: 500 from three rounds up
  int priority = -1;          // GFY_OPT_PRIORITY: windowed kernel, s_setprio levels (gine_layer_w.inc); -1: 4
  int timing = 0;
  // timing == 3: every layer launch records the device clock (s_memrealtime, 100 MHz) of its
  // first workgroup start and last workgroup end: the kernel's own duration, as a profiler
  // sees it, also when other streams keep the chip busy between two events of this one
  unsigned long long* device_spans = nullptr;   // [kMaxLayers][2] on the device
  mutable hipStream_t last_stream = nullptr;    // stream of the last encode (get_timing waits on it)
  mutable int last_layer_kernel = 0;            // layer kernel the last fp16 encode launched (1 / 3 / 4)
  hipEvent_t events[gfy::kMaxLayers + 3] = {};
  mutable int events_recorded = 0;
  void mark(hipStream_t s, int slot) const {
    if (!timing || timing == 3) return;
    if (timing == 2 && slot >= 2 && slot < layers) return;
    (void)hipEventRecord(events[slot], s);
    events_recorded = slot + 1;
  }
};

// ---- kernel launchers (one per .hip file) -------------------------------------------
namespace gfy {

// ---- a batch of shards in one sequence of launches ------------------------------------------
// Shards never share edges (graph.py:392-395), so k of them are ONE graph for the kernels: shard
// s owns the global rows [32 tile_base[s], 32 tile_base[s] + nodes[s]) — every shard starts on a
// tile boundary, the rows between its last node and the next boundary are padding (isolated,
// zero features, never stored to an output) — and the global edge ids [edge_base[s],
// edge_base[s + 1]).  Hidden states, CSR and tile plans live in the workspace in global
// numbering; only the kernels that touch the callers' arrays (edge list in, node features in,
// embeddings out) look a tile's or an edge's shard up in this table, which travels as a kernel
// argument.
constexpr int kMaxBatchShards = 16;
struct ShardTable {
  int shards;
  int tile_base[kMaxBatchShards + 1];         // [shards] = tiles of the batch
  int edge_base[kMaxBatchShards + 1];         // [shards] = edges of the batch
  int count_block_base[kMaxBatchShards + 1];  // blocks of the counting kernel, per shard
  int nodes[kMaxBatchShards];
  int edges[kMaxBatchShards];
  const float* x[kMaxBatchShards];            // [nodes][in_dim]
  const int32_t* edge_index[kMaxBatchShards]; // [2][edges], shard-local node ids
  const uint8_t* edge_types[kMaxBatchShards];
  const int32_t* out_rows[kMaxBatchShards];   // or nullptr
  void* out[kMaxBatchShards];
  __host__ __device__ int total_tiles() const { return tile_base[shards]; }
  __host__ __device__ int64_t total_rows() const { return (int64_t)tile_base[shards] * 32; }
  __host__ __device__ int total_edges() const { return edge_base[shards]; }
  __device__ int shard_of_tile(int tile) const {
    int s = 0;
    for (int k = 1; k < shards; ++k) s += tile >= tile_base[k] ? 1 : 0;
    return s;
  }
  __device__ int shard_of_edge(int edge) const {
    int s = 0;
    for (int k = 1; k < shards; ++k) s += edge >= edge_base[k] ? 1 : 0;
    return s;
  }
  __device__ int shard_of_count_block(int block) const {
    int s = 0;
    for (int k = 1; k < shards; ++k) s += block >= count_block_base[k] ? 1 : 0;
    return s;
  }
};

// Optional record boundaries of a batch's shards (gfy_shard.node_ptr / edge_ptr: records never
// share edges, graph.py:392-395, and a record's edges are one contiguous part of the edge
// list).  With them the COO -> plans stage needs no global atomics (csr_records.inc): a
// workgroup owns kRecRows consecutive rows of one shard and scans only the edges of the records
// that overlap them.  `range_base`: workgroups of that stage, per shard; `range_rows`: rows per
// workgroup, kRecRowsSmall for a batch of up to kRecSmallBatchRows rows (three 60,000-node
// micro-batches and fewer; a lone 60,000-node
// micro-batch: 235 workgroups, every one resident at once, and a workgroup's latency chain is the
// launch — 512 / 768 rows cost it 4 / 10 us), kRecRowsLarge above (four shards: the scan of a
// workgroup covers its records once for three times the rows, and fewer workgroups stand in the
// way of the input Linear's: 41.7 -> 37.2 us; 512 rows 37.9, 1,024 rows 41.5).
// A batch of up to kRecLoneBatchRows rows (a lone 60,000-node micro-batch) takes 512 rows per
// workgroup of 512 THREADS: the launch is one workgroup's latency chain there, and eight waves
// halve its scan and its turns (set-up 19.2-20.0 -> 18.0-18.7 us, a call 97.4 -> 95.9 us; 8,000
// nodes 81.6 -> 80.4; 120,000 nodes +1 us, 180,000 +3.3 us, 240,000 rows with 512 / 384 threads
// +6-7 us beside the Linear's workgroups, which get the same size: profiles/README.md).
constexpr int kRecRowsLone = 512, kRecRowsSmall = 256, kRecRowsLarge = 768;
constexpr int kRecThreadsLone = 512, kRecThreadsSmall = 256, kRecThreadsLarge = 256;
constexpr int64_t kRecLoneBatchRows = 90000;
constexpr int64_t kRecSmallBatchRows = 200000;   // 180,000 rows: 220.3 (256) against 222.0 us (768) per call; 210,000: 253 / 252; 240,000: 278 / 276
struct RecordTable {
  const int64_t* node_ptr[kMaxBatchShards];   // [records + 1], shard-local, ascending
  const int64_t* edge_ptr[kMaxBatchShards];   // [records + 1]
  int records[kMaxBatchShards];
  int range_base[kMaxBatchShards + 1];
  int range_rows;
  __device__ int shard_of_range(int block, int shards) const {
    int s = 0;
    for (int k = 1; k < shards; ++k) s += block >= range_base[k] ? 1 : 0;
    return s;
  }
};

// COO -> CSR scratch (csr_build.hip, csr_finish.inc), all in the caller's workspace
constexpr int kCsrSlots = 8;        // edge ids kept per row in the table (= plan slots)
constexpr int kCsrTileRows = 32;    // rows finished by one 256-thread block (= layer tile)
#ifndef GFY_COUNT_EDGES_PER_BLOCK
#define GFY_COUNT_EDGES_PER_BLOCK 256
#endif
// edges per block of k_csr_count, 256 = one per thread (E = 300,000: 1,024 per block 8.9 us,
// 256 per block 7.5 us); ShardTable::count_block_base is in these units
constexpr int kCsrCountEdgesPerBlock = GFY_COUNT_EDGES_PER_BLOCK;
constexpr int kCsrLocalScanTiles = 4096;   // up to here the finish stage derives row_ptr itself
constexpr int kCsrStageFarRows = 40;   // out-of-tile rows a gather stage holds (= kLFar, gine_layer.inc)
constexpr uint32_t kCsrNoSource = 0xFFFFFFu;   // table entry: the edge's source is outside its shard
constexpr int64_t kCsrMaxRows = 0xFFFFFF;      // ... so a COO call addresses rows 0 .. 0xFFFFFE (strictly fewer than 2^24)
struct CsrScratch {
  int32_t* count;              // [n + 1]   in-degree counters          (zero between calls)
  int32_t* overflow_count;     // [2]       list length, finish ticket  (zero between calls)
  int32_t* tile_sum;           // [tiles]   edges per 32 rows           (zero between calls)
  int2* table;                 // [n][kCsrSlots] first arrivals of every row: {edge id, global
                               // source row | type << 24} (source 0xFFFFFF: outside its shard)
  int32_t* overflow;           // [e]       edge ids that found their row's slots taken
  int32_t* perm;               // [e]       hub rows: ids collected for the rank sort
};
size_t csr_clear_bytes(int64_t n);   // leading bytes of the workspace that must be zero
bool csr_scan_free(int64_t n);       // the finish stage derives row_ptr itself (n: largest shard)
CsrScratch carve_csr(void* base, int64_t n, int64_t e, int32_t** scan_sums, size_t* bytes);
int launch_csr_clear(void* ws, int64_t n, hipStream_t s);
int launch_csr_count_scan(const CsrScratch& w, int32_t* scan_sums, const ShardTable& shards,
                          bool scan_free, int32_t* row_ptr, int64_t rows, hipStream_t s);
ShardTable single_shard(const float* x, const int32_t* edge_index, const uint8_t* edge_types,
                        int64_t n, int64_t e, const int32_t* out_rows, void* out);
int64_t largest_shard_nodes(const ShardTable& shards);

int launch_build_csr(const int32_t* edge_index, const uint8_t* edge_types,
                     int64_t n, int64_t e, int32_t* row_ptr, int32_t* col,
                     uint8_t* typ, void* ws, size_t ws_bytes, hipStream_t s);
size_t csr_workspace_bytes(int64_t n, int64_t e);

int launch_build_graphs(const uint8_t* bases, const uint8_t* marks, const int64_t* node_ptr,
                        const int64_t* edge_ptr, int64_t records, int64_t n, int64_t e,
                        int struct_states, int positional_cols, int skip2,
                        const float* positional, float* features, int32_t* edge_index,
                        uint8_t* edge_types, int32_t* first_invalid, hipStream_t s);

// > 64 KB of dynamic LDS is an opt-in per kernel and per DEVICE: done once when an encoder is
// created on a device, not per encode call
int prepare_device_f16();
int launch_encode_f16(const gfy_encoder* enc, const float* x,
                      const int32_t* row_ptr, const int32_t* col,
                      const uint8_t* typ, int64_t n, int64_t e,
                      const int32_t* out_rows, void* out, int out_dtype,
                      int normalise, int tap_stage, void* ws, size_t ws_bytes,
                      hipStream_t s);
size_t encode_f16_workspace_bytes(int64_t n, int64_t e);
// COO in -> embeddings out: CSR build and encode as one sequence of launches, the last CSR
// stage fused with the per-encode setup; the workspace's first csr_clear_bytes(n) bytes must
// be zero (they are zero again afterwards)
// `records`: nullptr, or the record boundaries of EVERY shard of the batch
int launch_encode_coo_f16(const gfy_encoder* enc, const ShardTable& shards,
                          const RecordTable* records, int out_dtype, int normalise, void* ws,
                          size_t ws_bytes, hipStream_t s);
size_t encode_coo_f16_workspace_bytes(int64_t padded_rows, int64_t e);

// one GINE layer on a given hidden state, one of its phase tensors out (parity tests)
size_t debug_layer_f16_workspace_bytes(int64_t n);
int launch_debug_layer_f16(const gfy_encoder* enc, int layer, const void* hidden_in,
                           const int32_t* row_ptr, const int32_t* col, const uint8_t* typ,
                           int64_t n, int64_t e, int tap, void* out, void* ws, size_t ws_bytes,
                           hipStream_t s);

int launch_encode_f32(const gfy_encoder* enc, const float* x,
                      const int32_t* row_ptr, const int32_t* col,
                      const uint8_t* typ, int64_t n, int64_t e,
                      const int32_t* out_rows, void* out, int out_dtype,
                      int normalise, int tap_stage, void* ws, size_t ws_bytes,
                      hipStream_t s);
size_t encode_f32_workspace_bytes(int64_t n, int64_t e);

int launch_pairwise_dense(const void* a, int64_t n, const void* b, int64_t m,
                          int metric, float* out, void* ws, size_t ws_bytes,
                          hipStream_t s);
int launch_pairwise_nearest(const void* a, int64_t n, const void* b, int64_t m,
                            int metric, int64_t exclude_offset, int exclude_on, float* best_val,
                            int32_t* best_idx, void* ws, size_t ws_bytes,
                            hipStream_t s);
size_t pairwise_workspace_bytes(int64_t n, int64_t m);

// ---- LDS-DMA (global_load_lds_dwordx4) ---------------------------------------------
// One wave instruction: lane L copies 16 bytes from its own global address to LDS
// address lds + 16 L (lds is wave-uniform, in M0); inactive lanes move nothing.  No
// VGPRs, and not visible to the compiler's waitcnt bookkeeping: the consumer waits with
// dma_wait_all() / dma_wait_but().
__device__ __forceinline__ void dma16(const void* gbase /* uniform */, uint32_t goff,
                                      uint32_t lds) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2"
      :
      : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(goff), "s"(gbase)
      : "memory");
}
__device__ __forceinline__ void dma16_at(const void* lane_ptr, uint32_t lds) {   // 64-bit form
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off"
      :
      : "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(lane_ptr)
      : "memory");
}
// The builtin (not an asm string) so that hipcc's own waitcnt bookkeeping learns that
// nothing is outstanding: otherwise it guards the first use of every earlier-loaded
// register with a vmcnt wait that would drain a look-ahead early.
__device__ __forceinline__ void dma_wait_all() {
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0); expcnt, lgkmcnt untouched
  asm volatile("" ::: "memory");
}

// W[n_out][k_in] (row-major fp16) -> fragments of the MFMA 32x32x16 A operand (out^T = W . in^T):
// frag[(block * ksteps + s) * 64 + lane][j] = W[row(block, lane & 31)][k(s, lane >> 5, j)]
//   pack_b_fragments      natural:  row = 32 block + m,  k = 16 s + 8 half + j
//   pack_k_chained        the B operand is an MFMA result used in place (gine_layer.inc):
//                         k = 16 s + 8 (j >> 2) + 4 half + (j & 3), rows natural
//   pack_chain_fragments  same k; rows dealt so that the result's registers are natural
//                         16-byte chunks of the output row (head.2 -> embedding)
void pack_b_fragments(const f16* w, int n_out, int k_in, f16* frag);
void pack_k_chained(const f16* w, int n_out, int k_in, f16* frag);
void pack_chain_fragments(const f16* w, int n_out, int k_in, f16* frag);
inline int hidden_layout_channel(int half, int ks, int j) { return 16 * ks + 8 * half + j; }

}  // namespace gfy
