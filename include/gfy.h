/* gfy.h — C ABI of the MI355X (gfx950) GINE encode / embedding-distance library.
 *
 * This is the drop-in boundary for the hot path of nicoaira/GINFINITY.  The
 * reference has no FFI: its boundary is the Python seam
 *     Ginfinity._run_graph_shard(shard, embedding_dtype)      src/ginfinity/api.py:232-260
 * called from encode_graphs (api.py:227-228).  Every entry point below names
 * the reference lines it replaces.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; all array arguments are DEVICE pointers unless the
 *     name ends in _host; the caller owns every buffer it passes in;
 *   - every function that launches work takes a hipStream_t as `void* stream`
 *     and only ENQUEUES: no hidden synchronisation, no allocation (graph-
 *     capturable); scratch memory is caller-provided (`workspace`), sized by
 *     the matching *_workspace_bytes query;
 *   - return value: 0 = GFY_OK, otherwise an error code; the message is kept
 *     per host thread and read with gfy_last_error().  The library never
 *     aborts.  (Python shim maps codes to ValueError / RuntimeError —
 *     reference error conventions: api.py:70-76,197-210.)
 *   - a gfy_encoder handle is not thread-safe ("a loaded instance is safe for
 *     serialized inference", docs/OPERATIONS.md:43-47).
 */
#ifndef GFY_H_
#define GFY_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFY_ABI_VERSION 4

enum gfy_status {
  GFY_OK = 0,
  GFY_ERR_INVALID = 1,     /* bad argument / malformed weight pack            */
  GFY_ERR_UNSUPPORTED = 2, /* architecture not compiled in (hidden != 128 …) */
  GFY_ERR_HIP = 3,         /* a HIP runtime call failed                       */
  GFY_ERR_WORKSPACE = 4    /* workspace too small                             */
};

enum gfy_dtype { GFY_F16 = 0, GFY_F32 = 1, GFY_F64 = 2 };

enum gfy_metric { GFY_L2 = 0, GFY_COSINE = 1 };

typedef struct gfy_encoder gfy_encoder;

/* Last error message of the calling host thread ("" if none). */
const char* gfy_last_error(void);
int gfy_abi_version(void);

/* ---- weight pack -----------------------------------------------------------
 * One little-endian blob: a 32-byte header followed by float32 tensors in
 * checkpoint order, row-major as torch stores them ([out_features][in_features]).
 *
 *   uint32 magic 'GFY1' (0x31594647), uint32 version (1), uint32 in_dim (7),
 *   uint32 hidden (128), uint32 layers (4), uint32 edge_dim (10),
 *   uint32 out_dim (128), uint32 flags (bit0 = residual)
 *   input.weight[hidden][in_dim]  input.bias[hidden]
 *   for l in 0..layers-1:
 *     convs.l.eps[1]
 *     convs.l.edge_lin.weight[hidden][edge_dim]  convs.l.edge_lin.bias[hidden]
 *     convs.l.mlp.0.weight[2h][h]  convs.l.mlp.0.bias[2h]
 *     convs.l.mlp.1.weight[2h] .bias[2h] .running_mean[2h] .running_var[2h]
 *     convs.l.mlp.4.weight[h][2h]  convs.l.mlp.4.bias[h]
 *     norms.l.weight[h]  norms.l.bias[h]
 *   head.0.weight[h][h] head.0.bias[h] head.2.weight[out][h] head.2.bias[out]
 * (reference: src/ginfinity/_model.py:29-63). */
size_t gfy_weight_pack_bytes(uint32_t in_dim, uint32_t hidden, uint32_t layers,
                             uint32_t edge_dim, uint32_t out_dim);

/* Build an encoder on HIP device `device` from a HOST weight pack.
 * model_dtype = GFY_F16 reproduces `model.half()` (api.py:111-112: parameters
 * and BatchNorm buffers rounded to fp16, fp16 activations with fp32 internals);
 * GFY_F32 is `full_precision=True`.  Replaces api.py:101-112 (module build,
 * .to(device), .half()).  Synchronous (uploads weights). */
int gfy_encoder_create(const void* weight_pack_host, size_t bytes,
                       int model_dtype, int device, gfy_encoder** out);
void gfy_encoder_destroy(gfy_encoder* encoder);

/* ---- records -> graph arrays (MI355X-side extension of the path's caller) ---------
 * GraphBuilder._build_full for every unsliced record of a micro-batch, concatenated as
 * GraphShard.from_graphs does (src/ginfinity/graph.py:494-561 — node features 496-514,
 * typed edges 516-546, pair table 737-747; concatenation 346-412).  Integer / one-hot
 * work, bit-identical to the reference's arrays including the edge order.
 *   bases, marks   uint8 [N]     the records' sequence / dot-bracket text, concatenated
 *                                (A C G U and ( . ) only — RNA.__post_init__ guarantees it)
 *   node_ptr       int64 [R+1]   record r owns nodes node_ptr[r]-node_ptr[0] ..
 *   edge_ptr       int64 [R+1]   and edges edge_ptr[r]-edge_ptr[0] ..; the caller computes
 *                                it: 2(L-1) + 2 pairs + (skip2 ? 2 max(L-2,0) : 0) per record
 *   struct_states  1: one "paired" flag (struct_feature "A"); 3: one-hot ( . )  ("B")
 *   positional     float32 [N][positional_columns] or NULL: the reference's host numpy
 *                  float32 sin/cos columns, copied into the feature rows unchanged
 *   node_features  float32 [N][4 + struct_states + positional_columns]   out
 *   edge_index     int32 [2][E], edge_types uint8 [E]                     out
 *   first_invalid  int32 [1] out: -1, or the first record whose text is not a balanced
 *                  structure over the alphabet, disagrees with edge_ptr, or nests deeper
 *                  than 2,048 levels (the reference caps records at 4,096 nt,
 *                  _validation.py MAXIMUM_LENGTH_NT); its rows are then unspecified,
 *                  nothing is written out of bounds.  No workspace.                 */
int gfy_build_graphs(const uint8_t* bases, const uint8_t* marks,
                     const int64_t* node_ptr, const int64_t* edge_ptr,
                     int64_t n_records, int64_t n_nodes, int64_t n_edges,
                     int struct_states, int positional_columns, int skip2,
                     const float* positional, float* node_features,
                     int32_t* edge_index, uint8_t* edge_types,
                     int32_t* first_invalid, void* stream);

/* ---- COO -> CSR ------------------------------------------------------------
 * Destination-major CSR of a shard's edges, edges of one destination kept in
 * their COO order (a stable counting sort; integer work, bit-exact, run-to-run
 * deterministic).  Replaces the int64 widening + index_select/index_add_
 * addressing of api.py:239-242 and _model.py:41-45.
 *   edge_index  int32 [2][E]   row 0 = source, row 1 = destination (graph.py:306-308)
 *   edge_types  uint8 [E]
 *   row_ptr     int32 [N+1]    out
 *   col         int32 [E]      out: source node of each in-edge
 *   typ         uint8 [E]      out: edge type of each in-edge                */
/* Limit of the COO entry points (gfy_build_csr, gfy_encode_coo, gfy_encode_coo_batch): the node
 * count of a call, rounded up to whole 32-row tiles (a batch: the sum over its shards), must be
 * below 16,777,215 — GFY_ERR_UNSUPPORTED otherwise, before anything is launched.  (The counting
 * kernel keeps an edge's source row in 24 bits.)  gfy_encode with a caller-built CSR takes up to
 * 16,777,216 nodes.  The reference's micro-batches hold 60,000 (api.py:147-148). */
size_t gfy_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int gfy_build_csr(const int32_t* edge_index, const uint8_t* edge_types,
                  int64_t n_nodes, int64_t n_edges, int32_t* row_ptr,
                  int32_t* col, uint8_t* typ, void* workspace,
                  size_t workspace_bytes, void* stream);

/* ---- encode ------------------------------------------------------------------
 * GINEEncoder.forward (+ optional float64 L2 normalise) for one micro-batch:
 * input Linear, `layers` x (GINE message/aggregate/update, BatchNorm MLP,
 * LayerNorm, residual), 2-layer head.  Replaces api.py:237-252 and
 * _model.py:39-46,65-72.
 *   node_features float32 [N][in_dim]          (graph.py:302-305)
 *   (edge values are the caller's to validate, as GraphShard does: graph.py:318-323.  The device
 *   entry points never read outside the arrays for any VALUE in them: an edge whose source is
 *   outside [0, N) or whose type is >= edge_dim is IGNORED — it contributes no message —; the
 *   host entry point gfy_host_encode returns GFY_ERR_INVALID for such an edge)
 *   row_ptr/col/typ                            from gfy_build_csr
 *   out_rows      int32 [N] or NULL: output row of node i, -1 = drop the node
 *                 (context nodes of sliced graphs, api.py:253-260); NULL = i
 *   out           [n_out_rows][out_dim] of out_dtype (f16 / f32 / f64)
 *   normalise     1: out = o / max(||o||_2, 1e-12) computed in float64 and
 *                 rounded once to out_dtype (api.py:250-252,258-259);
 *                 0: raw head output o
 * Output dtypes of the fp16 model are separate code paths: out_dtype f16 runs head + normalise
 * inside the last layer launch, f32 / f64 run the stand-alone head kernel, and the two sum the
 * head's 128-deep dot products in different k orders (fp32 accumulation either way).  Each is
 * deterministic and within 1e-3 of the reference, but an f32 result rounded to fp16 is NOT
 * guaranteed to be the f16 call's bytes: fewer than 2e-3 of the elements differ, by at most
 * 2.5e-4 (tests/test_gpu_parity.py::test_fused_head_equals_standalone_head).                */
size_t gfy_encode_workspace_bytes(const gfy_encoder* encoder, int64_t n_nodes,
                                  int64_t n_edges);
int gfy_encode(gfy_encoder* encoder, const float* node_features,
               const int32_t* row_ptr, const int32_t* col, const uint8_t* typ,
               int64_t n_nodes, int64_t n_edges, const int32_t* out_rows,
               void* out, int out_dtype, int normalise, void* workspace,
               size_t workspace_bytes, void* stream);

/* The whole seam in one call: COO in, embeddings out — Ginfinity._run_graph_shard
 * (api.py:236-252) for one micro-batch.  Same result as gfy_build_csr + gfy_encode with fewer
 * launches: the last stage of the CSR build runs inside the encoder's setup launch and no
 * counter is zeroed per call (3 + layers launches instead of 6 + layers for the fp16 model).
 *   edge_index / edge_types   as for gfy_build_csr; the other arguments as for gfy_encode
 *   workspace                 gfy_encode_coo_workspace_bytes(); its first
 *                             gfy_encode_coo_clear_bytes(n_nodes) bytes must be ZERO when the
 *                             call starts and are zero again when it has run: clear a new
 *                             workspace once with gfy_encode_coo_prepare (or hipMemset the
 *                             whole of it) and again whenever n_nodes / n_edges change or
 *                             anything else has written to it.                             */
size_t gfy_encode_coo_workspace_bytes(const gfy_encoder* encoder, int64_t n_nodes,
                                      int64_t n_edges);
size_t gfy_encode_coo_clear_bytes(int64_t n_nodes);
int gfy_encode_coo_prepare(void* workspace, size_t workspace_bytes, int64_t n_nodes,
                           void* stream);
int gfy_encode_coo(gfy_encoder* encoder, const float* node_features,
                   const int32_t* edge_index, const uint8_t* edge_types, int64_t n_nodes,
                   int64_t n_edges, const int32_t* out_rows, void* out, int out_dtype,
                   int normalise, void* workspace, size_t workspace_bytes, void* stream);

/* A BATCH of shards in one sequence of launches: what encode_graphs does micro-batch after
 * micro-batch (api.py:211-230 calling _run_graph_shard, api.py:232-260) for up to
 * GFY_MAX_BATCH_SHARDS micro-batches at once.  Shards never share edges (graph.py:392-395), so the
 * kernels treat the batch as one graph in which every shard starts on a 32-row tile boundary; the
 * result of every shard is bit-identical to gfy_encode_coo on that shard alone (the per-node
 * arithmetic does not depend on what else is in the launch).  Why: a 60,000-node shard gives each
 * of the 256 CUs less than one round of tiles, so a launch is mostly ramp, fill and drain; a
 * batch runs the persistent-rounds layer kernel (GFY_OPT_LAYER_KERNEL) and pays those once.
 *   shards_host   HOST array of n_shards descriptors; the pointers in them are DEVICE pointers
 *                 with the meaning of the gfy_encode_coo arguments of the same name
 *   workspace     gfy_encode_coo_batch_workspace_bytes(); its first
 *                 gfy_encode_coo_batch_clear_bytes() bytes must be ZERO when the call starts and
 *                 are zero again when it has run (hipMemset once; again when the shards' sizes
 *                 change or anything else has written to it)                                  */
#define GFY_MAX_BATCH_SHARDS 16
typedef struct gfy_shard {
  const float* node_features;
  const int32_t* edge_index;
  const uint8_t* edge_types;
  const int32_t* out_rows;   /* or NULL */
  void* out;
  int64_t n_nodes, n_edges;
  /* Optional (ABI 4): the shard's record boundaries as GraphShard keeps them (graph.py:268-271),
   * DEVICE arrays of n_records + 1 ascending int64 — record r owns the nodes
   * [node_ptr[r] - node_ptr[0], node_ptr[r + 1] - node_ptr[0]) and the edges
   * [edge_ptr[r] - edge_ptr[0], ...) of this shard, and no edge leaves its record
   * (graph.py:392-395).  When EVERY shard of a call has them, COO -> tile plans runs without
   * global atomics (csrc/csr_records.inc: one launch instead of two; a workgroup scans only the
   * edges of the records that overlap its rows).  NULL / 0: the counting kernel, as before.
   * Results are identical either way.  Pass them only where a record's edge list is short
   * against the shard (every workgroup of a record reads the record's whole edge list).      */
  const int64_t* node_ptr;
  const int64_t* edge_ptr;
  int64_t n_records;
} gfy_shard;
size_t gfy_encode_coo_batch_workspace_bytes(const gfy_encoder* encoder,
                                            const gfy_shard* shards_host, int n_shards);
size_t gfy_encode_coo_batch_clear_bytes(const gfy_shard* shards_host, int n_shards);
int gfy_encode_coo_batch(gfy_encoder* encoder, const gfy_shard* shards_host, int n_shards,
                         int out_dtype, int normalise, void* workspace, size_t workspace_bytes,
                         void* stream);

/* Debug/parity tap: copy the hidden state after `stage` into `out`
 * ([N][hidden] in the model dtype): stage 0 = input Linear, l+1 = after layer l.
 * Same arguments as gfy_encode; used by the stage-by-stage parity tests. */
int gfy_encode_hidden(gfy_encoder* encoder, const float* node_features,
                      const int32_t* row_ptr, const int32_t* col,
                      const uint8_t* typ, int64_t n_nodes, int64_t n_edges,
                      int stage, void* out, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Debug/parity tap of ONE GINE layer (GINEConv + LayerNorm + residual, _model.py:39-46,69-71)
 * run on a GIVEN hidden state — e.g. the reference's own recorded tensor, so that every layer
 * and every phase is pinned by itself instead of through the layers before it:
 *   hidden_in  fp16 [N][hidden], natural channel order (h of the previous layer, or h0)
 *   tap        GFY_TAP_H: h' = R(h + y) [N][hidden];  GFY_TAP_Z: z = R(R(s h) + agg) after the
 *              gather;  GFY_TAP_V: v = relu(R(BN(u))) [N][2 hidden];  GFY_TAP_W: w = R(v W1^T + b1);
 *              GFY_TAP_Y: y = R(LayerNorm(w))
 *   out        fp16, natural channel order.  fp16 model with the residual architecture only;
 *              runs the persistent-rounds kernel's tap instantiation (same code otherwise).   */
enum gfy_layer_tap { GFY_TAP_H = 0, GFY_TAP_Z = 1, GFY_TAP_V = 2, GFY_TAP_W = 3, GFY_TAP_Y = 4 };
size_t gfy_debug_layer_workspace_bytes(const gfy_encoder* encoder, int64_t n_nodes,
                                       int64_t n_edges);
int gfy_debug_layer(gfy_encoder* encoder, int layer, const void* hidden_in,
                    const int32_t* row_ptr, const int32_t* col, const uint8_t* typ,
                    int64_t n_nodes, int64_t n_edges, int tap, void* out, void* workspace,
                    size_t workspace_bytes, void* stream);

/* Per-kernel device timing of gfy_encode (diagnostics; bench.py's roofline
 * figure).  While enabled, gfy_encode brackets each kernel with hipEvents on the
 * caller's stream (do not enable under graph capture).  enable = 2 leaves out the
 * events BETWEEN layer launches 1 .. layers-1 and reports their mean: an event between
 * two dependent kernels adds ~2.5 us of stream time that a profiler's kernel duration
 * does not contain.  enable = 3 (fp16 model) records no events: every layer launch notes the
 * device clock of its first workgroup start and last workgroup end, and
 * gfy_encoder_get_timing returns those `layers` kernel durations (10 ns resolution) — what a
 * profiler reports, also when other streams' kernels run between two events of this stream.
 * gfy_encoder_get_timing
 * waits for the last gfy_encode and writes milliseconds to ms_host:
 * [0] per-encode setup (tile plans + input Linear), [1..layers] GINE layer
 * launches, [layers+1] stand-alone head+normalise (fp16-model fp16 output: the
 * last layer's launch runs the head too and this entry is ~0);
 * *count receives layers+2. */
int gfy_encoder_set_timing(gfy_encoder* encoder, int enable);
int gfy_encoder_get_timing(gfy_encoder* encoder, float* ms_host, int capacity,
                           int* count);

/* Diagnostic options of one encoder (no reference counterpart; results within the stated
 * tolerances for every setting).  Set between encodes, never read from the environment.
 *   GFY_OPT_SEPARATE_HEAD  1: head + normalise as its own launch even for fp16 output
 *   GFY_OPT_LAYER_KERNEL   -1 (default): by the launch — several rounds of tiles per CU (a batch,
 *                          a large micro-batch) run the windowed kernel (two 4-wave workgroups per
 *                          CU, weights streamed; edge_dim <= 12, else persistent rounds), one
 *                          round the one-round kernel whose last launch carries the head;
 *                          1 / 3 / 4 force one-round / persistent rounds / windowed
 *   GFY_OPT_STAGGER        rounds kernels: start offset between the workgroups of an XCD in
 *                          shader cycles; -1 (default): 500 (windowed: 250) from three rounds up
 *   GFY_OPT_PRIORITY       windowed kernel: s_setprio level of a CU's first (bits 1:0) / second
 *                          (bits 3:2) workgroup while it multiplies, of the second one elsewhere
 *                          (bits 5:4); -1 (default): 4                                            */
enum gfy_option { GFY_OPT_SEPARATE_HEAD = 2, GFY_OPT_LAYER_KERNEL = 3, GFY_OPT_STAGGER = 4,
                  GFY_OPT_PRIORITY = 6 };
int gfy_encoder_set_option(gfy_encoder* encoder, int option, int value);
/* Which layer kernel the encoder's last fp16-model encode launched (the values of
 * GFY_OPT_LAYER_KERNEL: 1 one round, 3 persistent rounds, 4 windowed; 0: none yet): lets a
 * caller — and the tests — see that a call took the batched path. */
int gfy_encoder_last_layer_kernel(const gfy_encoder* encoder);

/* ---- host (CPU) implementation --------------------------------------------------------
 * The reference's default device is the CPU (api.py:64-76: Ginfinity.load(device="cpu")); these
 * entry points serve it: the same rounding-point model in plain C++ (csrc/gine_host.cpp),
 * threads over node blocks, no HIP call.  For the drop-in surface on a box without a GPU
 * (BASELINE configs[0]) — not a fallback: a gfy_encoder never routes here.  ALL pointers are
 * HOST pointers; gfy_host_encode returns when the result is written.  Arguments as
 * gfy_encode_coo; `threads` <= 1 runs on the calling thread.                                */
typedef struct gfy_host_encoder gfy_host_encoder;
int gfy_host_encoder_create(const void* weight_pack_host, size_t bytes, int model_dtype,
                            gfy_host_encoder** out);
void gfy_host_encoder_destroy(gfy_host_encoder* encoder);
int gfy_host_encode(const gfy_host_encoder* encoder, const float* node_features,
                    const int32_t* edge_index, const uint8_t* edge_types, int64_t n_nodes,
                    int64_t n_edges, const int32_t* out_rows, void* out, int out_dtype,
                    int normalise, int threads);

/* ---- host-side packing of one micro-batch (no HIP call; exported by both libraries) ------
 * Records [start, stop) of a host shard -> one staging block, laid out as the device arrays of a
 * gfy_shard: what Ginfinity.encode_graphs does per micro-batch with GraphShard.slice
 * (api.py:211-230, graph.py:414-444: node rows cut, edge_index rebased by -node_ptr[start],
 * ptr arrays cut) plus the core-row map of api.py:253-257 — in ONE call that holds no
 * interpreter lock, so a pool of packer threads scales (the numpy form of it kept the packers of
 * 128 micro-batches behind one lock: 19-24 ms for 561 MB).
 *   slot + base      where the block starts (page-locked staging memory; base % 256 == 0)
 *   offsets[6]       out: byte offsets FROM `slot` of node_features, edge_index (2 x edges),
 *                    edge_types, out_rows (int32, -1 = dropped row; absent when every node is a
 *                    core node), node_ptr, edge_ptr (absent unless with_records and every
 *                    record has <= 65,536 edges); -1 = absent; every array starts at a multiple
 *                    of 256
 *   counts[4]        out: nodes, edges, records (0 = boundaries absent), kept rows
 * Returns GFY_ERR_INVALID (gfy_last_error: which) for an edge that leaves the records' node range
 * — the check of GraphShard.slice (graph.py:318-321).                                          */
int gfy_pack_microbatch(const float* node_features, int feature_dim, const int32_t* edge_index,
                        int64_t edges_total, const uint8_t* edge_types,
                        const uint8_t* node_roles, const int64_t* node_ptr,
                        const int64_t* edge_ptr, int64_t start, int64_t stop, int with_records,
                        void* slot, int64_t base, int64_t* offsets, int64_t* counts);

/* ---- one packed group of micro-batches -> the device (libgfy.so only) ---------------------
 * What follows gfy_pack_microbatch in the micro-batch loop of Ginfinity.encode_graphs
 * (api.py:211-230: `batch.to(device)`).  A gfy_upload_ring owns one event per staging slot of
 * its caller (`slots` of them, 1..64; made on the device that is current at creation).
 * gfy_upload_async: `bytes` of page-locked staging memory (slot `slot`) go up by ONE
 * asynchronous copy on `copy_stream`, and `consumer_stream` — the stream gfy_encode_coo_batch
 * of that group is issued on — waits for it; nothing blocks the host.  gfy_upload_wait returns
 * when the last upload from `slot` has left it, i.e. when a packer may write into it again (at
 * once if there was none).  One call per group instead of a copy, two event records, a stream
 * wait and a stream switch in the caller's interpreter (encode_shards_device: ~140 us of ~400
 * per group of the launching thread, which is what bounds a rank fed from host arrays).  A ring
 * is used by one thread at a time.                                                           */
typedef struct gfy_upload_ring gfy_upload_ring;
int gfy_upload_ring_create(int slots, gfy_upload_ring** ring_out);
void gfy_upload_ring_destroy(gfy_upload_ring* ring);
int gfy_upload_async(gfy_upload_ring* ring, int slot, void* device_dst, const void* pinned_src,
                     size_t bytes, void* copy_stream, void* consumer_stream);
int gfy_upload_wait(gfy_upload_ring* ring, int slot);

/* ---- all-pairs distance over 128-d embeddings ----------------------------------
 * No reference symbol (the aligner lives in the external `ginfinity-sw`;
 * only parameters are exported: api.py:47-50, data/alignment.json:6) — defined
 * here as D_ij = sqrt(max(|a_i|^2 + |b_j|^2 - 2 a_i.b_j, 0)) (GFY_L2) or
 * S_ij = a_i.b_j / (|a_i||b_j|) (GFY_COSINE); fp16 inputs, fp32 accumulation
 * on the matrix cores.
 *   a [n][128] fp16, b [m][128] fp16.                                           */

size_t gfy_pairwise_workspace_bytes(int64_t n, int64_t m);

/* Dense block: out float32 [n][m]  (small blocks only: n*m*4 bytes). */
int gfy_pairwise_dense(const void* a, int64_t n, const void* b, int64_t m,
                       int metric, float* out, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Fused row reduction, the N x M matrix is never materialised:
 *   best_val float32 [n], best_idx int32 [n]: nearest b-row of each a-row
 *   (smallest distance for GFY_L2, largest similarity for GFY_COSINE; ties ->
 *   lowest index).  exclude_offset >= 0 skips the pair (i, i + exclude_offset)
 *   — "self" when b is a with a row offset; -1 excludes nothing.               */
int gfy_pairwise_nearest(const void* a, int64_t n, const void* b, int64_t m,
                         int metric, int64_t exclude_offset, float* best_val,
                         int32_t* best_idx, void* workspace,
                         size_t workspace_bytes, void* stream);

/* The same when b IS the rows [window_first, window_first + m) of a (one rank's own piece in
 * the chunked cross-shard search): every a-row skips itself, i.e. the pair (window_first + j, j)
 * is excluded for every j — rows in front of the window, inside it and behind it in ONE call. */
int gfy_pairwise_nearest_window(const void* a, int64_t n, const void* b, int64_t m,
                                int metric, int64_t window_first, float* best_val,
                                int32_t* best_idx, void* workspace, size_t workspace_bytes,
                                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GFY_H_ */
