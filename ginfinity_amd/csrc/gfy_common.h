// Shared host/device helpers of libgfy (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>

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

// ---- fp16 layer parameters on the device (see gine_f16.hip for the layouts) ------
struct LayerF16 {
  const f16* edge_table;  // [kMaxEdgeTypes][128]  R(W_edge[:,t] + b_edge), rows >= edge_dim zero
  float scale;            // fp16 value R(1 + R(eps)) widened to fp32
  const f16* w0_frag;     // mlp.0.weight in MFMA B-fragment order
  const f16* b0;          // [256]
  const float* bn_alpha;  // [256]  invstd * gamma          (fp32, from fp16-rounded buffers)
  const float* bn_shift;  // [256]  beta - mean * alpha
  const f16* w1_frag;     // mlp.4.weight in MFMA B-fragment order
  const f16* b1;          // [128]
  const f16* ln_gamma;    // [128]
  const f16* ln_beta;     // [128]
  // third-generation layer kernel (gine_layer3.inc)
  const f16* w01_image;   // 128 KB: mlp.0.weight fragments | mlp.4.weight fragments for a B
                          // operand taken from the first product's result (pack_chain_fragments)
  const void* image3;     // 7,680-byte LDS image: edge table + -inf row, alpha, shift, b0,
                          // b1, gamma, beta in the orders the lanes read them
};

struct HeadF16 {
  const f16* wa_frag;  // head.0.weight fragments
  const f16* ba;       // [128]
  const f16* wb_frag;  // head.2.weight fragments
  const f16* bb;       // [128]
  const f16* w_chain;  // gine_layer3.inc: head.0 fragments (32 KB) | head.2 chained (32 KB)
  const void* image3;  // 512-byte LDS image: ba (GEMM result order) | bb (hidden order)
};

struct ModelF16 {
  const f16* w_in;  // [8][16][8]: [c & 7][c >> 3][k], k padded 7 -> 8 with zero
  const f16* b_in;  // [128]
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
  int layer_workgroups = 0;   // gfy_encoder_set_layer_workgroups (0 = default)
  int layer_kernel = 3;       // GFY_OPT_LAYER_KERNEL: 3 = gine_layer3.inc, 2 = gine_layer.inc
  int separate_head = 0;      // GFY_OPT_SEPARATE_HEAD
  int tune = 0;               // GFY_OPT_TUNE: diagnostic schedule switches of gine_layer3.inc
  int timing = 0;
  hipEvent_t events[gfy::kMaxLayers + 3] = {};
  mutable int events_recorded = 0;
  void mark(hipStream_t s, int slot) const {
    if (!timing) return;
    if (timing == 2 && slot >= 2 && slot < layers) return;
    (void)hipEventRecord(events[slot], s);
    events_recorded = slot + 1;
  }
};

// ---- kernel launchers (one per .hip file) -------------------------------------------
namespace gfy {

int launch_build_csr(const int32_t* edge_index, const uint8_t* edge_types,
                     int64_t n, int64_t e, int32_t* row_ptr, int32_t* col,
                     uint8_t* typ, void* ws, size_t ws_bytes, hipStream_t s);
size_t csr_workspace_bytes(int64_t n, int64_t e);

int launch_build_graphs(const uint8_t* bases, const uint8_t* marks, const int64_t* node_ptr,
                        const int64_t* edge_ptr, int64_t records, int64_t n, int64_t e,
                        int struct_states, int positional_cols, int skip2,
                        const float* positional, float* features, int32_t* edge_index,
                        uint8_t* edge_types, int32_t* first_invalid, hipStream_t s);

int launch_encode_f16(const gfy_encoder* enc, const float* x,
                      const int32_t* row_ptr, const int32_t* col,
                      const uint8_t* typ, int64_t n, int64_t e,
                      const int32_t* out_rows, void* out, int out_dtype,
                      int normalise, int tap_stage, void* ws, size_t ws_bytes,
                      hipStream_t s);
size_t encode_f16_workspace_bytes(int64_t n, int64_t e);

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
                            int metric, int64_t exclude_offset, float* best_val,
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

// W[n_out][k_in] (row-major fp16) -> MFMA 32x32x16 B-operand fragment order:
// frag[(ntile * ksteps + ks) * 64 + lane][8] = W[32*ntile + (lane & 31)][16*ks + 8*(lane >> 5) + j]
void pack_b_fragments(const f16* w, int n_out, int k_in, f16* frag);

// Fragments of the SECOND product of a chain (gine_layer3.inc): its B operand is the first
// product's 32x32 result converted in place, so k-step s holds, in element j of lane half h,
// input channel 16 s + 8 (j >> 2) + 4 h + (j & 3); and its rows are dealt so that the result
// lands in the hidden-state layout (lane half h, register i of block blk = output channel
// 32 blk + 16 (i >> 3) + 8 h + (i & 7)).
void pack_chain_fragments(const f16* w, int n_out, int k_in, f16* frag);
// index helpers shared by the constant images
inline int gemm_result_channel(int block, int half, int reg) {   // 32x32 C/D layout
  return 32 * block + (reg & 3) + 8 * (reg >> 2) + 4 * half;
}
inline int hidden_layout_channel(int half, int ks, int j) { return 16 * ks + 8 * half + j; }

}  // namespace gfy
