// C ABI of libgfy (include/gfy.h): argument checking, weight-pack parsing and
// the host-side derivation of everything the fp16 kernels consume.
//
// Reference behaviour reproduced here (host side, once per encoder):
//   model.half()                       src/ginfinity/api.py:111-112
//   edge_lin on a one-hot row          src/ginfinity/_model.py:43   -> 10-row table
//   (1 + eps) formed in fp16           src/ginfinity/_model.py:46
//   BatchNorm1d eval affine in fp32    src/ginfinity/_model.py:35
// The rounding points follow SURVEY.md §8-A / oracle/gine_numpy.py.
// Compiled with -ffp-contract=off: alpha/shift must round after every step.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "gfy_common.h"

namespace gfy {

// (error channel, ABI version, weight-pack size: gfy_base.cpp, shared with libgfy_host.so)
size_t pack_floats(uint32_t in_dim, uint32_t h, uint32_t layers, uint32_t edge_dim,
                   uint32_t out_dim);

void pack_b_fragments(const f16* w, int n_out, int k_in, f16* frag) {
  const int tiles = n_out / 32, ksteps = k_in / 16;
  for (int tile = 0; tile < tiles; ++tile)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j)
          frag[(((size_t)tile * ksteps + ks) * 64 + lane) * 8 + j] =
              w[(size_t)(32 * tile + (lane & 31)) * k_in + 16 * ks +
                8 * (lane >> 5) + j];
}

void pack_k_chained(const f16* w, int n_out, int k_in, f16* frag) {
  const int blocks = n_out / 32, ksteps = k_in / 16;
  for (int blk = 0; blk < blocks; ++blk)
    for (int s = 0; s < ksteps; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j)
          frag[(((size_t)blk * ksteps + s) * 64 + lane) * 8 + j] =
              w[(size_t)(32 * blk + (lane & 31)) * k_in + 16 * s + 8 * (j >> 2) +
                4 * (lane >> 5) + (j & 3)];
}

void pack_chain_fragments(const f16* w, int n_out, int k_in, f16* frag) {
  const int blocks = n_out / 32, ksteps = k_in / 16;
  for (int blk = 0; blk < blocks; ++blk)
    for (int s = 0; s < ksteps; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 31, half = lane >> 5;
        const int reg = (m & 3) + 4 * (m >> 3), row_half = (m >> 2) & 1;   // C/D row m
        const int out_channel = 32 * blk + 16 * (reg >> 3) + 8 * row_half + (reg & 7);
        for (int j = 0; j < 8; ++j) {
          const int k = 16 * s + 8 * (j >> 2) + 4 * half + (j & 3);
          frag[(((size_t)blk * ksteps + s) * 64 + lane) * 8 + j] =
              w[(size_t)out_channel * k_in + k];
        }
      }
}

namespace {

constexpr uint32_t kMagic = 0x31594647u;  // 'GFY1'

struct PackHeader {
  uint32_t magic, version, in_dim, hidden, layers, edge_dim, out_dim, flags;
};
static_assert(sizeof(PackHeader) == 32, "gfy_weight_pack_bytes (gfy_base.cpp) counts 32 bytes");

// bump allocator over a host staging image of the device blob
struct Blob {
  std::vector<char> host;
  size_t reserve(size_t bytes) {
    const size_t at = align_up(host.size(), 256);
    host.resize(at + bytes, 0);
    return at;
  }
  template <typename T>
  T* at(size_t off) { return reinterpret_cast<T*>(host.data() + off); }
};

inline f16 rh(float v) { return (f16)v; }  // RNE, as torch's .half()

// Makes `device` current for the life of the guard and gives the caller's device back: the
// library never leaves the calling thread on another device than it came with.
struct DeviceGuard {
  int previous = -1;
  hipError_t status;
  explicit DeviceGuard(int device) {
    status = hipGetDevice(&previous);
    if (status == hipSuccess && previous != device) status = hipSetDevice(device);
  }
  ~DeviceGuard() {
    if (previous >= 0) (void)hipSetDevice(previous);
  }
};

// Every launching entry point runs on the CALLER's current device (streams and buffers are the
// caller's): an encoder built for another device is refused instead of faulting or silently
// going through peer access.
static int check_current_device(const gfy_encoder* enc, const char* who) {
  int current = -1;
  GFY_CHECK_HIP(hipGetDevice(&current));
  GFY_REQUIRE(current == enc->device, GFY_ERR_INVALID,
              "%s: the encoder lives on HIP device %d but the calling thread's current device is "
              "%d (hipSetDevice / torch.cuda.device first)", who, enc->device, current);
  return GFY_OK;
}

struct Reader {
  const float* p;
  const float* take(size_t n) {
    const float* r = p;
    p += n;
    return r;
  }
};

}  // namespace
}  // namespace gfy

using namespace gfy;

struct gfy_upload_ring {     // include/gfy.h: one event per staging slot of the caller
  int slots = 0;
  hipEvent_t done[64] = {};
  bool used[64] = {};
};

extern "C" {

int gfy_encoder_create(const void* weight_pack_host, size_t bytes,
                       int model_dtype, int device, gfy_encoder** out) {
  clear_error();
  GFY_REQUIRE(out != nullptr, GFY_ERR_INVALID, "gfy_encoder_create: out is NULL");
  *out = nullptr;
  GFY_REQUIRE(weight_pack_host && bytes >= sizeof(PackHeader), GFY_ERR_INVALID,
              "gfy_encoder_create: weight pack missing or truncated");
  PackHeader hd;
  std::memcpy(&hd, weight_pack_host, sizeof hd);
  GFY_REQUIRE(hd.magic == kMagic && hd.version == 1, GFY_ERR_INVALID,
              "gfy_encoder_create: bad weight-pack magic/version");
  GFY_REQUIRE(hd.hidden == (uint32_t)kHidden && hd.in_dim == (uint32_t)kInDim &&
                  hd.out_dim == (uint32_t)kOutDim,
              GFY_ERR_UNSUPPORTED,
              "gfy_encoder_create: kernels are built for in_dim=7 hidden=128 "
              "out_dim=128 (got %u/%u/%u)",
              hd.in_dim, hd.hidden, hd.out_dim);
  GFY_REQUIRE(hd.layers >= 1 && hd.layers <= (uint32_t)kMaxLayers,
              GFY_ERR_UNSUPPORTED, "gfy_encoder_create: layers=%u outside 1..%d",
              hd.layers, kMaxLayers);
  GFY_REQUIRE(hd.edge_dim >= 1 && hd.edge_dim <= (uint32_t)kMaxEdgeTypes,
              GFY_ERR_UNSUPPORTED, "gfy_encoder_create: edge_dim=%u outside 1..%d",
              hd.edge_dim, kMaxEdgeTypes);
  GFY_REQUIRE(bytes == gfy_weight_pack_bytes(hd.in_dim, hd.hidden, hd.layers,
                                             hd.edge_dim, hd.out_dim),
              GFY_ERR_INVALID, "gfy_encoder_create: weight pack is %zu bytes, expected %zu",
              bytes,
              gfy_weight_pack_bytes(hd.in_dim, hd.hidden, hd.layers, hd.edge_dim,
                                    hd.out_dim));
  GFY_REQUIRE(model_dtype == GFY_F16 || model_dtype == GFY_F32, GFY_ERR_INVALID,
              "gfy_encoder_create: model_dtype must be GFY_F16 or GFY_F32");
  DeviceGuard guard(device);   // the caller's current device is restored on every way out
  GFY_CHECK_HIP(guard.status);
  if (model_dtype == GFY_F16)
    if (const int rc = prepare_device_f16()) return rc;

  const int H = kHidden, M = kMlp, L = (int)hd.layers, ED = (int)hd.edge_dim;
  Reader rd{reinterpret_cast<const float*>((const char*)weight_pack_host +
                                           sizeof(PackHeader))};
  gfy_encoder* enc = new gfy_encoder();
  enc->device = device;
  enc->model_dtype = model_dtype;
  enc->layers = L;
  enc->edge_dim = ED;
  enc->residual = (int)(hd.flags & 1u);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
        cus >= 8)
      enc->cus = cus & ~7;
  }

  Blob blob;
  // offsets first (pointers are fixed up after the single upload)
  struct LayerOff {
    size_t w01, image;                                                // f16 mode
    size_t ftable, fw0, fb0, falpha, fshift, fw1, fb1, flg, flb;      // f32 mode
    float scale, one_plus_eps;
  };
  size_t o_win = 0, o_bin = 0, o_wa = 0, o_ba = 0, o_wb = 0, o_bb = 0, o_hchain = 0,
         o_himage = 0;
  std::vector<LayerOff> lo(L);

  const float* w_in = rd.take((size_t)H * kInDim);
  const float* b_in = rd.take(H);
  if (model_dtype == GFY_F16) {
    o_win = blob.reserve((size_t)H * 8 * sizeof(f16));
    o_bin = blob.reserve(H * sizeof(f16));
    for (int p = 0; p < H; ++p) {
      // stored position p holds channel stored_channel(p); packed [p & 7][p >> 3][k]: see
      // k_input_linear_f16
      const int c = stored_channel(p);
      for (int k = 0; k < kInDim; ++k)
        blob.at<f16>(o_win)[((p & 7) * 16 + (p >> 3)) * 8 + k] = rh(w_in[c * kInDim + k]);
      blob.at<f16>(o_bin)[p] = rh(b_in[c]);
    }
  } else {
    o_win = blob.reserve((size_t)H * kInDim * 4);
    o_bin = blob.reserve(H * 4);
    for (int c = 0; c < H; ++c)
      for (int k = 0; k < kInDim; ++k)
        blob.at<float>(o_win)[k * H + c] = w_in[c * kInDim + k];  // [K][N]
    std::memcpy(blob.at<float>(o_bin), b_in, H * 4);
  }

  std::vector<f16> tmp;
  for (int l = 0; l < L; ++l) {
    const float* eps = rd.take(1);
    const float* ew = rd.take((size_t)H * ED);
    const float* eb = rd.take(H);
    const float* w0 = rd.take((size_t)M * H);
    const float* b0 = rd.take(M);
    const float* bg = rd.take(M);
    const float* bb = rd.take(M);
    const float* bm = rd.take(M);
    const float* bv = rd.take(M);
    const float* w1 = rd.take((size_t)H * M);
    const float* b1 = rd.take(H);
    const float* lg = rd.take(H);
    const float* lb = rd.take(H);
    LayerOff& o = lo[l];
    if (model_dtype == GFY_F16) {
      o.scale = (float)rh(1.0f + (float)rh(eps[0]));  // fp16 scalar (1 + eps)
      // [W0 | W1] fragments: every B operand of the layer kernel is an MFMA result (or a
      // stored row, which has the same order) used in place
      o.w01 = blob.reserve((size_t)2 * M * H * sizeof(f16));
      tmp.resize((size_t)M * H);
      for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = rh(w0[i]);
      pack_k_chained(tmp.data(), M, H, blob.at<f16>(o.w01));
      for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = rh(w1[i]);
      pack_k_chained(tmp.data(), H, M, blob.at<f16>(o.w01) + (size_t)M * H);
      // LDS image of gine_layer.inc, in the order its lanes read it
      o.image = blob.reserve(7680);
      char* im = blob.at<char>(o.image);
      f16* table = reinterpret_cast<f16*>(im);           // [17][128], stored order
      for (int t = 0; t < ED; ++t)
        for (int p = 0; p < H; ++p) {   // R(R(W[c][t]) + R(b[c]))  — one-hot Linear
          const int c = stored_channel(p);
          table[t * H + p] = rh((float)rh(ew[c * ED + t]) + (float)rh(eb[c]));
        }
      for (int p = 0; p < H; ++p) table[kMaxEdgeTypes * H + p] = (f16)(-INFINITY);
      float* ia = reinterpret_cast<float*>(im + 4352);
      float* is = reinterpret_cast<float*>(im + 4352 + 1024);
      f16* ib0 = reinterpret_cast<f16*>(im + 4352 + 2048);
      for (int blk = 0; blk < M / 32; ++blk)
        for (int half = 0; half < 2; ++half)
          for (int reg = 0; reg < 16; ++reg) {
            const int c = gemm_result_channel(blk, half, reg);
            const int at = (blk * 2 + half) * 16 + reg;
            const float g = (float)rh(bg[c]), b = (float)rh(bb[c]);
            const float mean = (float)rh(bm[c]), var = (float)rh(bv[c]);
            const float invstd = 1.0f / std::sqrt(var + 1e-5f);
            const float alpha = invstd * g;
            const float prod = mean * alpha;  // separate rounding (no fma)
            ia[at] = alpha;
            is[at] = b - prod;
          }
      // b0 and b1 are added on the matrix cores (gine_layer.inc, mlp_pipe_step): natural channel
      // order — row lane % 32 of A-operand block b is channel 32 b + row
      for (int c = 0; c < M; ++c) ib0[c] = rh(b0[c]);
      f16* ib1 = reinterpret_cast<f16*>(im + 4352 + 2048 + 512);
      f16* ig = ib1 + H;
      f16* ib = ig + H;
      for (int blk = 0; blk < H / 32; ++blk)
        for (int half = 0; half < 2; ++half)
          for (int reg = 0; reg < 16; ++reg) {
            const int c = gemm_result_channel(blk, half, reg);
            const int at = (blk * 2 + half) * 16 + reg;
            ib1[c] = rh(b1[c]);
            ig[at] = rh(lg[c]);
            ib[at] = rh(lb[c]);
          }
    } else {
      auto put = [&](const float* src, size_t n) {
        const size_t off = blob.reserve(n * 4);
        std::memcpy(blob.at<float>(off), src, n * 4);
        return off;
      };
      auto put_transposed = [&](const float* src, int rows, int cols) {
        const size_t off = blob.reserve((size_t)rows * cols * 4);
        for (int r = 0; r < rows; ++r)
          for (int c = 0; c < cols; ++c)
            blob.at<float>(off)[(size_t)c * rows + r] = src[(size_t)r * cols + c];
        return off;
      };
      o.one_plus_eps = 1.0f + eps[0];
      o.ftable = blob.reserve((size_t)kMaxEdgeTypes * H * 4);
      for (int t = 0; t < ED; ++t)
        for (int c = 0; c < H; ++c)
          blob.at<float>(o.ftable)[t * H + c] = ew[c * ED + t] + eb[c];
      o.fw0 = put_transposed(w0, M, H);   // -> [H][M]
      o.fb0 = put(b0, M);
      o.falpha = blob.reserve(M * 4);
      o.fshift = blob.reserve(M * 4);
      for (int c = 0; c < M; ++c) {
        const float invstd = 1.0f / std::sqrt(bv[c] + 1e-5f);
        const float alpha = invstd * bg[c];
        const float prod = bm[c] * alpha;
        blob.at<float>(o.falpha)[c] = alpha;
        blob.at<float>(o.fshift)[c] = bb[c] - prod;
      }
      o.fw1 = put_transposed(w1, H, M);   // -> [M][H]
      o.fb1 = put(b1, H);
      o.flg = put(lg, H);
      o.flb = put(lb, H);
    }
  }
  const float* wa = rd.take((size_t)H * H);
  const float* ba = rd.take(H);
  const float* wb = rd.take((size_t)kOutDim * H);
  const float* bbias = rd.take(kOutDim);
  if (model_dtype == GFY_F16) {
    // stand-alone head kernel: reads stored rows (k chained), writes natural rows
    tmp.resize((size_t)H * H);
    for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = rh(wa[i]);
    o_wa = blob.reserve(tmp.size() * sizeof(f16));
    pack_k_chained(tmp.data(), H, H, blob.at<f16>(o_wa));
    // fused head: head.0 k-chained, head.2 chained with its rows dealt into natural chunks
    o_hchain = blob.reserve((size_t)2 * H * H * sizeof(f16));
    std::memcpy(blob.at<f16>(o_hchain), blob.at<f16>(o_wa), (size_t)H * H * sizeof(f16));
    for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = rh(wb[i]);
    o_wb = blob.reserve(tmp.size() * sizeof(f16));
    pack_b_fragments(tmp.data(), kOutDim, H, blob.at<f16>(o_wb));
    pack_chain_fragments(tmp.data(), kOutDim, H, blob.at<f16>(o_hchain) + (size_t)H * H);
    o_ba = blob.reserve(H * sizeof(f16));
    o_bb = blob.reserve(kOutDim * sizeof(f16));
    for (int c = 0; c < H; ++c) {
      blob.at<f16>(o_ba)[c] = rh(ba[c]);
      blob.at<f16>(o_bb)[c] = rh(bbias[c]);
    }
    o_himage = blob.reserve(512);
    {
      f16* iba = blob.at<f16>(o_himage);
      f16* ibb = iba + H;
      for (int blk = 0; blk < H / 32; ++blk)
        for (int half = 0; half < 2; ++half)
          for (int reg = 0; reg < 16; ++reg)
            iba[(blk * 2 + half) * 16 + reg] =
                blob.at<f16>(o_ba)[gemm_result_channel(blk, half, reg)];
      for (int half = 0; half < 2; ++half)
        for (int ks = 0; ks < 8; ++ks)
          for (int j = 0; j < 8; ++j)
            ibb[(half * 8 + ks) * 8 + j] =
                blob.at<f16>(o_bb)[hidden_layout_channel(half, ks, j)];
    }
  } else {
    o_wa = blob.reserve((size_t)H * H * 4);
    o_wb = blob.reserve((size_t)kOutDim * H * 4);
    for (int r = 0; r < H; ++r)
      for (int c = 0; c < H; ++c) {
        blob.at<float>(o_wa)[(size_t)c * H + r] = wa[(size_t)r * H + c];
        blob.at<float>(o_wb)[(size_t)c * kOutDim + r] = wb[(size_t)r * H + c];
      }
    o_ba = blob.reserve(H * 4);
    std::memcpy(blob.at<float>(o_ba), ba, H * 4);
    o_bb = blob.reserve(kOutDim * 4);
    std::memcpy(blob.at<float>(o_bb), bbias, kOutDim * 4);
  }

  void* dev = nullptr;
  hipError_t err = hipMalloc(&dev, blob.host.size());
  if (err == hipSuccess)
    err = hipMemcpy(dev, blob.host.data(), blob.host.size(), hipMemcpyHostToDevice);
  if (err != hipSuccess) {
    if (dev) (void)hipFree(dev);
    delete enc;
    set_error("gfy_encoder_create: device upload failed: %s", hipGetErrorString(err));
    return GFY_ERR_HIP;
  }
  enc->device_blob = dev;
  enc->device_blob_bytes = blob.host.size();
  char* base = (char*)dev;
  auto H16 = [&](size_t off) { return (const f16*)(base + off); };
  auto F32p = [&](size_t off) { return (const float*)(base + off); };
  if (model_dtype == GFY_F16) {
    enc->f16.w_in = H16(o_win);
    enc->f16.b_in = H16(o_bin);
    for (int l = 0; l < L; ++l) {
      LayerF16& d = enc->f16.layer[l];
      const LayerOff& o = lo[l];
      d.scale = o.scale;
      d.edge_types = ED;
      d.w01_image = H16(o.w01);
      d.image = base + o.image;
    }
    enc->f16.head = HeadF16{H16(o_wa), H16(o_ba), H16(o_wb), H16(o_bb), H16(o_hchain),
                            base + o_himage};
  } else {
    enc->f32.w_in_t = F32p(o_win);
    enc->f32.b_in = F32p(o_bin);
    for (int l = 0; l < L; ++l) {
      LayerF32& d = enc->f32.layer[l];
      const LayerOff& o = lo[l];
      d.table = F32p(o.ftable);
      d.one_plus_eps = o.one_plus_eps;
      d.w0t = F32p(o.fw0);
      d.b0 = F32p(o.fb0);
      d.alpha = F32p(o.falpha);
      d.shift = F32p(o.fshift);
      d.w1t = F32p(o.fw1);
      d.b1 = F32p(o.fb1);
      d.ln_g = F32p(o.flg);
      d.ln_b = F32p(o.flb);
    }
    enc->f32.ha_wt = F32p(o_wa);
    enc->f32.ha_b = F32p(o_ba);
    enc->f32.hb_wt = F32p(o_wb);
    enc->f32.hb_b = F32p(o_bb);
  }
  *out = enc;
  return GFY_OK;
}

int gfy_encoder_set_timing(gfy_encoder* enc, int enable) {
  clear_error();
  GFY_REQUIRE(enc != nullptr, GFY_ERR_INVALID, "gfy_encoder_set_timing: encoder is NULL");
  GFY_REQUIRE(enable != 3 || enc->model_dtype == GFY_F16, GFY_ERR_UNSUPPORTED,
              "gfy_encoder_set_timing: mode 3 (device-clock spans) exists for the fp16 model only");
  DeviceGuard guard(enc->device);
  GFY_CHECK_HIP(guard.status);
  if (enable && !enc->events[0])
    for (auto& ev : enc->events) GFY_CHECK_HIP(hipEventCreate(&ev));
  if (enable == 3 && !enc->device_spans) {
    GFY_CHECK_HIP(hipMalloc((void**)&enc->device_spans, 2 * kMaxLayers * sizeof(unsigned long long)));
    // start = +inf, end = 0: a launch that never ran reads as "no span", not as garbage
    unsigned long long init[2 * kMaxLayers];
    for (int l = 0; l < kMaxLayers; ++l) init[2 * l] = ~0ull, init[2 * l + 1] = 0;
    GFY_CHECK_HIP(hipMemcpy(enc->device_spans, init, sizeof init, hipMemcpyHostToDevice));
  }
  enc->timing = enable == 2 || enable == 3 ? enable : enable != 0;
  enc->events_recorded = 0;
  return GFY_OK;
}

int gfy_encoder_set_option(gfy_encoder* enc, int option, int value) {
  clear_error();
  GFY_REQUIRE(enc != nullptr, GFY_ERR_INVALID, "gfy_encoder_set_option: encoder is NULL");
  switch (option) {
    case GFY_OPT_SEPARATE_HEAD:
      GFY_REQUIRE(value == 0 || value == 1, GFY_ERR_INVALID,
                  "gfy_encoder_set_option: GFY_OPT_SEPARATE_HEAD must be 0 or 1 (got %d)", value);
      enc->separate_head = value;
      return GFY_OK;
    case GFY_OPT_LAYER_KERNEL:
      GFY_REQUIRE(value == -1 || value == 1 || value == 3 || value == 4 || value == 5,
                  GFY_ERR_INVALID,
                  "gfy_encoder_set_option: GFY_OPT_LAYER_KERNEL must be -1 (by rounds), 1 (round-2 "
                  "kernel), 3 (persistent rounds), 4 (two windowed workgroups per CU) or 5 (three "
                  "workgroups per CU), got %d", value);
      enc->layer_kernel = value;
      return GFY_OK;
    case GFY_OPT_STAGGER:
      GFY_REQUIRE(value >= -1 && value <= 100000, GFY_ERR_INVALID,
                  "gfy_encoder_set_option: GFY_OPT_STAGGER must be -1 (default) or 0..100000 "
                  "cycles (got %d)", value);
      enc->stagger = value;
      return GFY_OK;
    case GFY_OPT_PRIORITY:
      GFY_REQUIRE(value >= -1 && value <= 63, GFY_ERR_INVALID,
                  "gfy_encoder_set_option: GFY_OPT_PRIORITY must be -1 (default) or 0..63 (got %d)",
                  value);
      enc->priority = value;
      return GFY_OK;
    default:
      set_error("gfy_encoder_set_option: unknown option %d", option);
      return GFY_ERR_INVALID;
  }
}

int gfy_upload_ring_create(int slots, gfy_upload_ring** ring_out) {
  clear_error();
  GFY_REQUIRE(ring_out && slots >= 1 && slots <= 64, GFY_ERR_INVALID,
              "gfy_upload_ring_create: NULL argument or slots %d outside 1..64", slots);
  gfy_upload_ring* ring = new gfy_upload_ring();
  ring->slots = slots;
  for (int slot = 0; slot < slots; ++slot) {
    hipError_t err = hipEventCreateWithFlags(&ring->done[slot], hipEventDisableTiming);
    if (err != hipSuccess) {
      set_error("hipEventCreateWithFlags failed: %s", hipGetErrorString(err));
      ring->slots = slot;
      gfy_upload_ring_destroy(ring);
      return GFY_ERR_HIP;
    }
  }
  *ring_out = ring;
  return GFY_OK;
}

void gfy_upload_ring_destroy(gfy_upload_ring* ring) {
  if (!ring) return;
  for (int slot = 0; slot < ring->slots; ++slot) (void)hipEventDestroy(ring->done[slot]);
  delete ring;
}

int gfy_upload_async(gfy_upload_ring* ring, int slot, void* device_dst, const void* pinned_src,
                     size_t bytes, void* copy_stream, void* consumer_stream) {
  clear_error();
  GFY_REQUIRE(ring && slot >= 0 && slot < ring->slots, GFY_ERR_INVALID,
              "gfy_upload_async: NULL ring or slot %d outside its slots", slot);
  GFY_REQUIRE(bytes == 0 || (device_dst && pinned_src), GFY_ERR_INVALID,
              "gfy_upload_async: NULL source or destination");
  hipStream_t copies = static_cast<hipStream_t>(copy_stream);
  if (bytes)
    GFY_CHECK_HIP(hipMemcpyAsync(device_dst, pinned_src, bytes, hipMemcpyHostToDevice, copies));
  GFY_CHECK_HIP(hipEventRecord(ring->done[slot], copies));
  ring->used[slot] = true;
  if (consumer_stream != copy_stream)
    GFY_CHECK_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream),
                                     ring->done[slot], 0));
  return GFY_OK;
}

int gfy_upload_wait(gfy_upload_ring* ring, int slot) {
  clear_error();
  GFY_REQUIRE(ring && slot >= 0 && slot < ring->slots, GFY_ERR_INVALID,
              "gfy_upload_wait: NULL ring or slot %d outside its slots", slot);
  if (ring->used[slot]) GFY_CHECK_HIP(hipEventSynchronize(ring->done[slot]));
  return GFY_OK;
}

int gfy_encoder_last_layer_kernel(const gfy_encoder* enc) {
  return enc ? enc->last_layer_kernel : 0;
}

int gfy_encoder_get_timing(gfy_encoder* enc, float* ms_host, int capacity, int* count) {
  clear_error();
  GFY_REQUIRE(enc && ms_host && count, GFY_ERR_INVALID, "gfy_encoder_get_timing: NULL argument");
  if (enc->timing == 3) {   // device clock spans of the layer launches, 100 MHz
    GFY_REQUIRE(capacity >= enc->layers, GFY_ERR_INVALID,
                "gfy_encoder_get_timing: capacity %d < %d", capacity, enc->layers);
    unsigned long long spans[2 * kMaxLayers];
    // the encode may have run on a non-blocking stream, which a plain hipMemcpy does not wait for
    if (const int rc = check_current_device(enc, "gfy_encoder_get_timing")) return rc;
    GFY_CHECK_HIP(hipStreamSynchronize(enc->last_stream));
    GFY_CHECK_HIP(hipMemcpy(spans, enc->device_spans, sizeof spans, hipMemcpyDeviceToHost));
    for (int l = 0; l < enc->layers; ++l) {
      GFY_REQUIRE(spans[2 * l + 1] >= spans[2 * l], GFY_ERR_INVALID,
                  "gfy_encoder_get_timing: layer launch %d left no span (no fp16 encode since "
                  "timing was set to 3?)", l);
      ms_host[l] = (float)((double)(spans[2 * l + 1] - spans[2 * l]) * 1e-5);   // 10 ns ticks
    }
    *count = enc->layers;
    return GFY_OK;
  }
  const int spans = enc->events_recorded - 1;
  GFY_REQUIRE(enc->timing && spans >= 1, GFY_ERR_INVALID,
              "gfy_encoder_get_timing: timing is off or nothing was recorded");
  GFY_REQUIRE(capacity >= spans, GFY_ERR_INVALID,
              "gfy_encoder_get_timing: capacity %d < %d", capacity, spans);
  GFY_CHECK_HIP(hipEventSynchronize(enc->events[spans]));
  const int L = enc->layers;
  for (int i = 0; i < spans; ++i) {
    if (enc->timing == 2 && i >= 1 && i < L) {   // layers 1 .. L-1: one span, reported as its mean
      if (i == 1) {
        float block = 0.f;
        GFY_CHECK_HIP(hipEventElapsedTime(&block, enc->events[1], enc->events[L]));
        for (int k = 1; k < L; ++k) ms_host[k] = block / (float)(L - 1);
      }
      continue;
    }
    GFY_CHECK_HIP(hipEventElapsedTime(&ms_host[i], enc->events[i], enc->events[i + 1]));
  }
  *count = spans;
  return GFY_OK;
}

void gfy_encoder_destroy(gfy_encoder* enc) {
  if (!enc) return;
  for (auto& ev : enc->events)
    if (ev) (void)hipEventDestroy(ev);
  if (enc->device_blob) (void)hipFree(enc->device_blob);
  if (enc->device_spans) (void)hipFree(enc->device_spans);
  delete enc;
}

size_t gfy_csr_workspace_bytes(int64_t n, int64_t e) {
  return csr_workspace_bytes(n < 1 ? 1 : n, e < 0 ? 0 : e);
}

int gfy_build_csr(const int32_t* edge_index, const uint8_t* edge_types,
                  int64_t n, int64_t e, int32_t* row_ptr, int32_t* col,
                  uint8_t* typ, void* ws, size_t ws_bytes, void* stream) {
  clear_error();
  GFY_REQUIRE(row_ptr && ws, GFY_ERR_INVALID, "gfy_build_csr: NULL output/workspace");
  GFY_REQUIRE(e == 0 || (edge_index && edge_types && col && typ), GFY_ERR_INVALID,
              "gfy_build_csr: NULL edge array with E=%lld", (long long)e);
  return launch_build_csr(edge_index, edge_types, n, e, row_ptr, col, typ, ws,
                          ws_bytes, (hipStream_t)stream);
}

int gfy_build_graphs(const uint8_t* bases, const uint8_t* marks, const int64_t* node_ptr,
                     const int64_t* edge_ptr, int64_t records, int64_t n, int64_t e,
                     int struct_states, int positional_columns, int skip2,
                     const float* positional, float* node_features, int32_t* edge_index,
                     uint8_t* edge_types, int32_t* first_invalid, void* stream) {
  clear_error();
  GFY_REQUIRE(records >= 0 && n >= 0 && n < INT32_MAX && e >= 0 && e < INT32_MAX,
              GFY_ERR_INVALID, "gfy_build_graphs: records=%lld n=%lld e=%lld out of range",
              (long long)records, (long long)n, (long long)e);
  GFY_REQUIRE(struct_states == 1 || struct_states == 3, GFY_ERR_INVALID,
              "gfy_build_graphs: struct_states must be 1 (paired flag) or 3 (one-hot)");
  GFY_REQUIRE(positional_columns == 0 || positional_columns == 2, GFY_ERR_INVALID,
              "gfy_build_graphs: positional_columns must be 0 or 2");
  GFY_REQUIRE(node_ptr && edge_ptr && first_invalid, GFY_ERR_INVALID,
              "gfy_build_graphs: NULL argument");
  GFY_REQUIRE(n == 0 || (bases && marks && node_features), GFY_ERR_INVALID,
              "gfy_build_graphs: NULL node array with N=%lld", (long long)n);
  GFY_REQUIRE(n == 0 || positional_columns == 0 || positional, GFY_ERR_INVALID,
              "gfy_build_graphs: positional columns requested but positional is NULL");
  GFY_REQUIRE(e == 0 || (edge_index && edge_types), GFY_ERR_INVALID,
              "gfy_build_graphs: NULL edge array with E=%lld", (long long)e);
  return launch_build_graphs(bases, marks, node_ptr, edge_ptr, records, n, e, struct_states,
                             positional_columns, skip2 ? 1 : 0, positional, node_features,
                             edge_index, edge_types, first_invalid, (hipStream_t)stream);
}

size_t gfy_encode_workspace_bytes(const gfy_encoder* enc, int64_t n, int64_t e) {
  if (!enc) return 0;
  n = n < 1 ? 1 : n;
  return enc->model_dtype == GFY_F16 ? encode_f16_workspace_bytes(n, e)
                                     : encode_f32_workspace_bytes(n, e);
}

static int encode_common(gfy_encoder* enc, const float* x, const int32_t* row_ptr,
                         const int32_t* col, const uint8_t* typ, int64_t n,
                         int64_t e, const int32_t* out_rows, void* out,
                         int out_dtype, int normalise, int tap, void* ws,
                         size_t ws_bytes, void* stream) {
  clear_error();
  GFY_REQUIRE(enc != nullptr, GFY_ERR_INVALID, "gfy_encode: encoder is NULL");
  GFY_REQUIRE(n > 0 && n < INT32_MAX && e >= 0 && e < INT32_MAX, GFY_ERR_INVALID,
              "gfy_encode: n=%lld e=%lld out of range", (long long)n, (long long)e);
  GFY_REQUIRE(x && row_ptr && out && ws, GFY_ERR_INVALID, "gfy_encode: NULL argument");
  GFY_REQUIRE(e == 0 || (col && typ), GFY_ERR_INVALID, "gfy_encode: NULL CSR arrays");
  GFY_REQUIRE(out_dtype == GFY_F16 || out_dtype == GFY_F32 || out_dtype == GFY_F64,
              GFY_ERR_INVALID, "gfy_encode: unsupported out_dtype %d", out_dtype);
  GFY_REQUIRE(tap < 0 || tap <= enc->layers, GFY_ERR_INVALID,
              "gfy_encode_hidden: stage %d outside 0..%d", tap, enc->layers);
  if (const int rc = check_current_device(enc, "gfy_encode")) return rc;
  enc->last_stream = (hipStream_t)stream;
  if (enc->model_dtype == GFY_F16)
    return launch_encode_f16(enc, x, row_ptr, col, typ, n, e, out_rows, out,
                             out_dtype, normalise, tap, ws, ws_bytes,
                             (hipStream_t)stream);
  return launch_encode_f32(enc, x, row_ptr, col, typ, n, e, out_rows, out,
                           out_dtype, normalise, tap, ws, ws_bytes,
                           (hipStream_t)stream);
}

static int64_t padded_rows(int64_t n) { return (n + 31) / 32 * 32; }   // whole 32-row tiles

size_t gfy_encode_coo_clear_bytes(int64_t n) { return csr_clear_bytes(padded_rows(n < 1 ? 1 : n)); }

// fp32 model: plain sequence (CSR build, then encode), CSR arrays in front
static size_t encode_coo_f32_workspace_bytes(int64_t n, int64_t e) {
  return align_up(csr_workspace_bytes(n, e), 256) + align_up((size_t)(n + 1) * 4, 256) +
         align_up((size_t)(e > 0 ? e : 1) * 4, 256) + align_up((size_t)(e > 0 ? e : 1), 256) +
         encode_f32_workspace_bytes(n, e);
}

size_t gfy_encode_coo_workspace_bytes(const gfy_encoder* enc, int64_t n, int64_t e) {
  if (!enc) return 0;
  n = n < 1 ? 1 : n;
  e = e < 0 ? 0 : e;
  if (enc->model_dtype == GFY_F16) return encode_coo_f16_workspace_bytes(padded_rows(n), e);
  return encode_coo_f32_workspace_bytes(n, e);
}

int gfy_encode_coo_prepare(void* ws, size_t ws_bytes, int64_t n, void* stream) {
  clear_error();
  GFY_REQUIRE(ws && n > 0 && n < INT32_MAX - 64, GFY_ERR_INVALID,
              "gfy_encode_coo_prepare: bad arguments");
  GFY_REQUIRE(ws_bytes >= csr_clear_bytes(padded_rows(n)), GFY_ERR_WORKSPACE,
              "gfy_encode_coo_prepare: workspace %zu < %zu", ws_bytes,
              csr_clear_bytes(padded_rows(n)));
  return launch_csr_clear(ws, padded_rows(n), (hipStream_t)stream);
}

// fp32 model (parity path, MFMA-bound): CSR build and encode behind one entry point
static int encode_coo_f32(gfy_encoder* enc, const float* x, const int32_t* edge_index,
                          const uint8_t* edge_types, int64_t n, int64_t e,
                          const int32_t* out_rows, void* out, int out_dtype, int normalise,
                          void* ws, size_t ws_bytes, hipStream_t stream) {
  char* at = (char*)ws;
  void* csr_ws = at;
  const size_t csr_bytes = align_up(csr_workspace_bytes(n, e), 256);
  at += csr_bytes;
  int32_t* row_ptr = (int32_t*)at;
  at += align_up((size_t)(n + 1) * 4, 256);
  int32_t* col = (int32_t*)at;
  at += align_up((size_t)(e > 0 ? e : 1) * 4, 256);
  uint8_t* typ = (uint8_t*)at;
  at += align_up((size_t)(e > 0 ? e : 1), 256);
  if (const int rc = launch_build_csr(edge_index, edge_types, n, e, row_ptr, col, typ, csr_ws,
                                      csr_bytes, stream))
    return rc;
  return launch_encode_f32(enc, x, row_ptr, col, typ, n, e, out_rows, out, out_dtype, normalise,
                           -1, at, ws_bytes - (size_t)(at - (char*)ws), stream);
}

int gfy_encode_coo(gfy_encoder* enc, const float* x, const int32_t* edge_index,
                   const uint8_t* edge_types, int64_t n, int64_t e, const int32_t* out_rows,
                   void* out, int out_dtype, int normalise, void* ws, size_t ws_bytes,
                   void* stream) {
  clear_error();
  GFY_REQUIRE(enc != nullptr, GFY_ERR_INVALID, "gfy_encode_coo: encoder is NULL");
  GFY_REQUIRE(n > 0 && n < INT32_MAX - 64 && e >= 0 && e < INT32_MAX, GFY_ERR_INVALID,
              "gfy_encode_coo: n=%lld e=%lld out of range", (long long)n, (long long)e);
  GFY_REQUIRE(x && out && ws, GFY_ERR_INVALID, "gfy_encode_coo: NULL argument");
  GFY_REQUIRE(e == 0 || (edge_index && edge_types), GFY_ERR_INVALID,
              "gfy_encode_coo: NULL edge array with E=%lld", (long long)e);
  GFY_REQUIRE(out_dtype == GFY_F16 || out_dtype == GFY_F32 || out_dtype == GFY_F64,
              GFY_ERR_INVALID, "gfy_encode_coo: unsupported out_dtype %d", out_dtype);
  const size_t need = gfy_encode_coo_workspace_bytes(enc, n, e);
  GFY_REQUIRE(ws_bytes >= need, GFY_ERR_WORKSPACE, "gfy_encode_coo: workspace %zu < required %zu",
              ws_bytes, need);
  if (const int rc = check_current_device(enc, "gfy_encode_coo")) return rc;
  enc->last_stream = (hipStream_t)stream;
  if (enc->model_dtype == GFY_F16)
    return launch_encode_coo_f16(enc, single_shard(x, edge_index, edge_types, n, e, out_rows, out),
                                 nullptr, out_dtype, normalise, ws, ws_bytes, (hipStream_t)stream);
  return encode_coo_f32(enc, x, edge_index, edge_types, n, e, out_rows, out, out_dtype, normalise,
                        ws, ws_bytes, (hipStream_t)stream);
}

// ---- a batch of shards in one sequence of launches ---------------------------------------
static int batch_table(const gfy_shard* shards, int count, const char* who, ShardTable* table,
                       RecordTable* records = nullptr, bool* has_records = nullptr) {
  GFY_REQUIRE(shards && count >= 1 && count <= GFY_MAX_BATCH_SHARDS, GFY_ERR_INVALID,
              "%s: 1..%d shards per call (got %d)", who, GFY_MAX_BATCH_SHARDS, count);
  ShardTable t{};
  t.shards = count;
  int64_t tiles = 0, edges = 0, blocks = 0;
  for (int s = 0; s < count; ++s) {
    const gfy_shard& one = shards[s];
    GFY_REQUIRE(one.n_nodes > 0 && one.n_edges >= 0, GFY_ERR_INVALID,
                "%s: shard %d has n=%lld e=%lld", who, s, (long long)one.n_nodes,
                (long long)one.n_edges);
    GFY_REQUIRE(one.node_features && one.out, GFY_ERR_INVALID, "%s: shard %d: NULL argument", who,
                s);
    GFY_REQUIRE(one.n_edges == 0 || (one.edge_index && one.edge_types), GFY_ERR_INVALID,
                "%s: shard %d: NULL edge array with E=%lld", who, s, (long long)one.n_edges);
    t.tile_base[s] = (int)tiles;
    t.edge_base[s] = (int)edges;
    t.count_block_base[s] = (int)blocks;
    t.nodes[s] = (int)one.n_nodes;
    t.edges[s] = (int)one.n_edges;
    t.x[s] = one.node_features;
    t.edge_index[s] = one.edge_index;
    t.edge_types[s] = one.edge_types;
    t.out_rows[s] = one.out_rows;
    t.out[s] = one.out;
    tiles += (one.n_nodes + 31) / 32;
    edges += one.n_edges;
    blocks += (one.n_edges + kCsrCountEdgesPerBlock - 1) / kCsrCountEdgesPerBlock;
    GFY_REQUIRE(tiles * 32 < kCsrMaxRows && edges < INT32_MAX, GFY_ERR_UNSUPPORTED,
                "%s: a batch holds at most 16,777,215 (padded) nodes and 2^31 - 1 edges", who);
  }
  t.tile_base[count] = (int)tiles;
  t.edge_base[count] = (int)edges;
  t.count_block_base[count] = (int)blocks;
  *table = t;
  if (records && has_records) {   // record boundaries: all shards or none
    RecordTable r{};
    bool all = true;
    int ranges = 0;
    r.range_rows = t.total_rows() <= kRecLoneBatchRows    ? kRecRowsLone
                   : t.total_rows() <= kRecSmallBatchRows ? kRecRowsSmall
                                                          : kRecRowsLarge;
    for (int s = 0; s < count; ++s) {
      const gfy_shard& one = shards[s];
      const bool given = one.node_ptr && one.edge_ptr && one.n_records > 0;
      GFY_REQUIRE(given || (!one.node_ptr && !one.edge_ptr), GFY_ERR_INVALID,
                  "%s: shard %d: node_ptr, edge_ptr and n_records go together", who, s);
      GFY_REQUIRE(!given || one.n_records <= one.n_nodes, GFY_ERR_INVALID,
                  "%s: shard %d: %lld records for %lld nodes", who, s, (long long)one.n_records,
                  (long long)one.n_nodes);
      all = all && given;
      r.node_ptr[s] = one.node_ptr;
      r.edge_ptr[s] = one.edge_ptr;
      r.records[s] = (int)one.n_records;
      r.range_base[s] = ranges;
      ranges += (int)((one.n_nodes + r.range_rows - 1) / r.range_rows);
    }
    r.range_base[count] = ranges;
    *records = r;
    *has_records = all;
  }
  return GFY_OK;
}

size_t gfy_encode_coo_batch_workspace_bytes(const gfy_encoder* enc, const gfy_shard* shards,
                                            int count) {
  clear_error();
  ShardTable t;
  if (!enc || batch_table(shards, count, "gfy_encode_coo_batch_workspace_bytes", &t)) return 0;
  if (enc->model_dtype == GFY_F16)
    return encode_coo_f16_workspace_bytes(t.total_rows(), t.total_edges());
  size_t most = 0;   // fp32 model: the shards run one after the other in one workspace
  for (int s = 0; s < count; ++s) {
    const size_t one = encode_coo_f32_workspace_bytes(shards[s].n_nodes, shards[s].n_edges);
    most = one > most ? one : most;
  }
  return most;
}

size_t gfy_encode_coo_batch_clear_bytes(const gfy_shard* shards, int count) {
  clear_error();
  ShardTable t;
  if (batch_table(shards, count, "gfy_encode_coo_batch_clear_bytes", &t)) return 0;
  return csr_clear_bytes(t.total_rows());
}

int gfy_encode_coo_batch(gfy_encoder* enc, const gfy_shard* shards, int count, int out_dtype,
                         int normalise, void* ws, size_t ws_bytes, void* stream) {
  clear_error();
  GFY_REQUIRE(enc != nullptr && ws != nullptr, GFY_ERR_INVALID,
              "gfy_encode_coo_batch: NULL encoder or workspace");
  GFY_REQUIRE(out_dtype == GFY_F16 || out_dtype == GFY_F32 || out_dtype == GFY_F64,
              GFY_ERR_INVALID, "gfy_encode_coo_batch: unsupported out_dtype %d", out_dtype);
  ShardTable t;
  RecordTable records;
  bool has_records = false;
  if (const int rc = batch_table(shards, count, "gfy_encode_coo_batch", &t, &records, &has_records))
    return rc;
  const size_t need = gfy_encode_coo_batch_workspace_bytes(enc, shards, count);
  GFY_REQUIRE(ws_bytes >= need, GFY_ERR_WORKSPACE,
              "gfy_encode_coo_batch: workspace %zu < required %zu", ws_bytes, need);
  if (const int rc = check_current_device(enc, "gfy_encode_coo_batch")) return rc;
  enc->last_stream = (hipStream_t)stream;
  if (enc->model_dtype == GFY_F16)
    return launch_encode_coo_f16(enc, t, has_records ? &records : nullptr, out_dtype, normalise, ws,
                                 ws_bytes, (hipStream_t)stream);
  for (int s = 0; s < count; ++s)
    if (const int rc = encode_coo_f32(enc, shards[s].node_features, shards[s].edge_index,
                                      shards[s].edge_types, shards[s].n_nodes, shards[s].n_edges,
                                      shards[s].out_rows, shards[s].out, out_dtype, normalise, ws,
                                      ws_bytes, (hipStream_t)stream))
      return rc;
  return GFY_OK;
}

int gfy_encode(gfy_encoder* enc, const float* x, const int32_t* row_ptr,
               const int32_t* col, const uint8_t* typ, int64_t n, int64_t e,
               const int32_t* out_rows, void* out, int out_dtype, int normalise,
               void* ws, size_t ws_bytes, void* stream) {
  return encode_common(enc, x, row_ptr, col, typ, n, e, out_rows, out, out_dtype,
                       normalise, -1, ws, ws_bytes, stream);
}

int gfy_encode_hidden(gfy_encoder* enc, const float* x, const int32_t* row_ptr,
                      const int32_t* col, const uint8_t* typ, int64_t n,
                      int64_t e, int stage, void* out, void* ws, size_t ws_bytes,
                      void* stream) {
  GFY_REQUIRE(stage >= 0, GFY_ERR_INVALID, "gfy_encode_hidden: negative stage");
  return encode_common(enc, x, row_ptr, col, typ, n, e, nullptr, out,
                       enc ? enc->model_dtype : GFY_F16, 0, stage, ws, ws_bytes,
                       stream);
}

size_t gfy_debug_layer_workspace_bytes(const gfy_encoder* enc, int64_t n, int64_t /*e*/) {
  if (!enc || enc->model_dtype != GFY_F16) return 0;
  return debug_layer_f16_workspace_bytes(n < 1 ? 1 : n);
}

int gfy_debug_layer(gfy_encoder* enc, int layer, const void* hidden_in, const int32_t* row_ptr,
                    const int32_t* col, const uint8_t* typ, int64_t n, int64_t e, int tap,
                    void* out, void* ws, size_t ws_bytes, void* stream) {
  clear_error();
  GFY_REQUIRE(enc != nullptr, GFY_ERR_INVALID, "gfy_debug_layer: encoder is NULL");
  GFY_REQUIRE(enc->model_dtype == GFY_F16, GFY_ERR_UNSUPPORTED,
              "gfy_debug_layer: fp16 model only");
  GFY_REQUIRE(layer >= 0 && layer < enc->layers, GFY_ERR_INVALID,
              "gfy_debug_layer: layer %d outside 0..%d", layer, enc->layers - 1);
  GFY_REQUIRE(tap >= GFY_TAP_H && tap <= GFY_TAP_Y, GFY_ERR_INVALID,
              "gfy_debug_layer: unknown tap %d", tap);
  GFY_REQUIRE(n > 0 && n < INT32_MAX - 64 && e >= 0 && e < INT32_MAX, GFY_ERR_INVALID,
              "gfy_debug_layer: n=%lld e=%lld out of range", (long long)n, (long long)e);
  GFY_REQUIRE(hidden_in && row_ptr && out && ws && (e == 0 || (col && typ)), GFY_ERR_INVALID,
              "gfy_debug_layer: NULL argument");
  if (const int rc = check_current_device(enc, "gfy_debug_layer")) return rc;
  enc->last_stream = (hipStream_t)stream;
  return launch_debug_layer_f16(enc, layer, hidden_in, row_ptr, col, typ, n, e, tap, out, ws,
                                ws_bytes, (hipStream_t)stream);
}

int gfy_pairwise_dense(const void* a, int64_t n, const void* b, int64_t m,
                       int metric, float* out, void* ws, size_t ws_bytes,
                       void* stream) {
  clear_error();
  GFY_REQUIRE(a && b && out && ws && n > 0 && m > 0, GFY_ERR_INVALID,
              "gfy_pairwise_dense: bad arguments");
  GFY_REQUIRE(metric == GFY_L2 || metric == GFY_COSINE, GFY_ERR_INVALID,
              "gfy_pairwise_dense: unknown metric %d", metric);
  return launch_pairwise_dense(a, n, b, m, metric, out, ws, ws_bytes, (hipStream_t)stream);
}

size_t gfy_pairwise_workspace_bytes(int64_t n, int64_t m) {
  return pairwise_workspace_bytes(n < 1 ? 1 : n, m < 1 ? 1 : m);
}

int gfy_pairwise_nearest(const void* a, int64_t n, const void* b, int64_t m,
                         int metric, int64_t exclude_offset, float* best_val,
                         int32_t* best_idx, void* ws, size_t ws_bytes,
                         void* stream) {
  clear_error();
  GFY_REQUIRE(a && b && best_val && best_idx && ws && n > 0 && m > 0 &&
                  m < INT32_MAX,
              GFY_ERR_INVALID, "gfy_pairwise_nearest: bad arguments");
  GFY_REQUIRE(metric == GFY_L2 || metric == GFY_COSINE, GFY_ERR_INVALID,
              "gfy_pairwise_nearest: unknown metric %d", metric);
  return launch_pairwise_nearest(a, n, b, m, metric, exclude_offset, exclude_offset >= 0 ? 1 : 0,
                                 best_val, best_idx, ws, ws_bytes, (hipStream_t)stream);
}

int gfy_pairwise_nearest_window(const void* a, int64_t n, const void* b, int64_t m, int metric,
                                int64_t window_first, float* best_val, int32_t* best_idx,
                                void* ws, size_t ws_bytes, void* stream) {
  clear_error();
  GFY_REQUIRE(a && b && best_val && best_idx && ws && n > 0 && m > 0 && m < INT32_MAX,
              GFY_ERR_INVALID, "gfy_pairwise_nearest_window: bad arguments");
  GFY_REQUIRE(metric == GFY_L2 || metric == GFY_COSINE, GFY_ERR_INVALID,
              "gfy_pairwise_nearest_window: unknown metric %d", metric);
  GFY_REQUIRE(window_first >= 0 && window_first + m <= n, GFY_ERR_INVALID,
              "gfy_pairwise_nearest_window: b must be rows [%lld, %lld) of the %lld rows of a",
              (long long)window_first, (long long)(window_first + m), (long long)n);
  return launch_pairwise_nearest(a, n, b, m, metric, -window_first, 1, best_val, best_idx, ws,
                                 ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
