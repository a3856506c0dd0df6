// Host (CPU) implementation of the encode path: the reference's DEFAULT device
// (src/ginfinity/api.py:64-76: Ginfinity.load(device="cpu")), so that the drop-in surface
// behaves like the reference on a box without a GPU (BASELINE configs[0]: the 8-nt README
// example).  Plain C++, no HIP call, threads over node blocks.
//
// It is NOT a fallback of the GPU path: a gfy_encoder never routes here, and nothing here is
// taken from oracle/ (test infrastructure).  It restates, from the reference's op sequence, the
// same rounding-point model the kernels implement:
//   _model.py:67      h  = R(R(x) Win^T + b)
//   _model.py:41-45   m  = relu(R(h[src] + T[type])),  agg = R(fp32 sum in edge order)
//   _model.py:46      z  = R(R(s h) + agg),  u = R(z W0^T + b0),  v = relu(R(BN(u))),
//                     w  = R(v W1^T + b1)
//   _model.py:69-71   y  = R(LayerNorm(w)),  h = R(h + y)
//   _model.py:72      o  = R(relu(R(h Wa^T + ba)) Wb^T + bb)
//   api.py:250-259    float64 normalise, one rounding to the output dtype
// with R = round-to-nearest-even to fp16 and fp32 arithmetic inside an op (dot products:
// fp32 accumulation in ascending k).  full_precision: fp32 activations, dot products
// accumulated in float64 and rounded once per op, as the fp32 kernels do.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/gfy.h"

namespace gfy {
void set_error(const char* fmt, ...);
void clear_error();
}   // namespace gfy

namespace {

constexpr int kH = 128, kM = 256, kIn = 7, kOut = 128, kMaxTypes = 16, kMaxLayers = 8;

// ---- fp16 <-> fp32 / fp64 by bit manipulation (round-to-nearest-even; no F16C needed) -------
inline float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {   // subnormal: value = man * 2^-24
      float f = (float)man * 5.9604644775390625e-08f;
      std::memcpy(&bits, &f, 4);
      bits |= sign;
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float out;
  std::memcpy(&out, &bits, 4);
  return out;
}

// double -> fp16 bits with ONE rounding (numpy's astype(float16) from float64)
inline uint16_t double_to_half_bits(double value) {
  uint64_t bits;
  std::memcpy(&bits, &value, 8);
  const uint16_t sign = (uint16_t)((bits >> 48) & 0x8000u);
  const int exp = (int)((bits >> 52) & 0x7FF);
  const uint64_t man = bits & 0xFFFFFFFFFFFFFull;
  if (exp == 0x7FF) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0u));
  const int e = exp - 1023;   // unbiased
  if (e > 15) return (uint16_t)(sign | 0x7C00u);
  if (e >= -14) {   // normal half: keep 10 mantissa bits, round on the 42 dropped ones
    uint32_t out = (uint32_t)((e + 15) << 10) | (uint32_t)(man >> 42);
    const uint64_t rest = man & ((1ull << 42) - 1), halfway = 1ull << 41;
    if (rest > halfway || (rest == halfway && (out & 1u))) ++out;   // may carry into the exponent
    return (uint16_t)(sign | out);
  }
  if (e < -25) return sign;   // below half of the smallest subnormal (2^-25 itself: tie -> 0 or odd)
  // subnormal half: value = m * 2^-24, m in [0, 1024)
  const uint64_t full = man | (1ull << 52);         // 53-bit significand, value = full * 2^(e-52)
  const int shift = 52 - (e + 24);                  // bits dropped: m = full >> shift
  if (shift > 63) return sign;
  uint32_t out = (uint32_t)(full >> shift);
  const uint64_t rest = full & ((1ull << shift) - 1), halfway = 1ull << (shift - 1);
  if (rest > halfway || (rest == halfway && (out & 1u))) ++out;
  return (uint16_t)(sign | out);
}
inline uint16_t float_to_half_bits(float value) { return double_to_half_bits((double)value); }
// R(.): fp32 -> fp16 -> fp32
inline float R(float value) { return half_bits_to_float(float_to_half_bits(value)); }

struct HostLayer {
  float scale;                        // fp16 model: R(1 + R(eps)); fp32 model: 1 + eps
  std::vector<float> table;           // [types][128]
  std::vector<float> w0t, b0;         // [128][256] (k-major), [256]
  std::vector<float> alpha, shift;    // BatchNorm eval affine
  std::vector<float> w1t, b1;         // [256][128], [128]
  std::vector<float> gamma, beta;
};

}   // namespace

struct gfy_host_encoder {
  int model_dtype = GFY_F16, layers = 0, edge_dim = 0, residual = 1;
  std::vector<float> w_in_t, b_in;    // [7][128], [128]
  HostLayer layer[kMaxLayers];
  std::vector<float> wa_t, ba, wb_t, bb;
};

namespace {

struct PackHeader {
  uint32_t magic, version, in_dim, hidden, layers, edge_dim, out_dim, flags;
};

std::vector<float> transposed(const float* w, int rows, int cols, bool half) {
  std::vector<float> out((size_t)rows * cols);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const float value = w[(size_t)r * cols + c];
      out[(size_t)c * rows + r] = half ? R(value) : value;
    }
  return out;
}
std::vector<float> copied(const float* w, int count, bool half) {
  std::vector<float> out(count);
  for (int i = 0; i < count; ++i) out[i] = half ? R(w[i]) : w[i];
  return out;
}

// out[c] (+)= sum_k a[k] * wt[k][c], fp32 accumulation in ascending k, vectorisable over c
template <int K, int C>
inline void dot_f32(const float* a, const float* wt, float* acc) {
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  for (int k = 0; k < K; ++k) {
    const float ak = a[k];
    const float* row = wt + (size_t)k * C;
    for (int c = 0; c < C; ++c) acc[c] = acc[c] + ak * row[c];
  }
}
template <int K, int C>
inline void dot_f64(const float* a, const float* wt, double* acc) {
  for (int c = 0; c < C; ++c) acc[c] = 0.0;
  for (int k = 0; k < K; ++k) {
    const double ak = a[k];
    const float* row = wt + (size_t)k * C;
    for (int c = 0; c < C; ++c) acc[c] = acc[c] + ak * (double)row[c];
  }
}

template <typename F>
void parallel_nodes(int64_t n, int threads, F&& body) {
  const int64_t block = 64;
  const int64_t blocks = (n + block - 1) / block;
  int workers = (int)std::min<int64_t>(std::max(threads, 1), blocks);
  if (workers <= 1) {
    body(0, n);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t per = (blocks + workers - 1) / workers * block;
  for (int w = 0; w < workers; ++w) {
    const int64_t lo = w * per, hi = std::min<int64_t>(n, lo + per);
    if (lo >= hi) break;
    pool.emplace_back([&body, lo, hi] { body(lo, hi); });
  }
  for (auto& t : pool) t.join();
}

// LayerNorm of one row: exact (float64) moments rounded to fp32, then the fp32 expression
// fma(fma(x, rstd, -rstd*mean), gamma, beta) evaluated in float64 and rounded once per fma
inline void layer_norm_row(const float* w, const float* gamma, const float* beta, float* y) {
  double sum = 0.0;
  for (int c = 0; c < kH; ++c) sum += w[c];
  const double mean64 = sum / kH;
  double sq = 0.0;
  for (int c = 0; c < kH; ++c) sq += ((double)w[c] - mean64) * ((double)w[c] - mean64);
  const float var = (float)(sq / kH), mean = (float)mean64;
  const float rstd = 1.0f / std::sqrt(var + 1e-5f);
  const float offset = -rstd * mean;
  for (int c = 0; c < kH; ++c) {
    const float inner = (float)((double)w[c] * (double)rstd + (double)offset);
    y[c] = (float)((double)inner * (double)gamma[c] + (double)beta[c]);
  }
}

}   // namespace

extern "C" {

int gfy_host_encoder_create(const void* weight_pack_host, size_t bytes, int model_dtype,
                            gfy_host_encoder** out) {
  gfy::clear_error();
  if (!out) {
    gfy::set_error("gfy_host_encoder_create: out is NULL");
    return GFY_ERR_INVALID;
  }
  *out = nullptr;
  PackHeader hd;
  if (!weight_pack_host || bytes < sizeof hd) {
    gfy::set_error("gfy_host_encoder_create: weight pack missing or truncated");
    return GFY_ERR_INVALID;
  }
  std::memcpy(&hd, weight_pack_host, sizeof hd);
  if (hd.magic != 0x31594647u || hd.version != 1) {
    gfy::set_error("gfy_host_encoder_create: bad weight-pack magic/version");
    return GFY_ERR_INVALID;
  }
  if (hd.hidden != (uint32_t)kH || hd.in_dim != (uint32_t)kIn || hd.out_dim != (uint32_t)kOut ||
      hd.layers < 1 || hd.layers > (uint32_t)kMaxLayers || hd.edge_dim < 1 ||
      hd.edge_dim > (uint32_t)kMaxTypes) {
    gfy::set_error("gfy_host_encoder_create: built for in_dim=7 hidden=128 out_dim=128, "
                   "1..8 layers, 1..16 edge types");
    return GFY_ERR_UNSUPPORTED;
  }
  if (bytes != gfy_weight_pack_bytes(hd.in_dim, hd.hidden, hd.layers, hd.edge_dim, hd.out_dim)) {
    gfy::set_error("gfy_host_encoder_create: weight pack is %zu bytes, expected %zu", bytes,
                   gfy_weight_pack_bytes(hd.in_dim, hd.hidden, hd.layers, hd.edge_dim, hd.out_dim));
    return GFY_ERR_INVALID;
  }
  if (model_dtype != GFY_F16 && model_dtype != GFY_F32) {
    gfy::set_error("gfy_host_encoder_create: model_dtype must be GFY_F16 or GFY_F32");
    return GFY_ERR_INVALID;
  }
  const bool half = model_dtype == GFY_F16;   // model.half(): parameters AND BatchNorm buffers
  const float* p = reinterpret_cast<const float*>((const char*)weight_pack_host + sizeof hd);
  auto take = [&](size_t count) {
    const float* r = p;
    p += count;
    return r;
  };
  gfy_host_encoder* enc = new gfy_host_encoder();
  enc->model_dtype = model_dtype;
  enc->layers = (int)hd.layers;
  enc->edge_dim = (int)hd.edge_dim;
  enc->residual = (int)(hd.flags & 1u);
  enc->w_in_t = transposed(take((size_t)kH * kIn), kH, kIn, half);
  enc->b_in = copied(take(kH), kH, half);
  const int ED = enc->edge_dim;
  for (int l = 0; l < enc->layers; ++l) {
    HostLayer& L = enc->layer[l];
    const float* eps = take(1);
    const float* ew = take((size_t)kH * ED);
    const float* eb = take(kH);
    const float* w0 = take((size_t)kM * kH);
    const float* b0 = take(kM);
    const float* bg = take(kM);
    const float* bb = take(kM);
    const float* bm = take(kM);
    const float* bv = take(kM);
    const float* w1 = take((size_t)kH * kM);
    const float* b1 = take(kH);
    const float* lg = take(kH);
    const float* lb = take(kH);
    L.scale = half ? R(1.0f + R(eps[0])) : 1.0f + eps[0];
    L.table.assign((size_t)kMaxTypes * kH, 0.f);
    for (int t = 0; t < ED; ++t)
      for (int c = 0; c < kH; ++c)   // edge_lin on a one-hot row = one weight column + bias
        L.table[(size_t)t * kH + c] =
            half ? R(R(ew[(size_t)c * ED + t]) + R(eb[c])) : ew[(size_t)c * ED + t] + eb[c];
    L.w0t = transposed(w0, kM, kH, half);
    L.b0 = copied(b0, kM, half);
    L.alpha.resize(kM);
    L.shift.resize(kM);
    for (int c = 0; c < kM; ++c) {   // BatchNorm1d eval as torch's CPU kernel evaluates it
      const float g = half ? R(bg[c]) : bg[c], b = half ? R(bb[c]) : bb[c];
      const float mean = half ? R(bm[c]) : bm[c], var = half ? R(bv[c]) : bv[c];
      const float invstd = 1.0f / std::sqrt(var + 1e-5f);
      const float alpha = invstd * g;
      const float prod = mean * alpha;
      L.alpha[c] = alpha;
      L.shift[c] = b - prod;
    }
    L.w1t = transposed(w1, kH, kM, half);
    L.b1 = copied(b1, kH, half);
    L.gamma = copied(lg, kH, half);
    L.beta = copied(lb, kH, half);
  }
  enc->wa_t = transposed(take((size_t)kH * kH), kH, kH, half);
  enc->ba = copied(take(kH), kH, half);
  enc->wb_t = transposed(take((size_t)kOut * kH), kOut, kH, half);
  enc->bb = copied(take(kOut), kOut, half);
  *out = enc;
  return GFY_OK;
}

void gfy_host_encoder_destroy(gfy_host_encoder* encoder) { delete encoder; }

static int host_encode(const gfy_host_encoder* enc, const float* x, const int32_t* edge_index,
                       const uint8_t* edge_types, int64_t n, int64_t e, const int32_t* out_rows,
                       void* out, int out_dtype, int normalise, int threads) {
  gfy::clear_error();
  if (!enc || !x || !out || n <= 0 || n >= INT32_MAX || e < 0 || e >= INT32_MAX ||
      (e > 0 && (!edge_index || !edge_types))) {
    gfy::set_error("gfy_host_encode: bad arguments (n=%lld e=%lld)", (long long)n, (long long)e);
    return GFY_ERR_INVALID;
  }
  if (out_dtype != GFY_F16 && out_dtype != GFY_F32 && out_dtype != GFY_F64) {
    gfy::set_error("gfy_host_encode: unsupported out_dtype %d", out_dtype);
    return GFY_ERR_INVALID;
  }
  const bool half = enc->model_dtype == GFY_F16;
  const int32_t* src = edge_index;
  const int32_t* dst = edge_index + e;
  // destination-major CSR, edges of one row in COO order (stable counting sort)
  std::vector<int32_t> row_ptr((size_t)n + 1, 0), order((size_t)e);
  for (int64_t i = 0; i < e; ++i) {
    if ((uint32_t)dst[i] >= (uint64_t)n || (uint32_t)src[i] >= (uint64_t)n ||
        edge_types[i] >= enc->edge_dim) {
      gfy::set_error("gfy_host_encode: edge %lld (%d -> %d, type %d) outside the shard",
                     (long long)i, src[i], dst[i], (int)edge_types[i]);
      return GFY_ERR_INVALID;
    }
    ++row_ptr[(size_t)dst[i] + 1];
  }
  for (int64_t i = 0; i < n; ++i) row_ptr[i + 1] += row_ptr[i];
  {
    std::vector<int32_t> fill(row_ptr.begin(), row_ptr.end() - 1);
    for (int64_t i = 0; i < e; ++i) order[fill[dst[i]]++] = (int32_t)i;
  }
  auto rnd = [half](float v) { return half ? R(v) : v; };

  std::vector<float> h((size_t)n * kH), next((size_t)n * kH);
  parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {   // input Linear (_model.py:67)
    for (int64_t i = lo; i < hi; ++i) {
      float xv[kIn];
      for (int k = 0; k < kIn; ++k) xv[k] = rnd(x[i * kIn + k]);   // api.py:237-238
      if (half) {
        float acc[kH];
        dot_f32<kIn, kH>(xv, enc->w_in_t.data(), acc);
        for (int c = 0; c < kH; ++c) h[i * kH + c] = R(acc[c] + enc->b_in[c]);
      } else {
        double acc[kH];
        dot_f64<kIn, kH>(xv, enc->w_in_t.data(), acc);
        for (int c = 0; c < kH; ++c) h[i * kH + c] = (float)(acc[c] + (double)enc->b_in[c]);
      }
    }
  });
  for (int l = 0; l < enc->layers; ++l) {
    const HostLayer& L = enc->layer[l];
    parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
      float z[kH], u[kM], w[kH], y[kH];
      for (int64_t i = lo; i < hi; ++i) {
        // message + aggregate (_model.py:41-45): fp32 sum in edge order, one rounding
        if (half) {
          float agg[kH];
          for (int c = 0; c < kH; ++c) agg[c] = 0.f;
          for (int32_t q = row_ptr[i]; q < row_ptr[i + 1]; ++q) {
            const int32_t id = order[q];
            const float* hs = &h[(size_t)src[id] * kH];
            const float* tt = &L.table[(size_t)edge_types[id] * kH];
            for (int c = 0; c < kH; ++c) {
              const float m = R(hs[c] + tt[c]);
              agg[c] = agg[c] + (m > 0.f ? m : 0.f);
            }
          }
          for (int c = 0; c < kH; ++c) z[c] = R(R(L.scale * h[i * kH + c]) + R(agg[c]));
        } else {
          double agg[kH];
          for (int c = 0; c < kH; ++c) agg[c] = 0.0;
          for (int32_t q = row_ptr[i]; q < row_ptr[i + 1]; ++q) {
            const int32_t id = order[q];
            const float* hs = &h[(size_t)src[id] * kH];
            const float* tt = &L.table[(size_t)edge_types[id] * kH];
            for (int c = 0; c < kH; ++c) {
              const float m = hs[c] + tt[c];
              agg[c] += (double)(m > 0.f ? m : 0.f);
            }
          }
          for (int c = 0; c < kH; ++c) z[c] = L.scale * h[i * kH + c] + (float)agg[c];
        }
        // update MLP (_model.py:34-36,46)
        if (half) {
          float acc[kM];
          dot_f32<kH, kM>(z, L.w0t.data(), acc);
          for (int c = 0; c < kM; ++c) {
            const float uc = R(acc[c] + L.b0[c]);
            const float bn = R((float)((double)uc * (double)L.alpha[c] + (double)L.shift[c]));
            u[c] = bn > 0.f ? bn : 0.f;
          }
          float acc1[kH];
          dot_f32<kM, kH>(u, L.w1t.data(), acc1);
          for (int c = 0; c < kH; ++c) w[c] = R(acc1[c] + L.b1[c]);
        } else {
          double acc[kM];
          dot_f64<kH, kM>(z, L.w0t.data(), acc);
          for (int c = 0; c < kM; ++c) {
            const float uc = (float)(acc[c] + (double)L.b0[c]);
            const float bn = (float)((double)uc * (double)L.alpha[c] + (double)L.shift[c]);
            u[c] = bn > 0.f ? bn : 0.f;
          }
          double acc1[kH];
          dot_f64<kM, kH>(u, L.w1t.data(), acc1);
          for (int c = 0; c < kH; ++c) w[c] = (float)(acc1[c] + (double)L.b1[c]);
        }
        // LayerNorm + residual (_model.py:69-71)
        layer_norm_row(w, L.gamma.data(), L.beta.data(), y);
        for (int c = 0; c < kH; ++c) {
          const float yc = rnd(y[c]);
          next[i * kH + c] = enc->residual ? rnd(h[i * kH + c] + yc) : yc;
        }
      }
    });
    h.swap(next);
  }
  // head (_model.py:72) + float64 normalise, one rounding (api.py:250-259)
  parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
    float t[kH], o[kOut];
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t dest = out_rows ? out_rows[i] : i;
      if (dest < 0) continue;
      if (half) {
        float acc[kH];
        dot_f32<kH, kH>(&h[i * kH], enc->wa_t.data(), acc);
        for (int c = 0; c < kH; ++c) {
          const float v = R(acc[c] + enc->ba[c]);
          t[c] = v > 0.f ? v : 0.f;
        }
        dot_f32<kH, kOut>(t, enc->wb_t.data(), acc);
        for (int c = 0; c < kOut; ++c) o[c] = R(acc[c] + enc->bb[c]);
      } else {
        double acc[kH];
        dot_f64<kH, kH>(&h[i * kH], enc->wa_t.data(), acc);
        for (int c = 0; c < kH; ++c) {
          const float v = (float)(acc[c] + (double)enc->ba[c]);
          t[c] = v > 0.f ? v : 0.f;
        }
        dot_f64<kH, kOut>(t, enc->wb_t.data(), acc);
        for (int c = 0; c < kOut; ++c) o[c] = (float)(acc[c] + (double)enc->bb[c]);
      }
      double scale = 1.0;
      if (normalise) {
        double ss = 0.0;
        for (int c = 0; c < kOut; ++c) ss += (double)o[c] * (double)o[c];
        const double norm = std::sqrt(ss);
        scale = norm > 1e-12 ? norm : 1e-12;
      }
      for (int c = 0; c < kOut; ++c) {
        const double value = normalise ? (double)o[c] / scale : (double)o[c];
        const size_t at = (size_t)dest * kOut + c;
        if (out_dtype == GFY_F16) ((uint16_t*)out)[at] = double_to_half_bits(value);
        else if (out_dtype == GFY_F32) ((float*)out)[at] = (float)value;
        else ((double*)out)[at] = value;
      }
    }
  });
  return GFY_OK;
}

// Nothing may unwind through the C ABI into the caller's interpreter: the buffers are
// std::vectors of ~1 KB per node and the node blocks run on std::threads (bad_alloc,
// system_error) — reported as a status, like every other failure.
int gfy_host_encode(const gfy_host_encoder* enc, const float* x, const int32_t* edge_index,
                    const uint8_t* edge_types, int64_t n, int64_t e, const int32_t* out_rows,
                    void* out, int out_dtype, int normalise, int threads) {
  try {
    return host_encode(enc, x, edge_index, edge_types, n, e, out_rows, out, out_dtype, normalise,
                       threads);
  } catch (const std::bad_alloc&) {
    gfy::set_error("gfy_host_encode: out of host memory for n=%lld e=%lld", (long long)n,
                   (long long)e);
    return GFY_ERR_WORKSPACE;
  } catch (const std::exception& failure) {
    gfy::set_error("gfy_host_encode: %s", failure.what());
    return GFY_ERR_INVALID;
  } catch (...) {
    gfy::set_error("gfy_host_encode: unknown failure");
    return GFY_ERR_INVALID;
  }
}

}   // extern "C"
