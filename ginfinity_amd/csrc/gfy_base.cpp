// What BOTH shared objects need and no HIP call is in: the per-thread error channel, the ABI
// version and the weight pack's size.  libgfy.so (HIP kernels + C ABI) links it, and so does
// libgfy_host.so = this file + gine_host.cpp, built with the host compiler alone: the
// reference's default device (Ginfinity.load(device="cpu"), src/ginfinity/api.py:64-76) then
// works on a box that has no ROCm runtime at all.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "../../include/gfy.h"

namespace gfy {

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
}
void clear_error() { g_error.clear(); }

// floats behind the 32-byte header of a weight pack (layout: ginfinity_amd/weights.py)
size_t pack_floats(uint32_t in_dim, uint32_t h, uint32_t layers, uint32_t edge_dim,
                   uint32_t out_dim) {
  const size_t per_layer = 1 + (size_t)h * edge_dim + h + (size_t)2 * h * h +
                           2 * h + 4 * (size_t)(2 * h) + (size_t)h * 2 * h + h +
                           2 * h;
  return (size_t)h * in_dim + h + layers * per_layer + (size_t)h * h + h +
         (size_t)out_dim * h + out_dim;
}

// Copies into page-locked staging memory (gfy_pack_microbatch).  The destination is written
// once and next read by the copy engine, never by this core: streaming stores leave it out of
// the cache and spare the read-for-ownership of every destination line (a third of the copy's
// memory traffic).  GFY_PACK_STREAM=0 keeps memcpy (A/B runs: tools/bench_host_feed.py).
static bool stream_stores() {   // read per call (once per ~4 MB): tests and A/B runs switch it
  const char* text = std::getenv("GFY_PACK_STREAM");
  return text == nullptr || text[0] != '0';
}

// to[i] = from[i] - shift for `count` 32-bit values; returns the OR of (value | (limit - value))
// over all of them: negative exactly when some value lies outside [0, limit]
static int32_t copy_rebased(int32_t* to, const int32_t* from, int64_t count, int32_t shift,
                            int32_t limit, bool stream) {
  int32_t seen = 0;
  int64_t i = 0;
#if defined(__SSE2__)
  if (stream) {
    for (; i < count && (reinterpret_cast<uintptr_t>(to + i) & 15) != 0; ++i) {
      const int32_t value = (int32_t)((uint32_t)from[i] - (uint32_t)shift);
      to[i] = value;
      seen |= value | (int32_t)((uint32_t)limit - (uint32_t)value);
    }
    const __m128i shifts = _mm_set1_epi32(shift), limits = _mm_set1_epi32(limit);
    __m128i any = _mm_setzero_si128();
    for (; i + 16 <= count; i += 16) {
      __m128i v[4];
      for (int k = 0; k < 4; ++k)
        v[k] = _mm_sub_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i*>(from + i + 4 * k)),
                             shifts);
      for (int k = 0; k < 4; ++k) {
        any = _mm_or_si128(any, _mm_or_si128(v[k], _mm_sub_epi32(limits, v[k])));
        _mm_stream_si128(reinterpret_cast<__m128i*>(to + i + 4 * k), v[k]);
      }
    }
    alignas(16) int32_t lanes[4];
    _mm_store_si128(reinterpret_cast<__m128i*>(lanes), any);
    seen |= lanes[0] | lanes[1] | lanes[2] | lanes[3];
  }
#endif
  for (; i < count; ++i) {
    const int32_t value = (int32_t)((uint32_t)from[i] - (uint32_t)shift);
    to[i] = value;
    seen |= value | (int32_t)((uint32_t)limit - (uint32_t)value);
  }
  return seen;
}

static void copy_bytes(void* to_, const void* from_, size_t bytes, bool stream) {
#if defined(__SSE2__)
  if (stream && bytes >= 4096) {
    char* to = static_cast<char*>(to_);
    const char* from = static_cast<const char*>(from_);
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(to) & 15)) & 15;
    memcpy(to, from, head);
    size_t at = head;
    for (; at + 64 <= bytes; at += 64) {
      __m128i v[4];
      for (int k = 0; k < 4; ++k)
        v[k] = _mm_loadu_si128(reinterpret_cast<const __m128i*>(from + at + 16 * k));
      for (int k = 0; k < 4; ++k)
        _mm_stream_si128(reinterpret_cast<__m128i*>(to + at + 16 * k), v[k]);
    }
    memcpy(to + at, from + at, bytes - at);
    return;
  }
#endif
  memcpy(to_, from_, bytes);
}

}  // namespace gfy

extern "C" {

const char* gfy_last_error(void) { return gfy::g_error.c_str(); }
int gfy_abi_version(void) { return GFY_ABI_VERSION; }

// One micro-batch of a host shard -> a staging block (include/gfy.h; the byte-for-byte twin of
// ginfinity_amd/api.py:_pack_microbatch_at + _Uploader.pack_at, which stay as the reference
// the tests compare it with).
int gfy_pack_microbatch(const float* node_features, int feature_dim, const int32_t* edge_index,
                        int64_t edges_total, const uint8_t* edge_types,
                        const uint8_t* node_roles, const int64_t* node_ptr,
                        const int64_t* edge_ptr, int64_t start, int64_t stop, int with_records,
                        void* slot, int64_t base, int64_t* offsets, int64_t* counts) {
  gfy::clear_error();
  if (!node_features || !node_ptr || !edge_ptr || !node_roles || !slot || !offsets || !counts ||
      feature_dim <= 0 || start < 0 || stop < start || base < 0 || (base & 255) != 0) {
    gfy::set_error("gfy_pack_microbatch: NULL argument, empty record range or unaligned base");
    return GFY_ERR_INVALID;
  }
  const int64_t n0 = node_ptr[start], n1 = node_ptr[stop], e0 = edge_ptr[start], e1 = edge_ptr[stop];
  const int64_t n = n1 - n0, e = e1 - e0;
  if (n < 0 || e < 0 || e0 < 0 || e1 > edges_total || (e > 0 && (!edge_index || !edge_types))) {
    gfy::set_error("gfy_pack_microbatch: records [%lld, %lld) have no valid node / edge range",
                   (long long)start, (long long)stop);
    return GFY_ERR_INVALID;
  }
  auto pad = [](int64_t bytes) { return (bytes + 255) / 256 * 256; };
  char* const out = static_cast<char*>(slot);
  const bool stream = gfy::stream_stores();
  int64_t at = base;
  // node rows
  offsets[0] = at;
  gfy::copy_bytes(out + at, node_features + n0 * feature_dim, (size_t)(n * feature_dim) * 4, stream);
  at += pad(n * feature_dim * 4);
  // edge_index, rebased; an index outside [0, n) in either row is the caller's error
  offsets[1] = at;
  int32_t seen = 0;
  for (int row = 0; row < 2; ++row) {
    const int32_t* from = edge_index + (int64_t)row * edges_total + e0;
    int32_t* to = reinterpret_cast<int32_t*>(out + at) + (int64_t)row * e;
    seen |= gfy::copy_rebased(to, from, e, (int32_t)n0, (int32_t)(n - 1), stream);
  }
  if (e > 0 && seen < 0) {
    gfy::set_error("edge index outside shard node range");
    return GFY_ERR_INVALID;
  }
  at += pad(2 * e * 4);
  offsets[2] = at;
  if (e > 0) gfy::copy_bytes(out + at, edge_types + e0, (size_t)e, stream);
  at += pad(e);
  // core rows: present only where a node of the range is not a core node (role != 0)
  bool any = false;
  for (int64_t i = n0; i < n1 && !any; ++i) any = node_roles[i] != 0;
  int64_t kept = n;
  offsets[3] = -1;
  if (any) {
    offsets[3] = at;
    int32_t* rows = reinterpret_cast<int32_t*>(out + at);
    int32_t next = 0;
    for (int64_t i = 0; i < n; ++i) rows[i] = node_roles[n0 + i] == 0 ? next++ : -1;
    kept = next;
    at += pad(n * 4);
  }
  // record boundaries, as they are (the set-up kernel reads them relative to their first entry)
  bool records = with_records != 0 && stop > start;
  for (int64_t r = start; r < stop && records; ++r)
    records = edge_ptr[r + 1] - edge_ptr[r] <= (int64_t)1 << 16;
  offsets[4] = offsets[5] = -1;
  if (records) {
    const int64_t bytes = (stop - start + 1) * 8;
    offsets[4] = at;
    memcpy(out + at, node_ptr + start, (size_t)bytes);
    at += pad(bytes);
    offsets[5] = at;
    memcpy(out + at, edge_ptr + start, (size_t)bytes);
    at += pad(bytes);
  }
  counts[0] = n, counts[1] = e, counts[2] = records ? stop - start : 0, counts[3] = kept;
#if defined(__SSE2__)
  if (stream) _mm_sfence();   // the streamed lines are in memory before the copy engine is told
#endif
  return GFY_OK;
}

size_t gfy_weight_pack_bytes(uint32_t in_dim, uint32_t hidden, uint32_t layers,
                             uint32_t edge_dim, uint32_t out_dim) {
  return 32 + 4 * gfy::pack_floats(in_dim, hidden, layers, edge_dim, out_dim);
}

}  // extern "C"
