// What BOTH shared objects need and no HIP call is in: the per-thread error channel, the ABI
// version and the weight pack's size.  libgfy.so (HIP kernels + C ABI) links it, and so does
// libgfy_host.so = this file + gine_host.cpp, built with the host compiler alone: the
// reference's default device (Ginfinity.load(device="cpu"), src/ginfinity/api.py:64-76) then
// works on a box that has no ROCm runtime at all.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

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
  int64_t at = base;
  // node rows
  offsets[0] = at;
  memcpy(out + at, node_features + n0 * feature_dim, (size_t)(n * feature_dim) * 4);
  at += pad(n * feature_dim * 4);
  // edge_index, rebased; an index outside [0, n) in either row is the caller's error
  offsets[1] = at;
  uint32_t largest = 0;
  for (int row = 0; row < 2; ++row) {
    const int32_t* from = edge_index + (int64_t)row * edges_total + e0;
    int32_t* to = reinterpret_cast<int32_t*>(out + at) + (int64_t)row * e;
    const int32_t shift = (int32_t)n0;
    for (int64_t i = 0; i < e; ++i) {
      const int32_t value = from[i] - shift;
      to[i] = value;
      largest = (uint32_t)value > largest ? (uint32_t)value : largest;
    }
  }
  if (e > 0 && (int64_t)largest >= n) {
    gfy::set_error("edge index outside shard node range");
    return GFY_ERR_INVALID;
  }
  at += pad(2 * e * 4);
  offsets[2] = at;
  if (e > 0) memcpy(out + at, edge_types + e0, (size_t)e);
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
  return GFY_OK;
}

size_t gfy_weight_pack_bytes(uint32_t in_dim, uint32_t hidden, uint32_t layers,
                             uint32_t edge_dim, uint32_t out_dim) {
  return 32 + 4 * gfy::pack_floats(in_dim, hidden, layers, edge_dim, out_dim);
}

}  // extern "C"
