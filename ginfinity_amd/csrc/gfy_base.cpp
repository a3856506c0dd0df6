// What BOTH shared objects need and no HIP call is in: the per-thread error channel, the ABI
// version and the weight pack's size.  libgfy.so (HIP kernels + C ABI) links it, and so does
// libgfy_host.so = this file + gine_host.cpp, built with the host compiler alone: the
// reference's default device (Ginfinity.load(device="cpu"), src/ginfinity/api.py:64-76) then
// works on a box that has no ROCm runtime at all.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
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

size_t gfy_weight_pack_bytes(uint32_t in_dim, uint32_t hidden, uint32_t layers,
                             uint32_t edge_dim, uint32_t out_dim) {
  return 32 + 4 * gfy::pack_floats(in_dim, hidden, layers, edge_dim, out_dim);
}

}  // extern "C"
