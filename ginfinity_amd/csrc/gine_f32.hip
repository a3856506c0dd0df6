// full_precision (fp32) encode path — placeholder until the fp64-accumulate kernels land.
#include "gfy_common.h"
namespace gfy {
size_t encode_f32_workspace_bytes(int64_t n, int64_t) { return 2 * align_up((size_t)n * kHidden * 4, 256); }
int launch_encode_f32(const gfy_encoder*, const float*, const int32_t*, const int32_t*,
                      const uint8_t*, int64_t, int64_t, const int32_t*, void*, int, int, int,
                      void*, size_t, hipStream_t) {
  set_error("gfy_encode: fp32 model path not built yet");
  return GFY_ERR_UNSUPPORTED;
}
}  // namespace gfy
