// all-pairs distance — placeholder until the MFMA kernel lands.
#include "gfy_common.h"
namespace gfy {
size_t pairwise_workspace_bytes(int64_t, int64_t) { return 256; }
int launch_pairwise_dense(const void*, int64_t, const void*, int64_t, int, float*, hipStream_t) {
  set_error("gfy_pairwise_dense: not built yet");
  return GFY_ERR_UNSUPPORTED;
}
int launch_pairwise_nearest(const void*, int64_t, const void*, int64_t, int, int64_t, float*,
                            int32_t*, void*, size_t, hipStream_t) {
  set_error("gfy_pairwise_nearest: not built yet");
  return GFY_ERR_UNSUPPORTED;
}
}  // namespace gfy
