// placeholder until the tiled kernel lands
#include "escoin_plan.h"
namespace escoin {
bool tiled_supported(const Geometry &) { return false; }
int tiled_build(escoin_plan *, hipStream_t) { return fail(ESCOIN_EINVAL, "tiled kernel not built"); }
int launch_tiled(const escoin_plan *, const float *, const float *, float *, int, hipStream_t) {
  return fail(ESCOIN_EINVAL, "tiled kernel not built");
}
const char *tiled_kernel_name(const escoin_plan *) { return "escoin_sconv_tiled_kernel"; }
}  // namespace escoin
