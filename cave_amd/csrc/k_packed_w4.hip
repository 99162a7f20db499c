// k_packed_w4.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_packed_w4, PackedParams, cone_packed_kernel<Ctx4>, Ctx4::NT)
}  // namespace cave
