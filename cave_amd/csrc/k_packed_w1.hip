// k_packed_w1.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_packed_w1, PackedParams, cone_packed_kernel<Ctx1>, Ctx1::NT)
}  // namespace cave
