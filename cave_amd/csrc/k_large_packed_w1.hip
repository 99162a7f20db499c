// k_large_packed_w1.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH_LARGE(launch_packed_large_w1, PackedParams, (cone_packed_large_kernel<Ctx1, 2>), Ctx1::NT)
}  // namespace cave
