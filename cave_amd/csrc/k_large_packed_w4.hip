// k_large_packed_w4.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH_LARGE(launch_packed_large_w4, PackedParams, (cone_packed_large_kernel<CtxL, 2>), CtxL::NT)
}  // namespace cave
