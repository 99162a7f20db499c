// k_large_packed_w2.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH_LARGE(launch_packed_large_w2, PackedParams, (cone_packed_large_kernel<CtxL2, 2>), CtxL2::NT)
}  // namespace cave
