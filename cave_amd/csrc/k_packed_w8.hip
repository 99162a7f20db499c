// k_packed_w8.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_packed_w8, PackedParams, cone_packed_kernel<CtxW>, CtxW::NT)
}  // namespace cave
