// k_large_pack.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH_LARGE(launch_pack_large, PackParams, cone_pack_large_kernel<CtxL>, CtxL::NT)
}  // namespace cave
