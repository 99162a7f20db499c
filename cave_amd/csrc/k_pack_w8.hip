// k_pack_w8.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_pack_w8, PackParams, cone_pack_kernel<CtxW>, CtxW::NT)
}  // namespace cave
