// k_pack_w4.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_pack_w4, PackParams, cone_pack_kernel<Ctx4>, Ctx4::NT)
}  // namespace cave
