// k_pack_w2.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_pack_w2, PackParams, cone_pack_kernel<Ctx2>, Ctx2::NT)
}  // namespace cave
