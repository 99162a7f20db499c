// k_pack_w1.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_pack_w1, PackParams, cone_pack_kernel<Ctx1>, Ctx1::NT)
}  // namespace cave
