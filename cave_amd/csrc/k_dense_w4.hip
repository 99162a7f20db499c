// k_dense_w4.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_dense_w4, DenseParams, cone_dense_kernel<Ctx4>, Ctx4::NT)
}  // namespace cave
