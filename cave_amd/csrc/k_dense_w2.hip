// k_dense_w2.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_dense_w2, DenseParams, cone_dense_kernel<Ctx2>, Ctx2::NT)
}  // namespace cave
