// k_dense_w1.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_dense_w1, DenseParams, cone_dense_kernel<Ctx1>, Ctx1::NT)
}  // namespace cave
