// k_dense_w8.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH(launch_dense_w8, DenseParams, cone_dense_kernel<CtxW>, CtxW::NT)
}  // namespace cave
