// k_large_dense.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
CAVE_DEFINE_LAUNCH_LARGE(launch_dense_large, DenseParams, cone_dense_large_kernel<CtxL>, CtxL::NT)
}  // namespace cave
