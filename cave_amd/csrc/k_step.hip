// k_step.hip — one kernel shape and its launch function (see kernels.h)
#include "kernels.h"

namespace cave {
using CtxStep = BlockCtx<2, true>;  // pack half: two waves per instance, 256-register budget
CAVE_DEFINE_LAUNCH(launch_step, StepParams, cone_step_kernel<CtxStep>, CtxStep::NT)
CAVE_DEFINE_LAUNCH(launch_lite_from_packed, LiteFromPackedParams, lite_from_packed_kernel<Ctx2>, Ctx2::NT)
}  // namespace cave
