// k_step.hip — one kernel shape and its launch function (see kernels.h)
// Wave priorities inside the fused launch (measured, one box, TSP-20 B = 1024 + 1024; tools/diag/build_variant.sh):
// pack waves at priority 1 from their start (without: 132 us per step -- the second round of pack workgroups is the
// tail), and a solve still iterating after five Newton rounds -- by then the tail of the launch: 12 % of the instances
// -- at priority 3 from there on: 118.9 us (from round 4: 119.9, from round 6: 120.4, never: 121.4).
#ifndef CAVE_LITE_TAIL_PRIO_IT
#define CAVE_LITE_TAIL_PRIO_IT 5
#endif
#include "kernels.h"

namespace cave {
using CtxStep = BlockCtx<2, true>;  // pack half: two waves per instance, 256-register budget
CAVE_DEFINE_LAUNCH(launch_step, StepParams, cone_step_kernel<CtxStep>, CtxStep::NT)
CAVE_DEFINE_LAUNCH(launch_lite_from_packed, LiteFromPackedParams, lite_from_packed_kernel<Ctx2>, Ctx2::NT)
}  // namespace cave
