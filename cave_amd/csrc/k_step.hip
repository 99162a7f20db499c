// k_step.hip — one kernel shape and its launch function (see kernels.h)
// Wave priorities inside the fused launch (measured, one box per line, TSP-20 B = 1024 + 1024; tools/diag/build_variant.sh):
// pack waves above the solve waves from their start (without: 132 us per step -- the second round of pack workgroups is
// the tail), and a solve still iterating after a few Newton rounds -- by then the tail of the launch -- at priority 3
// from there on.  Before the solver's exchange loop and its one-round-trip prologue: never 121.4, from round 4 119.9, from
// round 5 118.9, from round 6 120.4.  With them (the solves are 10 % shorter, fewer of them are still running when the
// pack half ends): from round 3 123.5, 4 121.0, 5 117.9, 6 115.9 - 116.4, 7 117.0, never 116.6; pack waves at priority 2
// instead of 1: 116.0 with round 6.
#ifndef CAVE_LITE_TAIL_PRIO_IT
#define CAVE_LITE_TAIL_PRIO_IT 6
#endif
#ifndef CAVE_STEP_PACK_PRIO
#define CAVE_STEP_PACK_PRIO 2
#endif
#include "kernels.h"

namespace cave {
using CtxStep = BlockCtx<2, true>;  // pack half: two waves per instance, 256-register budget
CAVE_DEFINE_LAUNCH(launch_step, StepParams, cone_step_kernel<CtxStep>, CtxStep::NT)
CAVE_DEFINE_LAUNCH(launch_lite_from_packed, LiteFromPackedParams, lite_from_packed_kernel<Ctx2>, Ctx2::NT)
}  // namespace cave
