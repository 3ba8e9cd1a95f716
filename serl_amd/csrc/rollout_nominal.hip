// rollout_nominal.hip -- rollout kernels for the 'nominal' dynamics code variant, in two flavours that differ in
// where the model's block signals live (citation_dev.h): `breg` (registers, any lanes per wave) and `blds`
// (LDS, <= SERL_LDS_B_LANES_PER_WAVE episodes per wavefront).  See rollout_variant.inc.
#include <stdlib.h>
#include "citation_dev.h"
#include "rollout_device.h"
// hand-written leaves (citation_leaves.h) replace the lifted ones of this variant
#define cit_nominal_rt_Lookup2D_Normal(ro, xr, nr, xc, nc, z, u0, u1) cit_lookup2d((xr), (int)(nr), (xc), (int)(nc), (z), (u0), (u1))
#define cit_nominal_rt_Lookup(ro, x, n, u, y) cit_lookup1d((x), (int)(n), (u), (y))
#define cit_nominal_ac_axes(ro, su, sy, mode) (c->err |= cit_axes_apply(&c->ax, (su), (sy), (mode)))
#define CIT_RO_LO_W cit_nominal_RO_LO_W
#define CIT_RO_HI_W cit_nominal_RO_HI_W
#define RO_BASE_W cit_nominal_RO_BASE_W
#define VARIANT nominal

namespace breg {
#define CIT_B_AT(i) (c->B[(i)])
#define SERL_FLAVOUR_LDS 0
#include "gen/citation_nominal.inc"
static_assert(cit_nominal_RO_HI_W - cit_nominal_RO_LO_W <= CIT_RO_LDS_WORDS, "LDS table window too small");
#define CIT_MODEL cit_nominal_model
#define CIT_DERIV cit_nominal_derivatives
#define CIT_STEP cit_step_nominal
#include "citation_step_dev.h"
#include "rollout_variant.inc"
#undef CIT_B_AT
#undef SERL_FLAVOUR_LDS
}  // namespace breg

namespace blds {
#define CIT_B_AT(i) (g_B[c->bslot + (i)])
#define SERL_FLAVOUR_LDS 1
#include "gen/citation_nominal.inc"
#define CIT_MODEL cit_nominal_model
#define CIT_DERIV cit_nominal_derivatives
#define CIT_STEP cit_step_nominal
#include "citation_step_dev.h"
#include "rollout_variant.inc"
#undef CIT_B_AT
#undef SERL_FLAVOUR_LDS
}  // namespace blds

// the same kernels around the model evaluation generated from the DAG (tools/dag/codegen_lane.py: ~1 170 branch-free nodes
// instead of the ~5 000 lifted statements); block signals do not exist there, c->B only carries the 19 derivatives
namespace bdag {
#define CIT_B_AT(i) (c->B[(i)])
#define SERL_FLAVOUR_LDS 0
#define CIT_NO_AXES 1
#include "gen/citation_nominal_lane.inc"
#define CIT_MODEL cit_nominal_dag_model
#define CIT_DERIV cit_nominal_dag_derivatives
#define CIT_STEP cit_step_nominal
#include "citation_step_dev.h"
#include "rollout_variant.inc"
#undef CIT_NO_AXES
#undef CIT_B_AT
#undef SERL_FLAVOUR_LDS
}  // namespace bdag

// SERL_LANE_DAG=0 selects the kernels compiled from the lifted code (a second, independently derived implementation)
static bool serl_lane_dag() { const char *e = getenv("SERL_LANE_DAG"); return !e || atoi(e) != 0; }

void serl_launch_rollout_nominal(const RolloutArgs &a, int grid, hipStream_t stream)
{
  if (serl_lane_dag()) bdag::serl_launch_rollout_nominal(a, grid, stream);
  else if (a.lanes <= SERL_LDS_B_LANES_PER_WAVE && a.block <= 256) blds::serl_launch_rollout_nominal(a, grid, stream);
  else breg::serl_launch_rollout_nominal(a, grid, stream);
}

void serl_launch_dyn_nominal(const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream)
{
  if (serl_lane_dag()) bdag::serl_launch_dyn_nominal(a, cmds, states, T, grid, stream);
  else if (a.lanes <= SERL_LDS_B_LANES_PER_WAVE && a.block <= 256) blds::serl_launch_dyn_nominal(a, cmds, states, T, grid, stream);
  else breg::serl_launch_dyn_nominal(a, cmds, states, T, grid, stream);
}
