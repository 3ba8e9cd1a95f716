// rollout_cg_timed.hip -- lane-per-episode rollout kernels (lanes_per_wave > 0: one episode per lane) for the 'cg_timed' dynamics
// code variant, around the branch-free model evaluation tools/dag/codegen_lane.py generates from the DAG (~1 170 nodes
// per lane); block signals do not exist there, c->B only carries the 19 derivatives.  See rollout_variant.inc.
// (The same kernels compiled around the LIFTED model are test infrastructure: oracle/xcheck/, libserl_xcheck.so.)
#include "citation_dev.h"
#include "rollout_device.h"
#define VARIANT cg_timed
#define CIT_RO_LO_W cit_cg_timed_RO_LO_W
#define CIT_RO_HI_W cit_cg_timed_RO_HI_W
#define RO_BASE_W cit_cg_timed_RO_BASE_W

namespace bdag {
#define CIT_B_AT(i) (c->B[(i)])
#define SERL_FLAVOUR_LDS 0
#define CIT_NO_AXES 1
// the banks live in the CALLER's stack frame: a private-segment pointer, so that the reads are scratch loads (vmcnt) and not flat loads, which also count on
// the LDS counter the table reads wait on (65 536 episodes: 219.6 -> 225.9 M env-steps/s)
#ifndef CIT_CTX_FLAT
#define CIT_DWM_PTR __attribute__((address_space(5))) double *
#define CIT_CMD_PTR const __attribute__((address_space(5))) double *
#else
#define CIT_DWM_PTR double *
#define CIT_CMD_PTR const double *
#endif
#include "gen/citation_cg_timed_lane.inc"
static_assert(cit_cg_timed_RO_HI_W - cit_cg_timed_RO_LO_W <= CIT_RO_LDS_WORDS, "LDS table window too small");
static_assert(8 * (CIT_RO_LDS_WORDS + cit_cg_timed_NSLOPE) <= 160 * 1024, "tables + interval quotients beyond the 160 KB of LDS");
#define CIT_MODEL cit_cg_timed_dag_model
#define CIT_DERIV cit_cg_timed_dag_derivatives
#define CIT_STEP cit_step_cg_timed
#define CIT_SLOPE_DESC cit_cg_timed_slope_desc      // (precomputed x-direction quotients of the tables: rollout_variant.inc stages them, citation_leaves.h cit_lookup2d_at_s)
#define CIT_SLOPE_TABLES cit_cg_timed_NSLOPE_TABLES
#define CIT_USE_HINTS cit_cg_timed_NSEARCH      // (the index searches verify the previous evaluation's interval first: CitCtx.hint travels with the state)
#define CIT_DW_IN_MEMORY 1
#ifdef CIT_WITH_CMD_IN_MEMORY      // (A/B builds around gen files made with CITW_LANE_CMDMEM=1: measured slower)
#define CIT_CMD_IN_MEMORY 1
#endif
#define CIT_Y_IS_STATE 1      // (gen/citation_cg_timed_lane.inc: `if (major) c->Y[i] = X[i]`, i < 12 -- the outputs of step() are the states in front of the integration)
#include "citation_step_dev.h"
#include "rollout_variant.inc"
#undef CIT_NO_AXES
#undef CIT_B_AT
#undef SERL_FLAVOUR_LDS
}  // namespace bdag

void serl_launch_rollout_cg_timed(const RolloutArgs &a, int grid, hipStream_t stream) { bdag::serl_launch_rollout_cg_timed(a, grid, stream); }

void serl_launch_dyn_cg_timed(const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream)
{
  bdag::serl_launch_dyn_cg_timed(a, cmds, states, T, grid, stream);
}
