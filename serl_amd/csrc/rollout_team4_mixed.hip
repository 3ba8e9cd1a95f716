// rollout_team4_mixed.hip -- ONE code object for the mixed-fault sweep (BASELINE config 5; round 5).
//
// A mixed-fault population (fault mode per episode over be / jr / sa / se / ice / cg: envs/phlabenv.py:114-165, envs/{be,jr,sa,se}/citation.py:71-79)
// needs the dynamics of several BUILDS in one evaluation: h2000_v90 and cg run the 'nominal' code variant on different tables, ice runs the 'ice'
// variant.  Until round 5 that was one launch per build, side by side on streams of their own -- and two DIFFERENT code objects side by side cost
// 12 % (profiles/r04_experiments.md section 4: be + cg, the same code on different tables, is free; as soon as the ice kernel runs beside the
// nominal one every launch slows down: neighbouring CUs share an instruction cache, and two 95 KB kernels do not fit what one does).
// Here the four-episodes-per-team device functions of BOTH code variants (rollout_team.inc + rollout_team_half.inc, lane groups of 16) are
// compiled into one kernel; a workgroup runs the variant and the tables of the LAUNCH PART it belongs to (contiguous workgroup ranges, so that
// the workgroups dispatched next to each other almost always run the same slices of code).  Every part is what a launch of its own would
// be -- its own descriptor, tables, episode range and work-queue counter -- so the results are those of the separate launches, bit for bit.
#define CITW_SEARCH_BATCH 1
#define CITW_GROUP_LANES 16
#define CITW_MAX_WAVES 4          // blackboard rows: one per episode of the team
#define CITW_M_ROWS 32            // libm result rows: one per (team wavefront, episode)
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
#define SERL_NO_CHUNKED_ACTOR 1      // (H = 32 actors only: serl_capi.hip)
#define SERL_TEAM_NO_ENTRY 1         // device functions only: the kernel is below
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_nominal_wave.inc"
#include "gen/citation_nominal_teamg.inc"
#define VARIANT nominal
#include "rollout_team.inc"
#undef VARIANT
#include "gen/citation_ice_wave.inc"
#include "gen/citation_ice_teamg.inc"
#define VARIANT ice
#include "rollout_team.inc"
#undef VARIANT
#include "serl_mixed.h"

static_assert(citw_nominal_team_WAVES == citw_ice_team_WAVES, "one workgroup shape for both variants");

__global__ void __launch_bounds__(64 * (citw_nominal_team_WAVES + SERL_ACTOR_WAVES)) serl_rollout_kernel_team4_mixed(SerlMixedArgs m)
{
  // which part of the launch this workgroup belongs to (wave-uniform; at most SERL_MIXED_MAX parts)
  int k = 0;
#pragma unroll
  for (int i = 1; i < SERL_MIXED_MAX; ++i) k = (i < m.n && (int)blockIdx.x >= m.first_wg[i]) ? i : k;
  k = __builtin_amdgcn_readfirstlane(k);
  // part k's arguments straight out of the kernel-argument segment (`m.a[k]` with a run-time k would copy the whole struct to scratch,
  // and every descriptor field read in the episode loop would come from there)
  const SerlMixedArgs *km = (const SerlMixedArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  const RolloutArgs &a = km->a[k];
  const bool actor = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= citw_nominal_team_WAVES;
  if (m.code[k] == SERL_DYN_ICE) {
    citw_team_stage_ice(a);
    if (actor) serl_teamg_actor_wave_ice(a);
    else serl_teamg_episodes_ice(a);
  } else {
    citw_team_stage_nominal(a);
    if (actor) serl_teamg_actor_wave_nominal(a);
    else serl_teamg_episodes_nominal(a);
  }
}

void serl_launch_rollout_team4_mixed(const SerlMixedArgs &m, int grid, hipStream_t stream)
{
  hipLaunchKernelGGL(serl_rollout_kernel_team4_mixed, dim3(grid), dim3(64 * (citw_nominal_team_WAVES + SERL_ACTOR_WAVES)), 0, stream, m);
}
