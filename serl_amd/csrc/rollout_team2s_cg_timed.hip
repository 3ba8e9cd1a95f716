// rollout_team2s_cg_timed.hip -- two episodes per team for actors that STREAM their weights (hidden != 32: SERL10's 72, the TD3 actor's
// 96), 'cg_timed' dynamics code variant: a team of SIX wavefronts (gen/citation_cg_timed_team6.inc) carries the two episodes in lane
// groups of 32, TWO actor wavefronts run one episode's forward pass each (rollout_team.inc + rollout_team_half.inc).
#define CITW_SEARCH_BATCH 1
#define CITW_GROUP_LANES 32
#define CITW_MAX_WAVES 2          // blackboard rows: one per episode of the team
#define CITW_M_ROWS 16            // libm result rows: one per (team wavefront, episode)
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
#define SERL_ACTOR_WAVES 2
#define SERL_TEAMG_TAG team2s_
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_cg_timed_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#include "gen/citation_cg_timed_team6.inc"
#define VARIANT cg_timed
#include "rollout_team.inc"
#undef VARIANT
