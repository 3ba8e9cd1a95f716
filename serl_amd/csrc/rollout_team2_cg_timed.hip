// rollout_team2_cg_timed.hip -- two episodes per team (rollout_team.inc + rollout_team_half.inc, lane groups of 32) for the 'cg_timed'
// dynamics code variant: between CUs and 4 x CUs episodes per launch, SERL50 actor shape (H = 32).
#define CITW_SEARCH_BATCH 1
#define CITW_GROUP_LANES 32
#define CITW_MAX_WAVES 2          // blackboard rows: one per episode of the team
#define CITW_M_ROWS 16            // libm result rows: one per (team wavefront, episode)
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
#define SERL_NO_CHUNKED_ACTOR 1      // (these kernels carry H = 32 actors only: serl_capi.hip)
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_cg_timed_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#include "gen/citation_cg_timed_team.inc"
#define VARIANT cg_timed
#include "rollout_team.inc"
#undef VARIANT
