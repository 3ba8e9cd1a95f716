// rollout_team_cg_timed.hip -- team (four wavefronts per episode) rollout kernels for the 'cg_timed' dynamics code variant
// (rollout_team.inc, gen/citation_cg_timed_team.inc): the latency-bound regime, fewer episodes than CUs.
#define CITW_SEARCH_BATCH 1
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_cg_timed_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#include "gen/citation_cg_timed_team.inc"
#define VARIANT cg_timed
#include "rollout_team.inc"
#undef VARIANT
