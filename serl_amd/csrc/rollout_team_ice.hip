// rollout_team_ice.hip -- team (four wavefronts per episode) rollout kernels for the 'ice' dynamics code variant
// (rollout_team.inc, gen/citation_ice_team.inc): the latency-bound regime, fewer episodes than CUs.
#define CITW_SEARCH_BATCH 1
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_ice_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#include "gen/citation_ice_team.inc"
#define VARIANT ice
#include "rollout_team.inc"
#undef VARIANT
