// rollout_team_ice.hip -- team (seven wavefronts + the actor wavefront per episode) rollout kernels for the 'ice' dynamics code variant
// (rollout_team.inc, gen/citation_ice_team.inc): the latency-bound regime, fewer episodes than CUs.
#define CITW_SEARCH_BATCH 1
#define CITW_MAX_WAVES 1          // one episode per workgroup: the team shares row 0 of every blackboard ...
#ifndef CITW_M_ROWS
#define CITW_M_ROWS 8             // ... except the libm results: one row per wavefront of the team
#endif
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_ice_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#ifndef CITW_TEAM_INC
#define CITW_TEAM_INC "gen/citation_ice_team.inc"
#endif
#include CITW_TEAM_INC
#define VARIANT ice
#include "rollout_team.inc"
#undef VARIANT
