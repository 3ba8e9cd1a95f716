// rollout_team_nominal.hip -- team (seven wavefronts + the actor wavefront per episode) rollout kernels for the 'nominal' dynamics code variant
// (rollout_team.inc, gen/citation_nominal_team.inc): the latency-bound regime, fewer episodes than CUs.
#define CITW_SEARCH_BATCH 1
#define CITW_MAX_WAVES 1          // one episode per workgroup: the team shares row 0 of every blackboard ...
#ifndef CITW_M_ROWS
#define CITW_M_ROWS 8             // ... except the libm results: one row per wavefront of the team
#endif
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_nominal_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#ifndef CITW_TEAM_INC
#define CITW_TEAM_INC "gen/citation_nominal_team.inc"
#endif
#include CITW_TEAM_INC
#define VARIANT nominal
#include "rollout_team.inc"
#undef VARIANT
