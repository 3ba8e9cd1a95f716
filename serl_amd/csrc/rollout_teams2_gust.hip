// rollout_teams2_gust.hip -- one episode per team for actors that STREAM their weights (hidden > 64: SERL10's 72, the TD3 actor's 96), 'gust'
// dynamics code variant: a team of SIX wavefronts (gen/citation_gust_team6.inc) integrates the model, TWO actor wavefronts share ONE forward
// pass beside it -- each owns half of every layer's rows (rollout_device.h: serl_actor_forward_split; rollout_team.inc).
#define CITW_SEARCH_BATCH 1
#define CITW_MAX_WAVES 1          // one episode per workgroup: the team shares row 0 of every blackboard ...
#define CITW_M_ROWS 8             // ... except the libm results: one row per wavefront of the team
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
#define SERL_ACTOR_WAVES 2
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_gust_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#include "gen/citation_gust_team6.inc"
#define VARIANT gust
#include "rollout_team.inc"
#undef VARIANT
