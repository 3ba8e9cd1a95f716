// rollout_teamr_gust.hip -- REMOTE-ACTOR team kernel (round 6: the seven team wavefronts + a courier on one CU, the episode's two actor wavefronts in a workgroup of their own on another; rollout_team.inc SERL_TEAM_REMOTE) for the 'gust' dynamics code variant
// (rollout_team.inc, gen/citation_gust_team.inc): the latency-bound regime, fewer episodes than CUs.
#define CITW_SEARCH_BATCH 1
#define SERL_ACTOR_WAVES 2          // (the LDS rows and flags of a forward pass shared by several wavefronts: rollout_team.inc)
#define SERL_TEAM_REMOTE 1
#define CITW_MAX_WAVES 1          // one episode per workgroup: the team shares row 0 of every blackboard ...
#ifndef CITW_M_ROWS
#define CITW_M_ROWS 8             // ... except the libm results: one row per wavefront of the team
#endif
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
// Which role of the partitioned evaluation runs on which hardware wavefront (w and w + 4 share a SIMD, the actor is wavefront 7), and a static
// issue priority.  Round 4, sweeps k / l (after the short libm and the new balance, 150 episodes, us per env step): identity 16.85 - 16.89;
// roles 1 and 6 exchanged -- role 1, last at B1, beside role 2, which waits longest there; role 5 beside role 6 -- 16.60; + role 1 at
// priority 1: 16.56 - 16.61.  Session ae (books on the actor wavefront, its forward pass a quarter shorter): all 105 pairings of the seven
// roles on the four SIMDs (tools/sweep_roles.py, profiles/r04_ae_roles.json: 15.80 - 18.40): the look-up role 0 beside the LDS-resident
// actor, (1, 2) (3, 4) (5, 6) on the others: 15.80 - 15.83 against 15.89 - 15.97 of the map below it.  Beside an actor that STREAMS its
// weights (SERL10, the TD3 actor: busy most of the step, at priority) the older map stays -- 17.55 against 17.8 with the new one (session af).
// One episode per team only: the lane-group kernels keep the identity.
#ifndef SERL_TEAM_ROLES
#define SERL_TEAM_ROLES {1, 3, 5, 0, 2, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}
#endif
#ifndef SERL_TEAMS_ROLES
#define SERL_TEAMS_ROLES {0, 6, 2, 3, 4, 5, 1, 7, 8, 9, 10, 11, 12, 13, 14, 15}
#endif
#ifndef CITW_ROLE_PRIO_MASK
#define CITW_ROLE_PRIO_MASK 0x02
#endif
// Round 5: the LDS-actor kernel keeps every role's f64 literals -- the glue's, the coefficients of the short sincos / pow bodies, the actor's
// activation polynomial -- in registers for the episode (citation_wave.h CITW_K, citation_libm.h CITW_LK, rollout_device.h DET_K): 143 -> 235 of
// the 256 VGPRs two wavefronts per SIMD allow, 9 481 -> 8 747 static instructions (nominal).
#define CITW_PROF_TEAM_ROLES 1      // profiling builds: the marks of role r fire on the hardware wavefront that runs it (citation_wave.h CITW_PROF_IS)
#ifndef SERL_TEAM_KREGS
#define SERL_TEAM_KREGS 1
#endif
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_gust_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#ifndef CITW_TEAM_INC
#define CITW_TEAM_INC "gen/citation_gust_team.inc"
#endif
#include CITW_TEAM_INC
#define VARIANT gust
#include "rollout_team.inc"
#undef VARIANT
