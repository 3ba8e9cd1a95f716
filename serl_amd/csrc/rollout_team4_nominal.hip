// rollout_team4_nominal.hip -- four episodes per team (rollout_team.inc + rollout_team_half.inc, lane groups of 16) for the 'nominal'
// dynamics code variant: between 2 x CUs and 4 x CUs episodes per launch, SERL50 actor shape (H = 32).
#define CITW_SEARCH_BATCH 1
#define CITW_GROUP_LANES 16
#define CITW_MAX_WAVES 4          // blackboard rows: one per episode of the team
#define CITW_M_ROWS 32            // libm result rows: one per (team wavefront, episode)
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
#define SERL_NO_CHUNKED_ACTOR 1      // (these kernels carry H = 32 actors only: serl_capi.hip)
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_nominal_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#ifndef CITW_TEAM_INC
#define CITW_TEAM_INC "gen/citation_nominal_teamg.inc"      // (tools/exp_build.py: A/B builds around another generated file)
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(CITW_LIBM_COLD_CALLS)      // (A/B builds: measured slower, profiles/r06_experiments.md section 5)
#define exp citw_general_exp            // (the generated model calls ocml's exp / log10 behind gates that are closed at the trimmed flight condition)
#define log10 citw_general_log10
#endif
#include CITW_TEAM_INC
#undef exp
#undef log10
#define VARIANT nominal
#include "rollout_team.inc"
#undef VARIANT
