// rollout_team_test.hip -- two-wavefront-per-episode rollout kernels for the 'test' dynamics code variant
// (rollout_team.inc, gen/citation_test_team.inc): the latency-bound regime, fewer episodes than CUs.
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_test_wave.inc"   // look-up descriptor tables (shared with the one-wave kernels)
#include "gen/citation_test_team.inc"
#define VARIANT test
#include "rollout_team.inc"
#undef VARIANT
