// rollout_half_gust.hip -- two episodes per wavefront (rollout_half.inc) for the 'gust' dynamics code variant: the
// regime beyond one wavefront per SIMD (more than 4 x CUs episodes per launch), SERL50 actor shape (H = 32).
#define CITW_GROUP_LANES 32
#define CITW_MAX_WAVES 8          // LDS rows: 4 wavefronts x 2 episodes
#define CITW_OUT2_ROWS 1          // (no variant has a third look-up round / an invariant round)
#define CITW_INV_SLOTS 8
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_gust_wave.inc"
#define VARIANT gust
#include "rollout_half.inc"
#undef VARIANT
