// rollout_wave_gust.hip -- wave-cooperative rollout kernels (one wavefront per episode) for the 'gust'
// dynamics code variant: build `gust` (vertical gust of 15 ft/s when the model clock passes 20 s; live Derivative block).  See rollout_wave.inc.
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_gust_wave.inc"
#define VARIANT gust
#include "rollout_wave.inc"
#undef VARIANT
