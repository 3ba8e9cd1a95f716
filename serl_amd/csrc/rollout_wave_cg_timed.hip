// rollout_wave_cg_timed.hip -- wave-cooperative rollout kernels (one wavefront per episode) for the 'cg_timed'
// dynamics code variant: build `cg_timed` (centre of gravity shifts aft when the model clock passes 20 s).  See rollout_wave.inc.
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_cg_timed_wave.inc"
#define VARIANT cg_timed
#include "rollout_wave.inc"
#undef VARIANT
