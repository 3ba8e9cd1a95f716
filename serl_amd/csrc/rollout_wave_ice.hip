// rollout_wave_ice.hip -- wave-cooperative rollout kernels (one wavefront per episode) for the 'ice'
// dynamics code variant: build `ice` (icing: lift-coefficient saturation, drag / lift offsets).  See rollout_wave.inc.
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_ice_wave.inc"
#define VARIANT ice
#include "rollout_wave.inc"
#undef VARIANT
