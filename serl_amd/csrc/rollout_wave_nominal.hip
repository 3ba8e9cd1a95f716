// rollout_wave_nominal.hip -- wave-cooperative rollout kernels (one wavefront per episode) for the 'nominal'
// dynamics code variant: builds h2000_v90, h2000_v150, h10000_v90, cg, cg_for and the Python-level fault
// wrappers be / jr / sa / se on top of h2000_v90.  See rollout_wave.inc.
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_nominal_wave.inc"
#define VARIANT nominal
#include "rollout_wave.inc"
#undef VARIANT
