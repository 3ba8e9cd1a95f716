// rollout_wave_test.hip -- wave-cooperative rollout kernels (one wavefront per episode) for the 'test'
// dynamics code variant: build `test` (the reference's 14th dynamics build, envs/test).  See rollout_wave.inc.
#include "citation_wave.h"
#include "rollout_device.h"
#include "gen/citation_test_wave.inc"
#define VARIANT test
#include "rollout_wave.inc"
#undef VARIANT
