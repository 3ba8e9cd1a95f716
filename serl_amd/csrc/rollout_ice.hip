// rollout_ice.hip -- rollout kernel for the 'ice' dynamics code variant (see rollout_variant.inc).
#include "citation_dev.h"
#include "gen/citation_ice.inc"
#define CIT_MODEL cit_ice_model
#define CIT_DERIV cit_ice_derivatives
#define CIT_STEP cit_step_ice
#include "citation_step_dev.h"
#include "rollout_device.h"
#define VARIANT ice
#include "rollout_variant.inc"
