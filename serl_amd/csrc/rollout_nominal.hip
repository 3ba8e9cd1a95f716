// rollout_nominal.hip -- rollout kernel for the 'nominal' dynamics code variant (see rollout_variant.inc).
#include "citation_dev.h"
#include "gen/citation_nominal.inc"
#define CIT_MODEL cit_nominal_model
#define CIT_DERIV cit_nominal_derivatives
#define CIT_STEP cit_step_nominal
#include "citation_step_dev.h"
#include "rollout_device.h"
#define VARIANT nominal
#include "rollout_variant.inc"
