// rollout_ice.hip -- rollout kernel for the 'ice' dynamics code variant (see rollout_variant.inc).
#include "citation_dev.h"
// hand-written leaves (citation_leaves.h) replace the lifted ones of this variant
#define cit_ice_rt_Lookup2D_Normal(ro, xr, nr, xc, nc, z, u0, u1) cit_lookup2d((xr), (int)(nr), (xc), (int)(nc), (z), (u0), (u1))
#define cit_ice_rt_Lookup(ro, x, n, u, y) cit_lookup1d((x), (int)(n), (u), (y))
#define cit_ice_ac_axes(ro, su, sy, mode) (c->err |= cit_axes_apply(&c->ax, (su), (sy), (mode)))
#define CIT_RO_LO_W cit_ice_RO_LO_W
#define CIT_RO_HI_W cit_ice_RO_HI_W
#define RO_BASE_W cit_ice_RO_BASE_W
#include "gen/citation_ice.inc"
static_assert(cit_ice_RO_HI_W - cit_ice_RO_LO_W <= CIT_RO_LDS_WORDS, "LDS table window too small");
#define CIT_MODEL cit_ice_model
#define CIT_DERIV cit_ice_derivatives
#define CIT_STEP cit_step_ice
#include "citation_step_dev.h"
#include "rollout_device.h"
#define VARIANT ice
#include "rollout_variant.inc"
