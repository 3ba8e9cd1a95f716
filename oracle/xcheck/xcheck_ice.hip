// xcheck_ice.hip -- TEST INFRASTRUCTURE: lane-per-episode kernels around the lifted 'ice' model (see xcheck_variant.inc).
#include <stdlib.h>
#include "citation_dev.h"
#include "rollout_device.h"
// hand-written leaves (citation_leaves.h) replace the lifted ones of this variant
#define cit_ice_rt_Lookup2D_Normal(ro, xr, nr, xc, nc, z, u0, u1) cit_lookup2d((xr), (int)(nr), (xc), (int)(nc), (z), (u0), (u1))
#define cit_ice_rt_Lookup(ro, x, n, u, y) cit_lookup1d((x), (int)(n), (u), (y))
#define cit_ice_ac_axes(ro, su, sy, mode) (c->err |= cit_axes_apply(&c->ax, (su), (sy), (mode)))
#define CIT_RO_LO_W cit_ice_RO_LO_W
#define CIT_RO_HI_W cit_ice_RO_HI_W
#define RO_BASE_W cit_ice_RO_BASE_W
#define VARIANT ice
#define XC_GEN_INC "../gen/citation_ice.inc"
#include "xcheck_variant.inc"
