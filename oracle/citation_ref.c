/* citation_ref.c -- ORACLE: CPU restatement of the reference's native dynamics library, exported
 * with a small C ABI for ctypes (tests/, bench.py cpu_baseline, smoke()).  TEST INFRASTRUCTURE ONLY;
 * the product never links this.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 *
 * Restates  initialize() / step(real_T *cmd, real_T *out)  of envs/<build>/_citation*.so
 * (DWARF citation_to_python.h:1600-1604; SWIG surface envs/h2000_v90/citation.py:65-72) for all five
 * code variants the 14 build directories contain.  Unlike the reference (file-scope
 * state, one instance per library image) a context is re-entrant and relocatable.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#define LIFT_FN static __attribute__((unused))
#include "citation_rt.h"

#include "gen/citation_nominal.inc"
#define CIT_MODEL cit_nominal_model
#define CIT_DERIV cit_nominal_derivatives
#define CIT_STEP cit_step_nominal
#include "citation_step.h"

#include "gen/citation_ice.inc"
#define CIT_MODEL cit_ice_model
#define CIT_DERIV cit_ice_derivatives
#define CIT_STEP cit_step_ice
#include "citation_step.h"

#include "gen/citation_cg_timed.inc"
#define CIT_MODEL cit_cg_timed_model
#define CIT_DERIV cit_cg_timed_derivatives
#define CIT_STEP cit_step_cg_timed
#include "citation_step.h"

#include "gen/citation_gust.inc"
#define CIT_MODEL cit_gust_model
#define CIT_DERIV cit_gust_derivatives
#define CIT_STEP cit_step_gust
#include "citation_step.h"

#include "gen/citation_test.inc"
#define CIT_MODEL cit_test_model
#define CIT_DERIV cit_test_derivatives
#define CIT_STEP cit_step_test
#include "citation_step.h"

/* ---- C ABI ------------------------------------------------------------------------------------ */
enum { CIT_CODE_NOMINAL = 0, CIT_CODE_ICE = 1, CIT_CODE_CG_TIMED = 2, CIT_CODE_GUST = 3, CIT_CODE_TEST = 4 };

typedef struct CitInstance {
  CitCtx c;
  int code;
} CitInstance;

int cit_instance_size(void) { return (int)sizeof(CitInstance); }

/* The effect of initialize() @0xb4e0: state images x0[19] (rtX) and dw0[31] (rtDW, IWORK in the last
 * 12 bytes) captured from the live reference library right after initialize(); ro = .rodata as f64;
 * t3 = table3 parameter arrays P1[3] P2[4] P3[3] P4[36]. */
void cit_reset(CitInstance *I, int code, const double *ro, const double *t3, const double *x0,
               const double *dw0, double dt)
{
  CitCtx *c = &I->c;
  memset(c, 0, sizeof(*c));
  I->code = code;
  memcpy(c->X, x0, sizeof(c->X));
  memcpy(c->DW, dw0, sizeof(c->DW));
  memcpy(c->IW, (const char *)dw0 + 29 * 8, 12);
  c->ro = ro; c->t3 = t3; c->dt = dt; c->major = 1; c->tick = 0; c->t = 0.0;
}

/* The reference's initialize() does not reset the model clock (clockTick0 @rtM+0xba18, t @+0xbac0 keep their
 * values; probed on the live cg_timed library): an episode that is not the first of its process starts here. */
void cit_set_clock(CitInstance *I, unsigned tick)
{
  I->c.tick = tick;
  I->c.t = (double)tick * I->c.dt;
}

int cit_step(CitInstance *I, const double *cmd, double *out)
{
  switch (I->code) {
    case CIT_CODE_NOMINAL: cit_step_nominal(&I->c, cmd, out); return 0;
    case CIT_CODE_ICE: cit_step_ice(&I->c, cmd, out); return 0;
    case CIT_CODE_CG_TIMED: cit_step_cg_timed(&I->c, cmd, out); return 0;
    case CIT_CODE_GUST: cit_step_gust(&I->c, cmd, out); return 0;
    case CIT_CODE_TEST: cit_step_test(&I->c, cmd, out); return 0;
  }
  return -1;
}

/* n steps with a fixed command (throughput measurement of the bare dynamics) */
int cit_step_n(CitInstance *I, const double *cmd, double *out, int n)
{
  for (int k = 0; k < n; ++k) if (cit_step(I, cmd, out)) return -1;
  return 0;
}

double *cit_B(CitInstance *I) { return I->c.B; }
double *cit_X(CitInstance *I) { return I->c.X; }
double *cit_DW(CitInstance *I) { return I->c.DW; }

/* One model evaluation at a caller-supplied state (test aid for tools/dag: per-stage comparison of the block
 * signals B and the derivative vector).  major != 0 also latches rtY and updates the Derivative-block banks. */
int cit_eval(CitInstance *I, const double *X, const double *cmd, double t, int major, double *xdot)
{
  CitCtx *c = &I->c;
  double out[12];
  memcpy(c->X, X, sizeof(c->X));
  c->t = t; c->major = major;
  switch (I->code) {
    case CIT_CODE_NOMINAL: cit_nominal_model(c, cmd, out); cit_nominal_derivatives(c, xdot); break;
    case CIT_CODE_ICE: cit_ice_model(c, cmd, out); cit_ice_derivatives(c, xdot); break;
    case CIT_CODE_CG_TIMED: cit_cg_timed_model(c, cmd, out); cit_cg_timed_derivatives(c, xdot); break;
    case CIT_CODE_GUST: cit_gust_model(c, cmd, out); cit_gust_derivatives(c, xdot); break;
    case CIT_CODE_TEST: cit_test_model(c, cmd, out); cit_test_derivatives(c, xdot); break;
    default: return -1;
  }
  c->major = 1;
  return 0;
}
