/* citation_ref.c -- ORACLE: CPU restatement of the reference's native dynamics library, exported
 * with a small C ABI for ctypes (tests/, bench.py cpu_baseline, smoke()).  TEST INFRASTRUCTURE ONLY;
 * the product never links this.  Build: see oracle/Makefile (gcc -O2 -ffp-contract=off).
 *
 * Restates:  initialize()/step(real_T*,real_T*) of envs/<build>/_citation*.so
 *            (DWARF citation_to_python.h:1600-1604; SWIG surface envs/h2000_v90/citation.py:65-72)
 * Unlike the reference (file-scope state, one instance per library image) this is re-entrant.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#define LIFT_FN static __attribute__((unused))
#include "citation_rt.h"

#define CIT_PASTE2(a, b) a##b
#define CIT_PASTE(a, b) CIT_PASTE2(a, b)

#include "gen/citation_h2000_v90.inc"
#define CIT_MODEL cit_h2000_v90_model
#include "citation_step.h"

/* ---- C ABI ------------------------------------------------------------------------------------ */
int cit_ctx_size(void) { return (int)sizeof(CitCtx); }

/* images: x0[19], dw0[31] (rtDW image incl. IWORK in the last 3 ints), ro (rodata f64), t3[46] */
void cit_reset(CitCtx *c, const double *ro, const double *t3, const double *x0, const double *dw0, double dt)
{
  memset(c, 0, sizeof(*c));
  memcpy(c->X, x0, sizeof(c->X));
  memcpy(c->DW, dw0, sizeof(c->DW));
  memcpy(c->IW, (const char *)dw0 + 29 * 8, 12);
  c->ro = ro; c->t3 = t3; c->dt = dt; c->major = 1; c->tick = 0; c->t = 0.0;
}

void cit_step_nominal(CitCtx *c, const double *cmd, double *out)
{
  cit_step(c, cmd, out, -0.25, 0.0);
}

double *cit_B(CitCtx *c) { return c->B; }
double *cit_X(CitCtx *c) { return c->X; }
double *cit_DW(CitCtx *c) { return c->DW; }
