/* citation_step.h -- ORACLE (test infrastructure; see citation_rt.h).  The ERT step() driver that
 * the lifter cuts out of the reference's step(): derivatives, ODE5, clock.  Included after the
 * generated model code with CIT_MODEL defined to the generated `<prefix>model` function.
 *
 * Re-includable: once per code variant, with CIT_MODEL / CIT_DERIV (generated functions) and
 * CIT_STEP (name of the step function to define) set by the includer.
 *
 * Reference (nominal build addresses):
 *   step() entry/major bookkeeping @0x6030..0x608c, 0x9e30;
 *   inlined rt_ertODEUpdateContinuousStates @0x9eb8..0xa4f3 (Dormand-Prince "ode5", 6 stages,
 *   coefficients from the literal pool 0x13688, 0x13898..0x13938).
 */
#if !defined(CIT_MODEL) || !defined(CIT_DERIV) || !defined(CIT_STEP)
#error "define CIT_MODEL, CIT_DERIV and CIT_STEP before including citation_step.h"
#endif

#ifndef CIT_POISON
#ifdef CIT_POISON_B   /* debug: prove that no block signal carries state from one model evaluation to the next */
#define CIT_POISON(c) do { for (int i_ = 0; i_ < CIT_MAX_NB; ++i_) (c)->B[i_] = NAN; } while (0)
#else
#define CIT_POISON(c) ((void)0)
#endif
#endif
#ifndef CIT_ODE5_TABLES
#define CIT_ODE5_TABLES
static const double cit_ode5_A[6] = {0.2, 0.3, 0.8, 0.8888888888888888, 1.0, 1.0};
static const double cit_ode5_B[6][6] = {
  {0.2, 0, 0, 0, 0, 0},
  {0.075, 0.225, 0, 0, 0, 0},
  {0.9777777777777777, -3.7333333333333334, 3.5555555555555554, 0, 0, 0},
  {2.9525986892242035, -11.595793324188385, 9.822892851699436, -0.2908093278463649, 0, 0},
  {2.8462752525252526, -10.757575757575758, 8.906422717743473, 0.2784090909090909, -0.2735313036020583, 0},
  {0.09114583333333333, 0.0, 0.44923629829290207, 0.6510416666666666, -0.322376179245283, 0.13095238095238096},
};

#endif /* CIT_ODE5_TABLES */

/* One call of the reference's exported step(cmd[10], out[12]) in major-step mode. */
static inline void CIT_STEP(CitCtx *c, const double *cmd, double *out)
{
  double y[19], f[6][19];
  c->major = 1;
  CIT_POISON(c);
  CIT_MODEL(c, cmd, out);            /* stop_time=(tick+1)*dt; outputs; rtY latch; Derivative banks; out=rtY */
  const double t0 = c->t, tnew = c->stop_time, h = c->dt;
  c->major = 0;
  for (int i = 0; i < 19; ++i) y[i] = c->X[i];
  CIT_DERIV(c, f[0]);
  for (int s = 0; s < 5; ++s) {
    double hB[6];
    for (int j = 0; j <= s; ++j) hB[j] = cit_ode5_B[s][j] * h;
    for (int i = 0; i < 19; ++i) {
      double acc = f[0][i] * hB[0];
      for (int j = 1; j <= s; ++j) acc = acc + f[j][i] * hB[j];
      c->X[i] = acc + y[i];
    }
    c->t = (s == 4) ? tnew : (s == 0 ? hB[0] + t0 : h * cit_ode5_A[s] + t0);
    CIT_POISON(c);
    CIT_MODEL(c, cmd, out);
    CIT_DERIV(c, f[s + 1]);
  }
  {
    double hB[6];
    for (int j = 0; j < 6; ++j) hB[j] = cit_ode5_B[5][j] * h;
    for (int i = 0; i < 19; ++i) {
      double acc = f[0][i] * hB[0];
      for (int j = 1; j < 6; ++j) acc = acc + f[j][i] * hB[j];
      c->X[i] = acc + y[i];
    }
  }
  c->major = 1;
  c->tick += 1;
  c->t = tnew;
}
#undef CIT_MODEL
#undef CIT_DERIV
#undef CIT_STEP
