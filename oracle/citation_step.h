/* citation_step.h -- ORACLE (test infrastructure; see citation_rt.h).  The ERT step() driver that
 * the lifter cuts out of the reference's step(): derivatives, ODE5, clock.  Included after the
 * generated model code with CIT_MODEL defined to the generated `<prefix>model` function.
 *
 * Reference (nominal build addresses):
 *   citation_to_python_derivatives @0x5f60;  step() entry/major bookkeeping @0x6030..0x608c, 0x9e30;
 *   inlined rt_ertODEUpdateContinuousStates @0x9eb8..0xa4f3 (Dormand-Prince "ode5", 6 stages,
 *   coefficients from the literal pool 0x13688, 0x13898..0x13938).
 */
#ifndef CIT_MODEL
#error "define CIT_MODEL before including citation_step.h"
#endif

/* citation_to_python_derivatives: XDot[0..11] = B.TmpSignalConversionAtIntegrator (B+0x13b8);
 * washout: -0.25*X[12] + 0.0 + B.Integrator[2];  the two "Parameter" states are constants;
 * engine states from B.Switch_bn / Switch / Switch_b / Switch_h (B+0x1428,0x1418,0x1420,0x1430).
 * The B offsets are those of the nominal layout; builds with a longer B (gust/test) shift them by
 * CIT_B_SHIFT doubles. */
#ifndef CIT_B_SHIFT
#define CIT_B_SHIFT 0
#endif
static inline void cit_derivatives(CitCtx *c, double *xdot, double washout_k, double washout_c)
{
  const double *tail = &c->B[0x13b8 / 8 + CIT_B_SHIFT];
  for (int i = 0; i < 12; ++i) xdot[i] = tail[i];
  double w = washout_k * c->X[12];
  w = w + washout_c;
  w = w + c->B[2];
  xdot[12] = w;
  xdot[13] = 0.0;
  xdot[14] = 0.0;
  xdot[15] = tail[14];   /* B.Switch_bn @0x1428 */
  xdot[16] = tail[12];   /* B.Switch    @0x1418 */
  xdot[17] = tail[13];   /* B.Switch_b  @0x1420 */
  xdot[18] = tail[15];   /* B.Switch_h  @0x1430 */
}

static const double cit_ode5_A[6] = {0.2, 0.3, 0.8, 0.8888888888888888, 1.0, 1.0};
static const double cit_ode5_B[6][6] = {
  {0.2, 0, 0, 0, 0, 0},
  {0.075, 0.225, 0, 0, 0, 0},
  {0.9777777777777777, -3.7333333333333334, 3.5555555555555554, 0, 0, 0},
  {2.9525986892242035, -11.595793324188385, 9.822892851699436, -0.2908093278463649, 0, 0},
  {2.8462752525252526, -10.757575757575758, 8.906422717743473, 0.2784090909090909, -0.2735313036020583, 0},
  {0.09114583333333333, 0.0, 0.44923629829290207, 0.6510416666666666, -0.322376179245283, 0.13095238095238096},
};

/* One call of the reference's exported step(cmd[10], out[12]) in major-step mode. */
static inline void cit_step(CitCtx *c, const double *cmd, double *out, double washout_k, double washout_c)
{
  double y[19], f[6][19];
  c->major = 1;
  CIT_MODEL(c, cmd, out);            /* stop_time=(tick+1)*dt; outputs; rtY latch; Derivative banks; out=rtY */
  const double t0 = c->t, tnew = c->stop_time, h = c->dt;
  c->major = 0;
  for (int i = 0; i < 19; ++i) y[i] = c->X[i];
  cit_derivatives(c, f[0], washout_k, washout_c);
  for (int s = 0; s < 5; ++s) {
    double hB[6];
    for (int j = 0; j <= s; ++j) hB[j] = cit_ode5_B[s][j] * h;
    for (int i = 0; i < 19; ++i) {
      double acc = f[0][i] * hB[0];
      for (int j = 1; j <= s; ++j) acc = acc + f[j][i] * hB[j];
      c->X[i] = acc + y[i];
    }
    c->t = (s == 4) ? tnew : (s == 0 ? hB[0] + t0 : h * cit_ode5_A[s] + t0);
    CIT_MODEL(c, cmd, out);
    cit_derivatives(c, f[s + 1], washout_k, washout_c);
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
