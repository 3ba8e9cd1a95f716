// citation_step_dev.h -- re-includable: defines CIT_STEP(c, cmd, out), one call of the reference's
// exported step(cmd[10], out[12]) (major step + ODE5 + clock), on top of the generated
// CIT_MODEL / CIT_DERIV functions of one code variant.  See citation_dev.h.
#if !defined(CIT_MODEL) || !defined(CIT_DERIV) || !defined(CIT_STEP)
#error "define CIT_MODEL, CIT_DERIV and CIT_STEP"
#endif
static __device__ __noinline__ void CIT_STEP(CitCtx *c, const double *cmd, double *out)
{
  double y[19], f[6][19];
  c->major = 1;
  CIT_MODEL(c, cmd, out);  // stop_time = (tick+1)*dt; outputs; rtY latch; Derivative banks; out = rtY
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
