// citation_step_dev.h -- re-includable: defines CIT_STEP(c, cmd, out), one call of the reference's
// exported step(cmd[10], out[12]) (major step + ODE5 + clock), on top of the generated
// CIT_MODEL / CIT_DERIV functions of one code variant.  See citation_dev.h.
//
// Structure for the GPU: the six model evaluations of one ODE5 macro step (1 major + 5 minor; the
// reference recurses into step() for the minor ones, @0xa041..0xa3e7) run as ONE loop around ONE inlined
// copy of the model body, on a function-local context.  Block signals B never carry state from one
// evaluation to the next (verified by poisoning them with NaN between evaluations on the CPU restatement
// for all five code variants), so B lives only in registers; X / DW / rtY / clock are copied in and out.
#if !defined(CIT_MODEL) || !defined(CIT_DERIV) || !defined(CIT_STEP)
#error "define CIT_MODEL, CIT_DERIV and CIT_STEP"
#endif
#ifndef CIT_STEP_ATTR
#define CIT_STEP_ATTR __noinline__
#endif
static __device__ CIT_STEP_ATTR void CIT_STEP(CitCtx *gc, const double *cmd_in, double *out_arg)
{
  CitCtx lc;
#ifdef CIT_CMD_IN_MEMORY
  double y[19], f[6][19], out[12];
#else
  double y[19], f[6][19], cmd[10], out[12];
#endif      // (y as a seventh row of f -- kept in the stack frame instead of 38 registers -- measured worse: 198 against 124 scratch loads per evaluation body)
  for (int i = 0; i < 19; ++i) { lc.X[i] = gc->X[i]; y[i] = lc.X[i]; }
#ifdef CIT_DW_IN_MEMORY      // (the DAG evaluation reads the banks where it needs them and writes them in the major step: gen/citation_<v>_lane.inc)
  lc.dwm = gc->DW;
#else
  for (int i = 0; i < 29; ++i) lc.DW[i] = gc->DW[i];
#endif
  for (int i = 0; i < 4; ++i) lc.IW[i] = gc->IW[i];
#ifndef CIT_Y_IS_STATE      // (the DAG evaluation latches rtY = the first twelve states of the major step, which y[] holds anyway: twenty-four registers less across six evaluations)
  for (int i = 0; i < 12; ++i) lc.Y[i] = gc->Y[i];
#endif
#ifdef CIT_USE_HINTS
  for (int i = 0; i < (CIT_USE_HINTS + 5) / 6; ++i) lc.hint[i] = gc->hint[i];
#endif
#ifdef CIT_CMD_IN_MEMORY      // (the DAG evaluation reads the command words from the caller's array where it needs them)
  const double *cmd = cmd_in;
#else
  for (int i = 0; i < 10; ++i) cmd[i] = cmd_in[i];
#endif
  lc.t = gc->t; lc.stop_time = gc->stop_time; lc.dt = gc->dt; lc.tick = gc->tick;
  lc.ro = gc->ro; lc.t3 = gc->t3; lc.err = gc->err; lc.bslot = gc->bslot;
  const double t0 = lc.t, h = lc.dt;
  for (int s = 0; s < 6; ++s) {
    if (s > 0) {
      double hB[6];
      for (int j = 0; j < s; ++j) hB[j] = cit_ode5_B[s - 1][j] * h;
      for (int i = 0; i < 19; ++i) {
        double acc = f[0][i] * hB[0];
        for (int j = 1; j < s; ++j) acc = acc + f[j][i] * hB[j];
        lc.X[i] = acc + y[i];
      }
      lc.t = (s == 5) ? lc.stop_time : (s == 1 ? hB[0] + t0 : h * cit_ode5_A[s - 1] + t0);
    }
    lc.major = (s == 0) ? 1 : 0;
#ifndef CIT_NO_AXES
    cit_axes_prepare(&lc.ax, lc.X[4], lc.X[5], lc.X[6], lc.X[7], lc.X[8]);
#endif
    CIT_MODEL(&lc, cmd, out);   // s == 0: stop_time = (tick+1)*dt; rtY latch; Derivative-block banks
    CIT_DERIV(&lc, f[s]);
  }
  {
    double hB[6];
    for (int j = 0; j < 6; ++j) hB[j] = cit_ode5_B[5][j] * h;
    for (int i = 0; i < 19; ++i) {
      double acc = f[0][i] * hB[0];
      for (int j = 1; j < 6; ++j) acc = acc + f[j][i] * hB[j];
      gc->X[i] = acc + y[i];
    }
  }
#ifndef CIT_DW_IN_MEMORY
  for (int i = 0; i < 29; ++i) gc->DW[i] = lc.DW[i];
#endif
  for (int i = 0; i < 4; ++i) gc->IW[i] = lc.IW[i];
#ifdef CIT_Y_IS_STATE
  for (int i = 0; i < 12; ++i) { gc->Y[i] = y[i]; out_arg[i] = y[i]; }
#else
  for (int i = 0; i < 12; ++i) { gc->Y[i] = lc.Y[i]; out_arg[i] = lc.Y[i]; }
#endif
#ifdef CIT_USE_HINTS
  for (int i = 0; i < (CIT_USE_HINTS + 5) / 6; ++i) gc->hint[i] = lc.hint[i];
#endif
  gc->major = 1;
  gc->err = lc.err;
  gc->tick = lc.tick + 1;
  gc->stop_time = lc.stop_time;
  gc->t = lc.stop_time;
}
#undef CIT_MODEL
#undef CIT_DERIV
#undef CIT_STEP
#undef CIT_USE_HINTS
#undef CIT_Y_IS_STATE
#undef CIT_DW_IN_MEMORY
#undef CIT_CMD_IN_MEMORY
