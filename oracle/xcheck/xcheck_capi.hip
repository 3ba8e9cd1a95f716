// xcheck_capi.hip -- TEST INFRASTRUCTURE: C entry points of oracle/_build/libserl_xcheck.so.
// The library holds GPU kernels compiled from the LIFTED model source (oracle/gen), i.e. the checker's own
// statement-per-instruction translation of the reference binary; the product library (serl_amd/csrc) does not
// contain that source.  Only tests/ loads this library.  The caller passes the device copy of a build's tables
// (ro | t3[46] | x0[19] | dw0[31], as serl_ctx_load_build lays them out) and a serl_rollout_desc of DEVICE pointers.
#include <hip/hip_runtime.h>
#include <string.h>
#include <string>
#include "rollout_device.h"

void serl_xcheck_launch_rollout_nominal(const RolloutArgs &a, int grid, hipStream_t stream);
void serl_xcheck_launch_rollout_ice(const RolloutArgs &a, int grid, hipStream_t stream);
void serl_xcheck_launch_dyn_nominal(const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream);
void serl_xcheck_launch_dyn_ice(const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream);

static thread_local std::string g_xerr;

// ---- unit checks of product device functions (tests/test_gpu_rollout.py): arrays in, arrays out ----------------------------------
#include "citation_libm.h"
// x / c by the reciprocal + fma correction of the product (citation_libm.h: citw_div_const, the function the generated kernels call) and by
// the IEEE division
__global__ void xcheck_div_const_kernel(const double *x, int n, double c, double rc, double *fast, double *ieee)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fast[i] = citw_div_const(x[i], c, rc);
  ieee[i] = x[i] / c;
}
// kind 0: citw_sincos -> (o0, o1); 1: citw_tan -> o0; 2: citw_pow(x, c) -> o0; 3: citw_atan -> o0
__global__ void xcheck_libm_kernel(int kind, const double *x, int n, double c, double *o0, double *o1)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (kind == 0) citw_sincos(x[i], &o0[i], &o1[i]);
  else if (kind == 1) o0[i] = citw_tan(x[i]);
  else if (kind == 3) o0[i] = citw_atan(x[i]);
  else o0[i] = citw_pow(x[i], c);
}

// the actor's activation of the product (rollout_device.h: serl_act -> det_tanhf / det_expm1f_neg / LeakyReLU) on an f32 array
__global__ void xcheck_act_kernel(int act, const float *x, int n, float *y)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = serl_act(x[i], act);
}

static void xcheck_args(RolloutArgs &a, const double *blob, int n_ro, double dt, int lanes, int waves, int *grid)
{
  a.ro = blob; a.t3 = blob + n_ro; a.x0 = a.t3 + 46; a.dw0 = a.x0 + 19;
  a.dyn_dt = dt;
  a.prof = nullptr;
  a.lanes = lanes;
  const int wpb = waves <= 256 ? 1 : 4;      // one table copy in LDS per workgroup; four wavefronts share it beyond one per CU
  a.block = 64 * wpb;
  *grid = (waves + wpb - 1) / wpb;
}

extern "C" {

const char *serl_xcheck_last_error(void) { return g_xerr.c_str(); }

// code: 0 nominal, 1 ice (enum serl_dyn_code); lanes = episodes per wavefront, 1..64
int serl_xcheck_rollout(int code, const double *blob, int n_ro, double dt, const serl_rollout_desc *d, int lanes, void *stream)
{
  if (!blob || !d || n_ro <= 0 || lanes < 1 || lanes > 64 || (code != SERL_DYN_NOMINAL && code != SERL_DYN_ICE)) {
    g_xerr = "serl_xcheck_rollout: bad argument";
    return SERL_E_INVALID;
  }
  if (d->env_config != SERL_ENV_ATTITUDE || d->incremental) { g_xerr = "serl_xcheck_rollout: attitude task only"; return SERL_E_UNSUPPORTED; }
  RolloutArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  a.e0 = 0; a.e_end = d->n_episodes;
  int grid = 0;
  xcheck_args(a, blob, n_ro, dt, lanes, (d->n_episodes + lanes - 1) / lanes, &grid);
  if (code == SERL_DYN_NOMINAL) serl_xcheck_launch_rollout_nominal(a, grid, (hipStream_t)stream);
  else serl_xcheck_launch_rollout_ice(a, grid, (hipStream_t)stream);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_xerr = std::string("serl_xcheck_rollout: ") + hipGetErrorString(e); return SERL_E_HIP; }
  return SERL_OK;
}

// initialize() + T calls of step(cmd): cmds f64 [n_episodes][T][10] -> states f64 [n_episodes][T][12] (device)
int serl_xcheck_dyn_open_loop(int code, const double *blob, int n_ro, double dt, int n_episodes, int T, const double *cmds,
                              double *states, int lanes, void *stream)
{
  if (!blob || !cmds || !states || n_ro <= 0 || n_episodes <= 0 || T <= 0 || lanes < 1 || lanes > 64 ||
      (code != SERL_DYN_NOMINAL && code != SERL_DYN_ICE)) {
    g_xerr = "serl_xcheck_dyn_open_loop: bad argument";
    return SERL_E_INVALID;
  }
  RolloutArgs a;
  memset(&a, 0, sizeof(a));
  a.d.n_episodes = n_episodes;
  int grid = 0;
  xcheck_args(a, blob, n_ro, dt, lanes, (n_episodes + lanes - 1) / lanes, &grid);
  if (code == SERL_DYN_NOMINAL) serl_xcheck_launch_dyn_nominal(a, cmds, states, T, grid, (hipStream_t)stream);
  else serl_xcheck_launch_dyn_ice(a, cmds, states, T, grid, (hipStream_t)stream);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_xerr = std::string("serl_xcheck_dyn_open_loop: ") + hipGetErrorString(e); return SERL_E_HIP; }
  return SERL_OK;
}

int serl_xcheck_div_const(const double *x, int n, double c, double rc, double *fast, double *ieee, void *stream)
{
  hipLaunchKernelGGL(xcheck_div_const_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, n, c, rc, fast, ieee);
  return hipGetLastError() == hipSuccess ? SERL_OK : SERL_E_HIP;
}

int serl_xcheck_libm(int kind, const double *x, int n, double c, double *o0, double *o1, void *stream)
{
  hipLaunchKernelGGL(xcheck_libm_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, kind, x, n, c, o0, o1);
  return hipGetLastError() == hipSuccess ? SERL_OK : SERL_E_HIP;
}

int serl_xcheck_act(int act, const float *x, int n, float *y, void *stream)
{
  hipLaunchKernelGGL(xcheck_act_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, act, x, n, y);
  return hipGetLastError() == hipSuccess ? SERL_OK : SERL_E_HIP;
}

}  // extern "C"
