// citation_dev.h -- device-side runtime of the PH-LAB Citation dynamics for gfx950 (product code).
//
// The reference ships the aircraft model only as a compiled Simulink/Embedded-Coder library
// (envs/<build>/_citation.cpython-38-x86_64-linux-gnu.so).  The model-outputs
// function is generated from that binary by tools/lift (gen/citation_<variant>.inc: one statement
// per x86-64 instruction, IEEE-754 order preserved, all memory references resolved to named
// regions); this header supplies what the generated code is written against and the hand-written
// pieces the lifter cuts out:
//   cit_table3 / cit_table2   `table3` S-function  mdlOutputs @0x10da0, Table2 @0x10a30
//   CitStepper<>::step        the ERT step(): major-step outputs, inlined
//                             rt_ertODEUpdateContinuousStates (Dormand-Prince ode5, @0x9eb8..0xa4f3),
//                             clock update
//   cit_reset                 the effect of initialize() @0xb4e0 (state images captured from the
//                             live library; serl_build_desc in include/serl_amd.h)
// Per-lane state lives in a CitCtx in private memory; read-only tables are shared by all lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "citation_libm.h"

#ifndef CIT_MAX_NB
#define CIT_MAX_NB 648
#endif
#define CIT_SINCOS(x, s, c_) citw_sincos((x), (s), (c_))
#include "citation_leaves.h"

// Read-only model tables (.rodata words [RO_LO_W, RO_HI_W) of the build: aero tables rtConstP, rtConstB,
// literal pool; ~94 KiB) are staged once per workgroup into LDS and shared by its wavefronts.
#define CIT_RO_LDS_WORDS 12040
__shared__ double g_ro[CIT_RO_LDS_WORDS];

struct CitCtx {
  double X[19];          // p q r V alpha beta phi theta psi he xe ye | washout, 2 consts, 4 engine states
  double B[CIT_MAX_NB];  // block signals (DWARF struct B)
  double DW[29];         // Derivative-block banks: TimeStampA, LastUAtTimeA[12], TimeStampB, LastUAtTimeB[12]; RWORK[3]
  int32_t IW[4];         // table3 IWORK[3]
  double Y[12];          // rtY, latched in the major step before integration
  double t, stop_time, dt;
  int32_t major;         // simTimeStep: 1 major, 0 minor
  uint32_t tick;         // clockTick0
  const double *ro;      // unused on the device (tables are staged in LDS: g_ro)
  const double *t3;      // table3 parameters P1[3] P2[4] P3[3] P4[36]
  CitAxes ax;            // trigonometry + rotation matrices of the current model evaluation
  int32_t bslot;         // first word of this episode's window in g_B (LDS flavour)
  int32_t err;           // CIT_ERR_* flags
  double *dwm;           // (lane-per-episode DAG kernels) the Derivative-block banks of the CALLER's context: the evaluation reads / writes them there (citation_step_dev.h CIT_DW_IN_MEMORY)
  uint32_t hint[8];      // (lane-per-episode DAG kernels) interval every index search found in the previous model evaluation, five bits each, six to a word (citation_leaves.h cit_lookup_index_h)
};

// inlining policy: the model body, derivatives and the S-function bodies are inlined into the step
// function so that the block-signal array B (a local there) is promoted to registers; table lookups and
// rt_powd_snf stay out of line (they only read the shared tables).
#define LIFT_INLINE static __device__ __forceinline__
#define LIFT_OUTLINE static __device__ __noinline__
#define LIFT_OMIT_rt_GetLookupIndex   // -> citation_leaves.h
#define LIFT_OMIT_rt_Lookup
#define LIFT_OMIT_rt_Lookup2D_Normal
#define LIFT_OMIT_matmultiply
#define LIFT_OMIT_ac_axes
#define LIFT_FN_rt_powd_snf LIFT_OUTLINE
#define LIFT_FN_ac_atmos LIFT_INLINE
#define LIFT_FN_derivatives LIFT_INLINE
#define LIFT_FN_model LIFT_INLINE
static __device__ __forceinline__ uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }
static __device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long)u); }

#define RO_D(o)    (g_ro[((uint64_t)(o) >> 3) - CIT_RO_LO_W])
// Block signals B (647 f64 per model instance) have two homes, chosen per kernel flavour (CIT_B_LDS):
//   registers  -- B is a function-local array of the step function, promoted to SSA values; what does not fit
//                 the 512-register budget spills to scratch.  Only choice when a wavefront carries many episodes.
//   LDS        -- one 5 KiB window per episode; used when a wavefront carries <= SERL_LDS_B_LANES_PER_WAVE
//                 episodes (the latency-bound regime): an LDS access costs ~64 cycles where a scratch spill
//                 costs a trip through the vector memory path.
#define SERL_LDS_B_LANES_PER_WAVE 2
#define SERL_LDS_B_SLOTS (4 * SERL_LDS_B_LANES_PER_WAVE)
__shared__ double g_B[SERL_LDS_B_SLOTS * CIT_MAX_NB];
#define B_D(o)     CIT_B_AT((uint64_t)(o) >> 3)
#define X_D(o)     (c->X[(uint64_t)(o) >> 3])
#define DW_D(o)    (c->DW[(uint64_t)(o) >> 3])
#define Y_D(o)     (c->Y[(uint64_t)(o) >> 3])
#define CMD_D(o)   (((double *)cmd)[(uint64_t)(o) >> 3])
#define OUT_D(o)   (out[(uint64_t)(o) >> 3])
#define SU_D(o)    (((double *)su)[(uint64_t)(o) >> 3])
#define SY_D(o)    (sy[(uint64_t)(o) >> 3])
#define MX0_D(o)   ((double)mode)
#define TPTR_D(o)  (c->t)
#define XDOT_D(o)  (xdot[(uint64_t)(o) >> 3])
#define RTINF_D(o) (__longlong_as_double(0x7ff0000000000000LL))
#define RTMINF_D(o) (__longlong_as_double((long long)0xfff0000000000000ULL))
#define RTNAN_D(o) (__longlong_as_double(0x7ff8000000000000LL))
#define STK_D(o)   (stk_d[(uint64_t)(o) >> 3])
#define STK_I(o)   (stk_i[(uint64_t)(o) >> 3])
#define STK_W(o)   (stk_w[(uint64_t)(o) >> 2])
#define P_rdi_D(o) (p_rdi[(int64_t)(o) >> 3])
#define P_rsi_D(o) (p_rsi[(int64_t)(o) >> 3])
#define P_rdx_D(o) (p_rdx[(int64_t)(o) >> 3])
#define P_r8_D(o)  (p_r8[(int64_t)(o) >> 3])
#define M_I32(o)   M_I32_##o
#define M_D(o)     M_D_##o
// table lookups: per-breakpoint-vector index caches are locals of the model function (LIFT_IDX_DECL)
#define LIFT_IDX_DECL(a) double ixu_##a = 0.0; int ixi_##a = -1;
#define LIFT_L2D(fn, xr, nr, xc, nc, z, u0, u1)                                                        \
  cit_lookup2d_at(&RO_D(xr), (nr), &RO_D(xc), &RO_D(z),                                                \
                  cit_lookup_index_cached(&RO_D(xr), (nr), (u0), &ixu_##xr, &ixi_##xr),               \
                  cit_lookup_index_cached(&RO_D(xc), (nc), (u1), &ixu_##xc, &ixi_##xc), (u0), (u1))
#define LIFT_L1D(fn, x, n, u, y) \
  cit_lookup1d_at(&RO_D(x), cit_lookup_index_cached(&RO_D(x), (n), (u), &ixu_##x, &ixi_##x), (u), &RO_D(y))
#define LIFT_SQRT(x) sqrt(x)
#define LIFT_POW(x, y) citw_pow(x, y)
#define LIFT_EXP(x) exp(x)
#define LIFT_LOG10(x) log10(x)
#define LIFT_LOG(x) log(x)
#define LIFT_SIN(x) citw_sin(x)
#define LIFT_COS(x) citw_cos(x)
#define LIFT_TAN(x) citw_tan(x)
#define LIFT_ATAN(x) citw_atan(x)
#define LIFT_ATAN2(x, y) atan2(x, y)
#define LIFT_ASIN(x) asin(x)
#define LIFT_ACOS(x) acos(x)
#define LIFT_FLOOR(x) floor(x)
#define LIFT_ISNAN(x) ((x) != (x))
#define LIFT_ISINF(x) (((x) == RTINF_D(0)) || ((x) == RTMINF_D(0)))
#define LIFT_SINCOS(x, s, c_) citw_sincos((x), (s), (c_))

// ---- table3 S-function: 3-D table, linear interpolation (mdlOutputs @0x10da0) --------------------
// The reference walks linearly from the interval cached in IWORK/RWORK; the interval it ends on is
// i = clamp(max{i : tab[i] < x}, 0, n-2) whatever the cache holds, so only the observable part of the
// cache (RWORK/IWORK values, which are part of rtDW) is kept.
static __device__ __forceinline__ int cit_t3_search(const double *tab, int n, double x, double *rwork, int32_t *iwork)
{
  int idx = *iwork, interval;
  if (x >= *rwork) {
    while (idx < n && tab[idx] < x) ++idx;
    if (idx >= n) { *iwork = n - 1; interval = n - 2; }
    else if (idx < 0) { *iwork = 0; interval = 0; }
    else { *iwork = idx; interval = (idx == 0) ? 0 : idx - 1; }
  } else {
    while (idx >= 0 && !(x > tab[idx])) --idx;
    if (idx < 0) { *iwork = 0; interval = 0; }
    else if (idx == n) { *iwork = n - 1; interval = n - 2; }
    else { *iwork = idx; interval = idx; }
  }
  *rwork = x;
  return interval;
}

// Table2 @0x10a30: bilinear interpolation on one slab, tab[(ix+k)*M + j]
static __device__ __forceinline__ double cit_table2(const double *xtab, const double *ytab, int ix, int iy,
                                           const double *tab, int M, double x, double y)
{
  double rows[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    double v0 = tab[(ix + 0) * M + (iy + j)], v1 = tab[(ix + 1) * M + (iy + j)];
    double x1 = xtab[ix + 1], x0 = xtab[ix];
    if (x == x1) rows[j] = v1;
    else { double d = v1 - v0, dx = x1 - x0, w = x - x0; d = d * w; d = d / dx; rows[j] = d + v0; }
  }
  double y1 = ytab[iy + 1];
  if (y == y1) return rows[1];
  double y0 = ytab[iy];
  double w = y - y0, d = rows[1] - rows[0], dy = y1 - y0;
  d = d * w; d = d / dy;
  return d + rows[0];
}

static __device__ __forceinline__ void cit_table3(CitCtx *c, const double *u0, const double *u1, const double *u2, double *y)
{
  const double *P1 = c->t3, *P2 = c->t3 + 3, *P3 = c->t3 + 7, *P4 = c->t3 + 10;
  int i0 = cit_t3_search(P1, 3, *u0, &c->DW[26], &c->IW[0]);
  int i1 = cit_t3_search(P2, 4, *u1, &c->DW[27], &c->IW[1]);
  int i2 = cit_t3_search(P3, 3, *u2, &c->DW[28], &c->IW[2]);
  const double *slab = P4 + i2 * 12;
  double a = cit_table2(P1, P2, i0, i1, slab, 4, *u0, *u1);
  double b = cit_table2(P1, P2, i0, i1, slab + 12, 4, *u0, *u1);
  double z = *u2, z1 = P3[i2 + 1], z0 = P3[i2];
  if (z == z1) { *y = b; return; }
  double w = z - z0, dz = z1 - z0, d = b - a;
  d = d * w; d = d / dz;
  *y = d + a;
}

// effect of initialize(): images captured from the live reference library
static __device__ inline void cit_reset(CitCtx *c, const double *ro, const double *t3, const double *x0,
                                        const double *dw0, double dt)
{
  for (int i = 0; i < 19; ++i) c->X[i] = x0[i];
  for (int i = 0; i < 29; ++i) c->DW[i] = dw0[i];
  const int32_t *iw = (const int32_t *)(dw0 + 29);
  c->IW[0] = iw[0]; c->IW[1] = iw[1]; c->IW[2] = iw[2]; c->IW[3] = 0;
  for (int i = 0; i < 12; ++i) c->Y[i] = 0.0;
  c->ro = ro; c->t3 = t3; c->dt = dt; c->major = 1; c->tick = 0; c->t = 0.0; c->stop_time = 0.0; c->err = 0;
  for (int i = 0; i < 8; ++i) c->hint[i] = 0u;
}

// Dormand-Prince "ode5" tableau as the reference's literal pool holds it (0x13688, 0x13898..0x13938)
__device__ static const double cit_ode5_A[6] = {0.2, 0.3, 0.8, 0.8888888888888888, 1.0, 1.0};
__device__ static const double cit_ode5_B[6][6] = {
  {0.2, 0, 0, 0, 0, 0},
  {0.075, 0.225, 0, 0, 0, 0},
  {0.9777777777777777, -3.7333333333333334, 3.5555555555555554, 0, 0, 0},
  {2.9525986892242035, -11.595793324188385, 9.822892851699436, -0.2908093278463649, 0, 0},
  {2.8462752525252526, -10.757575757575758, 8.906422717743473, 0.2784090909090909, -0.2735313036020583, 0},
  {0.09114583333333333, 0.0, 0.44923629829290207, 0.6510416666666666, -0.322376179245283, 0.13095238095238096},
};
