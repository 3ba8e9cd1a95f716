/* rollout_ref.c -- ORACLE (test infrastructure): CPU restatement of one SERL episode and of the GA
 * population-evaluate loop, with the same descriptor as the product's serl_rollout() but HOST
 * pointers.  Never linked into the product.
 *
 * Follows, line by line:
 *   base/core/agent.py:63-138        Agent.evaluate  (episode loop, fitness = sum of rewards)
 *   base/core/genetic_agent.py:104-109  Actor.select_action (obs f64 -> f32 -> MLP -> f32[3])
 *   base/core/mod_utils.py:14-18,39-50  activations ('relu' = LeakyReLU 0.01), custom LayerNorm
 *                                        gamma*(x-mean)/(std_unbiased + 1e-6) + beta
 *   envs/phlabenv.py:62-73           scale_action: 0.5*(a+1.0) in f32, then f64
 *   envs/phlabenv.py:347-399         calc_error / get_reward / get_cost / check_bounds
 *   envs/phlabenv.py:401-482         reset / step
 *   envs/{be,jr,sa,se}/citation.py:71-79   actuator faults seen by the plant only
 * Float-32 arithmetic of the actor: the fixed order include/serl_amd.h specifies (four interleaved fma partial
 * sums per dot product, pairwise LayerNorm trees; dot4 / tree_sum below), shared bit for bit with the HIP kernels.
 * torch's CPU kernels use another (vectorised) order -- agreement with the reference is to f32 rounding, measured
 * per shipped actor in tests/golden/make_sensitivity.py and asserted in tests/test_oracle_rollout.py.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "../include/serl_amd.h"

typedef struct CitInstance CitInstance;
int cit_instance_size(void);
void cit_reset(CitInstance *I, int code, const double *ro, const double *t3, const double *x0,
               const double *dw0, double dt);
int cit_step(CitInstance *I, const double *cmd, double *out);
void cit_set_clock(CitInstance *I, unsigned tick);

#define DET_FN inline
static inline double det_pow2(long long k) { union { unsigned long long u; double d; } c; c.u = (unsigned long long)(k + 1023) << 52; return c.d; }
#define DET_POW2(k) det_pow2(k)

/* tanh / expm1 of the f32 actor, evaluated in f64 with + - * / only (no libm, no FMA contraction), so that the
 * CPU oracle and the HIP kernel return bit-identical f32 activations (specified in include/serl_amd.h):
 *   z = 2|x| (tanh) or x (expm1, x <= 0);  k = round(z / ln2);  r = (z - k*LN2_HI) - k*LN2_LO;
 *   q = expm1(r) by the Taylor polynomial through r^13/13! in Estrin form (pairs, quads, octets);
 *   tanh = q/(q+2) if k == 0 else 1 - 2/(2^k (q+1) + 1);   expm1 = q if k == 0 else 2^k (q+1) - 1;
 * the f64 result (error ~1e-16) is rounded to f32 once. */
static DET_FN double det_expm1_reduced(double z, long long *kout)
{
  const double INVLN2 = 1.4426950408889634, LN2_HI = 0.6931471803691238, LN2_LO = 1.9082149292705877e-10;
  const double v = z * INVLN2;
  const long long k = v < 0.0 ? -(long long)(0.5 - v) : (long long)(v + 0.5);
  const double kd = (double)k;
  const double r = (z - kd * LN2_HI) - kd * LN2_LO;
  /* expm1(r) - r = r^2 P(r), P of degree 11 with the Taylor coefficients 1/2! .. 1/13!, in Estrin form (dependency depth
   * 7 instead of 24: a lone GPU wavefront waits out every dependent operation) */
  const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
  const double b0 = 0.5 + 0.16666666666666666 * r, b1 = 0.041666666666666664 + 0.008333333333333333 * r;
  const double b2 = 0.001388888888888889 + 0.0001984126984126984 * r, b3 = 2.48015873015873e-05 + 2.7557319223985893e-06 * r;
  const double b4 = 2.755731922398589e-07 + 2.505210838544172e-08 * r, b5 = 2.08767569878681e-09 + 1.6059043836821613e-10 * r;
  const double c0 = b0 + b1 * r2, c1 = b2 + b3 * r2, c2 = b4 + b5 * r2;
  const double p = (c0 + c1 * r4) + c2 * r8;
  *kout = k;
  return r + r2 * p;
}

static DET_FN float det_tanhf(float xf)
{
  if (xf != xf) return xf;
  const double x = (double)xf;
  const double ax = x < 0.0 ? -x : x;
  double t;
  if (ax > 20.0) t = 1.0;
  else {
    long long k;
    const double q = det_expm1_reduced(ax + ax, &k);
    if (k == 0) t = q / (q + 2.0);
    else t = 1.0 - 2.0 / (DET_POW2(k) * (q + 1.0) + 1.0);
  }
  return (float)(x < 0.0 ? -t : t);
}

static DET_FN float det_expm1f_neg(float xf)     /* x <= 0 (the ELU branch) */
{
  if (xf != xf) return xf;
  const double x = (double)xf;
  if (x < -104.0) return -1.0f;
  long long k;
  const double q = det_expm1_reduced(x, &k);
  return (float)(k == 0 ? q : DET_POW2(k) * (q + 1.0) - 1.0);
}

/* cos(pi s), 0 <= s <= 1 (include/serl_amd.h, serl_ref_spec) -- the same operations as the HIP kernels' det_cospi */
static DET_FN double det_cospi(double s)
{
  const double PI = 3.14159265358979323846;
  const int neg = s > 0.5;
  const double r = neg ? 1.0 - s : s;
  const int use_cos = r <= 0.25;
  const double x = use_cos ? PI * r : PI * (0.5 - r);
  const double x2 = x * x;
  static const double C[11] = {1.0, -0.5, 0.041666666666666664, -0.001388888888888889, 2.48015873015873e-05, -2.755731922398589e-07,
                               2.08767569878681e-09, -1.1470745597729725e-11, 4.779477332387385e-14, -1.5619206968586225e-16,
                               4.110317623312165e-19};
  static const double S[11] = {1.0, -0.16666666666666666, 0.008333333333333333, -0.0001984126984126984, 2.7557319223985893e-06,
                               -2.505210838544172e-08, 1.6059043836821613e-10, -7.647163731819816e-13, 2.8114572543455206e-15,
                               -8.22063524662433e-18, 1.9572941063391263e-20};
  double pc = C[10], ps = S[10];
  for (int k = 9; k >= 0; --k) { pc = C[k] + x2 * pc; ps = S[k] + x2 * ps; }
  const double c = use_cos ? pc : x * ps;
  return neg ? -c : c;
}

static double ref_channel(const double *tt, const double *aa, int n, double w, double t)
{
  double ti = 0.0, a = 0.0, prev = 0.0;
  int on = 0;
  for (int i = 0; i < n && i < SERL_REF_MAX_STEPS; ++i)
    if (t >= tt[i]) { prev = on ? a : 0.0; ti = tt[i]; a = aa[i]; on = 1; }
  if (!on) return 0.0;
  double s = (t - ti) / w;
  s = s < 1.0 ? s : 1.0;
  return prev + (a - prev) * (1.0 - det_cospi(s)) / 2.0;
}

static void ref_generate(const serl_ref_spec *r, double t, double t_max, double *rk)
{
  const double D2R = 3.14159265358979323846 / 180.0;
  const double th = ref_channel(r->t_theta, r->a_theta, r->n_theta, r->w_theta, t) + ((0.0 <= t && t <= t_max) ? r->trim_deg : 0.0);
  const double ph = ref_channel(r->t_phi, r->a_phi, r->n_phi, r->w_phi, t);
  rk[0] = th * D2R; rk[1] = ph * D2R; rk[2] = 0.0 * D2R;
}

static float act_f(float v, int act)
{
  switch (act) {
    case SERL_ACT_TANH: return det_tanhf(v);
    case SERL_ACT_ELU: return v > 0.0f ? v : det_expm1f_neg(v);
    default: return v > 0.0f ? v : 0.01f * v;
  }
}

/* the actor's activation on an array (unit check of the device's det_tanhf / det_expm1f_neg against this restatement, bit for bit) */
void serl_oracle_act(int act, const float *x, int n, float *y)
{
  for (int i = 0; i < n; ++i) y[i] = act_f(x[i], act);
}

/* Actor forward, f32 (genetic_agent.py:69-109).  hbuf: 3*H scratch floats. */
/* The f32 arithmetic of the actor as include/serl_amd.h specifies it (shared with the HIP kernels, bit for bit):
 *   dot product   four interleaved partial sums p[j & 3] = fmaf(w[j], h[j], p[j & 3]) over ascending j (exact products,
 *                 one rounding per step), then  bias + ((p0 + p1) + (p2 + p3))
 *   LayerNorm sum balanced pairwise tree over blocks of 16 consecutive rows (zero padded), the blocks added in order
 * Any fixed order is as faithful to the reference as any other (torch's CPU kernels use vector lanes and FMA); this one
 * has short dependency chains on a 64-lane wavefront and on a CPU alike. */
static float dot4(const float *w, const float *h, int n, float bias)
{
  float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int j = 0; j < n; ++j) p[j & 3] = fmaf(w[j], h[j], p[j & 3]);
  return bias + ((p[0] + p[1]) + (p[2] + p[3]));
}

static float tree16(const float *x, int n)        /* n <= 16 values, zero padded */
{
  float t[16];
  for (int i = 0; i < 16; ++i) t[i] = i < n ? x[i] : 0.0f;
  for (int s = 1; s < 16; s <<= 1)
    for (int i = 0; i < 16; i += 2 * s) t[i] = t[i] + t[i + s];
  return t[0];
}

static float tree_sum(const float *x, int n)
{
  float s = tree16(x, n < 16 ? n : 16);
  for (int b = 16; b < n; b += 16) s = s + tree16(x + b, n - b < 16 ? n - b : 16);
  return s;
}

static void actor_forward(const serl_rollout_desc *d, const float *w, const float *obs, float *act_out,
                          float *hbuf)
{
  const int S = d->state_dim, H = d->hidden, A = d->action_dim, L = d->num_layers;
  float *h0 = hbuf, *h1 = hbuf + H, *h2 = hbuf + 2 * H;
  const float *W = w, *b = w + (size_t)H * S;
  for (int i = 0; i < H; ++i) h0[i] = act_f(dot4(W + (size_t)i * S, obs, S, b[i]), d->activation);
  w = b + H;
  for (int l = 0; l < L; ++l) {
    const float *Wl = w, *bl = w + (size_t)H * H, *g = bl + H, *be = g + H;
    for (int i = 0; i < H; ++i) h1[i] = dot4(Wl + (size_t)i * H, h0, H, bl[i]);
    const float mean = tree_sum(h1, H) / (float)H;
    for (int i = 0; i < H; ++i) { const float dlt = h1[i] - mean; h2[i] = dlt * dlt; }
    const float var = tree_sum(h2, H);
    const float std = sqrtf(var / (float)(H - 1));
    const float den = std + 1e-6f;
    for (int i = 0; i < H; ++i) h0[i] = act_f(g[i] * (h1[i] - mean) / den + be[i], d->activation);
    w = be + H;
  }
  const float *Wo = w, *bo = w + (size_t)A * H;
  for (int i = 0; i < A; ++i) act_out[i] = det_tanhf(dot4(Wo + (size_t)i * H, h0, H, bo[i]));
}

int serl_param_count(int S, int H, int L, int A) { return H * S + H + L * (H * H + 3 * H) + A * H + A; }

static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* envs/noise/citation.py:71-82 (== envs/gust/citation.py:72-86): out[:3] += .., out[4] += .., out[5] += .., out[6:8] += ..;
 * the seven addends (bias + sd * randn, drawn by the caller in that order) arrive pre-computed */
static void add_sensor_noise(double *x, const double *sn)
{
  x[0] += sn[0]; x[1] += sn[1]; x[2] += sn[2]; x[4] += sn[3]; x[5] += sn[4]; x[6] += sn[5]; x[7] += sn[6];
}

/* envs/phlabenv.py:84-97 (obs_idx per configuration), :213-220 (n_obs), :415-428 / :462-468 (observation) */
int serl_env_action_dim(int env_config) { return env_config == SERL_ENV_SYMMETRIC ? 1 : 3; }
int serl_env_state_dim(int env_config, int incremental)
{
  if (env_config < 0 || env_config > 2) return 0;
  const int A = serl_env_action_dim(env_config);
  const int nx = env_config == SERL_ENV_ATTITUDE ? 4 : (env_config == SERL_ENV_SYMMETRIC ? 1 : 10);
  return A + nx + (incremental ? A : 0);
}

static int build_obs(double *obs, int cfg, int incr, int A, const double *err, const double *x, const double *last_u)
{
  int n = 0;
  for (int i = 0; i < A; ++i) obs[n++] = err[i];
  if (cfg == SERL_ENV_ATTITUDE) { obs[n++] = x[0]; obs[n++] = x[1]; obs[n++] = x[2]; obs[n++] = x[4]; }
  else if (cfg == SERL_ENV_SYMMETRIC) obs[n++] = x[1];
  else for (int i = 0; i < 10; ++i) obs[n++] = x[i];
  if (incr) for (int i = 0; i < A; ++i) obs[n++] = last_u[i];
  return n;
}

static int run_episode(const serl_rollout_desc *d, const serl_build_desc *bd, int e, CitInstance *I, float *hbuf)
{
  const double PI = 3.14159265358979323846;
  const double deg2rad = PI / 180.0, rad2deg = 180.0 / PI;
  const int cfg = d->env_config, incr = d->incremental != 0;
  const int A = serl_env_action_dim(cfg), S = serl_env_state_dim(cfg, incr);
  if (S == 0 || S != d->state_dim || A != d->action_dim) return SERL_E_INVALID;
  const double bound = (incr ? 25.0 : 10.0) * deg2rad; /* phlabenv.py:205-208: rate bound [rad/s] / deflection bound [rad] */
  const double low = -bound, high = bound;
  const double max_theta = 60.0 * deg2rad, max_phi = 75.0 * deg2rad;   /* :211-212 */
  const double scaler[3] = {6.0 / PI * 1.0, 6.0 / PI * 1.0, 6.0 / PI * 4.0};  /* :226-232 (one action: [1] * 6/pi) */
  const double dt = 0.01;                              /* phlabenv.py:81 (class attribute) */
  const serl_fault_row nominal = {1.0, INFINITY, INFINITY, 0.0, 0.0, 0, 0, 0};
  const serl_fault_row *f = d->faults ? &d->faults[e] : &nominal;
  const float *w = d->weights + (size_t)d->member_of_episode[e] * d->weight_stride;
  const double *ref = d->ref ? d->ref + (size_t)e * d->ref_stride : NULL;
  const serl_ref_spec *rspec = d->ref_spec ? d->ref_spec + (size_t)e * d->ref_spec_stride : NULL;
  const double *snoise = NULL;                         /* this episode's row of pre-drawn sensor noise, if any */
  if (d->sensor_noise) {
    const int sr = d->sensor_row ? d->sensor_row[e] : e;
    if (sr >= 0) snoise = d->sensor_noise + (size_t)sr * ((size_t)d->max_steps + 1) * 7;
  }
  const double *noise = NULL;                          /* this episode's row of pre-drawn action noise, if any */
  if (d->action_noise) {
    const int nr = d->noise_row ? d->noise_row[e] : e;
    if (nr >= 0) noise = d->action_noise + (size_t)nr * d->max_steps * 3;
  }
  double cmd[10], x[12], err[3] = {0, 0, 0}, obs[16], nobs[16];
  double u[3] = {0, 0, 0};                             /* env.last_u (phlabenv.py:412); incremental control integrates it */
  float obsf[16], a[3] = {0, 0, 0};
  const int TW = 2 * S + A + 3;                        /* floats per stored transition */

  /* reset (phlabenv.py:401-428) */
  cit_reset(I, bd->code, bd->ro, bd->t3, bd->x0, bd->dw0, bd->dt);
  if (d->tick0) cit_set_clock(I, (unsigned)d->tick0[e]);   /* initialize() leaves the model clock running */
  memset(cmd, 0, sizeof(cmd));
  cmd[0] = clipd(cmd[0] * f->elev_gain, -f->elev_clip, f->elev_clip);
  cmd[1] = clipd(cmd[1], -f->ail_clip, f->ail_clip);
  if (f->rudder_jam_on != 0.0) cmd[2] = f->rudder_jam;
  cit_step(I, cmd, x);
  if (snoise) add_sensor_noise(x, snoise);             /* envs/noise/citation.py:71-82 on the value step() returns */
  const double V0 = x[3];
  double t = 0.0;
  if (d->err0) for (int i = 0; i < A; ++i) err[i] = d->err0[(size_t)e * 3 + i];
  build_obs(obs, cfg, incr, A, err, x, u);

  double fitness = 0.0;
  int k = 0, cost_steps = 0, done = 0;
  while (!done) {
    if (k >= d->max_steps) return SERL_E_INVALID;     /* reference table too short */
    for (int i = 0; i < S; ++i) obsf[i] = (float)obs[i];
    actor_forward(d, w, obsf, a, hbuf);
    double sc[3] = {0, 0, 0};
    if (noise) {
      /* agent.py:90-93: f32 action + f64 noise -> f64, clipped; scale_action then runs in f64 */
      for (int i = 0; i < A; ++i) {
        double an = clipd((double)a[i] + noise[(size_t)k * 3 + i], -1.0, 1.0);
        sc[i] = low + 0.5 * (an + 1.0) * (high - low);
        a[i] = (float)an;                              /* agent.py:93,103: the transition holds the executed action */
      }
    } else {
      for (int i = 0; i < A; ++i) {
        float s = 0.5f * (a[i] + 1.0f);               /* phlabenv.py:72-73, f32 part */
        sc[i] = low + (double)s * (high - low);
      }
    }
    /* phlabenv.py:377-380,446-450: incremental control integrates the commanded rate */
    for (int i = 0; i < A; ++i) u[i] = incr ? u[i] + sc[i] * dt : sc[i];
    memset(cmd, 0, sizeof(cmd));
    cmd[0] = clipd(u[0] * f->elev_gain, -f->elev_clip, f->elev_clip);
    cmd[1] = clipd(u[1], -f->ail_clip, f->ail_clip);
    cmd[2] = (f->rudder_jam_on != 0.0) ? f->rudder_jam : u[2];
    cit_step(I, cmd, x);
    if (snoise) add_sensor_noise(x, snoise + (size_t)(k + 1) * 7);
    /* reward (phlabenv.py:347-367): reference at the pre-increment time */
    double rk[3];
    if (rspec) ref_generate(rspec, t, d->t_max, rk);
    else { rk[0] = ref[(size_t)k * 3]; rk[1] = ref[(size_t)k * 3 + 1]; rk[2] = ref[(size_t)k * 3 + 2]; }
    const double ctrl[3] = {x[7], x[6], x[5]};         /* :347-350 theta, phi, beta */
    for (int i = 0; i < A; ++i) err[i] = rk[i] - ctrl[i];
    double rsum = 0.0;
    for (int i = 0; i < A; ++i) rsum = rsum + fabs(clipd(scaler[i] * err[i], -1.0, 1.0));
    double reward = -rsum / (double)A;
    /* cost (phlabenv.py:369-375; degrees compared with 0.75*max_phi in radians -- reference quirk) */
    int cost = (rad2deg * fabs(x[4]) > 11.0) || (rad2deg * fabs(x[6]) > 0.75 * max_phi) || (x[3] < V0 / 3.0);
    build_obs(nobs, cfg, incr, A, err, x, u);
    /* bounds (phlabenv.py:391-399) */
    done = (t >= d->t_max) || (fabs(x[7]) > max_theta) || (fabs(x[6]) > max_phi) || (x[9] < 50.0);
    if (done) reward += -1.0 / dt * (d->t_max - t) * 2.0;
    t += dt;
    if (d->actions) for (int i = 0; i < 3; ++i) d->actions[((size_t)e * d->max_steps + k) * 3 + i] = u[i];
    if (d->states) for (int i = 0; i < 12; ++i) d->states[((size_t)e * d->max_steps + k) * 12 + i] = x[i];
    if (d->rewards) d->rewards[(size_t)e * d->max_steps + k] = reward;
    if (d->transitions) {
      float *tr = d->transitions + ((size_t)e * d->max_steps + k) * TW;
      for (int i = 0; i < S; ++i) tr[i] = (float)obs[i];
      for (int i = 0; i < A; ++i) tr[S + i] = a[i];
      for (int i = 0; i < S; ++i) tr[S + A + i] = (float)nobs[i];
      tr[2 * S + A] = (float)reward; tr[2 * S + A + 1] = done ? 1.0f : 0.0f; tr[2 * S + A + 2] = cost ? 1.0f : 0.0f;
    }
    fitness += reward;
    cost_steps += cost;
    memcpy(obs, nobs, sizeof(obs));
    ++k;
  }
  d->fitness[e] = fitness;
  d->length_steps[e] = k;
  d->length_t[e] = t;
  d->cost_steps[e] = cost_steps;
  return 0;
}

typedef struct { const serl_rollout_desc *d; const serl_build_desc *bd; int e0, e1, stride, rc; } job_t;

static void *worker(void *p)
{
  job_t *j = (job_t *)p;
  CitInstance *I = (CitInstance *)malloc((size_t)cit_instance_size());
  float *hbuf = (float *)malloc(sizeof(float) * 3 * (size_t)j->d->hidden);
  for (int e = j->e0; e < j->e1; e += j->stride) {
    int rc = run_episode(j->d, j->bd, e, I, hbuf);
    if (rc) j->rc = rc;
  }
  free(hbuf); free(I);
  return NULL;
}

/* Same descriptor as serl_rollout (include/serl_amd.h) with host pointers; `threads` host threads
 * split the episodes round-robin (each thread owns a private dynamics instance). */
int serl_oracle_rollout(const serl_build_desc *bd, const serl_rollout_desc *d, int threads)
{
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t th[256]; job_t jobs[256];
  for (int i = 0; i < threads; ++i) {
    jobs[i] = (job_t){d, bd, i, d->n_episodes, threads, 0};
    if (threads == 1) worker(&jobs[i]);
    else pthread_create(&th[i], NULL, worker, &jobs[i]);
  }
  int rc = 0;
  for (int i = 0; i < threads; ++i) {
    if (threads > 1) pthread_join(th[i], NULL);
    if (jobs[i].rc) rc = jobs[i].rc;
  }
  return rc;
}
