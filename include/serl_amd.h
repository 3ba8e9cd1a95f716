/* serl_amd.h -- C ABI of the MI355X-native population-rollout fitness evaluator for SERL.
 *
 * Drop-in boundary.  The reference has no FFI for this path: its seam is Python
 * (`Agent.evaluate`, base/core/agent.py:63-138, driven by the GA loop at :229-256) on top of the
 * SWIG surface of its native dynamics library,
 *       void initialize(void);  void step(real_T *cmd, real_T *out);  void ..._terminate(void);
 * (DWARF citation_to_python.h:1600-1604; envs/h2000_v90/citation.py:65-72).  This library replaces
 * that whole inner loop -- actor MLP forward (base/core/genetic_agent.py:104-109,
 * base/core/mod_utils.py:39-50) -> CitationEnv.step (envs/phlabenv.py:430-482) -> native step() ->
 * reward/cost/bounds (envs/phlabenv.py:347-399) -- with one fused HIP kernel batched over episodes.
 *
 * Conventions: every entry point returns 0 on success or a negative SERL_E_* code and never
 * throws; `serl_last_error()` returns a thread-local message.  All array arguments of
 * `serl_rollout` / `serl_ga_*` are DEVICE pointers owned by the caller (PyTorch allocates them);
 * the context owns only the read-only model tables it uploaded.  Calls are asynchronous on the
 * `hipStream_t` passed as `void *stream` (NULL = default stream); the caller synchronises.
 * The CPU oracle (oracle/rollout_ref.c, test infrastructure) implements `serl_oracle_rollout` with
 * the same descriptor and HOST pointers.
 */
#ifndef SERL_AMD_H
#define SERL_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SERL_ABI_VERSION 8

enum serl_error {
  SERL_OK = 0,
  SERL_E_INVALID = -1,      /* bad argument / descriptor */
  SERL_E_HIP = -2,          /* HIP runtime error (message in serl_last_error) */
  SERL_E_UNSUPPORTED = -3,  /* network shape / build variant not compiled in */
  SERL_E_NOMEM = -4
};

/* activation_actor of base/core/mod_utils.py:14-18 ('relu' IS LeakyReLU(0.01) there) */
enum serl_activation { SERL_ACT_TANH = 0, SERL_ACT_ELU = 1, SERL_ACT_LEAKY_RELU = 2 };

/* code variants of the dynamics library (14 build dirs = 5 code variants) */
enum serl_dyn_code { SERL_DYN_NOMINAL = 0, SERL_DYN_ICE = 1, SERL_DYN_CG_TIMED = 2, SERL_DYN_GUST = 3,
                     SERL_DYN_TEST = 4 };

typedef struct serl_ctx serl_ctx;

/* Tables + post-initialize() state image of ONE dynamics build (what the reference's
 * initialize() @0xb4e0 sets up).  HOST pointers; copied by serl_ctx_load_build. */
typedef struct serl_build_desc {
  int32_t code;            /* enum serl_dyn_code */
  int32_t n_ro;            /* number of f64 in ro[] */
  uint64_t ro_base;        /* virtual address of ro[0] in the reference binary (.rodata start) */
  const double *ro;        /* .rodata as f64: rtConstP aero tables, rtConstB, literal pool */
  const double *t3;        /* table3 S-function parameters P1[3] P2[4] P3[3] P4[36] = 46 */
  const double *x0;        /* rtX after initialize(): 19 continuous states */
  const double *dw0;       /* rtDW after initialize(): 31 f64 (IWORK in the last 12 bytes) */
  double dt;               /* fixed step, 0.01 */
} serl_build_desc;

/* Per-episode actuator-fault row (envs/{be,jr,sa,se}/citation.py:71-79): what the PLANT sees is
 *   cmd[0] = clip(cmd[0] * elev_gain, -elev_clip, +elev_clip)
 *   cmd[1] = clip(cmd[1], -ail_clip, +ail_clip)
 *   cmd[2] = rudder_jam_on ? rudder_jam : cmd[2]
 * while the logged action (env.last_u) stays the commanded one.  Nominal row: {1, +inf, +inf, 0, 0}. */
typedef struct serl_fault_row {
  double elev_gain, elev_clip, ail_clip, rudder_jam_on, rudder_jam, pad0, pad1, pad2;
} serl_fault_row;

/* A reference signal of one episode as PARAMETERS instead of a table: theta and phi are
 * `signals.SmoothedStepSequence(times, amplitudes, smooth_width)` objects (call sites envs/phlabenv.py:303-345,
 * base/evaluate.py:173-180, base/evaluation_utils.py:51), beta is Const(0): at the env's accumulated time t
 *   i = last index with t >= times[i] (none: the channel is 0);  prev = i ? amps[i-1] : 0;  s = min((t - times[i]) / w, 1)
 *   v = prev + (amps[i] - prev) * (1 - cos(pi * s)) / 2                                      [degrees]
 *   ref = (pi/180) * [v_theta + (0 <= t <= t_max ? trim_deg : 0),  v_phi,  0]
 * cos(pi s) is evaluated with + - * only (det_cospi: reflection to [0, 1/4], Taylor polynomials of cos / sin through
 * x^20 / x^21 in Horner form) so that the kernels and the CPU oracle agree bit for bit; against a libm cosine the
 * reference value differs by <= 2 ulp.  Per env step this replaces the 24 B read of ref[k] by arithmetic. */
#define SERL_REF_MAX_STEPS 8
typedef struct serl_ref_spec {
  int32_t n_theta, n_phi;            /* number of steps of each channel, <= SERL_REF_MAX_STEPS */
  double w_theta, w_phi;             /* smooth widths, seconds */
  double trim_deg;                   /* theta trim (envs/phlabenv.py:202,319-320,344) */
  double t_theta[SERL_REF_MAX_STEPS], a_theta[SERL_REF_MAX_STEPS];
  double t_phi[SERL_REF_MAX_STEPS], a_phi[SERL_REF_MAX_STEPS];
} serl_ref_spec;

typedef struct serl_rollout_desc {
  /* -- actor network (base/core/genetic_agent.py:69-102).  f32 arithmetic, part of this ABI (the HIP kernels and the CPU
   *    oracle implement it bit for bit): a dot product is four interleaved partial sums p[j & 3] = fmaf(w[j], h[j],
   *    p[j & 3]) over ascending j, then bias + ((p0 + p1) + (p2 + p3)); a LayerNorm sum is a balanced pairwise tree per
   *    block of 16 consecutive rows (zero padded), the blocks added in order; mean = sum / H, std = sqrtf(sum((x-mean)^2)
   *    / (H - 1)), y = gamma * (x - mean) / (std + 1e-6f) + beta; tanh / ELU through det_tanhf / det_expm1f_neg
   *    (f64 + - x / only, rounded to f32 once).
   *    Linear(S,H) act, L x [Linear(H,H)
   *    LayerNorm(H) act], Linear(H,A) tanh.  weights: f32 [n_members][param_count], packed in
   *    state_dict order: W0[H][S] b0[H]  { Wl[H][H] bl[H] gamma_l[H] beta_l[H] } x L  Wo[A][H] bo[A] */
  int32_t state_dim, action_dim, hidden, num_layers, activation;
  int32_t n_members;
  const float *weights;
  int64_t weight_stride;            /* in floats, >= param_count */
  /* -- episodes */
  int32_t n_episodes;
  int32_t build_slot;               /* slot given to serl_ctx_load_build */
  const int32_t *member_of_episode; /* [n_episodes] */
  const serl_fault_row *faults;     /* [n_episodes] or NULL (nominal) */
  const double *ref;                /* reference signals, radians, sampled at the accumulated env
                                       time t_k (envs/phlabenv.py:347-349,473): [.., max_steps, 3] */
  int64_t ref_stride;               /* doubles between consecutive episodes' tables (0 = shared) */
  const double *err0;               /* [n_episodes][3] tracking error carried into obs0
                                       (envs/phlabenv.py:401-428 never clears self.error) or NULL=0 */
  const double *action_noise;       /* [rows][max_steps][3] pre-drawn clipped Gaussian noise
                                       (base/core/agent.py:90-93) or NULL */
  const int32_t *noise_row;         /* [n_episodes] row of action_noise an episode adds to its actions, -1 = none
                                       (a population evaluation and the RL actor's exploration episode share one
                                       launch); NULL = row e for every episode */
  const double *sensor_noise;       /* [rows][max_steps + 1][7] additive sensor noise of the `noise` / `gust` wrappers
                                       (envs/noise/citation.py:71-82, envs/gust/citation.py:72-86): bias + sd * randn,
                                       pre-drawn in the wrapper's draw order, for the channels p q r | alpha | beta |
                                       phi theta of what step() RETURNS (the plant state stays clean); entry 0 belongs
                                       to the step reset() takes, entry k + 1 to env step k.  NULL = none */
  const int32_t *sensor_row;        /* [n_episodes] row of sensor_noise, -1 = none; NULL = row e for every episode */
  const int32_t *tick0;             /* [n_episodes] model clock (clockTick0) the episode starts with, or NULL = 0.
                                       The reference's initialize() @0xb4e0 resets the states but NOT the model
                                       clock (rtM clockTick0/1 and t keep counting across episodes of one process;
                                       probed on the live library), which only matters to the time-switched builds
                                       cg_timed ("CG aft after 20 s") and gust: there every episode after the first
                                       of a process starts at tick0 = steps simulated so far (incl. one per reset) */
  double t_max;                     /* 80 (eval) / 20 (train) seconds */
  int32_t max_steps;                /* rows in ref / trace buffers; 8001 for t_max = 80 */
  int32_t lanes_per_wave;           /* 0 = auto: wave-cooperative kernels chosen from the episode count -- a team of eight
                                       wavefronts per episode while every episode can have a CU of its own (episodes <=
                                       CUs), two / four episodes per team up to 4 x CUs (hidden 32; other hidden sizes: two
                                       per team up to 2 x CUs, six team + two actor wavefronts), beyond that ONE launch of
                                       four-episode teams with a work queue (a lane group takes the next episode when its
                                       own ends) -- and from 80 x CUs episodes on (hidden 32) the lane-per-episode kernels with 64 episodes per wavefront --
                                       or one wavefront per episode; 1..64 = lane-per-episode kernels with that
                                       many episodes per wavefront (every code variant, attitude task only) */
  int32_t concurrent_episodes;      /* episodes of OTHER serl_rollout calls expected to run at the same time on other
                                       streams (mixed-build sweeps: one call per dynamics build); the kernel and the
                                       wavefronts per workgroup are chosen for n_episodes + concurrent_episodes so that
                                       the launches fit the GPU side by side.  0 = this call has the GPU to itself */
  int32_t kernel_hint;              /* enum serl_kernel_hint: SERL_KERNEL_AUTO (0) = chosen from the episode count as
                                       described at lanes_per_wave; the others force one of the wave-cooperative kernel
                                       families (tests and A/B measurements compare them: results are bit-identical).
                                       Ignored when lanes_per_wave > 0.  SERL_E_UNSUPPORTED when the forced kernel does not
                                       exist for the actor shape (two / four episodes per team, half: hidden 32 only) */
  /* -- results (per episode) */
  double *fitness;                  /* sum of rewards incl. termination penalty */
  int32_t *length_steps;            /* number of env steps taken */
  double *length_t;                 /* info['t'] after the final increment */
  int32_t *cost_steps;              /* number of steps with get_cost() == 1 */
  /* -- optional traces (NULL = not exported) */
  double *actions;                  /* env.last_u per step:  [n_episodes][max_steps][3] */
  double *states;                   /* env.x per step:       [n_episodes][max_steps][12] */
  double *rewards;                  /* reward per step:      [n_episodes][max_steps] */
  float *transitions;               /* (obs7,a3,next_obs7,r,done,cost) f32 x20 per step:
                                       [n_episodes][max_steps][20]  (base/core/agent.py:101-112); other env
                                       configurations: 2 S + A + 3 floats per step, see env_config */
  /* -- reference generation in the kernel (NULL = table mode: `ref` is read) */
  const serl_ref_spec *ref_spec;    /* [n_episodes] (ref_spec_stride 1) or one shared spec (stride 0); `ref` may be NULL */
  int64_t ref_spec_stride;
  /* -- env configuration (CitationEnv(configuration, mode), envs/phlabenv.py:84-97,174-176,205-220,377-380,415-428,
   *    446-470); zeros = the attitude task every BASELINE configuration uses.
   *      env_config   SERL_ENV_ATTITUDE   A = 3 actions (de da dr), observed states x[0 1 2 4] (p q r alpha)
   *                   SERL_ENV_SYMMETRIC  A = 1 (de; one reference: theta), observed state x[1] (q)
   *                   SERL_ENV_FULL       A = 3, observed states x[0..9]
   *      incremental  the action is an actuator RATE: scaled to +-25 deg/s instead of +-10 deg, u = last_u + scaled * dt
   *                   (last_u = 0 at reset), and last_u is part of the observation
   *    observation = [error (A), observed states, last_u (A, incremental only)]; state_dim must equal its length and
   *    action_dim must equal A.  reward = -sum_i<A |clip(scaler_i * error_i, -1, 1)| / A.  The per-episode tables keep
   *    their 3-column layouts (ref, err0, action_noise, actions: the first A columns are used / written, the others
   *    are 0); a transition row is (obs S, action A, next_obs S, reward, done, cost) = 2 S + A + 3 floats.
   *    Configurations other than the default run on kernel instantiations of their own that read the widths from the
   *    descriptor (serl_rollout_teamx_kernel_<variant>: rounds of one-episode teams while the episodes fit two rounds of
   *    workgroups, i.e. <= 2 x CUs; serl_rollout_wavex_kernel_<variant>, one wavefront per episode, beyond). */
  int32_t env_config;
  int32_t incremental;
} serl_rollout_desc;

enum serl_env_config { SERL_ENV_ATTITUDE = 0, SERL_ENV_SYMMETRIC = 1, SERL_ENV_FULL = 2 };
/* serl_rollout_desc.kernel_hint -- which wave-cooperative kernel family runs the episodes (all bit-identical):
 *   TEAM   eight wavefronts = one episode (seven integrate, one runs the actor); rounds of CUs episodes
 *   TEAM2 / TEAM4   the same team carrying two / four episodes in lane groups of 32 / 16 (hidden 32)
 *   WAVE   one wavefront = one episode, up to four per workgroup
 *   HALF   one wavefront = two episodes (hidden 32) */
enum serl_kernel_hint { SERL_KERNEL_AUTO = 0, SERL_KERNEL_TEAM = 1, SERL_KERNEL_WAVE = 2, SERL_KERNEL_HALF = 3,
                        SERL_KERNEL_TEAM2 = 4, SERL_KERNEL_TEAM4 = 5 };
/* length of the observation of a configuration (0 = invalid) and its number of actions */
int serl_env_state_dim(int env_config, int incremental);
int serl_env_action_dim(int env_config);

int serl_abi_version(void);
/* Layout self-check for bindings that mirror the structs by hand (ctypes, cgo ...): fills out[0 .. n) with
 *   sizeof(serl_rollout_desc), then offsetof of each of its members in declaration order,
 *   then sizeof(serl_build_desc), sizeof(serl_fault_row), sizeof(serl_ref_spec), sizeof(serl_replay_job)
 * as this library was compiled, and returns the number of values (written or not: call with capacity 0 to size). */
int serl_abi_layout(int32_t *out, int32_t capacity);
/* number of f32 parameters of an actor: H*S+H + L*(H*H+3H) + A*H+A */
int serl_param_count(int state_dim, int hidden, int num_layers, int action_dim);
const char *serl_last_error(void);

int serl_ctx_create(int device, serl_ctx **out);
int serl_ctx_destroy(serl_ctx *ctx);
int serl_ctx_load_build(serl_ctx *ctx, int slot, const serl_build_desc *build);
/* Development overrides, read from the environment ONCE, by serl_ctx_create (the only getenv of the library):
 *   SERL_KERNEL=team|team2|team4|wave|half   kernel family for descriptors with kernel_hint == SERL_KERNEL_AUTO
 *   SERL_WAVES_PER_BLOCK=n                   wavefronts per workgroup of the one-wavefront kernels
 *   SERL_PROFILE=1                           cycle counters for serl_debug_profile
 *   SERL_SPLIT_ACTOR=1                       one-episode teams with a streamed actor (hidden > 64): two actor wavefronts share the forward pass
 *   SERL_REMOTE_ACTOR=0                      one-episode teams with a streamed actor: the actor stays on the team's CU (default 1: a workgroup of its own on another CU
 *                                            while two workgroups per episode fit the GPU at once -- serl_rollout_teamr_kernel_<variant>)
 *   SERL_JITTER_SEED=n, SERL_JITTER_SITES=m  acted on only by the TEST-ONLY stress build of the team kernels (libserl_amd_jitter.so:
 *                                            poisoned LDS blackboards, seeded pauses around every hand-over; the product ignores them;
 *                                            a development build -DSERL_DEV_ROLE_MAP=1 reads SERL_JITTER_SITES as the role <-> wavefront map of
 *                                            the one-episode team kernels, a nibble per hardware wavefront: tools/sweep_roles.py) */

/* One population evaluation: all episodes of the descriptor, one fused kernel launch per call. */
int serl_rollout(serl_ctx *ctx, const serl_rollout_desc *desc, void *stream);
/* ABI v7.  A mixed-fault population (fault mode per episode, /root/reference/envs/phlabenv.py:114-165: the builds be / jr / sa / se / cg share the
 * nominal code on different tables, ice has code of its own) as ONE launch of ONE code object: `n` descriptors (2 .. 4), one per dynamics build,
 * each exactly what serl_rollout would take (its own build_slot, episodes, outputs; concurrent_episodes is ignored).  Results are those of n
 * separate serl_rollout calls, bit for bit.  Eligible: attitude task, hidden 32, code variants nominal / ice, kernel_hint AUTO, more than
 * 2 x CUs episodes together; otherwise SERL_E_UNSUPPORTED and nothing was launched (the caller falls back to one serl_rollout per descriptor,
 * side by side on streams of their own).  `descs` is an array of n descriptors, contiguous in HOST memory.  The launch places its workgroups so that
 * CUs which share an instruction cache run the same code variant (SERL_MIXED_PLACE=0|1|2, read by serl_ctx_create, selects the placement for A/B;
 * results do not depend on it). */
int serl_rollout_multi(serl_ctx *ctx, int32_t n, const serl_rollout_desc *descs, void *stream);

/* Dynamics only (test / micro-benchmark entry): per episode initialize() followed by T calls of the
 * reference's step(cmd) -- cmds f64 [n_episodes][T][10] -> states f64 [n_episodes][T][12] (device). */
int serl_dyn_open_loop(serl_ctx *ctx, int slot, int32_t n_episodes, int32_t T, const double *cmds,
                       double *states, int32_t lanes_per_wave, int32_t kernel_hint /* AUTO, TEAM or WAVE */, void *stream);

/* Development aid: with SERL_PROFILE=1 in the environment serl_rollout records shader-clock cycles of wave 0 of
 * workgroup 0: out[0..3] = {actor forward, dynamics step, env bookkeeping, env steps}; out[4..31] = phase
 * counters of the model evaluation (non-zero only in builds compiled with -DCITW_PROFILE). */
int serl_debug_profile(serl_ctx *ctx, unsigned long long out[32]);
/* Development aid: how the most recent serl_rollout_multi launch placed its workgroups.  out[0]: 0 = by blockIdx range (SERL_MIXED_PLACE=0, or one
 * code variant only), 1 = by the census of the CU pairs, 2 = by tickets (SERL_MIXED_PLACE=1, or the census timed out: the GPU was shared);
 * out[1] = workgroups that registered in the census; out[2] = CU pairs that held two of them, out[3] = one.  Blocks until the device is idle. */
int serl_debug_mixed_placement(serl_ctx *ctx, int32_t out[4]);

/* ABI v8.  WHAT the most recent serl_rollout / serl_rollout_multi call of this context launched -- the kernel family is chosen by the library
 * (episode count, actor shape, env configuration, kernel_hint), so a caller that compares families, or a test that names one, asks here which
 * one actually ran (the seam being replaced is the one loop `for net in pop: for i in range(num_evals): evaluate(net)`,
 * /root/reference/base/core/agent.py:234-241, which has no such choice).  Host-side record of the call: does not wait for the device.
 *   out[0] enum serl_kernel_family        out[1] workgroups of the (last) launch     out[2] episodes per team / per wavefront (LANE: lanes)
 *   out[3] 1 = the launch drains a work queue (a lane group takes the next episode when its own ends)
 *   out[4] actor wavefronts beside the team wavefronts (0: the family has none)       out[5] 1 = the actor streams its weights from L2 (0: LDS-resident)
 *   out[6] launches the call made (rounds of workgroups)                              out[7] enum serl_dyn_code of the launch, -1 = several (mixed sweep)
 * serl_dyn_open_loop records its launch too (families TEAM / WAVE / LANE, no actor wavefront).  SERL_E_INVALID before the first launch of the context. */
enum serl_kernel_family { SERL_FAMILY_NONE = 0, SERL_FAMILY_TEAM = 1 /* eight wavefronts = one episode, actor weights in LDS */,
                          SERL_FAMILY_TEAMS = 2 /* ... the actor wavefront streams its weights (hidden 72 / 96) */,
                          SERL_FAMILY_TEAMS2 = 3 /* ... two actor wavefronts share the forward pass (SERL_SPLIT_ACTOR=1) */,
                          SERL_FAMILY_TEAMX = 4 /* ... env configurations other than the attitude task */,
                          SERL_FAMILY_TEAM2 = 5 /* two episodes per team */, SERL_FAMILY_TEAM2S = 6 /* ... six team + two streaming actor wavefronts */,
                          SERL_FAMILY_TEAM4 = 7 /* four episodes per team */, SERL_FAMILY_TEAM4_MIXED = 8 /* ... several code variants in one launch */,
                          SERL_FAMILY_HALF = 9 /* one wavefront = two episodes */, SERL_FAMILY_WAVE = 10 /* one wavefront = one episode */,
                          SERL_FAMILY_WAVEX = 11 /* ... other env configurations */, SERL_FAMILY_LANE = 12 /* one lane = one episode */,
                          SERL_FAMILY_TEAMR = 13 /* eight wavefronts = one episode, its streamed actor (two wavefronts) in a workgroup of its own on another CU */ };
int serl_last_rollout_info(serl_ctx *ctx, int32_t out[8]);

/* Duration (ms) of the most recent serl_rollout kernel on its stream, measured with HIP events
 * recorded around the launch; blocks until that kernel has finished. */
int serl_last_rollout_ms(serl_ctx *ctx, float *ms);

/* ---- SSNE weight-tensor edits (base/core/mod_neuro_evo.py) as elementwise kernels.  `weights` is
 * the same [n_members][stride] f32 tensor; index lists are DEVICE int32/float arrays generated by
 * the host from the reference's RNG streams so that selection stays bit-compatible. ------------ */
/* clone (mod_neuro_evo.py:371-382): weights[dst[i]] = weights[src[i]], i < n */
int serl_ga_clone(serl_ctx *ctx, float *weights, int64_t stride, int32_t param_count,
                  const int32_t *src, const int32_t *dst, int32_t n, void *stream);
/* crossover_inplace (mod_neuro_evo.py:61-93): n row/element swaps between two members:
 *   ops[i] = {offset, length, dir}: dir 0 copies member b -> a, dir 1 copies a -> b, at
 *   weights[.. + offset .. offset+length) */
int serl_ga_crossover(serl_ctx *ctx, float *weights, int64_t stride, int32_t member_a, int32_t member_b,
                      const int32_t *ops, int32_t n_ops, void *stream);
/* mutate_inplace (mod_neuro_evo.py:329-369): n sparse edits of one member applied IN ORDER:
 *   kind 0:  w[idx] += z * (strength * w[idx])   (normal / super mutation: random.gauss(0, strength*w) = z*sigma)
 *   kind 1:  w[idx]  = z                          (reset: random.gauss(0, 1))
 * each followed by the reference's hard clamp to +-1e6 (regularize_weight, :57-59,366). */
int serl_ga_mutate(serl_ctx *ctx, float *weights, int64_t stride, int32_t member,
                   const int32_t *idx, const int32_t *kind, const float *z, const float *strength,
                   int32_t n, void *stream);
/* proximal / safe mutation update (mod_neuro_evo.py:183-223, 254-298):
 *   theta[i] += delta[i] / scaling[i]   over the flat 2-D-weights genome given as (offset,length)
 *   segments of the packed parameter row */
int serl_ga_scaled_perturb(serl_ctx *ctx, float *weights, int64_t stride, int32_t member,
                           const int32_t *seg_offset, const int32_t *seg_length, int32_t n_seg,
                           const float *delta, const float *scaling, void *stream);

/* Output sensitivity of proximal_mutate / safe_mutate (mod_neuro_evo.py:183-223, 254-298): for every listed member,
 *   jacobian_i = d( sum_b actor(states[m][b])[i] ) / d genome,  i < action_dim   (the reference: one backward pass per output)
 *   scaling    = sqrt(sum_i jacobian_i^2);  scaling[scaling == 0] = 1;  scaling[scaling < 0.01] = 0.01
 * genome = the 2-D weights in named_parameters order (extract_parameters, genetic_agent.py:131-141), G = H*S + L*H*H + A*H.
 * states: f32 [n_members][batch][state_dim], scaling: f32 [n_members][G] (device).  The update itself is
 * serl_ga_scaled_perturb with delta drawn by the host (torch.distributions.Normal, the reference's generator). */
int serl_ga_sensitivity(serl_ctx *ctx, const float *weights, int64_t stride, int32_t state_dim, int32_t hidden,
                        int32_t num_layers, int32_t action_dim, int32_t activation, const int32_t *members,
                        int32_t n_members, const float *states, int32_t batch, float *scaling, void *stream);
/* Actor.get_novelty (genetic_agent.py:111-115) for n_pairs (actor, batch) pairs in one launch:
 *   novelty[p] = mean_b sum_a (actions[p][b][a] - actor_{members[p]}(states[p][b])[a])^2
 * -- the two halves of SSNE.get_distance (mod_neuro_evo.py:411-417), i.e. the keys of sort_groups_by_distance (:426-445). */
int serl_ga_novelty(serl_ctx *ctx, const float *weights, int64_t stride, int32_t state_dim, int32_t hidden,
                    int32_t num_layers, int32_t action_dim, int32_t activation, const int32_t *members, int32_t n_pairs,
                    const float *states, const float *actions, int32_t batch, float *novelty, void *stream);

/* ---- device replay rings (base/core/replay_memory.py:21-31 `add`, base/core/agent.py:101-112) ---------------------------
 * A ring is f32 [capacity][20] rows (obs7, a3, next_obs7, r, done, cost) in HBM.  One job appends the rows of one stored
 * episode -- staged[episode][0 .. length), what serl_rollout wrote to `transitions` -- to ring slots
 * (position + k) % capacity in step order, k = rank of the row among the rows taken: all of them, or (cost_only) the
 * cost-flagged ones, compacted (agent.critical_buffer).  The first `skip` ranks are not written (an episode longer than
 * the ring leaves only its tail, as sequential add() calls would).  The host keeps position / fill of every ring. */
typedef struct serl_replay_job {
  float *ring;
  int32_t capacity, position, episode, length, cost_only, skip;
} serl_replay_job;
int serl_replay_scatter(serl_ctx *ctx, const float *staged, int64_t rows_per_episode, const serl_replay_job *jobs /* device */,
                        int32_t n_jobs, void *stream);

/* calc_smoothness (base/core/utils.py:82-120) of n_episodes action traces of DIFFERENT lengths in one launch:
 *   Y = fft(y, N) per channel, N = |lengths[e]|;  S = sum_c sum_{1 <= i < N/2} |Y_i,c|^2 * dt * f_i * 2 / N  with
 *   f = linspace(dt, 1 / (2 dt), N/2 - 1);  out[e] = -sqrt(S) * 100 * (80 / (N dt))   (0 for N < 4)
 * evaluated as a direct DFT (one thread per frequency, the N twiddles in LDS), so there is no per-length FFT plan.
 * actions: f64 [n_episodes][episode_stride / 3][3] (env.last_u per step: serl_rollout_desc.actions), rows past N ignored;
 * work: f64 [serl_smoothness_work_size(n_episodes, max_len)] scratch; max_len >= every |length|, <= 8192. */
int serl_smoothness_work_size(int32_t n_episodes, int32_t max_len);
int serl_smoothness(serl_ctx *ctx, const double *actions, int64_t episode_stride, const int32_t *lengths, int32_t n_episodes,
                    int32_t max_len, double dt, double *work, double *out, void *stream);

/* The training loop of SSNE.distilation_crossover (base/core/mod_neuro_evo.py:131-147) for all pairs of an epoch in one
 * launch: n_steps[p] Adam steps (torch.optim.Adam defaults, lr) of GeneticAgent.update_parameters
 * (base/core/genetic_agent.py:22-59) per pair, on minibatches slots[p][step][0 .. batch[p]) of the child's buffer:
 *   loss = sum_kept (actor(state) - target)^2 + mean_kept(actor(state)^2)
 * `targets` (the better parent's action per state) and `keep` (1 where the critic's Q-filter keeps the state) are what
 * the parents and the critic contribute; they do not depend on the child and are computed once by the caller.
 * child: f32 [n_pairs][stride], in = the second parent's parameters, out = the trained child.  states [n_pairs][rows][S],
 * targets [n_pairs][rows][A], keep [n_pairs][rows], slots i32 [n_pairs][steps][128].  Compiled for hidden 32 x 3 layers
 * (SERL_E_UNSUPPORTED otherwise). */
int serl_ga_distill(serl_ctx *ctx, float *child, int64_t stride, int32_t n_pairs, int32_t state_dim, int32_t hidden, int32_t num_layers,
                    int32_t action_dim, int32_t activation, const float *states, const float *targets, const float *keep, int32_t rows,
                    const int32_t *slots, int32_t steps, const int32_t *n_steps, const int32_t *batch, float lr, void *stream);
/* Host helper (no GPU work): the rows `random.sample(memory, k)` of base/core/replay_memory.py:72-73,83-85 picks from n
 * transitions in `calls` consecutive calls, replayed from the generator's raw 32-bit outputs (CPython's selection
 * algorithm: set-based above its set-size threshold, pool-based below).  Returns the number of outputs consumed,
 * -1 = n_words too small, -2 = bad arguments.  out: i32 [calls][out_stride], out_stride >= k. */
long long serl_host_sample_slots(const uint32_t *words, long long n_words, int32_t n, int32_t k, int32_t calls, int32_t *out,
                                 int32_t out_stride);

#ifdef __cplusplus
}
#endif
#endif /* SERL_AMD_H */
