// serl_capi.hip -- C ABI (include/serl_amd.h) of the MI355X-native SERL population-rollout evaluator:
// context / build-table management, rollout dispatch with HIP-event timing, and the SSNE
// weight-tensor kernels.  No torch types cross this boundary: plain pointers, sizes, a hipStream_t.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "rollout_device.h"
#include "serl_ctx.h"

static thread_local std::string g_err;
int serl_fail(int code, const std::string &msg) { g_err = msg; return code; }
static int fail(int code, const std::string &msg) { return serl_fail(code, msg); }

#define SERL_LANE_DECL(v) \
  void serl_launch_rollout_##v(const RolloutArgs &a, int grid, hipStream_t stream); \
  void serl_launch_dyn_##v(const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream);
SERL_LANE_DECL(nominal) SERL_LANE_DECL(ice) SERL_LANE_DECL(cg_timed) SERL_LANE_DECL(gust) SERL_LANE_DECL(test)
#undef SERL_LANE_DECL
// the lane-per-episode kernels (rollout_<variant>.hip: one episode per lane around the branch-free DAG evaluation), one unit per code variant
static void serl_launch_rollout_lane(int code, const RolloutArgs &a, int grid, hipStream_t stream)
{
  switch (code) {
    case SERL_DYN_ICE: serl_launch_rollout_ice(a, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_rollout_cg_timed(a, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_rollout_gust(a, grid, stream); break;
    case SERL_DYN_TEST: serl_launch_rollout_test(a, grid, stream); break;
    default: serl_launch_rollout_nominal(a, grid, stream); break;
  }
}
static void serl_launch_dyn_lane(int code, const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream)
{
  switch (code) {
    case SERL_DYN_ICE: serl_launch_dyn_ice(a, cmds, states, T, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_dyn_cg_timed(a, cmds, states, T, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_dyn_gust(a, cmds, states, T, grid, stream); break;
    case SERL_DYN_TEST: serl_launch_dyn_test(a, cmds, states, T, grid, stream); break;
    default: serl_launch_dyn_nominal(a, cmds, states, T, grid, stream); break;
  }
}
#define SERL_DECL_WAVE(v)                                                                                     \
  void serl_launch_rollout_wave_##v(const RolloutArgs &a, int grid, hipStream_t stream);                          \
  void serl_launch_rollout_wavex_##v(const RolloutArgs &a, int grid, hipStream_t stream);                         \
  void serl_launch_dyn_wave_##v(const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream);
SERL_DECL_WAVE(nominal) SERL_DECL_WAVE(ice) SERL_DECL_WAVE(cg_timed) SERL_DECL_WAVE(gust) SERL_DECL_WAVE(test)

// a team of wavefronts per episode (rollout_team.inc: 7 + the actor wavefront, two per SIMD): the latency-bound regime
#define SERL_DECL_TEAM(v)                                                                                     \
  void serl_launch_rollout_team_##v(const RolloutArgs &a, int grid, hipStream_t stream);                          \
  void serl_launch_rollout_teamx_##v(const RolloutArgs &a, int grid, hipStream_t stream);                         \
  void serl_launch_rollout_teams_##v(const RolloutArgs &a, int grid, hipStream_t stream);                         \
  void serl_launch_dyn_team_##v(const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream);
SERL_DECL_TEAM(nominal) SERL_DECL_TEAM(ice) SERL_DECL_TEAM(cg_timed) SERL_DECL_TEAM(gust) SERL_DECL_TEAM(test)
// ... with one forward pass over two actor wavefronts (rollout_teams2_<v>.hip: hidden > 64)
#include "serl_mixed.h"
#define SERL_DECL_TEAMS2(v) void serl_launch_rollout_teams2_##v(const RolloutArgs &a, int grid, hipStream_t stream); \
                            void serl_launch_rollout_teamr_##v(const RolloutArgs &a, int grid, hipStream_t stream);
SERL_DECL_TEAMS2(nominal) SERL_DECL_TEAMS2(ice) SERL_DECL_TEAMS2(cg_timed) SERL_DECL_TEAMS2(gust) SERL_DECL_TEAMS2(test)

// two episodes per wavefront (rollout_half.inc): beyond one wavefront per SIMD
#define SERL_DECL_HALF(v) void serl_launch_rollout_half_##v(const RolloutArgs &a, int grid, hipStream_t stream);
SERL_DECL_HALF(nominal) SERL_DECL_HALF(ice) SERL_DECL_HALF(cg_timed) SERL_DECL_HALF(gust) SERL_DECL_HALF(test)

static void serl_launch_rollout_half(int code, const RolloutArgs &a, int grid, hipStream_t stream)
{
  switch (code) {
    case SERL_DYN_NOMINAL: serl_launch_rollout_half_nominal(a, grid, stream); break;
    case SERL_DYN_ICE: serl_launch_rollout_half_ice(a, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_rollout_half_cg_timed(a, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_rollout_half_gust(a, grid, stream); break;
    default: serl_launch_rollout_half_test(a, grid, stream); break;
  }
}

// two / four episodes per team (rollout_team_half.inc): between CUs and 4 x CUs episodes
#define SERL_DECL_TEAMG(v) void serl_launch_rollout_team2_##v(const RolloutArgs &a, int grid, hipStream_t stream); \
                           void serl_launch_rollout_team2s_##v(const RolloutArgs &a, int grid, hipStream_t stream); \
                           void serl_launch_rollout_team4_##v(const RolloutArgs &a, int grid, hipStream_t stream);
SERL_DECL_TEAMG(nominal) SERL_DECL_TEAMG(ice) SERL_DECL_TEAMG(cg_timed) SERL_DECL_TEAMG(gust) SERL_DECL_TEAMG(test)

static void serl_launch_rollout_teamg(int code, int groups, const RolloutArgs &a, int grid, hipStream_t stream)
{
  // (two per team with a streamed actor -- hidden != 32 -- is a kernel of its own: six team wavefronts + two actor wavefronts)
#define SERL_TEAMG_CASE(v) (groups == 4 ? serl_launch_rollout_team4_##v(a, grid, stream) : \
                            (a.d.hidden != 32 ? serl_launch_rollout_team2s_##v(a, grid, stream) : serl_launch_rollout_team2_##v(a, grid, stream)))
  switch (code) {
    case SERL_DYN_NOMINAL: SERL_TEAMG_CASE(nominal); break;
    case SERL_DYN_ICE: SERL_TEAMG_CASE(ice); break;
    case SERL_DYN_CG_TIMED: SERL_TEAMG_CASE(cg_timed); break;
    case SERL_DYN_GUST: SERL_TEAMG_CASE(gust); break;
    default: SERL_TEAMG_CASE(test); break;
  }
#undef SERL_TEAMG_CASE
}

// Episodes per team, 0 = use another kernel.  Measured per env step: 21.6 us with one episode per team, 22.2 with two,
// 25.6 with four (the scalar glue is paid once for all lane groups; only the lane-parallel passes multiply, and the team's
// helpers share them), and one team fits a CU -- so up to 2 x CUs episodes run two per team and up to 4 x CUs four per team,
// in ONE round of workgroups, against 57 us of the one-wavefront kernel (1 023 episodes: 39.2 M env-steps/s against 16).
// H = 32 only.  kernel_hint SERL_KERNEL_TEAM2 / TEAM4 force it.
static int serl_use_teamg(const serl_ctx *c, const serl_rollout_desc *d, int hint, int episodes)
{
  if (d->hidden != 32) {
    // actors without one hidden row per lane (SERL10's 72, the TD3 actor's 96): the actor wavefront runs the episodes' forward
    // passes one after the other -- two per team (~33 us per env step for the pair, actor-bound) beat one wavefront per episode
    // (57 - 61 us) while every pair has a CU: CUs < episodes <= 2 x CUs; four per team would be slower than four lone wavefronts
    if (hint == SERL_KERNEL_TEAM2) return 2;
    if (hint != SERL_KERNEL_AUTO) return 0;
    return (episodes > c->num_cus && episodes <= 2 * c->num_cus) ? 2 : 0;
  }
  if (hint == SERL_KERNEL_TEAM2) return 2;
  if (hint == SERL_KERNEL_TEAM4) return 4;
  if (hint != SERL_KERNEL_AUTO) return 0;
  if (episodes <= c->num_cus || episodes > 4 * c->num_cus) return 0;
  return episodes <= 2 * c->num_cus ? 2 : 4;
}

// Beyond 4 x CUs episodes the team kernels still win when the launch has the GPU to itself: 4 x CUs episodes per 30.8 us
// is the rate of the two-episodes-per-wavefront kernel (8 x CUs per 62.9 us) at half the granularity, and the remainder runs
// at the rate of its own size class.  Side-by-side launches (concurrent_episodes > 0) split the CUs by episode share (below; a
// team needs a whole CU).  H = 32 only; any kernel_hint other than AUTO switches it off.
static bool serl_use_team_rounds(const serl_ctx *c, const serl_rollout_desc *d, int hint, int episodes)
{
  if (d->hidden != 32 || hint != SERL_KERNEL_AUTO) return false;
  return episodes > 4 * c->num_cus;
}

// Work-queue counter of ONE launch: a ring of counters, so that queue launches in flight on different streams (a mixed-fault sweep;
// validation beside training) never share one -- a second launch's memset or atomicAdd on a shared counter would make the first
// skip or repeat episodes.  (64 launches later a counter is reused: far beyond what a context keeps in flight.)
static int32_t *serl_next_queue_counter(serl_ctx *c)
{
  int32_t *p = c->queue + c->queue_next;
  c->queue_next = (c->queue_next + 1) % SERL_QUEUE_COUNTERS;
  return p;
}

// The one-wavefront-per-episode kernel runs up to 4 x CUs episodes at once (one per SIMD); beyond that the launch needs a
// second round of wavefronts, and packing two episodes into a wavefront (1.1 x the time per env step) is the better deal.
// H = 32 only (the lane group of an episode holds one hidden row per lane).  kernel_hint SERL_KERNEL_HALF forces it.
static bool serl_use_half(const serl_ctx *c, const serl_rollout_desc *d, int hint, int episodes)
{
  if (d->hidden != 32) return false;
  if (hint != SERL_KERNEL_AUTO) return hint == SERL_KERNEL_HALF;
  return episodes > 4 * c->num_cus;
}

// One episode per workgroup and one workgroup per CU (the LDS copy of the tables): a team finishes an env step in
// ~0.37 of the time a lone wavefront needs (21.2 vs 57 us), but only one team fits a CU where four lone wavefronts
// would: teams while every episode gets a CU of its own (measured: 320 episodes as teams 88 us, 400 alone 59 us),
// lone wavefronts beyond.  kernel_hint SERL_KERNEL_TEAM forces it (any other hint excludes it).
static bool serl_use_team(const serl_ctx *c, int hint, int episodes)
{
  if (hint != SERL_KERNEL_AUTO) return hint == SERL_KERNEL_TEAM;
  return episodes <= c->num_cus;
}

// (rollout_device.h: serl_lds_actor_ok) the actor shape whose weights live in LDS for the episode
static bool serl_lds_actor_shape(const serl_rollout_desc &d) { return d.hidden == 32 && d.num_layers <= 3 && d.state_dim == 7 && d.action_dim == 3; }

// Round 6: streamed actors (hidden 65 .. 128: SERL10's 72, the TD3 actor's 96) of the attitude task with the actor in a workgroup of its own on another CU
// (rollout_teamr_<v>.hip): two workgroups per episode, pairs (b, b + 8) of a group of sixteen.  Only while every workgroup of the launch -- and of the launches it
// was told about -- can be resident at once: a team spins on its partner's mailbox.  0 = not eligible, else the grid.
static int serl_remote_actor_grid(const serl_ctx *c, const serl_rollout_desc *d, int together)
{
  if (!c->env_remote_actor || (d->hidden != 72 && d->hidden != 96 && d->hidden != 128) || d->state_dim != 7 || d->action_dim != 3) return 0;      // (rollout_device.h serl_cu_actor_ok)
  const int grid = 16 * ((d->n_episodes + 7) / 8), all = 16 * ((together + 7) / 8);
  return all <= c->num_cus ? grid : 0;
}

static void serl_launch_rollout_teamr(int code, const RolloutArgs &a, int grid, hipStream_t stream)
{
  switch (code) {
    case SERL_DYN_NOMINAL: serl_launch_rollout_teamr_nominal(a, grid, stream); break;
    case SERL_DYN_ICE: serl_launch_rollout_teamr_ice(a, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_rollout_teamr_cg_timed(a, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_rollout_teamr_gust(a, grid, stream); break;
    default: serl_launch_rollout_teamr_test(a, grid, stream); break;
  }
}

static void serl_launch_rollout_team(int code, const RolloutArgs &a, int grid, hipStream_t stream, bool split_actor = false)
{
  // (rollout_device.h: serl_lds_actor_ok) shapes whose weights the actor wavefront streams have a kernel of their own
  const bool lds_actor = serl_lds_actor_shape(a.d);
  if (!lds_actor && split_actor && a.d.hidden > 64 && a.d.hidden <= 128) {
    // (development override SERL_SPLIT_ACTOR=1) two actor wavefronts share ONE forward pass beside a six-wavefront team: round 4
    // built it for the shapes whose lone actor wavefront needed a whole step (H = 72: 43 k cycles) -- and then found that a
    // shape-specialised forward (serl_actor_forward_big: 25 k) on ONE wavefront beside the seven-wavefront team is faster
    switch (code) {
      case SERL_DYN_NOMINAL: serl_launch_rollout_teams2_nominal(a, grid, stream); break;
      case SERL_DYN_ICE: serl_launch_rollout_teams2_ice(a, grid, stream); break;
      case SERL_DYN_CG_TIMED: serl_launch_rollout_teams2_cg_timed(a, grid, stream); break;
      case SERL_DYN_GUST: serl_launch_rollout_teams2_gust(a, grid, stream); break;
      default: serl_launch_rollout_teams2_test(a, grid, stream); break;
    }
    return;
  }
  if (!lds_actor) {
    switch (code) {
      case SERL_DYN_NOMINAL: serl_launch_rollout_teams_nominal(a, grid, stream); break;
      case SERL_DYN_ICE: serl_launch_rollout_teams_ice(a, grid, stream); break;
      case SERL_DYN_CG_TIMED: serl_launch_rollout_teams_cg_timed(a, grid, stream); break;
      case SERL_DYN_GUST: serl_launch_rollout_teams_gust(a, grid, stream); break;
      default: serl_launch_rollout_teams_test(a, grid, stream); break;
    }
    return;
  }
  switch (code) {
    case SERL_DYN_NOMINAL: serl_launch_rollout_team_nominal(a, grid, stream); break;
    case SERL_DYN_ICE: serl_launch_rollout_team_ice(a, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_rollout_team_cg_timed(a, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_rollout_team_gust(a, grid, stream); break;
    default: serl_launch_rollout_team_test(a, grid, stream); break;
  }
}

static void serl_launch_dyn_team(int code, const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream)
{
  switch (code) {
    case SERL_DYN_NOMINAL: serl_launch_dyn_team_nominal(a, cmds, states, T, grid, stream); break;
    case SERL_DYN_ICE: serl_launch_dyn_team_ice(a, cmds, states, T, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_dyn_team_cg_timed(a, cmds, states, T, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_dyn_team_gust(a, cmds, states, T, grid, stream); break;
    default: serl_launch_dyn_team_test(a, cmds, states, T, grid, stream); break;
  }
}

// The wave-cooperative kernels (one wavefront per episode, rollout_wave.inc) exist for every code variant; the
// lane-per-episode kernels (rollout_variant.inc, lanes_per_wave > 0) for every code variant since round 6.
static bool serl_has_wave_kernel(int code) { return code >= SERL_DYN_NOMINAL && code <= SERL_DYN_TEST; }
static bool serl_has_lane_kernel(int code) { return code >= SERL_DYN_NOMINAL && code <= SERL_DYN_TEST; }

// Lane-per-episode kernels (rollout_device.h serl_actor_forward_lane32_t): the population [members][stride] regrouped to [ceil(P / 4)][mpad][4], so that parameter
// group g of 64 consecutive members is ONE run of 1 KB.  Reads run along a member's row, writes are 16 B pieces (30 MB for 2 048 SERL50 actors: microseconds in front of a
// launch of hundreds of milliseconds).
__global__ void serl_regroup_weights_kernel(const float *w, int64_t stride, int32_t n_members, int32_t P, float *out, int32_t mpad)
{
  const int g = blockIdx.y * blockDim.x + threadIdx.x, m = blockIdx.x;
  if (4 * g >= P) return;
  const float *row = w + (size_t)m * stride + 4 * (size_t)g;
  float4 v;
  v.x = row[0];
  v.y = 4 * g + 1 < P ? row[1] : 0.0f;
  v.z = 4 * g + 2 < P ? row[2] : 0.0f;
  v.w = 4 * g + 3 < P ? row[3] : 0.0f;
  *(float4 *)(out + ((size_t)g * mpad + m) * 4) = v;
}

static int serl_regroup_weights(serl_ctx *c, const serl_rollout_desc *d, RolloutArgs &a, hipStream_t stream)
{
  const int P = d->hidden * d->state_dim + d->hidden + d->num_layers * (d->hidden * d->hidden + 3 * d->hidden) + d->action_dim * d->hidden + d->action_dim;
  const int groups = (P + 3) / 4, mpad = (d->n_members + 63) / 64 * 64;
  const size_t bytes = (size_t)groups * mpad * 16;
  const int slot = c->wt_next;
  c->wt_next = (c->wt_next + 1) % SERL_WT_SLOTS;
  // a launch on another stream may still read the copy this slot held (more than SERL_WT_SLOTS lane launches in flight): order this launch behind it
  if (c->wt_done[slot]) HIP_TRY(hipStreamWaitEvent(stream, c->wt_done[slot], 0));
  else HIP_TRY(hipEventCreateWithFlags(&c->wt_done[slot], hipEventDisableTiming));
  c->wt_slot_of_launch = slot;
  if (c->wt_cap[slot] < bytes) {
    if (c->wt[slot]) HIP_TRY(hipFree(c->wt[slot]));      // (waits for the launches that read it)
    c->wt[slot] = nullptr; c->wt_cap[slot] = 0;
    HIP_TRY(hipMalloc(&c->wt[slot], bytes));
    c->wt_cap[slot] = bytes;
  }
  hipLaunchKernelGGL(serl_regroup_weights_kernel, dim3(d->n_members, (groups + 255) / 256), dim3(256), 0, stream, d->weights, d->weight_stride, d->n_members, P,
                     (float *)c->wt[slot], mpad);
  HIP_TRY(hipGetLastError());
  a.wt = (const float *)c->wt[slot];
  a.wt_members = mpad;
  return SERL_OK;
}

static void serl_launch_rollout_wave(int code, const RolloutArgs &a, int grid, hipStream_t stream)
{
  switch (code) {
    case SERL_DYN_NOMINAL: serl_launch_rollout_wave_nominal(a, grid, stream); break;
    case SERL_DYN_ICE: serl_launch_rollout_wave_ice(a, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_rollout_wave_cg_timed(a, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_rollout_wave_gust(a, grid, stream); break;
    default: serl_launch_rollout_wave_test(a, grid, stream); break;
  }
}

// env configurations other than the attitude task (serl_rollout_desc.env_config / incremental)
static void serl_launch_rollout_teamx(int code, const RolloutArgs &a, int grid, hipStream_t stream)
{
  switch (code) {
    case SERL_DYN_NOMINAL: serl_launch_rollout_teamx_nominal(a, grid, stream); break;
    case SERL_DYN_ICE: serl_launch_rollout_teamx_ice(a, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_rollout_teamx_cg_timed(a, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_rollout_teamx_gust(a, grid, stream); break;
    default: serl_launch_rollout_teamx_test(a, grid, stream); break;
  }
}

static void serl_launch_rollout_wavex(int code, const RolloutArgs &a, int grid, hipStream_t stream)
{
  switch (code) {
    case SERL_DYN_NOMINAL: serl_launch_rollout_wavex_nominal(a, grid, stream); break;
    case SERL_DYN_ICE: serl_launch_rollout_wavex_ice(a, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_rollout_wavex_cg_timed(a, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_rollout_wavex_gust(a, grid, stream); break;
    default: serl_launch_rollout_wavex_test(a, grid, stream); break;
  }
}

static void serl_launch_dyn_wave(int code, const RolloutArgs &a, const double *cmds, double *states, int T, int grid, hipStream_t stream)
{
  switch (code) {
    case SERL_DYN_NOMINAL: serl_launch_dyn_wave_nominal(a, cmds, states, T, grid, stream); break;
    case SERL_DYN_ICE: serl_launch_dyn_wave_ice(a, cmds, states, T, grid, stream); break;
    case SERL_DYN_CG_TIMED: serl_launch_dyn_wave_cg_timed(a, cmds, states, T, grid, stream); break;
    case SERL_DYN_GUST: serl_launch_dyn_wave_gust(a, cmds, states, T, grid, stream); break;
    default: serl_launch_dyn_wave_test(a, cmds, states, T, grid, stream); break;
  }
}

// Wavefronts per workgroup of the wave-cooperative kernels: one per CU while there are no more episodes than CUs,
// then up to four (one per SIMD) sharing the workgroup's LDS copy of the tables.
static int serl_wave_kernel_waves_per_block(const serl_ctx *c, int episodes)
{
  if (c->env_waves_per_block >= 1 && c->env_waves_per_block <= 8) return c->env_waves_per_block;     // (8 only with a library built with -DCITW_MAX_WAVES=8)
  int w = (episodes + c->num_cus - 1) / c->num_cus;
  return w < 1 ? 1 : (w > 4 ? 4 : w);
}

// Wavefronts per workgroup.  One workgroup stages one LDS copy of the tables (94 KiB of the CU's 160 KiB), so
// only one workgroup fits a CU: while there are no more wavefronts than CUs each gets a CU of its own (the
// code streams through the instruction cache and co-resident waves slow each other down); beyond that, waves
// share a CU four at a time (one per SIMD, each with the full 512-register budget).
static int serl_waves_per_block(const serl_ctx *c, int waves)
{
  if (c->env_waves_per_block >= 1 && c->env_waves_per_block <= 4) return c->env_waves_per_block;
  return waves <= c->num_cus ? 1 : 4;
}

// descriptor hint first, then the context's development override (SERL_KERNEL, read once by serl_ctx_create)
static int serl_resolve_hint(const serl_ctx *c, int desc_hint) { return desc_hint != SERL_KERNEL_AUTO ? desc_hint : c->env_kernel; }

// serl_last_rollout_info's record of a call (include/serl_amd.h)
static void serl_note_launch(serl_ctx *c, int family, int grid, int per_team, bool queue, int actor_waves, bool streamed, int launches, int code)
{
  const int32_t v[8] = {family, grid, per_team, queue ? 1 : 0, actor_waves, streamed ? 1 : 0, launches, code};
  for (int i = 0; i < 8; ++i) c->last_info[i] = v[i];
}

extern "C" {

int serl_abi_version(void) { return SERL_ABI_VERSION; }

int serl_abi_layout(int32_t *out, int32_t capacity)
{
#define SERL_OFF(m) (int32_t)offsetof(serl_rollout_desc, m)
  const int32_t v[] = {
    (int32_t)sizeof(serl_rollout_desc),
    SERL_OFF(state_dim), SERL_OFF(action_dim), SERL_OFF(hidden), SERL_OFF(num_layers), SERL_OFF(activation), SERL_OFF(n_members),
    SERL_OFF(weights), SERL_OFF(weight_stride), SERL_OFF(n_episodes), SERL_OFF(build_slot), SERL_OFF(member_of_episode),
    SERL_OFF(faults), SERL_OFF(ref), SERL_OFF(ref_stride), SERL_OFF(err0), SERL_OFF(action_noise), SERL_OFF(noise_row),
    SERL_OFF(sensor_noise), SERL_OFF(sensor_row), SERL_OFF(tick0), SERL_OFF(t_max), SERL_OFF(max_steps), SERL_OFF(lanes_per_wave),
    SERL_OFF(concurrent_episodes), SERL_OFF(kernel_hint), SERL_OFF(fitness), SERL_OFF(length_steps), SERL_OFF(length_t),
    SERL_OFF(cost_steps), SERL_OFF(actions), SERL_OFF(states), SERL_OFF(rewards), SERL_OFF(transitions), SERL_OFF(ref_spec),
    SERL_OFF(ref_spec_stride), SERL_OFF(env_config), SERL_OFF(incremental),
    (int32_t)sizeof(serl_build_desc), (int32_t)sizeof(serl_fault_row), (int32_t)sizeof(serl_ref_spec), (int32_t)sizeof(serl_replay_job)};
#undef SERL_OFF
  const int32_t n = (int32_t)(sizeof(v) / sizeof(v[0]));
  for (int32_t i = 0; out && i < n && i < capacity; ++i) out[i] = v[i];
  return n;
}

// envs/phlabenv.py:84-97 (obs_idx per configuration), :213-220 (n_obs)
int serl_env_action_dim(int env_config) { return env_config == SERL_ENV_SYMMETRIC ? 1 : 3; }
int serl_env_state_dim(int env_config, int incremental)
{
  if (env_config < 0 || env_config > 2) return 0;
  const int A = serl_env_action_dim(env_config);
  const int nx = env_config == SERL_ENV_ATTITUDE ? 4 : (env_config == SERL_ENV_SYMMETRIC ? 1 : 10);
  return A + nx + (incremental ? A : 0);
}
const char *serl_last_error(void) { return g_err.c_str(); }

int serl_ctx_create(int device, serl_ctx **out)
{
  if (!out) return fail(SERL_E_INVALID, "serl_ctx_create: out is NULL");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(SERL_E_INVALID, "serl_ctx_create: no such device");
  HIP_TRY(hipSetDevice(device));
  serl_ctx *c = new (std::nothrow) serl_ctx();
  if (!c) return fail(SERL_E_NOMEM, "serl_ctx_create: out of memory");
  c->device = device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
      if (prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
      if (prop.sharedMemPerBlockOptin > 0) c->lds_per_block = (int)prop.sharedMemPerBlockOptin;
    }
  }
  {   // development overrides: the library's only look at the process environment
    const char *e;
    c->env_kernel = SERL_KERNEL_AUTO;
    if ((e = getenv("SERL_KERNEL")) != nullptr) {
      const std::string k(e);
      c->env_kernel = k == "team" ? SERL_KERNEL_TEAM : k == "team2" ? SERL_KERNEL_TEAM2 : k == "team4" ? SERL_KERNEL_TEAM4
                    : k == "wave" ? SERL_KERNEL_WAVE : k == "half" ? SERL_KERNEL_HALF : SERL_KERNEL_AUTO;
    }
    c->env_waves_per_block = (e = getenv("SERL_WAVES_PER_BLOCK")) ? atoi(e) : -1;
    c->env_profile = getenv("SERL_PROFILE") != nullptr;
    c->env_split_actor = (e = getenv("SERL_SPLIT_ACTOR")) ? atoi(e) : 0;
    c->env_remote_actor = (e = getenv("SERL_REMOTE_ACTOR")) ? atoi(e) : 1;
    c->env_lane_regroup = ((e = getenv("SERL_LANE_WEIGHTS")) && !strcmp(e, "rows")) ? 0 : 1;
    c->env_mixed_place = (e = getenv("SERL_MIXED_PLACE")) ? atoi(e) : SERL_MIXED_PLACE_DEFAULT;
    c->env_jitter = (e = getenv("SERL_JITTER_SEED")) ? (unsigned)strtoul(e, nullptr, 0) : 0u;
    c->env_jitter_sites = (e = getenv("SERL_JITTER_SITES")) ? (unsigned)strtoul(e, nullptr, 0) : ~0u;
  }
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));
  HIP_TRY(hipMalloc((void **)&c->queue, SERL_QUEUE_COUNTERS * sizeof(int32_t)));
  *out = c;
  return SERL_OK;
}

int serl_ctx_destroy(serl_ctx *c)
{
  if (!c) return SERL_OK;
  (void)hipSetDevice(c->device);
  for (auto &s : c->slots) if (s.blob) (void)hipFree(s.blob);
  if (c->mail) (void)hipFree(c->mail);
  for (int i = 0; i < SERL_WT_SLOTS; ++i) { if (c->wt[i]) (void)hipFree(c->wt[i]); if (c->wt_done[i]) (void)hipEventDestroy(c->wt_done[i]); }
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->prof) (void)hipFree(c->prof);
  if (c->queue) (void)hipFree(c->queue);
  if (c->mixed_state) (void)hipFree(c->mixed_state);
  for (auto &e : c->mixed_ev) if (e) (void)hipEventDestroy(e);
  delete c;
  return SERL_OK;
}

int serl_ctx_load_build(serl_ctx *c, int slot, const serl_build_desc *b)
{
  if (!c || !b || slot < 0 || slot >= SERL_MAX_SLOTS) return fail(SERL_E_INVALID, "serl_ctx_load_build: bad argument");
  if (!b->ro || !b->t3 || !b->x0 || !b->dw0 || b->n_ro <= 0) return fail(SERL_E_INVALID, "serl_ctx_load_build: NULL table");
  if (!serl_has_wave_kernel(b->code))
    return fail(SERL_E_UNSUPPORTED, "serl_ctx_load_build: dynamics code variant not compiled into this library");
  HIP_TRY(hipSetDevice(c->device));
  BuildSlot &s = c->slots[slot];
  if (s.blob) { HIP_TRY(hipFree(s.blob)); s.blob = nullptr; s.loaded = false; }
  const size_t n = (size_t)b->n_ro + 46 + 19 + 31;
  std::vector<double> host(n);
  memcpy(host.data(), b->ro, sizeof(double) * b->n_ro);
  memcpy(host.data() + b->n_ro, b->t3, sizeof(double) * 46);
  memcpy(host.data() + b->n_ro + 46, b->x0, sizeof(double) * 19);
  memcpy(host.data() + b->n_ro + 65, b->dw0, sizeof(double) * 31);
  HIP_TRY(hipMalloc((void **)&s.blob, sizeof(double) * n));
  HIP_TRY(hipMemcpy(s.blob, host.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  s.n_ro = b->n_ro; s.code = b->code; s.ro_base = b->ro_base; s.dt = b->dt; s.loaded = true;
  return SERL_OK;
}

int serl_param_count(int S, int H, int L, int A) { return H * S + H + L * (H * H + 3 * H) + A * H + A; }

// argument checks of serl_rollout / serl_rollout_multi; *hint_out = the resolved kernel hint
static int serl_check_desc(serl_ctx *c, const serl_rollout_desc *d, int *hint_out)
{
  if (!c || !d) return fail(SERL_E_INVALID, "serl_rollout: NULL argument");
  if (d->build_slot < 0 || d->build_slot >= SERL_MAX_SLOTS || !c->slots[d->build_slot].loaded)
    return fail(SERL_E_INVALID, "serl_rollout: build slot not loaded");
  if (d->n_episodes <= 0) return fail(SERL_E_INVALID, "serl_rollout: n_episodes <= 0");
  if (!d->weights || !d->member_of_episode || (!d->ref && !d->ref_spec) || !d->fitness || !d->length_steps || !d->length_t || !d->cost_steps)
    return fail(SERL_E_INVALID, "serl_rollout: required device pointer is NULL");
  const bool general_env = d->env_config != SERL_ENV_ATTITUDE || d->incremental != 0;
  if (serl_env_state_dim(d->env_config, d->incremental) == 0)
    return fail(SERL_E_INVALID, "serl_rollout: env_config must be SERL_ENV_ATTITUDE, SERL_ENV_SYMMETRIC or SERL_ENV_FULL");
  if (d->state_dim != serl_env_state_dim(d->env_config, d->incremental) || d->action_dim != serl_env_action_dim(d->env_config))
    return fail(SERL_E_INVALID, "serl_rollout: state_dim / action_dim do not match the env configuration (attitude 7 / 3, symmetric 2 / 1, full 13 / 3; incremental adds action_dim observations)");
  if (general_env && d->lanes_per_wave > 0)
    return fail(SERL_E_UNSUPPORTED, "serl_rollout: the lane-per-episode kernels exist for the attitude task only");
  if (d->hidden < 2 || d->hidden > SERL_MAX_HIDDEN || d->num_layers < 0 || d->num_layers > 16)
    return fail(SERL_E_UNSUPPORTED, "serl_rollout: hidden size / layer count out of range");
  if (d->hidden % 4 != 0 || d->weight_stride % 4 != 0 || ((uintptr_t)d->weights & 15) != 0)
    return fail(SERL_E_UNSUPPORTED, "serl_rollout: hidden size and weight_stride must be multiples of 4 and weights 16-byte aligned (dwordx4 row loads)");
  if (d->activation < 0 || d->activation > 2) return fail(SERL_E_INVALID, "serl_rollout: activation");
  if (d->weight_stride < serl_param_count(d->state_dim, d->hidden, d->num_layers, d->action_dim))
    return fail(SERL_E_INVALID, "serl_rollout: weight_stride smaller than the parameter count");
  if (d->max_steps <= 0) return fail(SERL_E_INVALID, "serl_rollout: max_steps");
  if (d->kernel_hint < SERL_KERNEL_AUTO || d->kernel_hint > SERL_KERNEL_TEAM4) return fail(SERL_E_INVALID, "serl_rollout: kernel_hint");
  const int hint = d->lanes_per_wave > 0 ? SERL_KERNEL_AUTO : serl_resolve_hint(c, d->kernel_hint);
  if (d->hidden != 32 && (hint == SERL_KERNEL_TEAM4 || hint == SERL_KERNEL_HALF))
    return fail(SERL_E_UNSUPPORTED, "serl_rollout: kernel_hint TEAM4 / HALF needs hidden = 32 (other shapes: TEAM2, the actor wavefront runs the two episodes one after the other)");
  *hint_out = hint;
  return SERL_OK;
}

// RolloutArgs of one descriptor on its build's tables (what every launch path starts from)
static void serl_fill_args(serl_ctx *c, const serl_rollout_desc *d, RolloutArgs &a)
{
  const BuildSlot &s = c->slots[d->build_slot];
  memset(&a, 0, sizeof(a));
  a.d = *d;
  a.ro = s.blob; a.t3 = s.blob + s.n_ro; a.x0 = a.t3 + 46; a.dw0 = a.x0 + 19;
  a.dyn_dt = s.dt;
  a.jitter = c->env_jitter; a.jitter_sites = c->env_jitter_sites;
  a.prof = nullptr;
}

int serl_rollout(serl_ctx *c, const serl_rollout_desc *d, void *stream_)
{
  int hint = SERL_KERNEL_AUTO;
  { const int rc_ = serl_check_desc(c, d, &hint); if (rc_ != SERL_OK) return rc_; }
  const bool general_env = d->env_config != SERL_ENV_ATTITUDE || d->incremental != 0;
  HIP_TRY(hipSetDevice(c->device));
  const BuildSlot &s = c->slots[d->build_slot];
  hipStream_t stream = (hipStream_t)stream_;
  RolloutArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  a.ro = s.blob; a.t3 = s.blob + s.n_ro; a.x0 = a.t3 + 46; a.dw0 = a.x0 + 19;
  a.dyn_dt = s.dt;
  a.jitter = c->env_jitter; a.jitter_sites = c->env_jitter_sites;
  a.prof = nullptr;
  if (c->env_profile) {
    if (!c->prof) HIP_TRY(hipMalloc((void **)&c->prof, 32 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->prof, 0, 32 * sizeof(unsigned long long), stream));
    a.prof = c->prof;
  }
  int lanes = d->lanes_per_wave;
  // launches that share the GPU with others run on streams of their own: the context's one pair of timing events is
  // not recorded for them (an event in flight on one stream must not be re-recorded on another)
  const bool timed = d->concurrent_episodes <= 0;
  const int together = d->n_episodes + (d->concurrent_episodes > 0 ? d->concurrent_episodes : 0);   // episodes sharing the GPU
  a.e0 = 0; a.e_end = d->n_episodes;
  if (general_env) {
    // observation set / number of actions / incremental control from the descriptor: the team kernel while the episodes fit two
    // rounds of workgroups (2 x ~23 us per env step against 57 of one wavefront per episode), one wavefront per episode beyond
    if (!serl_has_wave_kernel(s.code)) return fail(SERL_E_UNSUPPORTED, "serl_rollout: code variant without a wave kernel");
    if (hint != SERL_KERNEL_AUTO && hint != SERL_KERNEL_TEAM && hint != SERL_KERNEL_WAVE)
      return fail(SERL_E_UNSUPPORTED, "serl_rollout: env configurations other than the attitude task run on the TEAM or WAVE kernels");
    const bool team = hint != SERL_KERNEL_AUTO ? hint == SERL_KERNEL_TEAM : (d->concurrent_episodes <= 0 ? together <= 2 * c->num_cus : together <= c->num_cus);
    if (timed) HIP_TRY(hipEventRecord(c->ev0, stream));
    if (team) {
      a.lanes = 1;
      a.block = 512;
      int launches = 0, last = 0;
      for (int e0 = 0; e0 < d->n_episodes; e0 += c->num_cus) {
        const int n = d->n_episodes - e0 < c->num_cus ? d->n_episodes - e0 : c->num_cus;
        a.e0 = e0; a.e_end = e0 + n;
        serl_launch_rollout_teamx(s.code, a, n, stream);
        HIP_TRY(hipGetLastError());
        ++launches; last = n;
      }
      serl_note_launch(c, SERL_FAMILY_TEAMX, last, 1, false, 1, true, launches, s.code);
    } else {
      const int wpb = serl_wave_kernel_waves_per_block(c, together + (timed ? 0 : 16));
      a.lanes = 1;
      a.block = 64 * wpb;
      serl_launch_rollout_wavex(s.code, a, (d->n_episodes + wpb - 1) / wpb, stream);
      HIP_TRY(hipGetLastError());
      serl_note_launch(c, SERL_FAMILY_WAVEX, (d->n_episodes + wpb - 1) / wpb, 1, false, 0, true, 1, s.code);
    }
    if (timed) HIP_TRY(hipEventRecord(c->ev1, stream));
    c->timed = timed;
    return SERL_OK;
  }
  if (lanes <= 0 && serl_use_team(c, hint, together)) {
    a.lanes = 1;
    a.block = 128;
    if (timed) HIP_TRY(hipEventRecord(c->ev0, stream));
    const int rgrid = serl_remote_actor_grid(c, d, together);
    if (rgrid > 0) {
      // mailboxes of this launch: a region of the ring, zeroed on the launch's stream in front of it
      const size_t region = (size_t)c->num_cus * sizeof(SerlMail);
      if (!c->mail) HIP_TRY(hipMalloc(&c->mail, SERL_MAIL_REGIONS * region));
      a.mail = (char *)c->mail + (size_t)c->mail_next * region;
      c->mail_next = (c->mail_next + 1) % SERL_MAIL_REGIONS;
      HIP_TRY(hipMemsetAsync(a.mail, 0, (size_t)d->n_episodes * sizeof(SerlMail), stream));
      a.block = 512;
      if (timed) HIP_TRY(hipEventRecord(c->ev0, stream));      // (re-recorded behind the memset: the kernel alone is timed)
      serl_launch_rollout_teamr(s.code, a, rgrid, stream);
      HIP_TRY(hipGetLastError());
      serl_note_launch(c, SERL_FAMILY_TEAMR, rgrid, 1, false, 4, true, 1, s.code);
      if (timed) HIP_TRY(hipEventRecord(c->ev1, stream));
      c->timed = timed;
      return SERL_OK;
    }
    serl_launch_rollout_team(s.code, a, d->n_episodes, stream, c->env_split_actor != 0);
    HIP_TRY(hipGetLastError());
    {
      const bool lds_actor = serl_lds_actor_shape(*d), split = !lds_actor && c->env_split_actor != 0 && d->hidden > 64 && d->hidden <= 128;
      serl_note_launch(c, lds_actor ? SERL_FAMILY_TEAM : split ? SERL_FAMILY_TEAMS2 : SERL_FAMILY_TEAMS, d->n_episodes, 1, false, split ? 2 : 1, !lds_actor, 1, s.code);
    }
    if (timed) HIP_TRY(hipEventRecord(c->ev1, stream));
    c->timed = timed;
    return SERL_OK;
  }
  const int teamg = (lanes <= 0 && serl_has_wave_kernel(s.code)) ? serl_use_teamg(c, d, hint, together) : 0;
  if (teamg) {
    a.lanes = 1;
    a.block = 512;
    if (timed) HIP_TRY(hipEventRecord(c->ev0, stream));
    // at most one team per CU; the episodes without a lane group at the start wait in the work queue (rollout_team_half.inc)
    int grid = (d->n_episodes + teamg - 1) / teamg;
    if (grid > c->num_cus && d->concurrent_episodes <= 0) {
      grid = c->num_cus;
      a.queue = serl_next_queue_counter(c);
      a.q0 = grid * teamg;
      HIP_TRY(hipMemsetAsync(a.queue, 0, sizeof(int32_t), stream));
    } else {
      a.queue = nullptr; a.q0 = d->n_episodes;
    }
    serl_launch_rollout_teamg(s.code, teamg, a, grid, stream);
    HIP_TRY(hipGetLastError());
    serl_note_launch(c, teamg == 4 ? SERL_FAMILY_TEAM4 : d->hidden != 32 ? SERL_FAMILY_TEAM2S : SERL_FAMILY_TEAM2, grid, teamg, a.queue != nullptr,
                     (teamg == 2 && d->hidden != 32) ? 2 : 1, true /* the lane-group kernels' actor wavefronts stream the weights */, 1, s.code);
    if (timed) HIP_TRY(hipEventRecord(c->ev1, stream));
    c->timed = timed;
    return SERL_OK;
  }
  // Round 6, the saturating regime: from 80 episodes per CU on, one episode per LANE -- 64 per wavefront, the lane's own actor in its registers over
  // the regrouped weights (rollout_device.h serl_actor_forward_lane32_t), at most two tables in flight and hinted index searches in the generated evaluation
  // (tools/dag/codegen_lane.py) -- outruns the queue launch of four-episode teams (episodes x 2 001 steps, M env-steps/s: 16 384: 47.4 against 49.1; 20 480: 53.0
  // against 49.0; 32 768: 82.8; 65 536: 165.6 against 49.6; profiles/r06_saturate.jsonl).  The SERL50 actor shape, nominal / ice code, the attitude task, alone on the GPU.
  if (lanes <= 0 && hint == SERL_KERNEL_AUTO && d->concurrent_episodes <= 0 && serl_has_lane_kernel(s.code) && d->hidden == 32 && d->state_dim == 7 &&
      d->action_dim == 3 && d->n_episodes >= 80 * c->num_cus)
    lanes = 64;
  if (lanes <= 0 && serl_has_wave_kernel(s.code) && serl_use_team_rounds(c, d, hint, together)) {
    // more than 4 x CUs episodes, alone on the GPU: rounds of 4 x CUs episodes (four per team, every CU busy), then the rest
    // with whichever team kernel suits its count -- 4 x CUs + 2 episodes cost 30.8 + 21.6 us per env step, not 62.9
    // ONE launch: a team of four lane groups per CU, the other episodes in the work queue (rollout_team_half.inc) -- a lane group
    // takes the next episode when its own ends, so episodes of different lengths (training: untrained actors crash within
    // seconds) keep every lane group busy instead of rounds that each wait for their longest episode.  (Round 2 ran rounds of
    // 4 x CUs episodes; with equal-length episodes the tail of a queue run costs the four-per-team step where a round of its own
    // could use the two-per-team kernel: 1 536 x 8 001 steps 7 % slower, any mix of lengths faster.)
    a.lanes = 1;
    a.block = 512;
    if (timed) HIP_TRY(hipEventRecord(c->ev0, stream));
    a.e0 = 0; a.e_end = d->n_episodes;
    // alone on the GPU: a team on every CU.  Beside other launches (a mixed-fault sweep: one launch per dynamics build on streams of
    // their own, each told the episodes of the others) the CUs are split by episode share, so that the teams of all launches are
    // resident together and every launch drains its own queue at the four-per-team rate (round 3 fell back to two episodes per
    // WAVEFRONT here: 24.1 M env-steps/s where the queue kernel gives 33.6 M on the nominal workload)
    int grid = c->num_cus;
    if (d->concurrent_episodes > 0) {
      grid = (int)((long long)c->num_cus * d->n_episodes / together);
      grid = grid < 1 ? 1 : grid;
    }
    if (grid * 4 >= d->n_episodes) {          // (a small share of a large sweep: every episode has a lane group, no queue)
      grid = (d->n_episodes + 3) / 4;
      a.queue = nullptr; a.q0 = d->n_episodes;
    } else {
      a.queue = serl_next_queue_counter(c);
      a.q0 = 4 * grid;
      HIP_TRY(hipMemsetAsync(a.queue, 0, sizeof(int32_t), stream));
    }
    serl_launch_rollout_teamg(s.code, 4, a, grid, stream);
    HIP_TRY(hipGetLastError());
    serl_note_launch(c, SERL_FAMILY_TEAM4, grid, 4, a.queue != nullptr, 1, true, 1, s.code);
    if (timed) HIP_TRY(hipEventRecord(c->ev1, stream));
    c->timed = timed;
    return SERL_OK;
  }
  if (lanes <= 0 && serl_has_wave_kernel(s.code) && serl_use_half(c, d, hint, together)) {
    const int waves = (d->n_episodes + 1) / 2;
    int wpb = (waves + c->num_cus - 1) / c->num_cus;
    wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
    a.lanes = 1;
    a.block = 64 * wpb;
    const int grid = (waves + wpb - 1) / wpb;
    if (timed) HIP_TRY(hipEventRecord(c->ev0, stream));
    serl_launch_rollout_half(s.code, a, grid, stream);
    HIP_TRY(hipGetLastError());
    serl_note_launch(c, SERL_FAMILY_HALF, grid, 2, false, 0, true, 1, s.code);
    if (timed) HIP_TRY(hipEventRecord(c->ev1, stream));
    c->timed = timed;
    return SERL_OK;
  }
  if (lanes <= 0 && serl_has_wave_kernel(s.code)) {
    // default: one wavefront per episode (model glue wave-uniform, look-ups / actor rows / ODE5 states per lane)
    // side-by-side launches round their workgroup counts up separately: leave room for a few partial workgroups
    const int wpb = serl_wave_kernel_waves_per_block(c, together + (timed ? 0 : 16));
    a.lanes = 1;
    a.block = 64 * wpb;
    const int grid = (d->n_episodes + wpb - 1) / wpb;
    if (timed) HIP_TRY(hipEventRecord(c->ev0, stream));
    serl_launch_rollout_wave(s.code, a, grid, stream);
    HIP_TRY(hipGetLastError());
    serl_note_launch(c, SERL_FAMILY_WAVE, grid, 1, false, 0, true, 1, s.code);
    if (timed) HIP_TRY(hipEventRecord(c->ev1, stream));
    c->timed = timed;
    return SERL_OK;
  }
  if (!serl_has_lane_kernel(s.code))
    return fail(SERL_E_UNSUPPORTED, "serl_rollout: lanes_per_wave > 0 (lane-per-episode kernels): unknown code variant");
  if (lanes <= 0) {
    // A wavefront takes the same time per env step whether it carries 1 or 64 episodes, and wavefronts that
    // share a CU slow each other down (measured: profiles/r01_microbench.md), so episodes are spread one
    // wavefront per CU first (256 CUs) and only then packed into lanes.
    lanes = (d->n_episodes + 255) / 256;
    if (lanes < 1) lanes = 1;
  }
  if (lanes > 64) lanes = 64;
  a.lanes = lanes;
  const int waves = (d->n_episodes + lanes - 1) / lanes;
  const int wpb = serl_waves_per_block(c, waves);
  a.block = 64 * wpb;
  const int grid = (waves + wpb - 1) / wpb;
  if (c->env_lane_regroup && d->hidden == 32 && d->state_dim == 7 && d->action_dim == 3 && d->n_members <= (1 << 22)) {      // (rollout_device.h serl_lane_actor_ok)
    const int rc = serl_regroup_weights(c, d, a, stream);
    if (rc != SERL_OK) return rc;
  }
  if (timed) HIP_TRY(hipEventRecord(c->ev0, stream));
  serl_launch_rollout_lane(s.code, a, grid, stream);
  HIP_TRY(hipGetLastError());
  if (a.wt) HIP_TRY(hipEventRecord(c->wt_done[c->wt_slot_of_launch], stream));
  serl_note_launch(c, SERL_FAMILY_LANE, grid, lanes, false, 0, true, 1, s.code);
  if (timed) HIP_TRY(hipEventRecord(c->ev1, stream));
  c->timed = timed;
  return SERL_OK;
}

// Several descriptors -- one per dynamics build of a mixed-fault population -- in ONE launch of one code object (rollout_team4_mixed.hip).
// Eligible: 2 .. SERL_MIXED_MAX descriptors of the attitude task with the LDS-sized actor shape (hidden 32), code variants nominal / ice, automatic
// kernel choice, more than 2 x CUs episodes together (the four-episodes-per-team size class).  Anything else returns SERL_E_UNSUPPORTED and the
// caller launches the descriptors one by one (serl_rollout with concurrent_episodes, streams of their own).
int serl_rollout_multi(serl_ctx *c, int32_t n, const serl_rollout_desc *descs, void *stream_)
{
  if (!c || !descs) return fail(SERL_E_INVALID, "serl_rollout_multi: NULL argument");
  if (n < 2 || n > SERL_MIXED_MAX) return fail(SERL_E_UNSUPPORTED, "serl_rollout_multi: 2 .. 4 descriptors");
  int together = 0;
  for (int k = 0; k < n; ++k) {
    const serl_rollout_desc *d = descs + k;
    int hint = SERL_KERNEL_AUTO;
    const int rc_ = serl_check_desc(c, d, &hint);
    if (rc_ != SERL_OK) return rc_;
    const int code = c->slots[d->build_slot].code;
    if (d->env_config != SERL_ENV_ATTITUDE || d->incremental != 0 || d->hidden != 32 || d->num_layers > 3 || d->lanes_per_wave > 0 ||
        hint != SERL_KERNEL_AUTO || (code != SERL_DYN_NOMINAL && code != SERL_DYN_ICE))
      return fail(SERL_E_UNSUPPORTED, "serl_rollout_multi: attitude task, hidden 32, code variants nominal / ice, kernel_hint AUTO");
    together += d->n_episodes;
  }
  if (together <= 2 * c->num_cus) return fail(SERL_E_UNSUPPORTED, "serl_rollout_multi: for more than 2 x CUs episodes (four per team)");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream = (hipStream_t)stream_;
  SerlMixedArgs m;
  memset(&m, 0, sizeof(m));
  m.n = n;
  m.place = c->env_mixed_place;
  int state_slot = -1;
  c->mixed_last = -1;
  if (m.place != 0) {
    if (!c->mixed_state) HIP_TRY(hipMalloc((void **)&c->mixed_state, SERL_MIXED_STATES * SERL_MIXED_STATE * sizeof(int32_t)));
    state_slot = c->mixed_state_next;
    c->mixed_last = state_slot;
    m.state = c->mixed_state + (size_t)state_slot * SERL_MIXED_STATE;
    c->mixed_state_next = (c->mixed_state_next + 1) % SERL_MIXED_STATES;
    // (two launches must never count into one census: the workgroups would disagree about the assignment)
    if (c->mixed_ev[state_slot]) HIP_TRY(hipStreamWaitEvent(stream, c->mixed_ev[state_slot], 0));
    else HIP_TRY(hipEventCreateWithFlags(&c->mixed_ev[state_slot], hipEventDisableTiming));
    HIP_TRY(hipMemsetAsync(m.state, 0, SERL_MIXED_STATE * sizeof(int32_t), stream));
  }
  const bool queue = together > 4 * c->num_cus;      // beyond four per CU: every part drains a work queue of its own on its share of the CUs
  // workgroups of every part: one per four episodes, or (queue) the part's share of the CUs
  int grids[SERL_MIXED_MAX], total_wg = 0;
  for (int k = 0; k < n; ++k) {
    const serl_rollout_desc *d = descs + k;
    grids[k] = (d->n_episodes + 3) / 4;
    if (queue) {
      int share = (int)((long long)c->num_cus * d->n_episodes / together);
      share = share < 1 ? 1 : share;
      if (share * 4 < d->n_episodes) grids[k] = share;
    }
    total_wg += grids[k];
  }
  // The census placement needs every workgroup resident, one per CU (a team fills a CU's LDS).  The parts round up on their own -- 341 / 341 / 342
  // episodes on 256 CUs are 86 + 86 + 86 = 258 workgroups, and a share bumped to 1 does the same in queue mode -- so the overflow is taken from the
  // largest part, which drains the episodes it loses through its work queue.
  while (total_wg > c->num_cus) {
    int big = 0;
    for (int k = 1; k < n; ++k) big = grids[k] > grids[big] ? k : big;
    if (grids[big] <= 1) break;
    const int cut = total_wg - c->num_cus < grids[big] - 1 ? total_wg - c->num_cus : grids[big] - 1;
    grids[big] -= cut; total_wg -= cut;
  }
  if (total_wg > c->num_cus && m.place == 2) m.place = 1;      // (more parts than CUs: never all resident -- tickets at once instead of the census' time-out)
  int wg = 0;
  for (int k = 0; k < n; ++k) {
    const serl_rollout_desc *d = descs + k;
    RolloutArgs &a = m.a[k];
    serl_fill_args(c, d, a);
    a.lanes = 1;
    a.block = 512;
    const int grid = grids[k];
    a.queue = nullptr; a.q0 = d->n_episodes;
    if (grid * 4 < d->n_episodes) {
      a.queue = serl_next_queue_counter(c);
      a.q0 = 4 * grid;
      HIP_TRY(hipMemsetAsync(a.queue, 0, sizeof(int32_t), stream));
    }
    m.first_wg[k] = wg;
    m.code[k] = c->slots[d->build_slot].code;
    // the lane-group kernels find their episodes at e0 + blockIdx.x * 4 + group: shift e0 (and the queue's first episode stays absolute)
    a.e0 = -4 * wg; a.e_end = d->n_episodes;
    wg += grid;
  }
  m.first_wg[n] = wg;
  HIP_TRY(hipEventRecord(c->ev0, stream));
  serl_launch_rollout_team4_mixed(m, wg, stream);
  HIP_TRY(hipGetLastError());
  {
    bool one_code = true;
    for (int k = 1; k < n; ++k) one_code = one_code && m.code[k] == m.code[0];
    bool any_queue = false;
    for (int k = 0; k < n; ++k) any_queue = any_queue || m.a[k].queue != nullptr;
    serl_note_launch(c, SERL_FAMILY_TEAM4_MIXED, wg, 4, any_queue, 1, true, 1, one_code ? m.code[0] : -1);
  }
  if (state_slot >= 0) HIP_TRY(hipEventRecord(c->mixed_ev[state_slot], stream));
  HIP_TRY(hipEventRecord(c->ev1, stream));
  c->timed = true;
  return SERL_OK;
}

int serl_dyn_open_loop(serl_ctx *c, int slot, int32_t n_episodes, int32_t T, const double *cmds, double *states,
                       int32_t lanes_per_wave, int32_t kernel_hint, void *stream_)
{
  if (!c || !cmds || !states || n_episodes <= 0 || T <= 0) return fail(SERL_E_INVALID, "serl_dyn_open_loop: bad argument");
  const int hint = lanes_per_wave > 0 ? SERL_KERNEL_AUTO : serl_resolve_hint(c, kernel_hint);
  if (hint != SERL_KERNEL_AUTO && hint != SERL_KERNEL_TEAM && hint != SERL_KERNEL_WAVE)
    return fail(SERL_E_UNSUPPORTED, "serl_dyn_open_loop: kernel_hint must be AUTO, TEAM or WAVE");
  if (slot < 0 || slot >= SERL_MAX_SLOTS || !c->slots[slot].loaded) return fail(SERL_E_INVALID, "serl_dyn_open_loop: build slot not loaded");
  HIP_TRY(hipSetDevice(c->device));
  const BuildSlot &s = c->slots[slot];
  RolloutArgs a;
  memset(&a, 0, sizeof(a));
  a.d.n_episodes = n_episodes;
  a.ro = s.blob; a.t3 = s.blob + s.n_ro; a.x0 = a.t3 + 46; a.dw0 = a.x0 + 19;
  a.dyn_dt = s.dt;
  a.jitter = c->env_jitter; a.jitter_sites = c->env_jitter_sites;
  hipStream_t stream = (hipStream_t)stream_;
  if (lanes_per_wave <= 0 && serl_use_team(c, hint, n_episodes)) {
    a.lanes = 1;
    a.block = 128;
    HIP_TRY(hipEventRecord(c->ev0, stream));
    serl_launch_dyn_team(s.code, a, cmds, states, T, n_episodes, stream);
    HIP_TRY(hipGetLastError());
    serl_note_launch(c, SERL_FAMILY_TEAM, n_episodes, 1, false, 0, false, 1, s.code);      // (dynamics only: the team without an actor wavefront)
    HIP_TRY(hipEventRecord(c->ev1, stream));
    c->timed = true;
    return SERL_OK;
  }
  if (lanes_per_wave <= 0 && serl_has_wave_kernel(s.code)) {
    const int wpb = serl_wave_kernel_waves_per_block(c, n_episodes);
    a.lanes = 1;
    a.block = 64 * wpb;
    HIP_TRY(hipEventRecord(c->ev0, stream));
    serl_launch_dyn_wave(s.code, a, cmds, states, T, (n_episodes + wpb - 1) / wpb, stream);
    HIP_TRY(hipGetLastError());
    serl_note_launch(c, SERL_FAMILY_WAVE, (n_episodes + wpb - 1) / wpb, 1, false, 0, false, 1, s.code);
    HIP_TRY(hipEventRecord(c->ev1, stream));
    c->timed = true;
    return SERL_OK;
  }
  if (!serl_has_lane_kernel(s.code))
    return fail(SERL_E_UNSUPPORTED, "serl_dyn_open_loop: lanes_per_wave > 0: unknown code variant");
  int lanes = lanes_per_wave <= 0 ? (n_episodes + 255) / 256 : lanes_per_wave;
  if (lanes > 64) lanes = 64;
  a.lanes = lanes;
  const int nwaves = (n_episodes + lanes - 1) / lanes;
  const int wpb = serl_waves_per_block(c, nwaves);
  a.block = 64 * wpb;
  const int grid = (nwaves + wpb - 1) / wpb;
  HIP_TRY(hipEventRecord(c->ev0, stream));
  serl_launch_dyn_lane(s.code, a, cmds, states, T, grid, stream);
  HIP_TRY(hipGetLastError());
  serl_note_launch(c, SERL_FAMILY_LANE, grid, lanes, false, 0, false, 1, s.code);
  HIP_TRY(hipEventRecord(c->ev1, stream));
  c->timed = true;
  return SERL_OK;
}

/* development aid (SERL_PROFILE=1): shader-clock cycles wave 0 of workgroup 0 spent in the actor forward, the
 * dynamics step and the env bookkeeping during the last serl_rollout, and its number of env steps */
int serl_debug_profile(serl_ctx *c, unsigned long long out[32])
{
  if (!c || !out || !c->prof) return fail(SERL_E_INVALID, "serl_debug_profile: profiling not enabled (SERL_PROFILE=1)");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, c->prof, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SERL_OK;
}

int serl_debug_mixed_placement(serl_ctx *c, int32_t out[4])
{
  if (!c || !out) return fail(SERL_E_INVALID, "serl_debug_mixed_placement: NULL argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!c->mixed_state || c->mixed_last < 0) return SERL_OK;          // no launch, or SERL_MIXED_PLACE=0
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  int32_t h[SERL_MIXED_STATE];
  HIP_TRY(hipMemcpy(h, c->mixed_state + (size_t)c->mixed_last * SERL_MIXED_STATE, sizeof(h), hipMemcpyDeviceToHost));
  out[0] = c->env_mixed_place == 1 ? 2 : h[SERL_MIXED_UNITS + 1]; out[1] = h[SERL_MIXED_UNITS];
  for (int u = 0; u < SERL_MIXED_UNITS; ++u) { out[2] += h[u] == 2; out[3] += h[u] == 1; }
  return SERL_OK;
}

int serl_last_rollout_ms(serl_ctx *c, float *ms)
{
  if (!c || !ms || !c->timed) return fail(SERL_E_INVALID, "serl_last_rollout_ms: no rollout recorded");
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return SERL_OK;
}

int serl_last_rollout_info(serl_ctx *c, int32_t out[8])
{
  if (!c || !out) return fail(SERL_E_INVALID, "serl_last_rollout_info: NULL argument");
  if (c->last_info[0] == SERL_FAMILY_NONE) return fail(SERL_E_INVALID, "serl_last_rollout_info: no rollout recorded");
  for (int i = 0; i < 8; ++i) out[i] = c->last_info[i];
  return SERL_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// SSNE weight-tensor edits (base/core/mod_neuro_evo.py) -- elementwise / row kernels
// ---------------------------------------------------------------------------------------------
__global__ void ga_clone_kernel(float *w, int64_t stride, int32_t P, const int32_t *src, const int32_t *dst, int32_t n)
{
  const int pair = blockIdx.y;
  if (pair >= n) return;
  const float *s = w + (size_t)src[pair] * stride;
  float *d = w + (size_t)dst[pair] * stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) d[i] = s[i];
}

// ops[i] = {offset, length, dir}; applied IN ORDER (later ops see earlier ones, like the reference's
// sequential row copies), one workgroup, threads stride over the row.
__global__ void ga_crossover_kernel(float *w, int64_t stride, int32_t ma, int32_t mb, const int32_t *ops, int32_t n_ops)
{
  float *a = w + (size_t)ma * stride, *b = w + (size_t)mb * stride;
  for (int o = 0; o < n_ops; ++o) {
    const int off = ops[3 * o], len = ops[3 * o + 1], dir = ops[3 * o + 2];
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
      if (dir == 0) a[off + i] = b[off + i]; else b[off + i] = a[off + i];
    }
    __syncthreads();
  }
}

// Sequential sparse edits of one member (edits may hit the same weight twice; order matters).
// kind 0: w += z * (strength * w)   [random.gauss(0, strength*w) = z*sigma]   kind 1: w = z
// followed by the reference's hard clamp to +-1e6 (mod_neuro_evo.py:57-59,366).
__global__ void ga_mutate_kernel(float *w, int64_t stride, int32_t member, const int32_t *idx, const int32_t *kind,
                                 const float *z, const float *strength, int32_t n)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float *p = w + (size_t)member * stride;
  for (int i = 0; i < n; ++i) {
    float v = p[idx[i]];
    if (kind[i] == 0) v = v + z[i] * (strength[i] * v); else v = z[i];
    v = fminf(fmaxf(v, -1000000.0f), 1000000.0f);
    p[idx[i]] = v;
  }
}

__global__ void ga_scaled_perturb_kernel(float *w, int64_t stride, int32_t member, const int32_t *seg_off,
                                         const int32_t *seg_len, int32_t n_seg, const float *delta, const float *scaling)
{
  float *p = w + (size_t)member * stride;
  int base = 0;
  for (int s = 0; s < n_seg; ++s) {
    const int off = seg_off[s], len = seg_len[s];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x)
      p[off + i] = p[off + i] + delta[base + i] / scaling[base + i];
    base += len;
  }
}

extern "C" {

int serl_ga_clone(serl_ctx *c, float *weights, int64_t stride, int32_t P, const int32_t *src, const int32_t *dst,
                  int32_t n, void *stream)
{
  if (!c || !weights || !src || !dst || n < 0 || P <= 0) return fail(SERL_E_INVALID, "serl_ga_clone: bad argument");
  if (n == 0) return SERL_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(ga_clone_kernel, dim3((P + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, weights, stride, P, src, dst, n);
  HIP_TRY(hipGetLastError());
  return SERL_OK;
}

int serl_ga_crossover(serl_ctx *c, float *weights, int64_t stride, int32_t ma, int32_t mb, const int32_t *ops,
                      int32_t n_ops, void *stream)
{
  if (!c || !weights || (!ops && n_ops > 0) || n_ops < 0) return fail(SERL_E_INVALID, "serl_ga_crossover: bad argument");
  if (n_ops == 0) return SERL_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(ga_crossover_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, weights, stride, ma, mb, ops, n_ops);
  HIP_TRY(hipGetLastError());
  return SERL_OK;
}

int serl_ga_mutate(serl_ctx *c, float *weights, int64_t stride, int32_t member, const int32_t *idx, const int32_t *kind,
                   const float *z, const float *strength, int32_t n, void *stream)
{
  if (!c || !weights || n < 0) return fail(SERL_E_INVALID, "serl_ga_mutate: bad argument");
  if (n == 0) return SERL_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(ga_mutate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, weights, stride, member, idx, kind, z, strength, n);
  HIP_TRY(hipGetLastError());
  return SERL_OK;
}

int serl_ga_scaled_perturb(serl_ctx *c, float *weights, int64_t stride, int32_t member, const int32_t *seg_offset,
                           const int32_t *seg_length, int32_t n_seg, const float *delta, const float *scaling, void *stream)
{
  if (!c || !weights || !seg_offset || !seg_length || !delta || !scaling || n_seg <= 0)
    return fail(SERL_E_INVALID, "serl_ga_scaled_perturb: bad argument");
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(ga_scaled_perturb_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, weights, stride, member,
                     seg_offset, seg_length, n_seg, delta, scaling);
  HIP_TRY(hipGetLastError());
  return SERL_OK;
}

}  // extern "C"
