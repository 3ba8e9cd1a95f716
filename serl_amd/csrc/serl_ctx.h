// serl_ctx.h -- internals shared by the translation units behind the C ABI (serl_capi.hip, serl_ga.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/serl_amd.h"

#define SERL_MAX_SLOTS 16
#define SERL_QUEUE_COUNTERS 64
#define SERL_MIXED_PLACE_DEFAULT 2
#define SERL_WT_SLOTS 4                  // regrouped weight copies of lane-per-episode launches in flight (a ring)
#define SERL_MAIL_REGIONS 8              // mailbox regions of remote-actor launches in flight (a ring, like the queue counters)
#define SERL_MIXED_STATES 8             // placement states of serl_rollout_multi launches in flight (a ring, like the queue counters)

int serl_fail(int code, const std::string &msg);      // records the thread-local message of serl_last_error()
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return serl_fail(SERL_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

struct BuildSlot {
  bool loaded = false;
  int32_t code = 0;
  uint64_t ro_base = 0;
  double dt = 0.01;
  double *blob = nullptr;   // one device allocation: ro | t3[46] | x0[19] | dw0[31]
  size_t n_ro = 0;
};

struct serl_ctx {
  int device = 0;
  int num_cus = 256;                    // multiProcessorCount of this context's device
  int lds_per_block = 65536;            // sharedMemPerBlockOptin
  // environment overrides, read once when the context is made (-1 = not set)
  int env_kernel = 0 /* serl_kernel_hint from SERL_KERNEL */, env_waves_per_block = -1, env_profile = 0;
  int env_split_actor = 0;              // SERL_SPLIT_ACTOR=1: streamed actors of one-episode teams (hidden > 64) on TWO actor wavefronts that share the forward pass
                                        // (rollout_teams2_<v>.hip; measured slower than one wavefront with the specialised forward: profiles/r04_experiments.md)
  unsigned env_jitter_sites = ~0u;      // SERL_JITTER_SITES: classes of sites that pause (citation_wave.h; all by default)
  int env_mixed_place = SERL_MIXED_PLACE_DEFAULT;      // SERL_MIXED_PLACE: how serl_rollout_multi places the parts' workgroups (serl_mixed.h)
  int32_t *mixed_state = nullptr;       // device [SERL_MIXED_STATES][SERL_MIXED_STATE]
  int mixed_state_next = 0, mixed_last = -1;            // (mixed_last: the state of the most recent launch, for serl_debug_mixed_placement)
  hipEvent_t mixed_ev[SERL_MIXED_STATES] = {};        // the launch that used a state last: a later launch on another stream waits for it before it zeroes the state
  unsigned env_jitter = 0;              // SERL_JITTER_SEED: seed of the hand-over stress builds' pauses (libserl_amd_jitter.so; the product ignores it)
  BuildSlot slots[SERL_MAX_SLOTS];
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  unsigned long long *prof = nullptr;   // device [32], allocated when SERL_PROFILE=1
  int32_t *queue = nullptr;             // device [SERL_QUEUE_COUNTERS]: work-queue counters of the multi-episode team kernels, one per LAUNCH (a ring)
  int queue_next = 0;
  int env_remote_actor = 1;             // SERL_REMOTE_ACTOR=0: streamed actors of one-episode teams stay on the team's CU (rollout_team_<v>.hip serl_rollout_teams_kernel_; A/B)
  void *mail = nullptr;                 // device [SERL_MAIL_REGIONS][num_cus] SerlMail: team <-> remote actor workgroup (rollout_teamr_<v>.hip)
  int mail_next = 0;
  int env_lane_regroup = 1;             // SERL_LANE_WEIGHTS=rows: the lane-per-episode kernels walk [members][P] rows instead of the regrouped copy (A/B)
  void *wt[SERL_WT_SLOTS] = {};         // lane-per-episode kernels: regrouped weights [ceil(P / 4)][members up to 64][4] of the launches in flight (a ring; grown on demand)
  size_t wt_cap[SERL_WT_SLOTS] = {};
  int wt_next = 0;
  hipEvent_t wt_done[SERL_WT_SLOTS] = {};      // recorded behind the launch that reads a copy: the next launch to take the slot waits for it on ITS stream (no host wait)
  int wt_slot_of_launch = -1;
  int32_t last_info[8] = {};            // serl_last_rollout_info: what the most recent rollout call launched (family, workgroups, episodes per team, queue, actor wavefronts, streamed, launches, code)
};
