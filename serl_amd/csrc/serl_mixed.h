// serl_mixed.h -- arguments of the one-code-object launch of a mixed-fault sweep (rollout_team4_mixed.hip, serl_capi.hip serl_rollout_multi).
#pragma once
#define SERL_MIXED_MAX 4
struct SerlMixedArgs {
  RolloutArgs a[SERL_MIXED_MAX];           // part k: what a launch of its own would get -- except e0, which is shifted by the part's first workgroup
                                           // (the lane-group kernels find their episodes at e0 + blockIdx.x * groups + group)
  int32_t first_wg[SERL_MIXED_MAX + 1];    // workgroups [first_wg[k], first_wg[k + 1]) run part k
  int32_t code[SERL_MIXED_MAX];            // SERL_DYN_NOMINAL / SERL_DYN_ICE
  int32_t n;
};
void serl_launch_rollout_team4_mixed(const SerlMixedArgs &m, int grid, hipStream_t stream);
