// serl_mixed.h -- arguments of the one-code-object launch of a mixed-fault sweep (rollout_team4_mixed.hip, serl_capi.hip serl_rollout_multi).
#pragma once
#define SERL_MIXED_MAX 4
#define SERL_MIXED_UNITS 256              // instruction-cache neighbourhoods: (XCC_ID * 4 + SE_ID) * 8 + (CU_ID + 1) / 2
#define SERL_MIXED_STATE (SERL_MIXED_UNITS + 2 + SERL_MIXED_MAX)      // int32 words of placement state per launch: census, arrivals, decision, tickets
struct SerlMixedArgs {
  RolloutArgs a[SERL_MIXED_MAX];           // part k: what a launch of its own would get -- except e0, which is shifted by the part's first workgroup
                                           // (the lane-group kernels find their episodes at e0 + index * groups + group)
  int32_t first_wg[SERL_MIXED_MAX + 1];    // workgroup INDICES [first_wg[k], first_wg[k + 1]) run part k (an index is blockIdx.x or what the placement hands out)
  int32_t code[SERL_MIXED_MAX];            // SERL_DYN_NOMINAL / SERL_DYN_ICE
  int32_t n;
  int32_t place;                           // 0: part by blockIdx range; 1 / 2: by where the workgroup runs (rollout_team4_mixed.hip)
  int32_t *state;                          // place != 0: [SERL_MIXED_STATE], zeroed per launch
};
void serl_launch_rollout_team4_mixed(const SerlMixedArgs &m, int grid, hipStream_t stream);
