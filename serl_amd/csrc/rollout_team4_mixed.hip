// rollout_team4_mixed.hip -- ONE code object for the mixed-fault sweep (BASELINE config 5; round 5).
//
// A mixed-fault population (fault mode per episode over be / jr / sa / se / ice / cg: envs/phlabenv.py:114-165, envs/{be,jr,sa,se}/citation.py:71-79)
// needs the dynamics of several BUILDS in one evaluation: h2000_v90 and cg run the 'nominal' code variant on different tables, ice runs the 'ice'
// variant.  Until round 5 that was one launch per build, side by side on streams of their own -- and as soon as the ice kernel runs beside the
// nominal one every launch is 12 % slower (be + cg, the same code on different tables, is free): two CUs share an instruction cache, one
// variant's loop fills it, and separate launches cannot choose their CUs.
// Here the four-episodes-per-team device functions of BOTH code variants (rollout_team.inc + rollout_team_half.inc, lane groups of 16) are
// compiled into one kernel, and a workgroup picks the LAUNCH PART it runs -- variant, tables, episode range, work queue -- by the CU it finds
// itself on, so that CUs which share an instruction cache run the same variant (serl_mixed_place below).  Every part is what a launch of its own
// would be -- its own descriptor, tables, episode range and work-queue counter -- so the results are those of the separate launches, bit for bit.
#define CITW_SEARCH_BATCH 1
#define CITW_GROUP_LANES 16
#define CITW_MAX_WAVES 4          // blackboard rows: one per episode of the team
#define CITW_M_ROWS 32            // libm result rows: one per (team wavefront, episode)
#define CITW_OUT2_ROWS 1
#define CITW_INV_SLOTS 8
#define SERL_NO_CHUNKED_ACTOR 1      // (H = 32 actors only: serl_capi.hip)
#define SERL_TEAM_NO_ENTRY 1         // device functions only: the kernel is below
#define SERL_TEAMG_WG g_mixed_wg     // the workgroup's index in the launch: blockIdx.x, or the one the placement below gave it
#include "citation_wave.h"
#include "rollout_device.h"
__shared__ int g_mixed_wg, g_mixed_part, g_mixed_census[256];
#include "gen/citation_nominal_wave.inc"
#include "gen/citation_nominal_teamg.inc"
#define VARIANT nominal
#include "rollout_team.inc"
#undef VARIANT
#include "gen/citation_ice_wave.inc"
#include "gen/citation_ice_teamg.inc"
#define VARIANT ice
#include "rollout_team.inc"
#undef VARIANT
#include "serl_mixed.h"

static_assert(citw_nominal_team_WAVES == citw_ice_team_WAVES, "one workgroup shape for both variants");

// Which part of the launch this workgroup runs, and as which of the part's workgroups (wavefront 0; result in g_mixed_part / g_mixed_wg).
//
// Two CUs share an instruction cache, and the loop of ONE code variant fills it (81 KB of code per variant against 64 KB): SQC_ICACHE_MISSES per
// workgroup and env step are 5 with the same variant on both CUs and 35 - 50 with the other variant next door, and that -- not "two code
// objects" -- is the "+12 % as soon as ice runs beside nominal" of rounds 4 / 5 (profiles/r05_experiments.md section 11: the pairs are CU_ID
// (1,2) (3,4) (5,6) (7,8) of a shader engine, CU 0 has no partner; found by A/B of the pairings, harvested ids differ from GPU to GPU).  The launch is
// as long as its slowest workgroup, so ONE mixed pair costs the whole launch the 12 %.  Workgroups therefore take their (part, index) by where they
// run (HW_REG_HW_ID: CU_ID[11:8], SE_ID[15:13]; HW_REG_XCC_ID[3:0]), and what a workgroup computes depends on (part, index) only: results do not
// depend on the placement.
//   place 0   by blockIdx range (parts contiguous; the dispatcher deals neighbouring blockIdx to different XCDs, so pairs mix)
//   place 1   tickets: a pair prefers one part (golden-ratio sequence over the pair's ordinal, in proportion to the parts' sizes) and its workgroups
//             take the part's next free index, else one of a part with the same code, else any.  A few pairs end up mixed, differently every launch.
//   place 2   census: every workgroup registers in its pair and waits until all have (the grid is at most one workgroup per CU: all resident),
//             then each computes the same assignment from the census -- the ice parts get whole pairs first, then single workgroups, so that no
//             pair is mixed unless the counts make it unavoidable.  If the workgroups do not all arrive within a millisecond (a GPU shared with another
//             launch), or a pair reports more than two, the launch falls back to place 1 -- decided once, by compare-and-swap, for all.
static __device__ __forceinline__ void serl_mixed_place(const SerlMixedArgs &m, const SerlMixedArgs *km)
{
  const int lane = threadIdx.x & 63;
  int k = 0, v = (int)blockIdx.x;
#pragma unroll
  for (int i = 1; i < SERL_MIXED_MAX; ++i) k = (i < m.n && (int)blockIdx.x >= m.first_wg[i]) ? i : k;
  const int total = km->first_wg[m.n];
  int need_ice = 0;
  for (int j = 0; j < m.n; ++j) need_ice += km->code[j] == SERL_DYN_ICE ? km->first_wg[j + 1] - km->first_wg[j] : 0;
  if (m.place != 0 && need_ice != 0 && need_ice != total) {
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4), xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u;
    const unsigned cu = (hw >> 8) & 15u, se = (hw >> 13) & 7u;
    const int unit = (int)((((xcc & 7u) * 4u + (se & 3u)) * 8u + (((cu + 1u) >> 1) & 7u)));
    int *census = m.state, *arrived = m.state + SERL_MIXED_UNITS, *decision = arrived + 1, *ticket = decision + 1;
    int mine = 0, dec = 2;
    if (m.place == 2) {
      if (lane == 0) {
        mine = atomicAdd(census + unit, 1);
        __threadfence();
        if (atomicAdd(arrived, 1) + 1 == total) atomicCAS(decision, 0, 1);
        const unsigned long long t0 = wall_clock64();                      // (100 MHz)
        while ((dec = __hip_atomic_load(decision, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
          if (wall_clock64() - t0 > 100000ull) atomicCAS(decision, 0, 2);
          __builtin_amdgcn_s_sleep(8);
        }
      }
      mine = __builtin_amdgcn_readfirstlane(mine);
      dec = __builtin_amdgcn_readfirstlane(dec);
    }
    if (dec == 1) {
      int most = 0;
#pragma unroll
      for (int i = 0; i < SERL_MIXED_UNITS / 64; ++i) {
        const int c = __hip_atomic_load(census + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g_mixed_census[lane + 64 * i] = c;
        most = c > most ? c : most;
      }
      if (__ballot(most > 2) != 0ull) {                 // (the same census, the same verdict in every workgroup)
        dec = 2;
        // ... and on the record: serl_debug_mixed_placement reports tickets, not the census (a device whose HW_ID-to-pair mapping collides
        // would otherwise lose the placement silently); a workgroup that has not read the decision yet finds 2 and takes a ticket at once
        if (lane == 0) __hip_atomic_store(decision, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (dec == 1) {
      if (lane == 0) {
        int pairs = 0, singles = 0;
        for (int u = 0; u < SERL_MIXED_UNITS; ++u) { const int c = g_mixed_census[u]; pairs += c == 2; singles += c == 1; }
        const int take_p = pairs < need_ice / 2 ? pairs : need_ice / 2;
        int rest = need_ice - 2 * take_p;
        const int take_s = singles < rest ? singles : rest;
        rest -= take_s;                                  // > 0: that many workgroups of the ice parts share a pair with the other code
        int p = 0, s1 = 0, ice_before = 0, nom_before = 0, ice_here = 0;
        for (int u = 0; u <= unit; ++u) {
          const int c = g_mixed_census[u];
          ice_here = 0;
          if (c == 2 && p < take_p) { ice_here = 2; ++p; }
          else if (c == 1 && s1 < take_s) { ice_here = 1; ++s1; }
          else if (rest > 0 && c > 0) { ice_here = c < rest ? c : rest; rest -= ice_here; }
          if (u < unit) { ice_before += ice_here; nom_before += c - ice_here; }
        }
        const bool ice = mine < ice_here;
        int ord = ice ? ice_before + mine : nom_before + (mine - ice_here);
        k = -1;
        for (int j = 0; j < m.n && k < 0; ++j) {
          if ((km->code[j] == SERL_DYN_ICE) != ice) continue;
          const int size = km->first_wg[j + 1] - km->first_wg[j];
          if (ord < size) { k = j; v = km->first_wg[j] + ord; } else ord -= size;
        }
      }
    } else if (lane == 0) {
      const int u = (int)(((unsigned long long)(((unsigned)unit + 1u) * 0x9e3779b9u) * (unsigned long long)total) >> 32);      // frac(unit * phi) * total
      int k0 = 0;
      for (int i = 1; i < m.n; ++i) k0 = u >= km->first_wg[i] ? i : k0;
      k = -1;
      for (int pass = 0; pass < 3 && k < 0; ++pass)
        for (int i = 0; i < m.n && k < 0; ++i) {
          const int j = (k0 + i) % m.n;
          if (pass == 0 ? j != k0 : pass == 1 ? km->code[j] != km->code[k0] : false) continue;
          const int t = atomicAdd(ticket + j, 1);
          if (t < km->first_wg[j + 1] - km->first_wg[j]) { k = j; v = km->first_wg[j] + t; }
        }
    }
  }
  if (lane == 0) { g_mixed_part = k; g_mixed_wg = v; }
}

__global__ void __launch_bounds__(64 * (citw_nominal_team_WAVES + SERL_ACTOR_WAVES)) serl_rollout_kernel_team4_mixed(SerlMixedArgs m)
{
  const SerlMixedArgs *km = (const SerlMixedArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  if (threadIdx.x < 64) serl_mixed_place(m, km);
  __syncthreads();
  const int k = __builtin_amdgcn_readfirstlane(g_mixed_part);
  // part k's arguments straight out of the kernel-argument segment (`m.a[k]` with a run-time k would copy the whole struct to scratch,
  // and every descriptor field read in the episode loop would come from there)
  const RolloutArgs &a = km->a[k];
  const bool actor = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= citw_nominal_team_WAVES;
  if (m.code[k] == SERL_DYN_ICE) {
    citw_team_stage_ice(a);
    if (actor) serl_teamg_actor_wave_ice(a);
    else serl_teamg_episodes_ice(a);
  } else {
    citw_team_stage_nominal(a);
    if (actor) serl_teamg_actor_wave_nominal(a);
    else serl_teamg_episodes_nominal(a);
  }
}

void serl_launch_rollout_team4_mixed(const SerlMixedArgs &m, int grid, hipStream_t stream)
{
  hipLaunchKernelGGL(serl_rollout_kernel_team4_mixed, dim3(grid), dim3(64 * (citw_nominal_team_WAVES + SERL_ACTOR_WAVES)), 0, stream, m);
}
