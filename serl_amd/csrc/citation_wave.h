// citation_wave.h -- runtime of the wave-cooperative Citation model evaluation (product code, gfx950).
//
// One wavefront owns one episode.  The generated evaluation function (gen/citation_<variant>_wave.inc, produced
// by tools/dag from the lifted model) computes the scalar "glue" of one model evaluation wave-uniformly and hands
// the table look-ups -- the bulk of the reference's work (61 rt_Lookup2D_Normal +
// 22 rt_Lookup + 144 rt_GetLookupIndex binary searches per evaluation) -- to the 64 lanes:
//
//   citw_search     one lane per distinct (breakpoint vector, input): rt_GetLookupIndex @0xf470 restated as a
//                   branch-free count (the vectors are strictly increasing, <= 22 entries)
//   citw_lookup2d   one lane per 2-D table: rt_Lookup2D_Normal @0xf590 (column-major z[ix + nx*iy],
//                   interpolate along x on both columns, then along y; operation order of the binary)
//   citw_lookup1d   one lane per 1-D table: rt_Lookup @0xf530   (y1-y0)/(x1-x0)*(u-x0)+y0
//   citw_table3     the `table3` S-function (mdlOutputs @0x10da0, Table2 @0x10a30), wave-uniform
//
// Inputs and results travel over per-wave LDS blackboards (g_in / g_sidx / g_out*); the model tables sit in LDS once per
// workgroup (g_ro, 94 KiB).  Within a wavefront LDS operations complete in order, so no barrier is needed
// between the phases.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "citation_libm.h"

#define CITW_MAX_ROUNDS 3

// Timing experiments (never in a shipped build): CITW_ABLATE_MASK -- team wavefronts that skip their part of the evaluation
// (rollout_team.inc); CITW_ABLATE_LOOK -- look-up passes that return at once (1: index searches, 2: 2-D, 4: 1-D, 8: the
// precomputed lanes); CITW_NO_BARRIER -- the evaluation's workgroup barriers left out (1: B1, 2: B2, 3: both).  Any of them freezes the state.
#ifndef CITW_ABLATE_MASK
#define CITW_ABLATE_MASK 0
#endif
#ifndef CITW_ABLATE_LOOK
#define CITW_ABLATE_LOOK 0
#endif
#ifndef CITW_NO_BARRIER
#define CITW_NO_BARRIER 0
#endif
#define CITW_ABLATE (CITW_ABLATE_MASK || CITW_ABLATE_LOOK || CITW_NO_BARRIER)
#if CITW_ABLATE_LOOK & 32       // ... 32: pow, 64: sincos replaced by one multiplication; 128: no ODE5 combination
#define pow(a, b) ((a) * (b))
#endif
#if CITW_ABLATE_LOOK & 64
#define sincos(a, s, c) (*(s) = (a) * 0.5, *(c) = 1.0 - (a))
#endif
// Hand-over stress builds (test-only library libserl_amd_jitter.so, serl_amd/build.py build_jitter; never in the product):
//   -DCITW_POISON=1   every LDS blackboard the wavefronts of a team exchange values through starts as a signalling NaN whose
//                     payload names the slot (array << 32 | index), the interval hints as 0x5a5a5a5a: a read that is not ordered
//                     behind its producer's write returns poison in EVERY launch, not only in the first one of a process
//   -DCITW_JITTER=1   a pseudo-random pause (hash of seed, site, wavefront, workgroup, sequence number; 0 .. 4 k cycles, now and
//                     then 32 k -- what a cold instruction cache does to one wavefront) in front of every flag raise, behind every
//                     flag wait and around every barrier; the seed comes from the context (SERL_JITTER_SEED, read by
//                     serl_ctx_create; 0 = no pauses).  A hand-over that is only right because its producer is usually early
//                     returns poison or a stale value under some seed; tests/test_gpu_rollout.py compares with the oracle.
#ifndef CITW_POISON
#define CITW_POISON 0
#endif
#ifndef CITW_JITTER
#define CITW_JITTER 0
#endif
#if CITW_JITTER
__shared__ unsigned g_jseed;          // RolloutArgs.jitter, staged by the kernels
__shared__ unsigned g_jsites;         // RolloutArgs.jitter_sites: which classes of sites pause (bit per class, below; diagnosis: tools/jitter_classes.py)
// class of a site: 0 .. 6 = flag raise / wait, iflag raise / wait, pflag raise / wait, value poll; 7 .. 11 = the actor hand-over (0xa00
// actor in front of its flag, 0xa10 team behind the action wait, 0xa20 in front of the observation flag, 0xa30 behind the ODE5
// combination, 0xa40 behind a step of a lane-group team); 12 = the actor wavefront's barriers; 13 / 14 = in front of / behind B1; 15 / 16 = B2
static __device__ __forceinline__ constexpr unsigned citw_jitter_class_(const unsigned site)
{
  return site < 0xa00u ? (site >> 8) - 1u : site < 0xac0u ? 7u + ((site >> 4) & 15u) : site < 0xb00u ? 12u : site == 0xb10u ? 13u : site == 0xb11u ? 14u : site == 0xb20u ? 15u : 16u;
}
static __device__ __forceinline__ void citw_jitter_(const unsigned site, const unsigned seq)
{
  unsigned h = g_jseed;
  if (h == 0u || ((g_jsites >> citw_jitter_class_(site)) & 1u) == 0u) return;
  h ^= seq * 2654435761u; h ^= site * 0x9e3779b1u; h ^= (unsigned)(threadIdx.x >> 6) * 0x85ebca6bu; h ^= (unsigned)blockIdx.x * 0xc2b2ae35u;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
  h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
  if ((h & 3u) != 0u) return;                       // one site in four pauses ...
  unsigned n = (h >> 8) & 31u;                      // ... 0 .. 31 x 128 cycles,
  if (((h >> 2) & 63u) == 0u) n *= 8u;              // one pause in 64 eight times as long
  for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(2);
}
#define CITW_JIT(site, seq) citw_jitter_((site), (seq))
#else
#define CITW_JIT(site, seq) ((void)0)
#endif
// LANES of one wavefront that hand values to each other through LDS -- one lane (or a few) stores, every lane loads -- are separate
// THREADS of the language's memory model.  The hardware completes the LDS operations of a wavefront in order, but nothing tells the
// COMPILER: round 3's early libm flag split "lanes 0 and 1 store their results" into two single-lane stores, whose addresses are
// constants -- LLVM's load-PRE then forwarded the stored value on the storing lane's path and sank the load of every other lane into
// the complementary path, which the structurizer places IN FRONT of the store: the other lanes read the slot before it was written
// (first launch of a process: NaN; later: the previous evaluation's value -- profiles/r04_experiments.md, found with the poison
// build).  A wavefront-scope fence is the ordering the code means: it emits no instruction, it makes the stores of all lanes
// "visible before" the loads that follow.  Every lane-subset store that the same wavefront reads back is followed by one.
#define CITW_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define CITW_TEAM_BARRIER() __syncthreads()
// B1 / B2 of the generated team code (FSEQ and stage are in scope there; jitter builds pause on both sides)
#define CITW_TEAM_BARRIER1() do { if (!(CITW_NO_BARRIER & 1)) { CITW_JIT(0xb10u, FSEQ * 8u + (unsigned)stage); __syncthreads(); CITW_JIT(0xb11u, FSEQ * 8u + (unsigned)stage); } } while (0)
#define CITW_TEAM_BARRIER2() do { if (!(CITW_NO_BARRIER & 2)) { CITW_JIT(0xb20u, FSEQ * 8u + (unsigned)stage); __syncthreads(); CITW_JIT(0xb21u, FSEQ * 8u + (unsigned)stage); } } while (0)

// Lanes that work for ONE episode.  64: the wavefront is the episode (rollout_wave.inc, rollout_team.inc).  32: two
// episodes per wavefront, lanes 0-31 / 32-63 (rollout_half.inc) -- the scalar "glue" of the model costs an instruction
// whether its 64 lanes hold one value or two, so packing two episodes into a wavefront nearly doubles the throughput in
// the regime where wavefronts, not CUs, are the scarce resource (257 .. 2 048 episodes per GPU).  `wv` (the LDS row of
// an episode) and every value derived from the state are then per-lane quantities; the lane-parallel phases run in
// ceil(count / 32) passes.
#ifndef CITW_GROUP_LANES
#define CITW_GROUP_LANES 64
#endif

// Stores of wave-uniform values (look-up inputs, exchanged values, derivatives, hand-over flags) in the generated team code:
// by lane 0 behind an exec-mask branch (0), or by every lane to the one address without a branch (1: the values, 2: the
// flags, 3: both).  ~100 branches per evaluation less: one episode per team 20.3 -> 19.8 us per env step (r03 sweeps 19 - 21),
// two per team 21.5 -> 21.0, four per team 24.8 -> 24.4 (lanes of a group store to the group's address).
#ifndef CITW_UNIFORM_STORE
#define CITW_UNIFORM_STORE 3
#endif
#define CITW_LANE0 ((CITW_UNIFORM_STORE & 1) != 0 || lane == 0)
#define CITW_LANE ((int)(threadIdx.x & (CITW_GROUP_LANES - 1)))
// Rows of the generated TEAM code (gen/citation_<variant>_team.inc): one episode per workgroup -> row 0 of every shared
// blackboard, wave q's libm results in g_m[q], exchanged values from g_x[0]; with two episodes per team (lane groups,
// rollout_team_half.inc) the row is the lane group.
#if CITW_GROUP_LANES == 64
#define CITW_TROW 0
#define CITW_MROW(q) (q)
#define CITW_XOFF 0
#define CITW_YOFF 0
#define CITW_GROUPS 1
#else
#define CITW_GROUPS (64 / CITW_GROUP_LANES)                      // 2 or 4 episodes per wavefront
#define CITW_TROW ((int)((threadIdx.x & 63) / CITW_GROUP_LANES))
#define CITW_MROW(q) ((q) * CITW_GROUPS + CITW_TROW)
#define CITW_XOFF (CITW_TROW * (256 / CITW_GROUPS))
#define CITW_YOFF (CITW_TROW * 32)
#endif
#ifndef CITW_SEARCH_BATCH
#define CITW_SEARCH_BATCH 0     // 1 (team kernels): index-search compares in batches of eight; costs 44 VGPRs, which the one-wave kernels lack
#endif
#define CITW_RO_LDS_WORDS 12040

struct CitwSearch { uint16_t row, n, in, pad; };                                // 8 B: row of g_bp, entries, input slot
struct CitwBpVec { uint16_t xw, n; };                                           // a distinct breakpoint vector: word offset in g_ro, entries
struct CitwLookup { uint16_t xrw, nr, xcw, zw, sx, sy, in0, in1, out, p0, p1, p2; };   // 24 B; 1-D: x = xrw, y = zw

// Per-wavefront LDS scratch, one row per wavefront of the workgroup.
#ifndef CITW_MAX_WAVES
#define CITW_MAX_WAVES 4          // wavefronts (episodes) per workgroup; 8 = two per SIMD with 256 registers each
#endif
#ifndef CITW_M_ROWS
#define CITW_M_ROWS CITW_MAX_WAVES
#endif
#ifndef CITW_OUT2_ROWS
#define CITW_OUT2_ROWS CITW_MAX_WAVES
#endif
#ifndef CITW_INV_SLOTS
#define CITW_INV_SLOTS 128
#endif
#ifndef CITW_LDS_EXTRA_FLOATS
#define CITW_LDS_EXTRA_FLOATS 4         // team kernels: the actor hand-over words and the LDS-resident actor weights
#endif
#define CITW_MAX_CONSTS 192
#define CITW_MAX_BPVEC 48
#define CITW_BP_PAD 24

#ifdef CITW_LDS_STRUCT
// ONE LDS object with an explicit member order.  A DS instruction reaches the first 64 KB of LDS with its 16-bit immediate
// offset; anything above needs its address in a VGPR (a v_mov per base in the ISA).  Left to itself the compiler put the
// 96 KB table block first and the per-episode blackboards -- the hot, constant-address accesses of the generated code --
// behind it (377 of 466 DS instructions of the team kernel without an immediate offset).  Hence: blackboards, exchange
// rows, flags and descriptors first (~45 KB), the table block last.
struct CitwLds {
  double xs[CITW_MAX_WAVES][20];          // continuous states X[19] of the current stage (lane i writes state i)
  double cmd[CITW_MAX_WAVES][12];         // command vector of the current env step
  double in[CITW_MAX_WAVES][32];          // look-up inputs of the current round
  int sidx[CITW_MAX_WAVES][64];           // interval indices of the current round
  double m[CITW_M_ROWS][64];              // results of the lane-parallel libm calls: [2j] / [2j+1] of call j (team kernels: one row per wavefront)
  double out0[CITW_MAX_WAVES][128];       // look-up results of round 1: [0..63] 2-D pass, [64..127] 1-D pass
  double out1[CITW_MAX_WAVES][128];       // ... round 2
  double out2[CITW_OUT2_ROWS][128];       // ... round 3 / the per-step invariant round (no current variant has one)
  double dw[CITW_MAX_WAVES][32];          // Derivative-block banks (rtDW): TimeStampA, LastUAtTimeA[12], TimeStampB, LastUAtTimeB[12]
  double f[CITW_MAX_WAVES][6][20];        // ODE5 stage derivatives
  double act[CITW_MAX_WAVES][16][3];      // action trace of the last <= 16 env steps (flushed as one coalesced store)
  double inv[CITW_MAX_WAVES][CITW_INV_SLOTS];   // per-step invariants of the model (citw_<v>_step_invariants)
  double x[256];                          // team kernels: values that cross between the wavefronts at barrier B1
  double t3[48];
  unsigned pflag[16]; double y[32 * CITW_GROUPS]; unsigned smiss;
  unsigned flag[16], iflag[16];           // hand-over flags of the team kernels (citw_flag_*, citw_iflag_*)
  alignas(16) float extra[CITW_LDS_EXTRA_FLOATS];     // unit-specific words (team kernels: actor hand-over + LDS-resident actor weights)
  double k[CITW_MAX_CONSTS];              // f64 literals of the model (only when generated with --lds-consts)
  CitwSearch S[CITW_MAX_ROUNDS][64];
  CitwLookup L[CITW_MAX_ROUNDS][2][64];
  double bp[CITW_MAX_BPVEC][CITW_BP_PAD]; // the distinct breakpoint vectors, padded with +inf (index search without bounds tests)
  double ro[CITW_RO_LDS_WORDS];           // the model's tables (.rodata words the evaluation reads): 94 KB, last
};
__shared__ CitwLds citw_lds;
#define g_xs citw_lds.xs
#define g_cmd citw_lds.cmd
#define g_in citw_lds.in
#define g_sidx citw_lds.sidx
#define g_m citw_lds.m
#define g_out0 citw_lds.out0
#define g_out1 citw_lds.out1
#define g_out2 citw_lds.out2
#define g_dw citw_lds.dw
#define g_f citw_lds.f
#define g_act citw_lds.act
#define g_inv citw_lds.inv
#define g_x citw_lds.x
#define g_t3 citw_lds.t3
#define g_flag citw_lds.flag
#define g_iflag citw_lds.iflag
#define g_pflag citw_lds.pflag
#define g_smiss citw_lds.smiss
#define g_y citw_lds.y
#define g_k citw_lds.k
#define g_S citw_lds.S
#define g_L citw_lds.L
#define g_bp citw_lds.bp
#define g_ro citw_lds.ro

#else
// Separate objects: distinct objects let the compiler tell the blackboards apart (results of look-up round 1 stay loadable
// across the stores of round 2, a store to g_f[..][stage] does not order the loads around it).  The compiler lays the LDS
// objects of a kernel out by DESCENDING ALIGNMENT (then size): the blackboards are over-aligned to 64 bytes so that they come
// first and the 94 KB table block (and the breakpoint / descriptor tables, 8-byte aligned) last -- a DS instruction reaches
// the first 64 KB with its 16-bit immediate offset, anything above needs its address in a VGPR (left to the default order
// 377 of the team kernel's 466 DS instructions had none).
__shared__ alignas(64) double g_in[CITW_MAX_WAVES][32];       // look-up inputs of the current round
__shared__ alignas(64) int g_sidx[CITW_MAX_WAVES][64];        // interval indices of the current round
__shared__ alignas(64) double g_m[CITW_M_ROWS][64];           // results of the lane-parallel libm calls: [2j] / [2j+1] of call j (team kernels: one row per wavefront)
__shared__ alignas(64) double g_out0[CITW_MAX_WAVES][128];    // look-up results of round 1: [0..63] 2-D pass, [64..127] 1-D pass
__shared__ alignas(64) double g_out1[CITW_MAX_WAVES][128];    // ... round 2
__shared__ alignas(64) double g_out2[CITW_OUT2_ROWS][128];    // ... round 3 / the per-step invariant round (no current variant has one)
__shared__ alignas(64) double g_dw[CITW_MAX_WAVES][32];       // Derivative-block banks (rtDW): TimeStampA, LastUAtTimeA[12], TimeStampB, LastUAtTimeB[12]
__shared__ alignas(64) double g_f[CITW_MAX_WAVES][6][20];     // ODE5 stage derivatives
__shared__ alignas(64) double g_xs[CITW_MAX_WAVES][20];       // continuous states X[19] of the current stage (lane i writes state i)
__shared__ alignas(64) double g_cmd[CITW_MAX_WAVES][12];      // command vector of the current env step
__shared__ alignas(64) double g_act[CITW_MAX_WAVES][16][3];   // action trace of the last <= 16 env steps (flushed as one coalesced store)
__shared__ alignas(64) double g_inv[CITW_MAX_WAVES][CITW_INV_SLOTS];     // per-step invariants of the model (citw_<v>_step_invariants)
__shared__ alignas(64) double g_x[256];                       // team kernels: values that cross between the wavefronts at barrier B1

__shared__ alignas(64) double g_k[CITW_MAX_CONSTS];           // f64 literals of the model (only when generated with --lds-consts)
__shared__ double g_ro[CITW_RO_LDS_WORDS];
__shared__ alignas(64) double g_t3[48];
__shared__ double g_bp[CITW_MAX_BPVEC][CITW_BP_PAD];   // the distinct breakpoint vectors, padded with +inf (index search without bounds tests)
__shared__ CitwSearch g_S[CITW_MAX_ROUNDS][64];
__shared__ CitwLookup g_L[CITW_MAX_ROUNDS][2][64];

__shared__ alignas(64) unsigned g_flag[16];      // hand-over flags of the team kernels, one per producing wavefront (citw_flag_*)
__shared__ alignas(64) unsigned g_iflag[16];     // ... and for look-up inputs computed by helper wavefronts (citw_iflag_*)
__shared__ alignas(64) unsigned g_smiss;         // sequence number of the last evaluation whose first search round had to repair an interval (citw_round_spec)
__shared__ alignas(64) unsigned g_pflag[16];     // ... and for the values of the task graph behind the look-ups (citw_pflag_*)
__shared__ alignas(64) double g_y[32 * CITW_GROUPS];   // team kernels: values that cross between the wavefronts BEHIND barrier B1 (task graph)
#endif

// ---- f64 literals of the generated team code in registers (round 5; tools/dag/codegen_team.py assign_kregs).  An f64 literal with a
// non-zero low dword costs two 32-bit moves at every use (the build keeps the machine LICM from hoisting them: as SGPR pairs they
// spilled) -- 15 % of the headline kernel's static instructions.  A wavefront's role is fixed for the episode, so it loads the
// literals of ITS role once, from the role's row of an LDS table (g_klit, staged by the kernel), into a register set that stays live
// across the episode loop: an LDS load at a wave-uniform address is wave-uniform for the compiler (scalar branches on what is
// computed from it stay scalar), its result is a VGPR pair that cannot be rematerialised, and a VGPR operand needs no
// constant-bus slot.  The generated text reads CITW_K(slot, literal): the register when the caller passed a set, the literal otherwise
// (HAVE_K is a compile-time fact after inlining; a flag of its own because a comparison of the set's address with null is not one: null is a valid private address on this target) -- the lane-group kernels, at 254 / 256 VGPRs, pass none and compile what they always did.
// (CitwKRegs, citw_no_kregs: serl_kregs.h)
#define CITW_K(j, lit) (HAVE_K ? KR.k[j] : (lit))

// Phase profile of the model evaluation (profiling builds only, -DCITW_PROFILE): shader-clock cycles of wave 0 of
// workgroup 0 between the CITW_T marks of the generated code, accumulated in LDS and copied out by the kernel.
#ifdef CITW_PROFILE
// (the marks of role r fire on the hardware wavefront that RUNS role r -- rollout_team_<v>.hip maps roles to wavefronts; units without a map: role = wavefront)
#ifdef CITW_PROF_TEAM_ROLES      // (one-episode team units: rollout_team.inc defines the map; wavefront 7 is the actor)
static __device__ __forceinline__ int serl_team_role(const bool stream);
#define CITW_PROF_ROLE() ((threadIdx.x >> 6) < 7 ? serl_team_role(false) : 8)
#else
#define CITW_PROF_ROLE() ((int)(threadIdx.x >> 6))
#endif
#define CITW_PROF_IS(r) (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && CITW_PROF_ROLE() == (r))
__shared__ unsigned long long g_prof[32];
__shared__ unsigned long long g_tlast;
#define CITW_T0() do { if (CITW_PROF_IS(0)) g_tlast = __builtin_readcyclecounter(); } while (0)
#define CITW_T(k) do { if (CITW_PROF_IS(0)) { const unsigned long long t_ = __builtin_readcyclecounter(); \
                         g_prof[(k)] += t_ - g_tlast; g_tlast = t_; } } while (0)
__shared__ unsigned long long g_tlast1;       // same for wave 1 of the team kernels
__shared__ unsigned long long g_tlastw[16];    // ... and for the other waves (barrier arrival / departure only)
#if CITW_PROFILE == 3      // LIGHT profile: only how long every team wavefront waits at the two barriers (slots 0 .. 6: B1, 8 .. 14: B2) -- two
                           // clock reads and one LDS update per barrier and wavefront, ~1 % (the full phase profile costs 15 % and distorts)
static __device__ __forceinline__ constexpr int citw_mark_k(int s) { return s == 10 || s == 15 || s == 19 ? 0 : s == 11 || s == 16 || s == 21 ? 1 : s == 12 || s == 17 || s == 24 ? 2 : s == 13 || s == 18 || s == 25 ? 3 : -1; }
#define CITW_MARK_(w, k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
                           if ((k) == 0 || (k) == 2) g_tlastw[(w)] = t_; else if ((k) == 1) g_prof[(w)] += t_ - g_tlastw[(w)]; else if ((k) == 3) g_prof[8 + (w)] += t_ - g_tlastw[(w)]; } } while (0)
#undef CITW_T0
#undef CITW_T
#define CITW_T0() ((void)0)
#define CITW_T(k) do { if ((k) <= 3) CITW_MARK_(0, (k)); } while (0)
#define CITW_U0() ((void)0)
#define CITW_U(k) do { if ((k) >= 10 && (k) <= 13) CITW_MARK_(1, (k) - 10); } while (0)
#define CITW_W0(w) ((void)0)
#define CITW_W(w, s) do { if (citw_mark_k(s) >= 0) CITW_MARK_((w), citw_mark_k(s)); } while (0)
#define CITW_V0(w) ((void)0)
#define CITW_V(w, s) do { if (citw_mark_k(s) >= 0) CITW_MARK_((w), citw_mark_k(s)); } while (0)
#elif CITW_PROFILE == 2      // second profiling build: waves 4, 5, 6 report into the slots waves 1, 2, 3 use in the first
#define CITW_U0() ((void)0)
#define CITW_U(k) ((void)0)
#define CITW_W0(w) ((void)0)
#define CITW_W(w, k) ((void)0)
#define CITW_V0(w) do { if (CITW_PROF_IS(w)) g_tlastw[(w)] = __builtin_readcyclecounter(); } while (0)
#define CITW_V(w, k) do { if (CITW_PROF_IS(w)) { const unsigned long long t_ = __builtin_readcyclecounter(); \
                         g_prof[(k)] += t_ - g_tlastw[(w)]; g_tlastw[(w)] = t_; } } while (0)
#else
#define CITW_V0(w) ((void)0)
#define CITW_V(w, k) ((void)0)
#define CITW_U0() do { if (CITW_PROF_IS(1)) g_tlast1 = __builtin_readcyclecounter(); } while (0)
#define CITW_U(k) do { if (CITW_PROF_IS(1)) { const unsigned long long t_ = __builtin_readcyclecounter(); \
                         g_prof[(k)] += t_ - g_tlast1; g_tlast1 = t_; } } while (0)
#define CITW_W0(w) do { if (CITW_PROF_IS(w)) g_tlastw[(w)] = __builtin_readcyclecounter(); } while (0)
#define CITW_W(w, k) do { if (CITW_PROF_IS(w)) { const unsigned long long t_ = __builtin_readcyclecounter(); \
                         g_prof[(k)] += t_ - g_tlastw[(w)]; g_tlastw[(w)] = t_; } } while (0)
#endif
#else
#define CITW_W0(w) ((void)0)
#define CITW_W(w, k) ((void)0)
#define CITW_V0(w) ((void)0)
#define CITW_V(w, k) ((void)0)
#define CITW_T0() ((void)0)
#define CITW_T(k) ((void)0)
#define CITW_U0() ((void)0)
#define CITW_U(k) ((void)0)
#endif

#ifndef CITW_POLL_SLEEP
#define CITW_POLL_SLEEP 0        // s_sleep argument between two polls of a hand-over flag (0: poll back to back)
#endif
#if CITW_POLL_SLEEP > 0
#define CITW_POLL_PAUSE() __builtin_amdgcn_s_sleep(CITW_POLL_SLEEP)
#else
#define CITW_POLL_PAUSE() ((void)0)
#endif
// Hand-over of a look-up input from a helper wavefront to wave 0 without a barrier (team kernels): the helper stores the
// value(s), then the sequence number of the evaluation (release); wave 0 polls the number (acquire) before it reads.
// All wavefronts of a workgroup are resident, so the poll cannot starve the writer; numbers only grow within an episode.
static __device__ __forceinline__ void citw_flag_raise(int q, unsigned seq)
{
  seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq);     // (two episodes per team: the first one's model clock counts for both)
  CITW_JIT(0x100u + q, seq);
  if ((CITW_UNIFORM_STORE & 2) || (threadIdx.x & 63) == 0) __hip_atomic_store(&g_flag[q], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void citw_flag_wait(int q, unsigned seq)
{
  seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq);     // (two episodes per team: the first one's model clock counts for both)
  // "reached", not "equal": a producer can never be an evaluation ahead (barrier B2 separates evaluations), but a poll that
  // tolerates it cannot hang either
  if (CITW_ABLATE_LOOK & 256) return;      // (timing experiment: values "ready at once")
  while ((int)(__hip_atomic_load(&g_flag[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - seq) < 0) CITW_POLL_PAUSE();
  CITW_JIT(0x200u + q, seq);
}

// ... and a second set for the look-up inputs a helper wavefront computes for wave 0 (spread-input partitions)
static __device__ __forceinline__ void citw_iflag_raise(int q, unsigned seq)
{
  seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq);     // (two episodes per team: the first one's model clock counts for both)
  CITW_JIT(0x300u + q, seq);
  if ((CITW_UNIFORM_STORE & 2) || (threadIdx.x & 63) == 0) __hip_atomic_store(&g_iflag[q], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void citw_iflag_wait(int q, unsigned seq)
{
  seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq);     // (two episodes per team: the first one's model clock counts for both)
  while ((int)(__hip_atomic_load(&g_iflag[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - seq) < 0) CITW_POLL_PAUSE();
  CITW_JIT(0x400u + q, seq);
}

// ... and a third for the task graph behind the look-ups (gen/citation_<v>_team.inc): wave q announces its k-th published value
// of the evaluation with sequence number SEQ by g_pflag[q] = SEQ * 16 + k; the value itself is in g_y
static __device__ __forceinline__ void citw_pflag_raise(int q, unsigned seq)
{
  seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq);
  CITW_JIT(0x500u + q, seq);
  if ((CITW_UNIFORM_STORE & 2) || (threadIdx.x & 63) == 0) __hip_atomic_store(&g_pflag[q], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void citw_pflag_wait(int q, unsigned seq)
{
  seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq);
  if (CITW_ABLATE_LOOK & 512) return;
  while ((int)(__hip_atomic_load(&g_pflag[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - seq) < 0) CITW_POLL_PAUSE();
  CITW_JIT(0x600u + q, seq);
}

// lane k's copy of a per-lane double, for every lane (k wave-uniform): two v_readlane_b32
#ifndef CITW_STATE_BCAST
#define CITW_STATE_BCAST 0      // 1: single-episode team kernels read the states of an evaluation from the lanes that combined them (gen/..._team.inc).
                                // Measured slower (r03 sweep 29: 18.38 against 18.12 us per env step; the states then live in SGPR pairs: 67 scalar spills for 39)
#endif
// issue priority of a wavefront while it makes libm calls the other wavefronts wait for (0: leave it alone)
#ifndef CITW_LIBM_PRIO_LEVEL
#define CITW_LIBM_PRIO_LEVEL 1      // (r03 sweep 32: 18.04 -> 17.93 us per env step; level 3 the same)
#endif
#define CITW_LIBM_PRIO(up) do { if (CITW_LIBM_PRIO_LEVEL) __builtin_amdgcn_s_setprio((up) ? CITW_LIBM_PRIO_LEVEL : 0); } while (0)
#ifndef CITW_LIBM_DIRECT
#define CITW_LIBM_DIRECT 0      // 1: ... and libm calls whose arguments are states are made by the lanes that hold those states (no argument slots).
                                // Measured slower as well (sweep 30: 18.38 against 18.15)
#endif
static __device__ __forceinline__ double citw_bcast(const double v, const int k)
{
#if CITW_STATE_BCAST == 2      // through the LDS crossbar (ds_bpermute): no scalar registers, no store in front of the load
  const int lo = __builtin_amdgcn_ds_bpermute(k * 4, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(k * 4, __double2hiint(v));
#else
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
#endif
  return __hiloint2double(hi, lo);
}

// A flag wait that returns the first value behind the flag.  The value is loaded BEHIND an acquire load of the raised flag.
// (Round 3 tried to issue the value's load in the same poll iteration as the flag's, both relaxed, for one LDS round trip on a hit:
// nothing orders two relaxed loads of different addresses, the compiler may issue the value's first -- with a flag that is raised
// right behind its data (the early libm flag) a consumer then read the slot BEFORE the data: NaNs in the first launch of a
// process, found by tools/repeat_check.py.  It had measured +-0.05 us anyway.)
static __device__ __forceinline__ double citw_poll_load_(const unsigned *flag, unsigned seq, const double *p)
{
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - seq) < 0) CITW_POLL_PAUSE();
  CITW_JIT(0x700u, seq);
  return *p;
}
static __device__ __forceinline__ double citw_pflag_wait_load(int q, unsigned seq, const double *p)
{
  seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq);
  if (CITW_ABLATE_LOOK & 512) return *p;
  return citw_poll_load_(&g_pflag[q], seq, p);
}
static __device__ __forceinline__ double citw_flag_wait_load(int q, unsigned seq, const double *p)
{
  seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq);
  if (CITW_ABLATE_LOOK & 256) return *p;
  return citw_poll_load_(&g_flag[q], seq, p);
}

static __device__ __forceinline__ unsigned long long citw_d2u(double d) { return (unsigned long long)__double_as_longlong(d); }
static __device__ __forceinline__ double citw_u2d(unsigned long long u) { return __longlong_as_double((long long)u); }

// interval index (rt_GetLookupIndex semantics):
//   u <= x[0] -> 0 ; u >= x[n-1] -> n-2 ; u < 0: x[i] <= u < x[i+1] ; u >= 0: x[i] < u <= x[i+1]
// The vectors are strictly increasing, so  lt = #{x_i < u}  is a position and  le = #{x_i <= u} = lt + (x[lt] == u);
// rows of g_bp are padded with +inf, which no comparison below counts for finite u (and the clamp absorbs u = +inf).
//
// HINTED search.  The states move a little per RK stage and env step, so an input almost never leaves its interval between
// two evaluations.  Every search has a slot of its own in g_sidx (SBASE + its number within the round; the rounds of an
// evaluation do not overlap), which therefore still holds the index the previous evaluation found.  With c = (u < 0 ? le : lt)
// the result is idx = clamp(c - 1, 0, n - 2), and idx == h holds exactly when
//     (h == 0     or  x[h]   (<, <= for u < 0)  u)       [c - 1 >= h, or the lower clamp]
// and (h == n - 2 or  not x[h+1] (<, <=) u)                [c - 1 <= h, or the upper clamp]
// -- two LDS words and two compares instead of the whole row and 2 x MAXN operations.  The index is unique, so a verified hint
// IS the result of the full count; if any lane of the wavefront fails the test (ballot) the wavefront runs the full count,
// which also repairs the slot.  Slots start at 0 (staged by the kernels), any value is re-verified, NaN inputs fail every
// compare and fall back to the count (idx 0, as before).
// (round 5) every write that CHANGES an interval index counts g_smiss up: the lane-group kernels' wavefronts keep the interval-dependent half of
// their look-up lanes in registers across evaluations (CitwPassCache below) and recompute it when the count has moved
static __device__ __forceinline__ void citw_sidx_changed() { atomicAdd(&g_smiss, 1u); }
template <int MAXN, int COUNT, int SBASE>
static __device__ __forceinline__ void citw_search_pass(const int wv, const CitwSearch *S, int ln);
// COUNT searches of one round, one per lane of the episode's lane group, in ceil(COUNT / CITW_GROUP_LANES) passes
template <int MAXN, int COUNT = 64, int SBASE = 0>
static __device__ __forceinline__ void citw_search(const int wv, const CitwSearch *S, int lane)
{
#pragma unroll
  for (int base = 0; base < COUNT; base += CITW_GROUP_LANES) citw_search_pass<MAXN, COUNT, SBASE>(wv, S, lane + base);
  CITW_WAVE_FENCE();      // the look-up lanes that follow read the indices other lanes stored
}

// The passes PART, PART + NPARTS, ... of a search round (several episodes per team: helper wavefronts take passes beside wave 0)
#ifndef CITW_L2_SHARE
#define CITW_L2_SHARE (64 / CITW_GROUP_LANES)
#endif
#define CITW_SEARCH_SHARE(count) (((count) + CITW_GROUP_LANES - 1) / CITW_GROUP_LANES < 3 ? ((count) + CITW_GROUP_LANES - 1) / CITW_GROUP_LANES : 3)
template <int MAXN, int COUNT, int PART, int NPARTS, int SBASE = 0>
static __device__ __forceinline__ void citw_search_part(const int wv, const CitwSearch *S, int lane)
{
#pragma unroll
  for (int base = PART * CITW_GROUP_LANES; base < COUNT; base += NPARTS * CITW_GROUP_LANES) citw_search_pass<MAXN, COUNT, SBASE>(wv, S, lane + base);
  CITW_WAVE_FENCE();
}

#ifndef CITW_SPEC_LOOKUP
#define CITW_SPEC_LOOKUP 1       // 1: single-episode team kernels precompute the hint-dependent half of wave 0's look-up lanes (citw_spec_pre / citw_spec_tail below)
#endif
#ifndef CITW_FUSED_LATER
#define CITW_FUSED_LATER 0       // 1: ... in the look-up rounds behind the first one only (two 2-D + two 1-D tables on the chain every wavefront waits for)
#endif
#ifndef CITW_SEARCH_HINT
#define CITW_SEARCH_HINT 1
#endif
#ifndef CITW_FUSED_LOOKUP
#define CITW_FUSED_LOOKUP 0      // 1: look-up lanes verify their own hints (citw_lookup*_fused), no search pass.  Measured SLOWER (r03 sweep 6: 21.4 vs
                                 // 20.3 us per env step; intervals change in 0.2 % of the evaluations, so it is not the fall-back): off
#endif

// the full count (the fallback of the hinted search, and the whole search with CITW_SEARCH_HINT = 0)
template <int MAXN>
static __device__ __forceinline__ int citw_search_count(const double *x, const int n, const double u)
{
  typedef double v2d __attribute__((ext_vector_type(2)));
#if CITW_SEARCH_BATCH
  int lt = 0;
  // compares in batches of eight, then their additions: a compare result (an SGPR pair) may not be consumed by the very
  // next VALU instruction on gfx950 -- one by one every compare drags an s_nop behind it
  constexpr int NP = (MAXN + 1) & ~1;
  double xv[NP];
#pragma unroll
  for (int i = 0; i < NP; i += 2) {
    const v2d v = *(const v2d *)(x + i);
    xv[i] = v.x; xv[i + 1] = v.y;
  }
#pragma unroll
  for (int i0 = 0; i0 < NP; i0 += 8) {
    bool c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] = (i0 + q < NP) && (xv[i0 + q < NP ? i0 + q : 0] < u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) lt += c[q] ? 1 : 0;
    __builtin_amdgcn_sched_barrier(0);
  }
#else
  int lt = 0;
#pragma unroll
  for (int i = 0; i < ((MAXN + 1) & ~1); i += 2) {
    const v2d v = *(const v2d *)(x + i);
    lt += (v.x < u) ? 1 : 0;
    lt += (v.y < u) ? 1 : 0;
  }
#endif
  const int le = lt + ((x[lt < CITW_BP_PAD - 1 ? lt : CITW_BP_PAD - 1] == u) ? 1 : 0);
  int idx = ((u < 0.0) ? le : lt) - 1;
  idx = idx < 0 ? 0 : idx;
  idx = idx > n - 2 ? n - 2 : idx;
  return idx;
}

template <int MAXN, int COUNT, int SBASE>
static __device__ __forceinline__ void citw_search_pass(const int wv, const CitwSearch *S, int ln)
{
  if (CITW_ABLATE_LOOK & 1) return;
  const bool valid = ln < COUNT;                      // (lanes beyond the round's searches hold a filler descriptor: they never store)
  const int lc = valid ? ln : 0;
  const CitwSearch d = S[lc];
  int *slot = &g_sidx[wv][SBASE + lc];
#if CITW_SEARCH_HINT
  const int stored = *slot;
#endif
  const double u = g_in[wv][d.in];
  const double *x = g_bp[d.row];
  const int n = d.n;
#if CITW_SEARCH_HINT
  int h = stored < 0 ? 0 : stored;
  h = h > n - 2 ? n - 2 : h;
  const double xl = x[h], xh = x[h + 1];
  const bool neg = u < 0.0;
  const bool below = neg ? (xl <= u) : (xl < u);      // x[h] counts
  const bool above = neg ? (xh <= u) : (xh < u);      // x[h + 1] counts too: the interval lies further up
  const bool ok = ((h == 0) || below) && ((h == n - 2) || !above);
  if (__builtin_expect(__ballot(valid && !ok) != 0ULL, 0)) {
    const int idx = citw_search_count<MAXN>(x, n, u);
    if (valid) *slot = idx;
    if (valid && idx != stored) citw_sidx_changed();
  } else if (valid && h != stored) {
    *slot = h;
    citw_sidx_changed();
  }
#else
  // (no hint: the full count every time.  An interval that moves must still be counted, or the table lanes of CitwPassCache keep stale
  // x0 / quotients / corners -- citw_lookup2d_part_c, citw_lookup1d_part_c refill only when g_smiss has moved)
  const int idx = citw_search_count<MAXN>(x, n, u);
  const int old = *slot;
  if (valid) *slot = idx;
  if (valid && idx != old) citw_sidx_changed();
#endif
}

// Searches FIRST .. FIRST + N - 1 of a round on the first N lanes of the lane group: the tables keyed on a look-up input that a
// helper wavefront computes (gen/citation_<v>_team.inc: the air-data chain) are searched and interpolated by that helper, in
// slots no other wavefront touches
template <int MAXN, int FIRST, int N, int SBASE>
static __device__ __forceinline__ void citw_search_range(const int wv, const CitwSearch *S, int lane)
{
  static_assert(N <= CITW_GROUP_LANES, "one pass");
  citw_search_pass<MAXN, FIRST + N, SBASE>(wv, S, lane < N ? lane + FIRST : FIRST + N);
  CITW_WAVE_FENCE();
}

template <typename OUT>
static __device__ __forceinline__ void citw_lookup2d_pass(const int wv, const CitwLookup *L, OUT &out, int lane);
template <int COUNT = 64, typename OUT>
static __device__ __forceinline__ void citw_lookup2d(const int wv, const CitwLookup *L, OUT &out, int lane)
{
#pragma unroll
  for (int base = 0; base < COUNT; base += CITW_GROUP_LANES) citw_lookup2d_pass(wv, L, out, lane + base);
  CITW_WAVE_FENCE();      // every lane reads the results back as wave-uniform loads
}

// The passes PART, PART + NPARTS, ... of a 2-D round: with fewer lanes per episode than tables (two / four episodes per team)
// the passes are independent and the team's idle helpers take some of them beside wave 0 (gen/citation_<v>_team.inc)
#ifndef CITW_L2_SHARE
#define CITW_L2_SHARE (64 / CITW_GROUP_LANES)
#endif
template <int COUNT, int PART, int NPARTS, typename OUT>
static __device__ __forceinline__ void citw_lookup2d_part(const int wv, const CitwLookup *L, OUT &out, int lane)
{
#pragma unroll
  for (int base = PART * CITW_GROUP_LANES; base < COUNT; base += NPARTS * CITW_GROUP_LANES) citw_lookup2d_pass(wv, L, out, lane + base);
  CITW_WAVE_FENCE();
}

template <typename OUT>
static __device__ __forceinline__ void citw_lookup2d_pass(const int wv, const CitwLookup *L, OUT &out, int lane)
{
  if (CITW_ABLATE_LOOK & 2) return;
  const CitwLookup d = L[lane];
  const int ix = g_sidx[wv][d.sx], iy = g_sidx[wv][d.sy];
  const double u0 = g_in[wv][d.in0], u1 = g_in[wv][d.in1];
  const double *xr = g_ro + d.xrw, *xc = g_ro + d.xcw, *z = g_ro + d.zw;
  const int nr = d.nr;
  const double x0 = xr[ix], x1 = xr[ix + 1];
  const double dx = x1 - x0, wx = u0 - x0;
  const double z00 = z[ix + nr * iy], z10 = z[ix + 1 + nr * iy];
  const double z01 = z[ix + nr * (iy + 1)], z11 = z[ix + 1 + nr * (iy + 1)];
  double a = z10 - z00; a = a / dx; a = a * wx; a = a + z00;
  double b = z11 - z01; b = b / dx; b = b * wx; b = b + z01;
  const double y0 = xc[iy];
  const double dy = xc[iy + 1] - y0;
  double r = b - a; r = r / dy; r = r * (u1 - y0);
  out[wv][d.out] = r + a;
}

// ---- FUSED hinted look-ups (one episode per wavefront / team: the lane group is the whole wavefront).  A look-up lane
// re-verifies the hints of its own one or two index searches -- the words x[h], x[h+1] it needs for that ARE the interval
// ends the interpolation uses -- and interpolates at once: no search pass, no round trip through g_sidx.  If any lane of the
// wavefront fails the test the wavefront falls back to the search pass (which repairs the slots) and the plain pass; the
// interval is unique, so both paths compute the same expression on the same operands.
static __device__ __forceinline__ bool citw_hint_ok(const int h, const int n, const double xl, const double xh, const double u)
{
  const bool neg = u < 0.0;
  const bool below = neg ? (xl <= u) : (xl < u);      // x[h] counts
  const bool above = neg ? (xh <= u) : (xh < u);      // x[h + 1] counts too: the interval lies further up
  return ((h == 0) || below) && ((h == n - 2) || !above);
}

template <int COUNT, int SMAXN, int SCOUNT, int SBASE, typename OUT>
static __device__ __forceinline__ void citw_lookup2d_fused(const int wv, const CitwSearch *S, const CitwLookup *L, OUT &out, int lane)
{
  static_assert(COUNT <= 64, "one pass");
  const bool valid = lane < COUNT;
  const CitwLookup d = L[valid ? lane : 0];
  int hx = g_sidx[wv][d.sx], hy = g_sidx[wv][d.sy];
  const double u0 = g_in[wv][d.in0], u1 = g_in[wv][d.in1];
  const int nr = d.nr, nc = d.p0;
  hx = hx < 0 ? 0 : hx; hx = hx > nr - 2 ? nr - 2 : hx;
  hy = hy < 0 ? 0 : hy; hy = hy > nc - 2 ? nc - 2 : hy;
  const double *xr = g_ro + d.xrw, *xc = g_ro + d.xcw, *z = g_ro + d.zw;
  const double x0 = xr[hx], x1 = xr[hx + 1];
  const double y0 = xc[hy], y1 = xc[hy + 1];
  const double z00 = z[hx + nr * hy], z10 = z[hx + 1 + nr * hy];
  const double z01 = z[hx + nr * (hy + 1)], z11 = z[hx + 1 + nr * (hy + 1)];
  const bool ok = citw_hint_ok(hx, nr, x0, x1, u0) && citw_hint_ok(hy, nc, y0, y1, u1);
  const double dx = x1 - x0, wx = u0 - x0;
  double a = z10 - z00; a = a / dx; a = a * wx; a = a + z00;
  double b = z11 - z01; b = b / dx; b = b * wx; b = b + z01;
  const double dy = y1 - y0;
  double r = b - a; r = r / dy; r = r * (u1 - y0);
  if (__builtin_expect(__ballot(valid && !ok) == 0ULL, 1)) {
    if (valid) out[wv][d.out] = r + a;
  } else {
    citw_search<SMAXN, SCOUNT, SBASE>(wv, S, lane);
    citw_lookup2d<COUNT>(wv, L, out, lane);
  }
}

template <typename OUT>
static __device__ __forceinline__ void citw_lookup1d_pass(const int wv, const CitwLookup *L, OUT &out, int lane);
template <int COUNT = 64, typename OUT>
static __device__ __forceinline__ void citw_lookup1d(const int wv, const CitwLookup *L, OUT &out, int lane);

template <int COUNT, int SMAXN, int SCOUNT, int SBASE, typename OUT>
static __device__ __forceinline__ void citw_lookup1d_fused(const int wv, const CitwSearch *S, const CitwLookup *L, OUT &out, int lane)
{
  static_assert(COUNT <= 64, "one pass");
  const bool valid = lane < COUNT;
  const CitwLookup d = L[valid ? lane : 0];
  int h = g_sidx[wv][d.sx];
  const double u = g_in[wv][d.in0];
  const int n = d.nr;
  h = h < 0 ? 0 : h; h = h > n - 2 ? n - 2 : h;
  const double *x = g_ro + d.xrw, *y = g_ro + d.zw;
  const double x0 = x[h], x1 = x[h + 1], y0 = y[h], y1 = y[h + 1];
  const bool ok = citw_hint_ok(h, n, x0, x1, u);
  double r = y1 - y0;
  r = r / (x1 - x0);
  r = r * (u - x0);
  if (__builtin_expect(__ballot(valid && !ok) == 0ULL, 1)) {
    if (valid) out[wv][d.out] = r + y0;
  } else {
    citw_search<SMAXN, SCOUNT, SBASE>(wv, S, lane);
    citw_lookup1d<COUNT>(wv, L, out, lane);
  }
}

template <int COUNT, typename OUT>
static __device__ __forceinline__ void citw_lookup1d(const int wv, const CitwLookup *L, OUT &out, int lane)
{
#pragma unroll
  for (int base = 0; base < COUNT; base += CITW_GROUP_LANES) citw_lookup1d_pass(wv, L, out, lane + base);
  CITW_WAVE_FENCE();
}

// 1-D tables FIRST .. FIRST + N - 1 of a round on the first N lanes of the lane group (citw_search_range; the other lanes run
// the filler descriptor 63)
template <int FIRST, int N, typename OUT>
static __device__ __forceinline__ void citw_lookup1d_range(const int wv, const CitwLookup *L, OUT &out, int lane)
{
  static_assert(N <= CITW_GROUP_LANES && FIRST + N <= 63, "one pass, and a filler row");
  citw_lookup1d_pass(wv, L, out, lane < N ? lane + FIRST : 63);
  CITW_WAVE_FENCE();
}

// ... and of a 1-D round (two wavefronts at most)
#define CITW_L1_SHARE(count) (((count) + CITW_GROUP_LANES - 1) / CITW_GROUP_LANES < 2 ? 1 : 2)
template <int COUNT, int PART, int NPARTS, typename OUT>
static __device__ __forceinline__ void citw_lookup1d_part(const int wv, const CitwLookup *L, OUT &out, int lane)
{
#pragma unroll
  for (int base = PART * CITW_GROUP_LANES; base < COUNT; base += NPARTS * CITW_GROUP_LANES) citw_lookup1d_pass(wv, L, out, lane + base);
  CITW_WAVE_FENCE();
}

template <typename OUT>
static __device__ __forceinline__ void citw_lookup1d_pass(const int wv, const CitwLookup *L, OUT &out, int lane)
{
  if (CITW_ABLATE_LOOK & 4) return;
  const CitwLookup d = L[lane];
  const int i = g_sidx[wv][d.sx];
  const double u = g_in[wv][d.in0];
  const double *x = g_ro + d.xrw, *y = g_ro + d.zw;
  const double x0 = x[i], x1 = x[i + 1], y0 = y[i], y1 = y[i + 1];
  double r = y1 - y0;
  r = r / (x1 - x0);
  r = r * (u - x0);
  out[wv][d.out] = r + y0;
}

// ---- The passes of the LANE-GROUP kernels with their interval-dependent half kept across evaluations (round 5; two / four episodes per team).
// A helper wavefront runs ONE pass of a round -- lane l of a lane group always the same search or table of its episode -- and a pass is three
// dependent LDS round trips (descriptor -> stored interval -> breakpoints / corners) and, for a table, two of its three divisions before the
// inputs even matter.  None of that changes while the stored intervals do not (4 of 2 400 evaluations move one).  CitwPassCache holds one search
// pass and one table pass of THIS wavefront: interval ends; x0, both x-direction quotients, two corners, y0, dy (a 1-D table: x0, its quotient,
// y0).  The search lanes verify their hints against the cached ends (one LDS round trip: the input); a miss runs the plain pass -- which rewrites
// g_sidx and counts g_smiss up -- and refills.  The table lanes read g_smiss with their inputs (every search flag of the round has been waited
// for by then) and refill when it has moved since they were filled.  Same operations on the same operands in the same order as the plain passes.
#ifndef CITW_PC_SEARCH
#define CITW_PC_SEARCH 1      // 0: only the table passes are cached (five registers less per wavefront)
#endif
struct CitwPassCache {
  double sxl, sxh; int sq;                       // search lane: ends of the stored interval; n | in << 8 | h << 16
  double x0, sa, z00, sb, z01, y0, dy; int tq;   // table lane (as CitwSpec); tq: in0 | in1 << 8 | out << 16
  unsigned epoch;                                // g_smiss when the table lane was filled
  bool s_valid, t_valid;
};
static __device__ CitwPassCache citw_no_pass_cache;      // what the default arguments bind (never touched: HAVE_PC is false there)

template <int MAXN, int COUNT, int PART, int NPARTS, int SBASE = 0>
static __device__ __forceinline__ void citw_search_part_c(const int wv, const CitwSearch *S, int lane, const bool HAVE_PC, CitwPassCache &c)
{
  if (!HAVE_PC || !CITW_PC_SEARCH) { citw_search_part<MAXN, COUNT, PART, NPARTS, SBASE>(wv, S, lane); return; }
  static_assert(PART * CITW_GROUP_LANES < COUNT && (PART + NPARTS) * CITW_GROUP_LANES >= COUNT, "one pass per part");
  const int ln = lane + PART * CITW_GROUP_LANES;
  const bool valid = ln < COUNT;
  if (c.s_valid) {
    const double u = g_in[wv][(c.sq >> 8) & 255];
    const bool ok = citw_hint_ok((c.sq >> 16) & 255, c.sq & 255, c.sxl, c.sxh, u);
    if (__builtin_expect(__ballot(valid && !ok) == 0ULL, 1)) return;      // every stored interval still holds: g_sidx is right as it is
  }
  citw_search_pass<MAXN, COUNT, SBASE>(wv, S, ln);       // first use, or an input left its interval: the plain pass repairs the slots (and counts g_smiss up)
  {
    const int lc = valid ? ln : 0;
    const CitwSearch d = S[lc];
    const int n = d.n;
    int h = g_sidx[wv][SBASE + lc];                        // (this lane's own slot: what it has just stored or confirmed)
    h = h < 0 ? 0 : h; h = h > n - 2 ? n - 2 : h;
    const double *x = g_bp[d.row];
    c.sxl = x[h]; c.sxh = x[h + 1];
    c.sq = n | (int)d.in << 8 | h << 16;
    c.s_valid = true;
  }
  CITW_WAVE_FENCE();
}

template <int COUNT, int PART, int NPARTS, typename OUT>
static __device__ __forceinline__ void citw_lookup2d_part_c(const int wv, const CitwLookup *L, OUT &out, int lane, const bool HAVE_PC, CitwPassCache &c)
{
  if (!HAVE_PC) { citw_lookup2d_part<COUNT, PART, NPARTS>(wv, L, out, lane); return; }
  static_assert(PART * CITW_GROUP_LANES < COUNT && (PART + NPARTS) * CITW_GROUP_LANES >= COUNT, "one pass per part");
  const unsigned ep = g_smiss;
  if (!c.t_valid || c.epoch != ep) {
    const CitwLookup d = L[lane + PART * CITW_GROUP_LANES];
    const int ix = g_sidx[wv][d.sx], iy = g_sidx[wv][d.sy];
    const double *xr = g_ro + d.xrw, *xc = g_ro + d.xcw, *z = g_ro + d.zw;
    const int nr = d.nr;
    const double x0 = xr[ix], x1 = xr[ix + 1];
    const double dx = x1 - x0;
    const double z00 = z[ix + nr * iy], z10 = z[ix + 1 + nr * iy];
    const double z01 = z[ix + nr * (iy + 1)], z11 = z[ix + 1 + nr * (iy + 1)];
    double a = z10 - z00; a = a / dx;
    double b = z11 - z01; b = b / dx;
    const double y0 = xc[iy];
    c.x0 = x0; c.sa = a; c.z00 = z00; c.sb = b; c.z01 = z01; c.y0 = y0; c.dy = xc[iy + 1] - y0;
    c.tq = (int)d.in0 | (int)d.in1 << 8 | (int)d.out << 16;
    c.epoch = ep; c.t_valid = true;
  }
  const double u0 = g_in[wv][c.tq & 255], u1 = g_in[wv][(c.tq >> 8) & 255];
  const double wx = u0 - c.x0;
  double a = c.sa * wx; a = a + c.z00;
  double b = c.sb * wx; b = b + c.z01;
  double r = b - a; r = r / c.dy; r = r * (u1 - c.y0);
  out[wv][(c.tq >> 16) & 255] = r + a;
  CITW_WAVE_FENCE();
}

template <int COUNT, int PART, int NPARTS, typename OUT>
static __device__ __forceinline__ void citw_lookup1d_part_c(const int wv, const CitwLookup *L, OUT &out, int lane, const bool HAVE_PC, CitwPassCache &c)
{
  if (!HAVE_PC) { citw_lookup1d_part<COUNT, PART, NPARTS>(wv, L, out, lane); return; }
  static_assert(PART * CITW_GROUP_LANES < COUNT && (PART + NPARTS) * CITW_GROUP_LANES >= COUNT, "one pass per part");
  const unsigned ep = g_smiss;
  if (!c.t_valid || c.epoch != ep) {
    const CitwLookup d = L[lane + PART * CITW_GROUP_LANES];
    const int i = g_sidx[wv][d.sx];
    const double *x = g_ro + d.xrw, *y = g_ro + d.zw;
    const double x0 = x[i], x1 = x[i + 1], y0 = y[i], y1 = y[i + 1];
    double r = y1 - y0;
    r = r / (x1 - x0);
    c.x0 = x0; c.sa = r; c.z00 = y0;
    c.tq = (int)d.in0 | (int)d.out << 16;
    c.epoch = ep; c.t_valid = true;
  }
  const double u = g_in[wv][c.tq & 255];
  double r = c.sa * (u - c.x0);
  out[wv][(c.tq >> 16) & 255] = r + c.z00;
  CITW_WAVE_FENCE();
}

// (citw_div_const -- x / c for a literal c in four instructions -- lives in citation_libm.h, free of LDS declarations, so that the device unit
// check of oracle/xcheck exercises the shipped text)

// ---- look-up lanes with the hint-dependent half PRECOMPUTED (single-episode team kernels, wave 0).  Of a 2-D interpolation
//     a = (z10 - z00) / dx * (u0 - x0) + z00,  b = (z11 - z01) / dx * (u0 - x0) + z01,  r = (b - a) / dy * (u1 - y0) + a
// only the underlined products depend on the inputs: the descriptor, the interval indices (the hints the previous evaluation
// left in g_sidx), the four corners, the interval ends and BOTH x-direction quotients depend on the interval alone -- five
// dependent LDS round trips and two of the three divisions.  citw_spec_pre() computes them at the top of the evaluation, where
// they overlap the arithmetic of wave 0's input cones (one basic block); once the inputs are in g_in, citw_spec_tail() verifies
// the hints on the search lanes (two compares against the preloaded interval ends) and finishes the interpolations: one LDS
// round trip, one division.  Same operations on the same operands in the same order as citw_lookup2d_pass / citw_lookup1d_pass
// (a 1-D table is the `a` half of a 2-D lane), so the bits are the same; if any search lane fails its test the caller runs the
// plain passes, which repair the indices.  The lanes of ALL rounds are precomputed at once (merged descriptor row, one table per
// lane): the later rounds' tails run behind barrier B1 on the registers of their lanes.
struct CitwSpec {
  double sxl, sxh;                           // search lane: ends of the stored interval
  int sq;                                    // ... n | in << 8 | h << 16 | (h == stored) << 24
  double x0, sa, z00, sb, z01, y0, dy;       // table lane
  int tq;                                    // ... in0 | in1 << 8 | out << 16 | is1d << 24
  double bx0, ba, by0;                       // second table lane (round 5): a 1-D table of round 1 -- x0, (y1 - y0) / (x1 - x0), y0
  int bq;                                    // ... in0 | out << 8
};

// the lanes, kept across evaluations by the kernels that can afford the registers (one-episode team: rollout_team.inc): valid until an
// evaluation repairs an interval (the generated code clears it in front of the plain search)
struct CitwSpecCache { CitwSpec p; bool valid; };
static __device__ CitwSpecCache citw_no_spec_cache;      // what the default arguments bind when the caller keeps none (never touched: HAVE_SC is false there)

// N1: the first N1 lanes also carry one 1-D table of round 1 each (descriptor row L1): until round 5 that pass ran on a helper wavefront BEHIND
// the hint verification of this one (flag, poll, descriptor -> interval -> table: three dependent LDS round trips and a division on the path
// to barrier B1); here its interval-dependent half -- everything but the last multiply-add -- is part of the cached lanes.
template <int NS, int NT, int N1 = 0>
static __device__ __forceinline__ CitwSpec citw_spec_pre(const int wv, const CitwSearch *S, const CitwLookup *L, const int lane, const CitwLookup *L1 = nullptr)
{
  static_assert(CITW_GROUP_LANES == 64 && NS <= 64 && NT <= 64 && N1 <= 64, "one lane per search / table");
  CitwSpec p;
  p.bx0 = p.ba = p.by0 = 0.0; p.bq = 0;
  if (N1 > 0 && !(CITW_ABLATE_LOOK & 8)) {      // the operations of citw_lookup1d_pass, up to the quotient
    const CitwLookup d = L1[lane < N1 ? lane : 0];
    const int i = g_sidx[wv][d.sx];
    const double *x = g_ro + d.xrw, *y = g_ro + d.zw;
    const double x0 = x[i], x1 = x[i + 1], y0 = y[i], y1 = y[i + 1];
    double r = y1 - y0;
    r = r / (x1 - x0);
    p.bx0 = x0; p.ba = r; p.by0 = y0;
    p.bq = (int)d.in0 | (int)d.out << 8;
  }
  if (CITW_ABLATE_LOOK & 8) { p.sxl = p.sxh = p.x0 = p.sa = p.z00 = p.sb = p.z01 = p.y0 = p.dy = 0.0; p.sq = p.tq = 0; return p; }
  {
    const CitwSearch d = S[lane < NS ? lane : 0];
    const int stored = g_sidx[wv][d.pad];
    const double *x = g_bp[d.row];
    const int n = d.n;
    int h = stored < 0 ? 0 : stored;
    h = h > n - 2 ? n - 2 : h;
    p.sxl = x[h]; p.sxh = x[h + 1];
    p.sq = n | (int)d.in << 8 | h << 16 | (h == stored ? 1 : 0) << 24;
  }
  {
    const CitwLookup d = L[lane < NT ? lane : 0];
    const int ix = g_sidx[wv][d.sx], iy = g_sidx[wv][d.sy];
    const double *xr = g_ro + d.xrw, *xc = g_ro + d.xcw, *z = g_ro + d.zw;
    const int nr = d.nr;
    const double x0 = xr[ix], x1 = xr[ix + 1];
    const double dx = x1 - x0;
    const double z00 = z[ix + nr * iy], z10 = z[ix + 1 + nr * iy];
    const double z01 = z[ix + nr * (iy + 1)], z11 = z[ix + 1 + nr * (iy + 1)];
    double a = z10 - z00; a = a / dx;
    double b = z11 - z01; b = b / dx;
    const double y0 = xc[iy];
    p.x0 = x0; p.sa = a; p.z00 = z00; p.sb = b; p.z01 = z01; p.y0 = y0; p.dy = xc[iy + 1] - y0;
    p.tq = (int)d.in0 | (int)d.in1 << 8 | (int)d.out << 16 | (int)d.p1 << 24;
  }
  return p;
}

// search lanes [S0, S1) verify, table lanes [T0, T1) interpolate and -- if every search lane confirmed its interval -- store;
// returns whether a lane missed (wave-uniform): then nothing was stored and the caller runs the plain passes of the round
template <int S0, int S1, int T0, int T1, int N1 = 0, typename OUT>
static __device__ __forceinline__ bool citw_spec_tail(const int wv, const CitwSpec &p, OUT &out, const int lane)
{
  if (CITW_ABLATE_LOOK & 8) return false;
  const double su = g_in[wv][(p.sq >> 8) & 255];
  const double u0 = g_in[wv][p.tq & 255], u1 = g_in[wv][(p.tq >> 8) & 255];
  const double ub = N1 > 0 ? g_in[wv][p.bq & 255] : 0.0;
  const bool ok = citw_hint_ok((p.sq >> 16) & 255, p.sq & 255, p.sxl, p.sxh, su) && (p.sq >> 24) != 0;
  const double wx = u0 - p.x0;
  double a = p.sa * wx; a = a + p.z00;
  double b = p.sb * wx; b = b + p.z01;
  double r = b - a; r = r / p.dy; r = r * (u1 - p.y0);
  const double res = (p.tq >> 24) != 0 ? a : r + a;
  double rb = p.ba * (ub - p.bx0);           // (citw_lookup1d_pass: r = r * (u - x0); out = r + y0)
  rb = rb + p.by0;
  const bool miss = __ballot(lane >= S0 && lane < S1 && !ok) != 0ULL;
  if (!miss && lane >= T0 && lane < T1) out[wv][(p.tq >> 16) & 255] = res;
  if (N1 > 0 && !miss && lane < N1) out[wv][(p.bq >> 8) & 255] = rb;
  CITW_WAVE_FENCE();
  return miss;
}

// ---- table3 S-function: 3-D table, linear interpolation.  The reference walks linearly from an interval cached in
// IWORK/RWORK (part of rtDW, not observable through step()); the interval it ends on is
// clamp(max{i : tab[i] < x}, 0, n-2) whatever the cache holds.
static __device__ __forceinline__ int citw_t3_interval(const double *tab, int n, double x)
{
  int i = -1;
  for (int k = 0; k < n; ++k) i = (tab[k] < x) ? k : i;
  i = i < 0 ? 0 : i;
  return i > n - 2 ? n - 2 : i;
}

static __device__ __forceinline__ double citw_table2(const double *xt, const double *yt, int ix, int iy, const double *tab,
                                                     int M, double x, double y)
{
  double rows[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const double v0 = tab[(ix + 0) * M + (iy + j)], v1 = tab[(ix + 1) * M + (iy + j)];
    const double x1 = xt[ix + 1], x0 = xt[ix];
    double d = v1 - v0; const double dx = x1 - x0, wgt = x - x0;
    d = d * wgt; d = d / dx;
    rows[j] = (x == x1) ? v1 : d + v0;
  }
  const double y1 = yt[iy + 1], y0 = yt[iy];
  const double wgt = y - y0, dy = y1 - y0;
  double d = rows[1] - rows[0];
  d = d * wgt; d = d / dy;
  return (y == y1) ? rows[1] : d + rows[0];
}

static __device__ __noinline__ double citw_table3(const double *t3, double u0, double u1, double u2)
{
  const double *P1 = t3, *P2 = t3 + 3, *P3 = t3 + 7, *P4 = t3 + 10;
  const int i0 = citw_t3_interval(P1, 3, u0), i1 = citw_t3_interval(P2, 4, u1), i2 = citw_t3_interval(P3, 3, u2);
  const double *slab = P4 + i2 * 12;
  const double a = citw_table2(P1, P2, i0, i1, slab, 4, u0, u1);
  const double b = citw_table2(P1, P2, i0, i1, slab + 12, 4, u0, u1);
  const double z1 = P3[i2 + 1], z0 = P3[i2];
  const double wgt = u2 - z0, dz = z1 - z0;
  double d = b - a;
  d = d * wgt; d = d / dz;
  return (u2 == z1) ? b : d + a;
}

// Dormand-Prince "ode5" tableau as the reference's literal pool holds it (0x13688, 0x13898..0x13938): citw_ode5_a / citw_ode5_b below

// ODE5 stage combination of one state component:  X_i = y_i + sum_{j <= st} f_j,i * (B[st][j] * h),  accumulated in j order
// like the reference (rt_ertODEUpdateContinuousStates @0x9eb8..0xa4f3).  Specialised per stage: the products B[st][j] * h
// fold to literals (the same IEEE product the run-time expression gives) and all stage derivatives are fetched from LDS
// together, instead of a variable-length loop with a constant-memory load and an LDS round trip per term.
static __host__ __device__ constexpr double citw_ode5_b(int st, int j)
{
  constexpr double B[6][6] = {
    {0.2, 0, 0, 0, 0, 0},
    {0.075, 0.225, 0, 0, 0, 0},
    {0.9777777777777777, -3.7333333333333334, 3.5555555555555554, 0, 0, 0},
    {2.9525986892242035, -11.595793324188385, 9.822892851699436, -0.2908093278463649, 0, 0},
    {2.8462752525252526, -10.757575757575758, 8.906422717743473, 0.2784090909090909, -0.2735313036020583, 0},
    {0.09114583333333333, 0.0, 0.44923629829290207, 0.6510416666666666, -0.322376179245283, 0.13095238095238096},
  };
  return B[st][j];
}

// A[st] of the tableau as scalar selects of literals (no constant-memory load on the path into the next evaluation)
static __device__ __forceinline__ double citw_ode5_a(int st)
{
  double a = 0.2;
  a = st == 1 ? 0.3 : a;
  a = st == 2 ? 0.8 : a;
  a = st == 3 ? 0.8888888888888888 : a;
  a = st >= 4 ? 1.0 : a;
  return a;
}

template <int ST>
static __device__ __forceinline__ double citw_ode5_combine_st(const double (*f)[20], int li, double yi)
{
  constexpr double h = 0.01;
  double fv[ST + 1];
#pragma unroll
  for (int j = 0; j <= ST; ++j) fv[j] = f[j][li];
  double acc = fv[0] * (citw_ode5_b(ST, 0) * h);
#pragma unroll
  for (int j = 1; j <= ST; ++j) acc = acc + fv[j] * (citw_ode5_b(ST, j) * h);
  return acc + yi;
}

// The same sum in two pieces (team kernels): the terms of the EARLIER stages are in LDS since the previous evaluation's last
// barrier, so their loads and multiply-adds are issued in front of the evaluation and overlap it; behind the evaluation's
// last barrier only the new stage's term and the state are added.  Same operations in the same order as citw_ode5_combine_st.
template <int ST>
static __device__ __forceinline__ double citw_ode5_partial_st(const double (*f)[20], int li)
{
  constexpr double h = 0.01;
  if constexpr (ST == 0) return 0.0;
  else {
    double fv[ST];
#pragma unroll
    for (int j = 0; j < ST; ++j) fv[j] = f[j][li];
    double acc = fv[0] * (citw_ode5_b(ST, 0) * h);
#pragma unroll
    for (int j = 1; j < ST; ++j) acc = acc + fv[j] * (citw_ode5_b(ST, j) * h);
    return acc;
  }
}
static __device__ __forceinline__ double citw_ode5_partial(int st, const double (*f)[20], int li)
{
  switch (st) {           // wave-uniform
    case 0: return 0.0;
    case 1: return citw_ode5_partial_st<1>(f, li);
    case 2: return citw_ode5_partial_st<2>(f, li);
    case 3: return citw_ode5_partial_st<3>(f, li);
    case 4: return citw_ode5_partial_st<4>(f, li);
    default: return citw_ode5_partial_st<5>(f, li);
  }
}
static __device__ __forceinline__ double citw_ode5_finish(int st, double part, const double (*f)[20], int li, double yi)
{
  constexpr double h = 0.01;
  const double fs = f[st][li];
  double c = citw_ode5_b(0, 0) * h;                  // B[st][st] * h as scalar selects of literals (the products fold)
  c = st == 1 ? citw_ode5_b(1, 1) * h : c;
  c = st == 2 ? citw_ode5_b(2, 2) * h : c;
  c = st == 3 ? citw_ode5_b(3, 3) * h : c;
  c = st == 4 ? citw_ode5_b(4, 4) * h : c;
  c = st >= 5 ? citw_ode5_b(5, 5) * h : c;
  const double t = fs * c;
  const double acc = st == 0 ? t : part + t;
  return acc + yi;
}

static __device__ __forceinline__ double citw_ode5_combine(int st, const double (*f)[20], int li, double yi)
{
  switch (st) {           // wave-uniform
    case 0: return citw_ode5_combine_st<0>(f, li, yi);
    case 1: return citw_ode5_combine_st<1>(f, li, yi);
    case 2: return citw_ode5_combine_st<2>(f, li, yi);
    case 3: return citw_ode5_combine_st<3>(f, li, yi);
    case 4: return citw_ode5_combine_st<4>(f, li, yi);
    default: return citw_ode5_combine_st<5>(f, li, yi);
  }
}

// Per-episode dynamics state of the wave kernel: lane i < 19 keeps continuous state i (ODE5 combination per lane);
// the model evaluation reads all 19 as wave-uniform LDS loads from g_xs.
struct CitwState {
  double xi;          // this lane's own state component (lane < 19); the wave-uniform copy lives in g_xs
  double xj;          // 16-lane groups only: state component lane + 16 (lanes 0..2)
  double t;           // model time
  unsigned tick;      // clockTick0
};
