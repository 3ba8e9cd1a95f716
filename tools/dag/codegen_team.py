#!/usr/bin/env python3
"""Generate the multi-wavefront ("team") model evaluation: serl_amd/csrc/gen/citation_<variant>_team.inc.

With one wavefront per SIMD every instruction -- VALU, SALU, LDS, waitcnt -- costs a 4-cycle issue slot, so one model
evaluation (~4 500 instructions) is issue-bound on a single wavefront.  When there are fewer episodes than CUs, the
other SIMDs of the CU take a share of the work: a team of K wavefronts per episode.  K = 7 by default (round 2): with the
actor wavefront of rollout_team.inc that makes eight wavefronts, TWO per SIMD -- a wavefront alone on a SIMD pays 7.6 cycles
per dependent f64 operation and ~85 per LDS round trip with nothing to fill them; a neighbour on the same SIMD issues into
those gaps, and a wavefront parked at a barrier leaves the whole SIMD to its neighbour, which evens out what the static
balance below misses.  Measured (150 episodes, us per env step): K = 4: 22.8 - 23.7, 5: 22.9, 6: 21.7, 7: 21.2.

  wave 0 (main)     cones of the round-1 look-up inputs (all but the one behind the libm pow chain) -> [poll the flag of the
                    wave that hands that input over] -> index search -> [raise its flag: wave 1 runs the 1-D interpolation
                    pass from here on] -> 2-D interpolation pass -> [barrier B1]
                    -> later look-up rounds, its share of the derivative cones -> [barrier B2]
  waves 1..K-1      the libm calls of the evaluation, each made by exactly ONE wave (lane-parallel), results in its g_m row,
  (helpers)         published by an LDS flag (release store of the evaluation's sequence number; a consumer polls it
                    once -- acquire -- before its first read); the last wave starts with the pow / exp chain of the air
                    data and puts the look-up input behind it into wave 0's input row; then a share of the glue that does
                    NOT depend on any look-up (rotation matrices, gravity, engines, kinematic equations ...) -> [B1] -> a
                    share of the derivative cones behind the round-1 look-ups (results are in LDS for everybody),
                    Derivative-block banks -> [B2]

Every sink (a value needed later, with its cone of ancestors) goes to the wave that minimises load + AFFINITY x the nodes
it would have to add (it leans towards the wave that already holds most of the cone); libm results are imports for
everybody but their maker, light sub-expressions shared by several cones are recomputed rather than exchanged; what a wave
needs from another one's pre-barrier values crosses through LDS (g_x) around B1.  Every wave executes exactly two workgroup
barriers per evaluation; only consumers ever wait on a flag, and only if the producer is late.  The arithmetic (operation
order per value) is that of the single-wave code: results are bit-identical.

Knobs (environment, for sweeps; defaults are the measured best): CITW_TEAM_WAVES, CITW_TEAM_LOOKUP_COST,
CITW_TEAM_ROUND2_COST, CITW_TEAM_AFFINITY, CITW_TEAM_SHARE_LIBM, CITW_TEAM_SPLIT_INPUTS, CITW_TEAM_FN_SCALE,
CITW_TEAM_LIBM_SCALE, CITW_TEAM_POST_BIAS, CITW_TEAM_IMPORT_COST, CITW_TEAM_AFFINITY_POST; and two that were measured
and stay off (profiles/r02_team_experiments.md): CITW_TEAM_SPREAD_INPUTS (every round-1 input cone on a helper, handed to
wave 0 by flags: 24.0 - 26.0), CITW_TEAM_SIMD_PAIRS (balance per SIMD instead of per wavefront: 22.6 - 23.4).

Usage: python tools/dag/codegen_team.py [variant ...] [--suffix=_tag]      (CITW_TEAM_WAVES=2..7 overrides the team size;
       --suffix writes gen/citation_<variant>_team_tag.inc for tools/exp_build.py A/B builds)
"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_dag, codegen, symex
from codegen import LOOKUPS, hexf

LEAF = ('cf', 'ci', 'in', 'in_i', 'true', 'false')
TASKS_UNSPLIT = set()
TEAM_WAVES = int(os.environ.get('CITW_TEAM_WAVES', 7))
# instruction estimates used by the balancer (wave 0's fixed work: search + 2-D + 1-D passes; later look-up rounds)
LOOKUP_PHASES = int(os.environ.get('CITW_TEAM_LOOKUP_COST', 1400))
POST_ROUND2_COST = int(os.environ.get('CITW_TEAM_ROUND2_COST', 300))
IMPORT_COST = float(os.environ.get('CITW_TEAM_IMPORT_COST', 0.0))         # units per value a wave fetches from LDS behind B1
# initial load per wave behind B1 (round 1, K = 4: the wave that hands the pow chain over needed a bias of 100 units; with two
# wavefronts per SIMD the hardware evens that out and no bias measures best)
POST_BIAS = [float(v) for v in os.environ.get('CITW_TEAM_POST_BIAS', '0').split(',') if v]
PRE_BIAS_GROUPS = '0,0,0,-100'      # the lane-group kernels' partition (gen/citation_<v>_teamg.inc, --lane-groups): there wave 1 still runs round 1's 1-D interpolation pass
PRE_BIAS = [float(v) for v in os.environ.get('CITW_TEAM_PRE_BIAS', PRE_BIAS_GROUPS if '--lane-groups' in sys.argv else '0,-200,0,-100').split(',') if v]        # initial load per wave in front of B1 (negative: the wave takes more; wave 3 shares its SIMD with the mostly parked actor wavefront)
AFFINITY_POST = float(os.environ.get('CITW_TEAM_AFFINITY_POST', 0.0))
SHARE_LIBM = int(os.environ.get('CITW_TEAM_SHARE_LIBM', 1))           # 1: every libm call is made by one wave, the others read the result (flag hand-over)
SPREAD_IN = int(os.environ.get('CITW_TEAM_SPREAD_INPUTS', 0))         # 1: EVERY round-1 input cone runs on a helper; wave 0 waits for their input flags (g_iflag), then looks up
L2_WAVES = [int(x) for x in os.environ.get('CITW_TEAM_L2_WAVES', '6,4,5').split(',')]           # the helper waves that take 2-D passes 1, 2, 3
SEARCH_WAVES = [int(x) for x in os.environ.get('CITW_TEAM_SEARCH_WAVES', '1,4').split(',')]   # the helper waves that take search passes 1 (and 2)
SEARCH_AT = float(os.environ.get('CITW_TEAM_SEARCH_AT', 1.0))         # where in a helper's own glue (fraction of its sinks) its shared search pass sits (1.0: behind all of it)
SHARE_SEARCH = int(os.environ.get('CITW_TEAM_SHARE_SEARCH', 1))    # 1: ... and the passes of round 1's index search with the helper waves SEARCH_WAVES (on the lightly loaded waves 1 and 3: four per team 26.8 -> 25.5 us; on waves 2 and 4, which also interpolate: 28.4)
SHARE_1D_WAVE = int(os.environ.get('CITW_TEAM_SHARE_1D_WAVE', 3))   # ... this helper
SHARE_1D = int(os.environ.get('CITW_TEAM_SHARE_1D', 1))            # 1: ... and the second pass of the 1-D interpolation (16 lanes per episode) runs on wave 3 beside wave 1's first
SHARE_2D = int(os.environ.get('CITW_TEAM_SHARE_2D', 1))            # 1: with several episodes per team (lane groups) the passes of round 1's 2-D interpolation are shared with helper waves 2, 4, 6 (CITW_L2_SHARE)
H1D_WAVE = int(os.environ.get('CITW_TEAM_H1D_WAVE', 1))               # ... this helper
OFFLOAD_1D = int(os.environ.get('CITW_TEAM_OFFLOAD_1D', 1))          # 1: the 1-D interpolation pass of round 1 runs on helper wave 1 (after wave 0's index search, by flag) beside wave 0's 2-D pass
SPREAD_MIN = float(os.environ.get('CITW_TEAM_SPREAD_MIN', 0))         # with SPREAD_INPUTS: only input cones at least this heavy (units) leave wave 0
SIMD_PAIRS = int(os.environ.get('CITW_TEAM_SIMD_PAIRS', 0))            # 1: the balancer counts the load of a SIMD (waves b and b + 4 share one) instead of a wave's
ACTOR_UNITS = float(os.environ.get('CITW_TEAM_ACTOR_UNITS', 550))     # the actor wavefront (wave index K) as load on the SIMD it shares, units per evaluation
ACTOR_POST = float(os.environ.get('CITW_TEAM_ACTOR_POST', 0))
COLD_COST = float(os.environ.get('CITW_TEAM_COLD_COST', 1.0))        # balancer: cost factor of gated nodes that the trimmed flight condition does not execute
LIBM_WAVE = {kv.split(':')[0]: int(kv.split(':')[1]) for kv in os.environ.get('CITW_TEAM_LIBM_WAVE', '').split(',') if ':' in kv}     # libm function -> the helper that makes its calls (default: the least loaded one)
EARLY_FLAG = int(os.environ.get('CITW_TEAM_EARLY_FLAG', 2))             # 2: the first libm function group of a wavefront (sincos in front of tan) is announced by a flag of its own, g_flag[8 + q] -- except on the wavefront of the handed-over chain (1: there too -- with lane groups that produced NaNs in the first launch of a process, r03 sweeps 39 / 41, cause not found; 0: off)
TAN_MERGE = int(os.environ.get('CITW_TEAM_TAN_MERGE', 1))               # 1: the tan lanes of a wavefront ride its sincos pass and divide behind it (no second body)
TWO_PASS = int(os.environ.get('CITW_TEAM_TWO_PASS', 1))                 # 1: a helper computes what needs no foreign libm result before its first flag wait, node by node
SCALAR_LIBM = int(os.environ.get('CITW_TEAM_SCALAR_LIBM', 1))          # 1: one or two individually guarded libm calls of a wavefront are made wave-uniformly under their guards' scalar branches (no LDS trip of argument and result)
SPEC_1D = int(os.environ.get('CITW_TEAM_SPEC_1D', 1))                  # 1: ... and round 1's 1-D tables ride those lanes too (one episode per team; no helper pass behind the hint verification)
SPEC = int(os.environ.get('CITW_TEAM_SPEC', 1))                        # 1: emit the merged descriptor row + the precomputed look-up lanes of wave 0 (citw_spec_pre / citw_spec_tail; compiled in with -DCITW_SPEC_LOOKUP=1)
STAGE0 = int(os.environ.get('CITW_TEAM_STAGE0', 1))                    # 1: glue that depends on the command vector alone runs in the first of the six evaluations only (its look-up inputs / exchanged values keep their LDS slots)
OWN_LOOKUPS = int(os.environ.get('CITW_TEAM_OWN_LOOKUPS', 0))          # 1: the helper that computes a look-up input also searches / interpolates the (1-D) tables keyed on it: wave 0 never waits for it
SPLIT_IN = int(os.environ.get('CITW_TEAM_SPLIT_INPUTS', 1))           # 1: the heaviest round-1 input cone (pow chain) runs on a helper, handed over by flag
FN_SCALE = float(os.environ.get('CITW_TEAM_FN_SCALE', 0.6))            # libm bodies relative to the first estimates in FN (round 4: the short bodies of citation_libm.h -- 1.0: 17.21 us per env step, 0.6 with LIBM_SCALE 1.5: 16.85; 39 random settings around it 16.79 - 18.99, profiles/r04_experiments.md)
LIBM_SCALE = float(os.environ.get('CITW_TEAM_LIBM_SCALE', 1.5))        # extra factor for the handed-over cone's libm bodies (2.5 with ocml's pow)
AFFINITY = float(os.environ.get('CITW_TEAM_AFFINITY', 1.0))            # > 0: a sink leans towards the wave that already holds most of its cone
SPLIT_CHAIN = int(os.environ.get('CITW_TEAM_SPLIT_CHAIN', 0))      # 1: the engine look-up chain (chain round -> second round -> N1 / N2 derivatives) on a wavefront of its own
ENGINE_WAVE = int(os.environ.get('CITW_TEAM_ENGINE_WAVE', 4))      # ... this one
CHAIN_COST = float(os.environ.get('CITW_TEAM_CHAIN_COST', 900))    # its look-up passes in balancer units (chain round in front of B1)
CHAIN_B1 = os.environ.get('CITW_TEAM_CHAIN_B1', 'rc')              # 'rc': barrier B1 behind the chain round; 'r1': behind the second round too
PRELOAD = int(os.environ.get('CITW_TEAM_PRELOAD', 1))             # 1: every wave loads the states / commands / table constants it reads ONCE, at the top of the evaluation, into locals (a flag poll or a barrier makes the compiler re-load an inline g_xs[..] otherwise)
POST_TASKS = int(os.environ.get('CITW_TEAM_POST_TASKS', 1))        # 1: the glue behind the look-ups as a task graph (fine-grained, values handed over by flags) instead of one cone per derivative
TASK_MAX = float(os.environ.get('CITW_TEAM_TASK_MAX', 50))         # a task heavier than this (cost units) is split at an inner node
TASK_MIN = float(os.environ.get('CITW_TEAM_TASK_MIN', 12))         # ... into pieces no lighter than this; lighter shared sub-expressions are recomputed
TASK_COMM = float(os.environ.get('CITW_TEAM_TASK_COMM', 24))       # cost units between "value stored" and "value usable on another wavefront" (LDS store, flag, poll, load)
KREGS_MAX = int(os.environ.get('CITW_TEAM_KREGS_MAX', 24))                # ... at most this many per role (two VGPRs each)
FMA = int(os.environ.get('CITW_TEAM_FMA', 0))                            # TIMING EXPERIMENT (results change): products feeding an add / sub on the same wavefront fused into fma
KREGS_ONE_MOVE = int(os.environ.get('CITW_TEAM_KREGS_ONE_MOVE', 1))        # 1: ... literals that cost ONE move (low dword zero) too, behind the others (r05b / r05d: 15.26 -> 15.03 us per env step together with the unpadded g_w)
KREGS = int(os.environ.get('CITW_TEAM_KREGS', 1))                      # 1: f64 literals that cost two 32-bit moves go through CITW_K(slot, literal): registers loaded once per episode (citation_wave.h CitwKRegs) when the kernel passes them, the literal itself otherwise
CW = dict(div=11, sqrt=15, sel=3, unord=2, table3=120)
FN = dict(sc_sin=100, sc_cos=100, sin=100, cos=100, tan=120, exp=40, log10=60, log=60, atan=80, pow=250)


def n_search(R):
    """index searches of a round that wave 0's passes cover (the rest belongs to the helper that computes their input)"""
    return R.get('ns_main', len(R['searches']))


def n_l1(R):
    return R.get('n1_main', len(R['L1']))


class TeamGen(codegen.Gen):
    def __init__(self, variant, waves=None, **kw):
        self.K = waves or TEAM_WAVES
        super().__init__(variant, split_chain=bool(SPLIT_CHAIN and self.K > 2), **kw)
        self.EW = min(ENGINE_WAVE, self.K - 1) if self.chain_round is not None else 0     # the wavefront that walks the later look-up rounds

    def closure(self, sinks, within):
        out, st = set(), list(sinks)
        while st:
            m = st.pop()
            if m in out or m not in within:
                continue
            out.add(m)
            if m in getattr(self, 'shared_libm', ()):
                continue            # a libm result made once for the team: its argument cone belongs to the wave that makes the call
            st.extend(build_dag.children(self.g, m))
        return out

    def plan(self):
        g, K = self.g, self.K
        users = collections.defaultdict(list)
        for n in self.order:
            for c in build_dag.children(g, n):
                users[c].append(n)
        self.users = users
        glue = lambda n: g.nodes[n][0] not in LEAF + LOOKUPS and not self.inv[n]      # per-step invariants come from LDS
        S0 = set(n for n in self.order if self.rnd[n] == 0 and glue(n))
        A0 = self.closure([n for n in self.rounds[0]['ins'] if n in S0], S0)
        roots = list(dict.fromkeys(list(self.xdot) + list(self.dw_out.values())))
        rootset = set(roots)
        later_use = lambda n: any((u not in S0) for u in users[n])
        sinks = [n for n in self.order if n in S0 and n not in A0 and (later_use(n) or n in rootset)]
        cdiv = lambda n: (g.nodes[n][0] == 'div' and codegen.CONST_DIV and g.nodes[g.nodes[n][2]][0] == 'cf'
                          and self.const_div_ok(g.nodes[g.nodes[n][2]][1]))
        cost0 = lambda n: 0 if g.nodes[n][0] in FN else (2 * 4 if cdiv(n) else 2 * CW.get(g.nodes[n][0], 1))
        cost = lambda n: cost0(n) * (COLD_COST if n in self.cold else 1.0)
        fns = [set() for _ in range(K)]
        self.shared_libm = set(self.libm_slot) if (SHARE_LIBM and K > 1) else set()
        shared = self.shared_libm

        def fn_cost(nodes, b, commit=False):
            c = 0
            for m in nodes:
                f = g.nodes[m][0]
                if f in FN and m not in shared:
                    f2 = {'sc_sin': 'sincos', 'sc_cos': 'sincos', 'sin': 'sincos', 'cos': 'sincos'}.get(f, f)
                    if f2 not in fns[b]:
                        c += FN_SCALE * FN[f]
                        if commit:
                            fns[b].add(f2)
            return c
        # ---- glue in front of the look-ups.  The cone of the heaviest round-1 look-up input -- the libm pow chain of the air
        # data -- runs on helper wave P, which puts the input into wave 0's g_in row and raises a flag (LDS, release /
        # acquire); wave 0 computes the other inputs meanwhile and polls the flag before the index search.  No barrier:
        # only wave 0 ever waits, and only if P is late.
        ins0 = [n for n in self.rounds[0]['ins']]
        full_cone = lambda n: set(m for m in self.closure_all(n) if m in S0)
        weight = lambda n: sum(cost(m) for m in full_cone(n)) + sum(FN.get(g.nodes[m][0], 0) for m in full_cone(n))
        self.in_owner = {n: 0 for n in ins0}
        self.P = None
        if SPLIT_IN and K > 1 and ins0:
            heavy = max(ins0, key=lambda n: (weight(n), -ins0.index(n)))
            if any(g.nodes[m][0] in FN for m in full_cone(heavy)):
                self.P = K - 1
                self.in_owner[heavy] = self.P
        self.spread = bool(SPREAD_IN and K > 2 and ins0 and SHARE_LIBM)
        # ---- the tables keyed on P's input.  If they are all 1-D (nominal: two 3-point tables on the calibrated airspeed), P
        # searches and interpolates them itself as soon as it has the input -- on its first lanes, with the same pass functions,
        # in slots of their own at the END of the round's search / 1-D lists -- and wave 0 starts its passes without polling P.
        self.own_lk = None
        R0 = self.rounds[0]
        if OWN_LOOKUPS and self.P is not None and not (SPREAD_IN and K > 2) and self.chain_round is None and 'ns_main' not in R0:
            kh = [k for k, n in enumerate(R0['ins']) if self.in_owner[n] == self.P]
            mv_l = [e for e in R0['L1'] if e['in0'] in kh]
            if not any(e['in0'] in kh or e['in1'] in kh for e in R0['L2']) and 0 < len(mv_l) <= 16:
                mv_s = [i for i, sr in enumerate(R0['searches']) if sr[2] in kh]
                keep_s = [i for i in range(len(R0['searches'])) if i not in mv_s]
                newpos = {old: new for new, old in enumerate(keep_s + mv_s)}
                R0['searches'] = [R0['searches'][i] for i in keep_s + mv_s]
                for e in R0['L2']:
                    e['sx'], e['sy'] = newpos[e['sx']], newpos[e['sy']]
                for e in R0['L1']:
                    e['sx'] = newpos[e['sx']]
                R0['L1'] = [e for e in R0['L1'] if e['in0'] not in kh] + mv_l
                for k, e in enumerate(R0['L1']):
                    self.outslot[e['node']] = (R0['oarr'], 64 + R0['obase1'] + k)
                R0['ns_main'], R0['n1_main'] = len(keep_s), len(R0['L1']) - len(mv_l)
                assert len(mv_s) <= 16
        if 'ns_main' in R0:
            self.own_lk = (R0['ns_main'], len(R0['searches']) - R0['ns_main'], R0['n1_main'], len(R0['L1']) - R0['n1_main'])
        if self.spread:
            # every input cone goes to a helper (heaviest first, to the least loaded one; the pow chain stays with P); wave 0
            # keeps the look-up phases only
            hl = [0.0] * K
            hh = [set() for _ in range(K)]
            if self.P is not None:
                cone = full_cone(next(n for n in ins0 if self.in_owner[n] == self.P))
                hh[self.P] |= cone; hl[self.P] += sum(cost(m) for m in cone) + sum(FN.get(g.nodes[m][0], 0) for m in cone)
            for n in sorted([n for n in ins0 if self.in_owner[n] == 0 and weight(n) > SPREAD_MIN], key=lambda n: (-weight(n), ins0.index(n))):
                cone = full_cone(n)
                b = min(range(1, K), key=lambda q: (hl[q] + sum(cost(m) for m in cone if m not in hh[q]), q))
                hl[b] += sum(cost(m) for m in cone if m not in hh[b]); hh[b] |= cone
                self.in_owner[n] = b
        A0w = self.closure([n for n in ins0 if self.in_owner[n] == 0 and n in S0], S0)       # wave 0's share of A0
        have = [set(A0w)] + [set() for _ in range(K - 1)]
        load = [sum(cost(m) for m in A0w) + fn_cost(A0w, 0, True) + LOOKUP_PHASES] + [0.0] * (K - 1)
        for q, v in enumerate(PRE_BIAS[:K]):
            load[q] += v
        if self.chain_round is not None:
            # the engine wave: cones of the chain round's inputs and of what the second round's inputs need from in front of B1
            EW = self.EW
            seeds = [n for n in self.chain_round['ins'] if n in S0]
            for R in self.rounds[1:]:
                for n in R['ins']:
                    seeds += [m for m in self.closure_all(n) if m in S0]
            AE = self.closure(seeds, S0)
            load[EW] += sum(cost(m) for m in AE if m not in have[EW]) + CHAIN_COST
            have[EW] |= AE
        self.h1d = min(H1D_WAVE, K - 1) if (OFFLOAD_1D and K > 2 and self.rounds[0]['L1']) else None
        self.l2_helpers = L2_WAVES if (SHARE_2D and K >= 7 and self.h1d is not None and self.rounds[0]['L2']) else []
        if self.h1d is not None:
            load[self.h1d] += 220.0            # the 1-D pass it takes over from wave 0
            load[0] -= 220.0
        self.handed = set()
        self.handed_of = {b: set() for b in range(K)}
        for n in ins0:
            b = self.in_owner[n]
            if b != 0:
                self.handed_of[b] |= self.closure([n], S0)
        if self.P is not None:
            self.handed = set(self.handed_of[self.P])
        for b in range(1, K):
            if self.handed_of[b]:
                have[b] |= self.handed_of[b]
                load[b] += sum(cost(m) for m in self.handed_of[b]) + (LIBM_SCALE if b == self.P else 1.0) * fn_cost(self.handed_of[b], b, True)
        def simd_extra(q, ld, actor=ACTOR_UNITS):
            """what else runs on wave q's SIMD (wavefronts q and q +- 4 share one; the actor wavefront is wave K)"""
            if not SIMD_PAIRS:
                return 0.0
            x = 0.0
            for p in (q - 4, q + 4):
                if 0 <= p < K:
                    x += ld[p]
                elif p == K:
                    x += actor
            return x
        # ---- who makes which libm call (shared results): the calls of the handed-over cone belong to P, the other groups
        # (one function body each) go to the helpers, heaviest first
        self.call_owner = {}
        if shared:
            grp = lambda j: self.libm_calls[j][0][0]
            gcost = lambda f: FN_SCALE * FN[{'sincos': 'sc_sin'}.get(f, f)]

            def give(b, j):
                self.call_owner[j] = b
                cone = self.closure([self.libm_calls[j][0][1]], S0)
                load[b] += sum(cost(m) for m in cone if m not in have[b])
                have[b] |= cone
                have[b] |= set(self.libm_calls[j][1].values())
            for j in range(len(self.libm_calls)):
                own = [b for b in range(1, K) if any(nd in self.handed_of[b] for nd in self.libm_calls[j][1].values())]
                if own:
                    b = self.P if self.P in own else own[0]
                    if grp(j) not in fns[b]:
                        load[b] += (LIBM_SCALE if b == self.P else 1.0) * gcost(grp(j))
                        fns[b].add(grp(j))
                    give(b, j)
            groups = collections.defaultdict(list)
            for j in range(len(self.libm_calls)):
                if j not in self.call_owner:
                    groups[grp(j)].append(j)
            helpers = list(range(1, K))
            for f, calls in sorted(groups.items(), key=lambda kv: (-gcost(kv[0]), kv[0])):
                b = min(helpers, key=lambda q: (load[q] + simd_extra(q, load) + (0.01 * load[q] if SIMD_PAIRS else 0.0), q))
                # sincos of the five angles feeds ~300 nodes in front of B1: every helper waits for it right at its start.  Its
                # producer is the wavefront that shares its SIMD with the (mostly parked) actor wavefront -- wave 3: the body
                # runs at the speed of a lone wavefront (r03 sweep 36: 17.97 -> 17.72 us per env step; on waves 1 / 4 / 5: 17.97 - 18.1)
                if f == 'sincos' and 'sincos' not in LIBM_WAVE and K >= 5:
                    b = 3
                if f in LIBM_WAVE and 0 <= LIBM_WAVE[f] < K:
                    b = LIBM_WAVE[f]          # (experiments: CITW_TEAM_LIBM_WAVE=sincos:3,tan:1)
                load[b] += gcost(f)
                for j in calls:
                    give(b, j)
        owner = {}
        cones = {n: self.closure([n], S0) for n in sinks}
        for n in sorted(sinks, key=lambda n: (-sum(cost(m) for m in cones[n]), n)):
            res = []
            for b in range(K):
                add = [m for m in cones[n] if m not in have[b]]
                res.append(load[b] + sum(cost(m) for m in add) + fn_cost(add, b))
            # AFFINITY > 0 leans towards the wave that already holds most of the cone (less recomputation overall)
            b = min(range(K), key=lambda q: (res[q] + simd_extra(q, load) + AFFINITY * (res[q] - load[q]) + (0.01 * res[q] if SIMD_PAIRS else 0.0), q))
            add = [m for m in cones[n] if m not in have[b]]
            load[b] = res[b]
            fn_cost(add, b, True)
            have[b].update(add)
            owner[n] = b
        self.S0, self.A0, self.have, self.owner, self.load_estimate = S0, A0, have, owner, load
        self.pre_sinks = [[n for n in sinks if owner[n] == b] for b in range(K)]
        # ---- glue behind the look-ups: derivative cones; cones that need a later look-up round stay on wave 0
        post = set(n for n in self.order if self.rnd[n] >= 1 and glue(n))
        psinks = [n for n in roots if n in post]
        pcones = {n: self.closure([n], post) for n in psinks}
        phave = [set() for _ in range(K)]
        pins = [set() for _ in range(K)]          # what a wave fetches from LDS behind B1: look-up results, values of other waves
        pload = [0.0] * K
        if self.nrounds > 1:
            pload[self.EW] += POST_ROUND2_COST if (self.chain_round is None or CHAIN_B1 == 'rc') else 40.0
        for q, v in enumerate(POST_BIAS[:K]):
            pload[q] += v
        powner = {}
        fetched = lambda cone: set(c for m in cone for c in build_dag.children(g, m) if c not in post and g.nodes[c][0] not in LEAF)

        def pcost(n, b):
            new = [m for m in pcones[n] if m not in phave[b]]
            ins = [c for c in fetched(new) if c not in pins[b] and not (c in have[b] and g.nodes[c][0] not in LOOKUPS)]
            return sum(cost(m) for m in new) + IMPORT_COST * len(ins), ins
        self.tasks = None
        if POST_TASKS and K > 2:
            self.post = post
            self.plan_post_tasks(psinks, cost, pload)
            phave, powner, pload = self.phave, self.powner, self.post_load
        else:
            for n in sorted(psinks, key=lambda n: (-sum(cost(m) for m in pcones[n]), n)):
                res = [pload[b] + pcost(n, b)[0] for b in range(K)]
                b = self.EW if self.rnd[n] >= 2 else min(range(K), key=lambda q: (res[q] + simd_extra(q, pload, ACTOR_POST) + AFFINITY_POST * (res[q] - pload[q]) + (0.01 * res[q] if SIMD_PAIRS else 0.0), q))
                pins[b].update(pcost(n, b)[1])
                phave[b].update(pcones[n]); pload[b] = res[b]; powner[n] = b
        self.post, self.phave, self.powner, self.post_load = post, phave, powner, pload
        self.post_sinks = [[n for n in psinks if powner[n] == b] for b in range(K)]
        # ---- what crosses between the waves around B1
        pre_val = lambda c: c not in post and glue(c)
        need = [set() for _ in range(K)]
        for b in range(K):
            for m in phave[b]:
                for c in build_dag.children(g, m):
                    if pre_val(c):
                        need[b].add(c)
        for R in self.rounds[1:]:
            for n in R['ins']:
                for c in [n] + [m for m in self.closure_all(n) if m not in post]:
                    if pre_val(c) and (c == n or any(u in post for u in users[c])):
                        need[self.EW].add(c)
        self.exp = [[] for _ in range(K)]
        self.imp = [[] for _ in range(K)]
        self.xslot = {}
        for b in range(K):
            for n in self.order:
                if n in need[b] and n not in have[b]:
                    p = min(q for q in range(K) if n in have[q])
                    if n not in self.xslot:
                        self.xslot[n] = len(self.xslot)
                        self.exp[p].append(n)
                    self.imp[b].append(n)
        assert len(self.xslot) <= 256
        # ---- per-step invariants.  Actuator saturations, the gear / flap logic ... depend on the command vector (and constants)
        # alone: 54 of the 120 nodes wave 0 computes in front of its index search.  They are the same in all six evaluations of
        # an env step, so a wavefront computes them in the first one only (`if (stage == 0)`, a scalar branch) -- provided every
        # consumer ON THAT WAVEFRONT is part of the block: the results leave through LDS slots that nothing else writes (look-up
        # inputs in g_in: the later rounds' inputs move behind round 1's; exchanged values in g_x), registers do not survive.
        self.stage0 = [set() for _ in range(K)]
        if STAGE0 and K > 1 and not self.inv_frontier:
            const_states = [i for i, n in enumerate(self.xdot) if g.nodes[n] == ('cf', 0)]
            sinv = {}
            for n in self.order:
                t = g.nodes[n]
                if t[0] == 'in':
                    sinv[n] = t[1] in ('CMD', 'RO') or (t[1] == 'X' and t[2] in const_states)
                elif t[0] in ('cf', 'ci', 'true', 'false'):
                    sinv[n] = True
                elif t[0] in LOOKUPS or t[0] in FN or t[0] == 'in_i' or n in shared or n in self.libm_slot:
                    sinv[n] = False
                else:
                    sinv[n] = all(sinv[c] for c in build_dag.children(g, n))
            later = set()
            for R in self.rounds[1:]:
                for n in R['ins']:
                    later |= set(self.closure_all(n))
            for b in range(K):
                mine = set(have[b]) | set(phave[b]) | (later if b == self.EW else set())
                if b == 0:
                    mine |= set(self.rounds[0]['ins'])
                blk = set(n for n in have[b] if sinv[n] and g.nodes[n][0] not in LEAF)
                changed = True
                while changed:
                    changed = False
                    for n in list(blk):
                        gated_by_n = [m for (c_, p_), ms in self.gate_nodes.items() if c_ == n for m in ms]      # (a gate's block reads its condition)
                        if any(u in mine and u not in blk for u in users[n] + gated_by_n) or n in rootset or n == self.stop:
                            blk.discard(n); changed = True
                self.stage0[b] = blk
            if any(n in self.stage0[0] for n in self.rounds[0]['ins']):
                off = len(self.rounds[0]['ins'])
                for R in self.rounds[1:]:
                    R['ibase'] = off
                    off += len(R['ins'])
                assert off <= 32
        # ---- precomputed look-up lanes (citation_wave.h: citw_spec_pre / citw_spec_tail): one lane per search and per table of
        # ALL rounds in a merged descriptor row behind the rounds' own rows
        self.spec = None
        if SPEC and K > 1 and self.chain_round is None and len(self.all_rounds) < 3 and not (SPREAD_IN and K > 2):
            R0 = self.rounds[0]
            ns, nt = [(0, n_search(R0))], [(0, len(R0['L2']))]
            for R in self.rounds[1:]:
                ns.append((ns[-1][1], ns[-1][1] + len(R['searches'])))
                nt.append((nt[-1][1], nt[-1][1] + len(R['L2']) + len(R['L1'])))
            if ns[-1][1] <= 64 and nt[-1][1] <= 64:
                self.spec = dict(ns=ns, nt=nt, NS=ns[-1][1], NT=nt[-1][1], tidx=len(self.all_rounds))
        # ---- owners of the outputs
        def out_owner(n):
            if n in post:
                return powner[n]
            if n in owner:
                return owner[n]
            return 0
        self.xdot_owner = [out_owner(n) for n in self.xdot]
        self.dw_owner = {k: out_owner(n) for k, n in self.dw_out.items() if g.nodes[n] != ('in', 'DW', k)}

    # ---- the glue behind the look-ups as a task graph -----------------------------------------------------------------------
    def plan_post_tasks(self, psinks, cost, pload0):
        """The derivative cones behind the first look-up round overlap heavily (p-dot and r-dot share 79 of their 83 nodes, the
        six force / moment derivatives 152 of 281 nodes): one cone per derivative and wavefront computes ~560 nodes for 281
        distinct ones, and the longest cone (134 nodes on ONE wavefront, latency bound) sets the time between the two barriers.
        Here the phase is cut into TASKS -- a task = one published value (or one derivative) with the part of its cone that lies
        above other published values: first at the nodes where the sharing between the derivatives ends, then any task heavier
        than TASK_MAX at the inner node that halves it -- scheduled over the K wavefronts by list scheduling (longest remaining
        path first, earliest finish, TASK_COMM units for a value that crosses wavefronts).  A value crosses through LDS (g_y);
        wave q announces its k-th published value of evaluation SEQ by the release store g_pflag[q] = SEQ * 16 + k, a consumer
        polls (acquire) once before its first read.  Every wavefront runs its tasks in scheduled order, the schedule is a
        topological order of the task graph and all wavefronts of the workgroup are resident: a poll cannot dead-lock.  Arithmetic
        per value is unchanged (same expression trees): results are bit-identical."""
        g, K, post = self.g, self.K, self.post
        users = self.users
        LK = LEAF + LOOKUPS
        inner = lambda m: m in post and g.nodes[m][0] not in LK
        r2 = set(n for n in post if self.rnd[n] >= 2)            # needs a later look-up round: stays on wave 0, never published
        roots = list(psinks)

        def residual(t, cuts):
            out, st = set(), [t]
            while st:
                m = st.pop()
                if m in out or not inner(m):
                    continue
                if m != t and m in cuts:
                    continue
                out.add(m)
                st.extend(build_dag.children(g, m))
            return out
        # 1. where the sharing between the derivative cones ends
        member = collections.defaultdict(set)
        for r in roots:
            for m in self.closure([r], post):
                member[m].add(r)
        cuts = set()
        for m in post:
            if m in r2 or g.nodes[m][0] in LK or len(member[m]) < 2:
                continue
            if any(member[u] != member[m] for u in users[m] if u in post):
                cuts.add(m)
        w = lambda nodes: sum(cost(m) for m in nodes)
        # drop frontier values too light to be worth a hand-over (their consumers recompute them)
        for m in sorted(cuts, key=lambda m: w(residual(m, cuts))):
            if w(residual(m, cuts - {m})) < TASK_MIN and m not in roots:
                cuts.discard(m)
        # 2. split heavy tasks at the inner node whose own cone is closest to half of the task
        for _ in range(200):
            tasks = list(dict.fromkeys(list(cuts) + roots))
            res = {t: residual(t, cuts) for t in tasks}
            heavy = [t for t in tasks if t not in r2 and w(res[t]) > TASK_MAX]
            if not heavy:
                break
            t = max(heavy, key=lambda t: (w(res[t]), t))
            best, bestd = None, None
            for m in res[t]:
                if m == t or m in r2 or g.ty[m] != 'f':
                    continue
                sub = residual(m, cuts) & res[t]
                ws = w(sub)
                if ws < TASK_MIN or w(res[t]) - ws < TASK_MIN:
                    continue
                d = abs(ws - w(res[t]) / 2.0)
                if bestd is None or (d, m) < (bestd, best):
                    best, bestd = m, d
            if best is None:
                cuts_before = len(cuts)
                # cannot be split: leave it
                res[t] = res[t]
                # mark by a sentinel so that it is not tried again
                TASKS_UNSPLIT.add(t)
                if all(h in TASKS_UNSPLIT for h in heavy):
                    break
                continue
            cuts.add(best)
        tasks = list(dict.fromkeys([t for t in self.order if t in cuts] + roots))
        res = {t: residual(t, cuts) for t in tasks}
        deps = {t: sorted(set(c for m in res[t] for c in build_dag.children(g, m) if c in cuts and c != t)) for t in tasks}
        cons = collections.defaultdict(list)
        for t in tasks:
            for d in deps[t]:
                cons[d].append(t)
        tcost = {t: w(res[t]) + 2.0 for t in tasks}
        blevel = {}
        for t in reversed([t for t in self.order if t in set(tasks)]):
            blevel[t] = tcost[t] + max([blevel[u] + TASK_COMM for u in cons[t]], default=0.0)
        # 3. list scheduling
        free = list(pload0)                                   # wave 0 starts behind its later look-up rounds
        wave_of, start, finish = {}, {}, {}
        done = set()
        pending = set(tasks)
        while pending:
            ready = [t for t in pending if all(d in done for d in deps[t])]
            t = max(ready, key=lambda t: (blevel[t], -t))
            best = None
            for b in ([self.EW] if (t in r2 or any(m in r2 for m in res[t])) else range(K)):
                est = max([free[b]] + [finish[d] + (0.0 if wave_of[d] == b else TASK_COMM) for d in deps[t]])
                key = (est + tcost[t], b)
                if best is None or key < best[0]:
                    best = (key, b, est)
            _, b, est = best
            wave_of[t], start[t], finish[t] = b, est, est + tcost[t]
            free[b] = finish[t]
            done.add(t); pending.discard(t)
        self.tasks = tasks
        self.task_res, self.task_deps, self.task_wave, self.task_start = res, deps, wave_of, start
        self.task_cuts = cuts
        self.task_order = [sorted([t for t in tasks if wave_of[t] == b], key=lambda t: (start[t], t)) for b in range(K)]
        # publication index of every value that another wavefront reads
        self.pub = {}
        self.yslot = {}
        for b in range(K):
            k = 0
            for t in self.task_order[b]:
                if any(wave_of[u] != b for u in cons[t]):
                    k += 1
                    self.pub[t] = (b, k)
                    self.yslot[t] = len(self.yslot)
        assert len(self.yslot) <= 32 and all(k <= 15 for _, k in self.pub.values()), (len(self.yslot), self.pub)
        self.phave = [set().union(*[res[t] for t in self.task_order[b]]) if self.task_order[b] else set() for b in range(K)]
        self.powner = {t: wave_of[t] for t in tasks}
        self.post_load = free
        self.task_makespan = max(finish.values()) if finish else 0.0

    # ---- libm phase for an explicit set of nodes (level-1 libm nodes among `needed`)
    # ---- f64 literals as registers ------------------------------------------------------------------------------------
    # An f64 literal whose low dword is not zero costs two 32-bit moves at EVERY use (the build disables the machine LICM: hoisted
    # out of the stage loop as SGPR pairs they spilled), 15 % of the headline kernel's instructions in round 4.  The role of a
    # wavefront is fixed for the episode, so each role gets a register set of its own: slot j of role b holds the j-th most
    # executed literal of eval_w<b>.  The generated text says CITW_K(j, literal); with a register set (HAVE_K, a
    # compile-time fact after inlining) that is KR.k[j] -- a VGPR pair loaded from the role's row of the LDS table before the
    # episode loop: wave-uniform for the compiler (an LDS load at a uniform address), not rematerialisable, no constant-bus slot --
    # and without one the literal itself, so the lane-group kernels (no registers to spare) compile exactly what they did.
    _LIT = None

    def assign_kregs(self, b, text):
        import re, struct
        if TeamGen._LIT is None:
            TeamGen._LIT = re.compile(r'\(-0x1\.[0-9a-f]+p[+-]\d+\)|(?<![\w.])0x1\.[0-9a-f]+p[+-]\d+')
        inline = {0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 4.0, -4.0} | {float(i) for i in range(-16, 65)}

        def val(tok):
            return float.fromhex(tok.strip('()'))

        def two_moves(x):
            return x not in inline and (struct.unpack('<Q', struct.pack('<d', x))[0] & 0xffffffff) != 0
        weight = collections.Counter()
        depth, regions = 0, []          # (depth at entry, weight factor)
        for line in text.split('\n'):
            w = 6.0
            for d0, f in regions:
                w = min(w, f)
            if re.match(r'\s*if \(stage == 0\) \{', line):
                regions.append((depth, 1.0))
            elif 'only the selects on this condition read these' in line:
                regions.append((depth, 0.05))
            for tok in TeamGen._LIT.findall(line):
                if two_moves(val(tok)):
                    weight[val(tok)] += w
                elif KREGS_ONE_MOVE and val(tok) not in inline:
                    weight[val(tok)] += 0.5 * w       # (one 32-bit move: its high dword; they fill what a role leaves of the set -- the set's size is the largest role's)
            depth += line.count('{') - line.count('}')
            while regions and depth <= regions[-1][0]:
                regions.pop()
        # the coefficient blocks of the libm bodies this role runs (citation_libm.h CITW_LK(i, literal)) come first -- they sit at the head of
        # the hand-over chains every wavefront waits for --, then the glue's literals by how often they execute
        blocks = self.libm_kblocks()
        order, base = [], {}
        for fam, fns in (('citw_sincos', ('citw_sincos', 'citw_tan', 'citw_sin', 'citw_cos')), ('citw_pow', ('citw_pow',))):
            if any(re.search(r'\b%s\(' % f, text) for f in fns) and len(order) + len(blocks[fam]) <= KREGS_MAX:
                for f in fns:
                    base[f] = len(order)
                order += blocks[fam]
        nlib = len(order)
        glue = [x for x, _ in sorted(weight.items(), key=lambda kv: (-kv[1], kv[0]))][:max(0, KREGS_MAX - nlib)]
        slot = {x: nlib + j for j, x in enumerate(glue)}
        self.klit[b] = order + glue

        def sub(m):
            x = val(m.group(0))
            return 'CITW_K(%d, %s)' % (slot[x], m.group(0)) if x in slot else m.group(0)
        text = TeamGen._LIT.sub(sub, text)
        # ... and every call of such a body is told where its block starts
        for f, kb in base.items():
            out, pos = [], 0
            for m in re.finditer(r'\b%s\(' % f, text):
                i, depth = m.end(), 1
                while depth:
                    depth += {'(': 1, ')': -1}.get(text[i], 0)
                    i += 1
                out.append(text[pos:i - 1] + ', HAVE_K, KR, %d)' % kb)
                pos = i
            text = ''.join(out) + text[pos:]
        return text

    _KBLOCKS = None

    def libm_kblocks(self):
        """{function: [coefficient, ...]} of serl_amd/csrc/citation_libm.h, slot order (CITW_LK(i, literal))"""
        if TeamGen._KBLOCKS is None:
            import re
            src = open(os.path.join(build_dag.ROOT, 'serl_amd', 'csrc', 'citation_libm.h')).read()
            blocks, cur = {}, None
            for line in src.split('\n'):
                m = re.match(r'CITW_LIBM_FN \w+ (\w+)\(', line)
                if m:
                    cur = m.group(1)
                for i, lit in re.findall(r'CITW_LK\((\d+), (-?0x1\.[0-9a-f]+p[+-]\d+)\)', line):
                    blocks.setdefault(cur, {})[int(i)] = float.fromhex(lit)
            TeamGen._KBLOCKS = {f: [d[i] for i in range(len(d))] for f, d in blocks.items()}
        return TeamGen._KBLOCKS

    # ---- TIMING EXPERIMENT (CITW_TEAM_FMA=1; results change in the last bits): every add / sub one of whose operands is a product computed
    # by the same wavefront becomes an fma -- what would a "fast DAG" specification buy?
    def fuse_fma(self, text):
        import re
        OPND = r'(?:CITW_K\(\d+, \(?-?0x[0-9a-fp.+-]+\)?\)|\(-?0x[0-9a-fp.+-]+\)|-?0x[0-9a-fp.+-]+|[A-Za-z_]\w*)'
        mul = {}
        for m in re.finditer(r'^  const double (v\d+) = (%s) \* (%s);$' % (OPND, OPND), text, re.M):
            mul[m.group(1)] = (m.group(2), m.group(3))
        n = [0]

        def sub(m):
            name, a, op, b = m.group(1), m.group(2), m.group(3), m.group(4)
            if a in mul:
                x, y = mul[a]
                n[0] += 1
                return '  const double %s = __builtin_fma(%s, %s, %s%s);' % (name, x, y, '-' if op == '-' else '', b)
            if b in mul:
                x, y = mul[b]
                n[0] += 1
                return '  const double %s = __builtin_fma(%s%s, %s, %s);' % (name, '-' if op == '-' else '', x, y, a)
            return m.group(0)
        text = re.sub(r'^  const double (v\d+) = (%s) ([+-]) (%s);$' % (OPND, OPND), sub, text, flags=re.M)
        self.n_fused = getattr(self, 'n_fused', 0) + n[0]
        return text

    def libm_plan(self, needed):
        calls = {}
        for n in self.order:
            if n not in needed or n not in self.libm_slot_all:
                continue
            calls.setdefault(self.libm_key[n], {})[self.libm_which[n]] = n
        order_fn = ['sincos', 'tan', 'exp', 'log10', 'log', 'atan', 'pow']
        lst = sorted(calls.items(), key=lambda kv: (order_fn.index(kv[0][0]), kv[0][1], kv[0][2]))
        slot = {}
        for j, (key, outs) in enumerate(lst):
            for which, node in outs.items():
                slot[node] = 2 * j + (0 if which == 'r0' else 1)
        return lst, slot

    def emit_team(self):
        g, V, K = self.g, self.variant, self.K
        self.plan()
        self.libm_key, self.libm_which = {}, {}
        for (key, outs) in self.libm_calls:
            for which, node in outs.items():
                self.libm_key[node] = key
                self.libm_which[node] = which
        self.libm_slot_all = dict(self.libm_slot)
        # shared libm results: (owner wave, slot in its g_m row) of every call result
        self.row_slot, self.calls_of = {}, collections.defaultdict(list)
        for j in sorted(self.call_owner):
            self.calls_of[self.call_owner[j]].append(j)
        for q, calls in self.calls_of.items():
            for jl, j in enumerate(calls):
                for which, node in self.libm_calls[j][1].items():
                    self.row_slot[node] = (q, 2 * jl + (0 if which == 'r0' else 1))
        # A wavefront that makes calls of several functions (sincos + tan) runs their bodies one after the other: the results of
        # the FIRST group are announced by a flag of their own, g_flag[8 + q], as soon as they are stored
        self.flag_of = {}
        for q, calls in self.calls_of.items():
            fns = [self.libm_calls[j][0][0] for j in calls]
            multi = EARLY_FLAG and len(set(fns)) > 1 and (EARLY_FLAG == 1 or (EARLY_FLAG == 2 and q != self.P) or (EARLY_FLAG == 3 and q == self.P))
            for j in calls:
                early = multi and self.libm_calls[j][0][0] == fns[0]
                for node in self.libm_calls[j][1].values():
                    self.flag_of[node] = (8 + q) if early else q
        shared = self.shared_libm
        # does anybody but the owner read a wave's libm results in front of B1?  (then the owner raises its flag)
        # sequence number of the evaluation for the hand-over flags: FSEQ counts the team's env steps (NOT the model clock TICK: the
        # lane groups of a multi-episode team carry episodes with different clocks, and a lane group that takes a new episode from
        # the work queue starts its clock again)
        SEQ = 'FSEQ * 8u + (unsigned)stage + 1u'
        out = []
        P = out.append
        P('/* GENERATED by tools/dag/codegen_team.py from gen/citation_%s.inc -- do not edit.' % V)
        P(' * %d-wavefront evaluation of the %s model.  Glue nodes computed before the first barrier per wave: %s' %
          (K, V, [len(h) for h in self.have]))
        P(' * (the union has %d: the rest is recomputed), behind it: %s; %d values cross through g_x. */' %
          (len(set().union(*self.have)), [len(h) for h in self.phave], len(self.xslot)))
        if self.chain_round is not None:
            RC = self.chain_round
            P('/* engine look-up chain on wave %d: chain round (%d inputs, %d searches, %d 2-D + %d 1-D tables of round 1), then round 2, then the N1 / N2 derivatives; barrier B1 behind %s */'
              % (self.EW, len(RC['ins']), len(RC['searches']), len(RC['L2']), len(RC['L1']), 'the chain round' if CHAIN_B1 == 'rc' else 'round 2'))
        # the team's own descriptor tables (the rounds differ from the one-wave kernels' when the engine chain is split off)
        P('#define CITW_TEAM_TABLES 1')
        P('enum { citw_%s_team_ROUNDS = %d };' % (V, len(self.all_rounds) + (1 if self.spec else 0)))
        for line in self.table_lines('citw_%s_team' % V):
            P(line)
        if self.tasks is not None:
            P('/* glue behind the look-ups: %d tasks, %d values handed over by flags; per wave %s; nodes computed %d for %d distinct */' %
              (len(self.tasks), len(self.yslot), [len(o) for o in self.task_order], sum(len(h) for h in self.phave), len(set().union(*self.phave))))
        P('enum { citw_%s_team_WAVES = %d, citw_%s_team_NX = %d, citw_%s_team_BARRIERS = 2 };   /* workgroup barriers per evaluation and wavefront */' % (V, K, V, len(self.xslot), V))

        def function(b):
            body = []
            B = body.append
            emitted = set()
            self.libm_slot = dict(self.libm_slot_all) if shared else {}
            self.in_override = {}
            waited, after_b1 = set(), [False]
            done_rounds = set()
            if b == 0:
                TM = lambda k: 'CITW_T(%d)' % k
            elif b == 1:
                TM = lambda k: 'CITW_U(%d)' % (10 + k)
            else:
                slots = {2: {0: 15, 1: 16, 2: 17, 3: 18}, 3: {0: 19, 1: 21, 2: 24, 3: 25},       # pre-B1, wait B1, post, wait B2
                         4: {0: 10, 1: 11, 2: 12, 3: 13}, 5: {0: 15, 1: 16, 2: 17, 3: 18}, 6: {0: 19, 1: 21, 2: 24, 3: 25}}.get(b, {})
                mac = 'CITW_W' if b < 4 else 'CITW_V'      # waves 4 .. 6 report in the second profiling build (-DCITW_PROFILE=2), into the slots of waves 1 .. 3
                TM = lambda k: ('%s(%d, %d)' % (mac, b, slots[k])) if k in slots else '((void)0)'

            def emit_gated(m0, allowed):
                """the not yet emitted part of gate (c, pol) that m0 needs, in one conditional block; what it reads from outside first"""
                c, pol = self.gate[m0]
                exk = self.gate_nodes[(c, pol)]
                need, st = set(), [m0]
                while st:
                    m = st.pop()
                    if m in need or m in emitted or m not in exk:
                        continue
                    need.add(m)
                    st.extend(build_dag.children(g, m))
                R = [m for m in self.order if m in need]
                emit_node(c, allowed)
                for r in R:
                    if allowed is not None:
                        assert r in allowed, 'wave %d would compute node %d %s outside its share' % (b, r, g.nodes[r][:1])
                    for ch in build_dag.children(g, r):
                        if ch not in exk and ch not in emitted and g.nodes[ch][0] not in LEAF:
                            emit_node(ch, allowed)
                import re as _re
                body = []
                for r in R:
                    st_ = self.stmt(r)
                    mt = _re.match(r'^  const (double|bool|long long) (\w+) = (.*);$', st_)
                    assert mt, st_
                    B('  %s %s = %s;' % (mt.group(1), mt.group(2), {'double': '0.0', 'bool': 'false', 'long long': '0'}[mt.group(1)]))
                    body.append('    %s = %s;' % (mt.group(2), mt.group(3)))
                    emitted.add(r)
                B('  if (%s%s) {   /* only the selects on this condition read these */' % ('' if pol == 'T' else '!', self.ref(c)))
                for line in body:
                    B(line)
                B('  }')

            def emit_node(n, allowed=None):
                stack = [(n, False)]
                while stack:
                    m, done = stack.pop()
                    if m in emitted:
                        continue
                    if m in self.gate and not done:
                        c_ = self.gate[m][0]
                        # (a gated value that leaves the wave on its own -- its select sits on another wavefront -- is computed
                        # unconditionally unless this wavefront has the condition anyway)
                        if c_ in emitted or allowed is None or c_ in allowed or g.nodes[c_][0] in LEAF:
                            emit_gated(m, allowed)
                            continue
                    if done:
                        emitted.add(m)
                        t = g.nodes[m]
                        if m in shared:
                            q, sl = self.row_slot[m]
                            fl = self.flag_of.get(m, q)
                            if q != b and fl not in waited and q not in waited and not after_b1[0]:
                                B('  const double v%d = citw_flag_wait_load(%d, %s, &g_m[%d][%d]);   /* libm results of wave %d */' % (m, fl, SEQ, q, sl, q))
                                waited.add(fl)
                            else:
                                B('  const double v%d = g_m[%d][%d];' % (m, q, sl))
                            continue
                        if t[0] in LOOKUPS:
                            assert self.outslot[m][0] in done_rounds, 'look-up result used before its round'
                        if t[0] in ('sc_sin', 'sc_cos') and m not in self.libm_slot:
                            s_, c_ = g.memo.get(('sc_sin', t[1])), g.memo.get(('sc_cos', t[1]))
                            B('  double v%d, v%d; citw_sincos(%s, &v%d, &v%d);' % (s_, c_, self.ref(t[1]), s_, c_))
                            emitted.add(s_); emitted.add(c_)
                            continue
                        s = self.stmt(m)
                        if s:
                            B(s)
                        continue
                    if allowed is not None and g.nodes[m][0] not in LEAF and m not in shared:
                        assert m in allowed, 'wave %d would compute node %d %s outside its share' % (b, m, g.nodes[m][:1])
                    stack.append((m, True))
                    if g.nodes[m][0] in LOOKUPS or m in self.libm_slot:
                        continue
                    for c in build_dag.children(g, m):
                        if c not in emitted:
                            stack.append((c, False))

            def libm_phase_shared(only=None, raise_flag=True):
                calls = [j for j in self.calls_of.get(b, []) if j not in made and (only is None or j in only)]
                if not calls:
                    return
                assert len(self.calls_of[b]) <= 16
                B('  /* ---- libm calls this wave makes for the team: one lane per call, results in g_m[%d] */' % b)
                B('  CITW_LIBM_PRIO(1);')
                for j in calls:
                    emit_node(self.libm_calls[j][0][1])
                for j in calls:
                    if j in self.call_guard:
                        emit_node(self.call_guard[j][0])     # the condition under which this call's result is used at all
                lo = self.calls_of[b].index(calls[0])
                assert [self.calls_of[b].index(j) for j in calls] == list(range(lo, lo + len(calls)))
                keys = [self.libm_calls[j][0] for j in calls]
                direct = (len(set(k_[0] for k_ in keys)) == 1 and keys[0][0] != 'pow' and not any(j in self.call_guard for j in calls)
                          and all(g.nodes[k_[1]][0] == 'in' and g.nodes[k_[1]][1] == 'X' for k_ in keys) and len(set(g.nodes[k_[1]][2] for k_ in keys)) == len(keys))
                scalar = (SCALAR_LIBM and not direct and len(calls) <= 2 and all(j in self.call_guard for j in calls)
                          and all(k_[0] in ('pow', 'exp', 'log10', 'log') for k_ in keys) and all(set(self.libm_calls[j][1]) == {'r0'} for j in calls))
                if scalar:
                    # Round 5: one or two calls, each under a wave-uniform guard of its own (the ISA atmosphere: the troposphere's power law OR the
                    # stratosphere's exponential): made by the whole wavefront under the guard's scalar branch, on the argument it holds in registers --
                    # no trip of the argument through LDS to "its" lane and of the result back (two round trips on the chain that hands the air-data
                    # look-up input to wave 0).  Same function, same argument: the same bits; the results still go to g_m for the other wavefronts.
                    for j in calls:
                        (fn, arg, prm) = self.libm_calls[j][0]
                        nd = self.libm_calls[j][1]['r0']
                        gd = self.call_guard[j]
                        call = {'pow': 'citw_pow(%s, %s)' % (self.ref(arg), hexf(prm))}.get(fn, '%s(%s)' % ({'atan': 'citw_atan'}.get(fn, fn), self.ref(arg)))
                        B('  double v%d = 0.0;' % nd)
                        B('  if (%s%s) { v%d = %s; }   /* (wave-uniform: the whole wavefront makes the call) */' % ('' if gd[1] else '!', self.ref(gd[0]), nd, call))
                        emitted.add(nd)
                    B('  if (CITW_LANE0) {')
                    for j in calls:
                        jl = self.calls_of[b].index(j)
                        B('    g_m[%d][%d] = v%d; g_m[%d][%d] = 0.0;' % (b, 2 * jl, self.libm_calls[j][1]['r0'], b, 2 * jl + 1))
                    B('  }')
                    B('  CITW_WAVE_FENCE();')
                    made.update(calls)
                    if raise_flag:
                        B('  citw_flag_raise(%d, %s);' % (b, SEQ))
                        B('  CITW_LIBM_PRIO(0);')
                        B('  __builtin_amdgcn_sched_barrier(0);     /* nothing of what follows may be scheduled in front of the hand-over */')
                    return
                if direct:
                    # every argument IS a state: lane i of XL holds state i (the ODE5 combination this wavefront has just made), so the
                    # lanes of those states make the calls at once -- no trip through g_xs and the argument slots in front of the chain
                    # every other wavefront waits for
                    fn = keys[0][0]
                    B('#if CITW_GROUP_LANES == 64 && CITW_LIBM_DIRECT')
                    B('  {')
                    B('    const int j_ = %s-1;' % ''.join('lane == %d ? %d : ' % (g.nodes[k_[1]][2], i_) for i_, k_ in enumerate(keys)))
                    B('    double r0_ = 0.0, r1_ = 0.0;')
                    B('    if (j_ >= 0) {')
                    B('      %s;' % ('citw_sincos(XL, &r0_, &r1_)' if fn == 'sincos' else 'r0_ = %s(XL)' % {'tan': 'citw_tan', 'atan': 'citw_atan'}.get(fn, fn)))
                    B('      g_m[%d][2 * (%d + j_)] = r0_; g_m[%d][2 * (%d + j_) + 1] = r1_;' % (b, lo, b, lo))
                    B('    }')
                    B('  }')
                    B('#else')
                B('  if (CITW_LANE0) {')
                for j in calls:
                    B('    g_m[%d][%d] = %s;' % (b, 48 + self.calls_of[b].index(j), self.ref(self.libm_calls[j][0][1])))
                B('  }')
                B('  CITW_WAVE_FENCE();')
                B('  {')
                B('    const int l_ = lane - %d;' % lo)
                B('    const double a_ = g_m[%d][48 + (lane >= %d && lane < %d ? lane : %d)];' % (b, lo, lo + len(calls), lo))
                B('    double r0_ = 0.0, r1_ = 0.0;')
                i, first, first_done = 0, True, 0
                # tan lanes ride the sincos pass (citw_tan IS sin / cos of citw_sincos, citation_libm.h): one body for both, the
                # tan lanes divide behind it -- same bits as a call of citw_tan, ~65 instructions less on the wavefront every helper waits for
                fns_ = [self.libm_calls[j][0][0] for j in calls]
                n_sc = sum(1 for f_ in fns_ if f_ == 'sincos')
                n_tan = sum(1 for f_ in fns_ if f_ == 'tan')
                tan_merge = (TAN_MERGE and n_sc > 0 and n_tan > 0 and fns_[:n_sc + n_tan] == ['sincos'] * n_sc + ['tan'] * n_tan
                             and not any(calls[q] in self.call_guard for q in range(n_sc + n_tan)))
                while i < len(calls):
                    (fn, arg, prm) = self.libm_calls[calls[i]][0]
                    k = i
                    while k < len(calls) and self.libm_calls[calls[k]][0][0] == fn and (fn != 'pow' or self.libm_calls[calls[k]][0][2] == prm):
                        k += 1
                    cond = '(l_ >= %d && l_ < %d)' % (i, k) if k - i > 1 else '(l_ == %d)' % i
                    if tan_merge and fn == 'sincos':
                        cond = '(l_ >= %d && l_ < %d)' % (i, k + n_tan)           # ... the tan lanes too
                    if tan_merge and fn == 'tan':
                        B('    if %s { r0_ = r0_ / r1_; r1_ = 0.0; }   /* tan = sin / cos of the pass above */' % cond)
                        first = True
                        i = k
                        continue
                    if k - i == 1 and calls[i] in self.call_guard:
                        gd = self.call_guard[calls[i]]
                        cond = '(l_ == %d && %s%s)' % (i, '' if gd[1] else '!', self.ref(gd[0]))
                    call = {'sincos': 'citw_sincos(a_, &r0_, &r1_)', 'pow': 'r0_ = citw_pow(a_, %s)' % hexf(prm), 'tan': 'r0_ = citw_tan(a_)'}.get(fn, 'r0_ = %s(a_)' % {'atan': 'citw_atan'}.get(fn, fn))
                    early = any(self.flag_of.get(nd, b) == 8 + b for nd in self.libm_calls[calls[i]][1].values())
                    if early and first:
                        # the first function's results leave at once, behind a flag of their own
                        B('    if %s { %s; }' % (cond, call))
                        B('    if (l_ >= %d && l_ < %d) { g_m[%d][2 * lane] = r0_; g_m[%d][2 * lane + 1] = r1_; }   /* (zeros for a call its guard skipped: the slot is never left unwritten) */' % (i, k, b, b))
                        B('    CITW_WAVE_FENCE();')
                        B('    citw_flag_raise(%d, %s);' % (8 + b, SEQ))
                        first_done = k
                        i = k
                        first = True
                        continue
                    B('    %sif %s { %s; }' % ('' if first else 'else ', cond, call))
                    first = False
                    i = k
                B('    if (l_ >= %d && l_ < %d) { g_m[%d][2 * lane] = r0_; g_m[%d][2 * lane + 1] = r1_; }' % (first_done, len(calls), b, b))
                B('  }')
                B('  CITW_WAVE_FENCE();   /* the lanes that made the calls stored; every lane of this wavefront loads the results it needs (citation_wave.h) */')
                if direct:
                    B('#endif')
                made.update(calls)
                if raise_flag:
                    B('  citw_flag_raise(%d, %s);' % (b, SEQ))
                    B('  CITW_LIBM_PRIO(0);')
                    B('  __builtin_amdgcn_sched_barrier(0);     /* nothing of what follows may be scheduled in front of the hand-over */')

            made = set()

            def libm_phase(needed):
                lst, slot = self.libm_plan(needed)
                if not lst:
                    return
                B('  /* ---- libm calls of this wave: one lane per call */')
                for (fn, arg, prm), outs in lst:
                    emit_node(arg)
                self.libm_slot = slot
                B('  if (CITW_LANE0) {')
                for j, ((fn, arg, prm), outs) in enumerate(lst):
                    B('    g_m[wv][%d] = %s;' % (48 + j, self.ref(arg)))
                B('  }')
                B('  CITW_WAVE_FENCE();')
                B('  {')
                B('    const double a_ = g_m[wv][48 + (lane < %d ? lane : 0)];' % len(lst))
                B('    double r0_ = 0.0, r1_ = 0.0;')
                j, first = 0, True
                while j < len(lst):
                    fn, prm = lst[j][0][0], lst[j][0][2]
                    k = j
                    while k < len(lst) and lst[k][0][0] == fn and (fn != 'pow' or lst[k][0][2] == prm):
                        k += 1
                    cond = '(lane >= %d && lane < %d)' % (j, k) if k - j > 1 else '(lane == %d)' % j
                    call = {'sincos': 'citw_sincos(a_, &r0_, &r1_)', 'pow': 'r0_ = citw_pow(a_, %s)' % hexf(prm), 'tan': 'r0_ = citw_tan(a_)'}.get(fn, 'r0_ = %s(a_)' % {'atan': 'citw_atan'}.get(fn, fn))
                    B('    %sif %s { %s; }' % ('' if first else 'else ', cond, call))
                    first = False
                    j = k
                B('    if (lane < %d) { g_m[wv][2 * lane] = r0_; g_m[wv][2 * lane + 1] = r1_; }' % len(lst))
                B('  }')
                B('  CITW_WAVE_FENCE();')
                for (fn, arg, prm), outs in lst:
                    for node in outs.values():
                        emitted.add(node)
                        B(self.stmt(node))

            def import_value(n):
                if g.ty[n] == 'b':
                    B('  const bool b%d = g_x[%d] != 0.0;' % (n, self.xslot[n]))
                else:
                    B('  const double v%d = g_x[%d];' % (n, self.xslot[n]))
                emitted.add(n)

            def lookup_round(r, R, allowed):
                # r == 0: the main round on wave 0; 'c': the chain round, 1 ..: later rounds (on the engine wave)
                B('  /* ---- look-up round %s */' % ('1' if r == 0 else ('chain (engine tables of round 1)' if r == 'c' else str(r + 1))))
                row = 'wv' if b == 0 else '0'           # (text substitution below: wave 0's own row IS row 0 of the shared blackboards)
                mine = [(k, n) for k, n in enumerate(R['ins']) if r != 0 or self.in_owner[n] == 0]
                if r == 0 and b == 0 and self.spec is not None:
                    B('#if CITW_GROUP_LANES == 64 && CITW_SPEC_LOOKUP   /* the hint-dependent half of every look-up lane (all rounds), in front of the input cones it overlaps */')
                    # Round 5: what citw_spec_pre computes depends on the STORED INTERVALS alone (g_sidx: descriptor, interval ends, four corners, the
                    # x-direction quotients -- five dependent LDS round trips and two divisions per evaluation on the wavefront every other one
                    # waits for), and an interval moves in 4 of 2 400 evaluations: with a cache (HAVE_SC: the caller keeps it for the episode) the
                    # lanes are recomputed only behind an evaluation that had to repair an interval -- the same values, not computed again.
                    n1s = n_l1(R) if (self.h1d is not None and SPEC_1D) else 0      # round 1's 1-D tables ride the same lanes (no helper pass behind the verification)
                    pre = 'citw_spec_pre<%d, %d, %d>(%s, S[%d], L[%d][0], lane, L[%d][1])' % (self.spec['NS'], self.spec['NT'], n1s, row, self.spec['tidx'], self.spec['tidx'], R['tidx'])
                    B('  if (HAVE_SC && !SC.valid) { SC.p = %s; SC.valid = true; }' % pre)
                    B('  const CitwSpec sp_ = HAVE_SC ? SC.p : %s;' % pre)
                    B('#endif')
                for k, n in mine:
                    emit_node(n, allowed)
                B('  if (CITW_LANE0) {')
                for k, n in mine:
                    if not (r == 0 and n in self.stage0[b]):
                        B('    g_in[%s][%d] = %s;' % (row, R['ibase'] + k, self.ref(n)))
                B('  }')
                B('  CITW_WAVE_FENCE();   /* (one lane may have stored the inputs alone -- CITW_UNIFORM_STORE --, the search / look-up lanes of this wavefront load them) */')
                if r == 0:
                    B('  %s;' % TM(5))
                    if self.spread:
                        for q in sorted(set(self.in_owner.values()) - {0}):
                            B('  citw_iflag_wait(%d, %s);   /* the look-up inputs wave %d computes are in g_in[0] */' % (q, SEQ, q))
                        B('  %s;' % TM(9))
                    elif self.P is not None and self.own_lk is None:
                        if self.P not in waited:
                            B('  citw_flag_wait(%d, %s);   /* the input(s) wave %d computes are in g_in[0] */' % (self.P, SEQ, self.P))
                            waited.add(self.P)
                        B('  %s;' % TM(9))
                sa = (R['maxn'], n_search(R), R['sbase'])
                off1d = (r == 0 and self.h1d is not None)
                sp = self.spec if (b == 0 and self.spec is not None) else None
                if sp is not None:
                    ri = 0 if r == 0 else r
                    B('#if CITW_GROUP_LANES == 64 && CITW_SPEC_LOOKUP   /* one episode per team: the look-up lanes were precomputed on the stored intervals (citw_spec_pre above) */')
                    if r == 0:
                        B('  %s;' % TM(6))
                    n1s = n_l1(R) if (r == 0 and off1d and SPEC_1D) else 0
                    B('  if (citw_spec_tail<%d, %d, %d, %d, %d>(%s, sp_, g_out%d, lane)) {   /* (rare) an interval moved: the plain passes, which repair the stored indices */'
                      % (sp['ns'][ri] + sp['nt'][ri] + (n1s, row, R['oarr'])))
                    B('    if (HAVE_SC) SC.valid = false;   /* the plain search rewrites g_sidx: the cached lanes are recomputed at the top of the next evaluation */')
                    B('    citw_search<%d, %d, %d>(%s, S[%d], lane);' % (sa + (row, R['tidx'])))
                    if R['L2']:
                        B('    citw_lookup2d<%d>(%s, L[%d][0], g_out%d, lane);' % (len(R['L2']), row, R['tidx'], R['oarr']))
                    if R['L1'] and r != 0:
                        B('    citw_lookup1d<%d>(%s, L[%d][1], g_out%d, lane);' % (len(R['L1']), row, R['tidx'], R['oarr']))
                    if off1d and SPEC_1D:
                        B('    citw_lookup1d<%d>(%s, L[%d][1], g_out%d, lane);' % (n_l1(R), row, R['tidx'], R['oarr']))
                    B('  }')
                    if off1d and SPEC_1D:
                        pass                      # (nobody waits: the 1-D tables rode the lanes above)
                    elif off1d:
                        B('  citw_iflag_raise(0, %s);   /* the interval indices in g_sidx[0] are verified: wave %d runs the 1-D pass */' % (SEQ, self.h1d))
                    elif r == 0 and R['L1']:
                        B('  citw_lookup1d<%d>(%s, L[%d][1], g_out%d, lane);' % (n_l1(R), row, R['tidx'], R['oarr']))
                    if r == 0:
                        B('  %s;' % TM(7))
                        B('  %s;' % TM(8))
                    B('#else')
                if r == 0 and self.l2_helpers and SHARE_SEARCH:
                    ns = n_search(R)
                    B('#if CITW_SEARCH_SHARE(%d) > 1   /* several episodes per team: the search passes are shared with waves 2 (and 4) */' % ns)
                    B('  citw_iflag_raise(7, %s);   /* the look-up inputs are in g_in[0] */' % SEQ)
                    B('  citw_search_part_c<%d, %d, 0, CITW_SEARCH_SHARE(%d), %d>(wv, S[%d], lane, HAVE_PC, PC);' % (R['maxn'], ns, ns, R['sbase'], R['tidx']))
                    B('  citw_iflag_raise(0, %s);' % SEQ)
                    B('  citw_iflag_wait(%d, %s);' % (SEARCH_WAVES[0], SEQ))
                    B('#if CITW_SEARCH_SHARE(%d) > 2' % ns)
                    B('  citw_iflag_wait(%d, %s);' % (SEARCH_WAVES[1], SEQ))
                    B('#endif')
                    B('#else')
                B('  citw_search<%d, %d, %d>(%s, S[%d], lane);' % (R['maxn'], n_search(R), R['sbase'], row, R['tidx']))
                if r == 0 and self.h1d is not None:
                    B('  citw_iflag_raise(0, %s);   /* interval indices are in g_sidx[0]: wave %d runs the 1-D pass beside the 2-D pass */' % (SEQ, self.h1d))
                if r == 0 and self.l2_helpers and SHARE_SEARCH:
                    B('#endif')
                if r == 0:
                    B('  %s;' % TM(6))
                if R['L2'] and r == 0 and self.l2_helpers:
                    B('  citw_lookup2d_part_c<%d, 0, CITW_L2_SHARE>(wv, L[%d][0], g_out%d, lane, HAVE_PC, PC);   /* (lane groups: waves %s take the other passes) */' % (len(R['L2']), R['tidx'], R['oarr'], self.l2_helpers))
                elif R['L2']:
                    B('  citw_lookup2d<%d>(%s, L[%d][0], g_out%d, lane);' % (len(R['L2']), row, R['tidx'], R['oarr']))
                if r == 0:
                    B('  %s;' % TM(7))
                if R['L1'] and not (r == 0 and self.h1d is not None):
                    B('  citw_lookup1d<%d>(%s, L[%d][1], g_out%d, lane);' % (n_l1(R), row, R['tidx'], R['oarr']))
                if r == 0:
                    B('  %s;' % TM(8))
                if sp is not None:
                    B('#endif')

            B('static __device__ CITW_EVAL_INLINE double citw_%s_team_eval_w%d(const int stage, const double T, const unsigned TICK, const unsigned FSEQ, const double XL, const bool HAVE_K = false, const CitwKRegs &KR = citw_no_kregs, const bool HAVE_SC = false, CitwSpecCache &SC = citw_no_spec_cache, const bool HAVE_PC = false, CitwPassCache &PC = citw_no_pass_cache)' % (V, b))
            B('{')
            B('  const CitwSearch (*S)[64] = g_S; const CitwLookup (*L)[2][64] = g_L;')
            B('  const bool major = stage == 0;')
            B('  double STOP = 0.0;')
            B('  const int lane = threadIdx.x & 63;')
            B('  %s;' % ('CITW_T0()' if b == 0 else ('CITW_U0()' if b == 1 else ('CITW_W0(%d)' if b < 4 else 'CITW_V0(%d)') % b)))
            # Derivative-block bank inputs: every wave reads the ones it needs before B1, the banks are rewritten after B1
            mine, st, seen = set(), list(self.pre_sinks[b]) + list(self.post_sinks[b]) + list(self.have[b]), set()
            st += [self.dw_out[k] for k, o in self.dw_owner.items() if o == b]
            if b == 0:
                st += [n for n in self.rounds[0]['ins']]
            if b == self.EW:
                st += [n for R in self.rounds[1:] + ([self.chain_round] if self.chain_round else []) for n in R['ins']]
            imported = set(self.imp[b])
            while st:
                m = st.pop()
                if m in seen or m in imported:
                    continue
                seen.add(m)
                if g.nodes[m][0] == 'in' and g.nodes[m][1] == 'DW':
                    mine.add(m)
                st.extend(build_dag.children(g, m))
            for m in sorted(mine):
                B('  const double dw%d = g_dw[0][%d];' % (g.nodes[m][2], g.nodes[m][2]))
                self.in_override[m] = 'dw%d' % g.nodes[m][2]
            if self.inv_frontier:
                B('  /* ---- per-step invariants (computed by every wave of the team into its own rows) */')
                done_rounds.add(len(self.rounds))
                for n in self.inv_frontier:
                    B(self.inv_load(n))
                    emitted.add(n)
            blk = self.stage0[b]
            if blk:
                for n in [m for m in self.order if m in blk]:
                    # (what the block reads from outside, a gate's condition included -- if this wavefront has the condition at all: otherwise
                    # emit_node computes the gated value unconditionally, like everywhere else)
                    for c in list(build_dag.children(g, n)) + ([self.gate[n][0]] if (n in self.gate and self.gate[n][0] in self.have[b]) else []):
                        if c not in blk and g.nodes[c][0] not in LEAF:
                            emit_node(c, self.have[b])
                B('  if (stage == 0) {   /* ---- what depends on the command vector alone: once per env step; its look-up inputs and exchanged values keep their LDS slots */')
                for n in [m for m in self.order if m in blk]:
                    emit_node(n, self.have[b])
                B('  if (CITW_LANE0) {')
                if b == 0:
                    for k, n in enumerate(self.rounds[0]['ins']):
                        if n in blk:
                            B('    g_in[wv][%d] = %s;' % (self.rounds[0]['ibase'] + k, self.ref(n)))
                for n in self.exp[b]:
                    if n in blk:
                        B('    g_x[%d] = %s;' % (self.xslot[n], ('%s ? 1.0 : 0.0' % self.ref(n)) if g.ty[n] == 'b' else self.ref(n)))
                B('  }')
                B('  }')
                B('  CITW_WAVE_FENCE();')
            if self.spread:
                pass          # (inputs follow the libm phase below)
            elif b == self.P:
                B('  /* ---- first of all: the look-up input(s) wave 0 waits for (libm pow chain of the air data) */')
                if shared:
                    libm_phase_shared(only=set(j for j in self.calls_of.get(b, []) if any(nd in self.handed for nd in self.libm_calls[j][1].values())), raise_flag=False)
                else:
                    libm_phase(self.handed)
                for k, n in enumerate(self.rounds[0]['ins']):
                    if self.in_owner[n] == b:
                        emit_node(n, self.have[b])
                B('  if (CITW_LANE0) {')
                for k, n in enumerate(self.rounds[0]['ins']):
                    if self.in_owner[n] == b:
                        B('    g_in[0][%d] = %s;' % (k, self.ref(n)))
                B('  }')
                B('  citw_flag_raise(%d, %s);' % (b, SEQ))
                B('  CITW_LIBM_PRIO(0);')
                B('  __builtin_amdgcn_sched_barrier(0);     /* nothing of what follows may be scheduled in front of the hand-over */')
                if self.own_lk is not None:
                    R0_ = self.rounds[0]
                    ns_, nso_, n1_, n1o_ = self.own_lk
                    B('  /* ---- ... and the %d 1-D table(s) keyed on it: searched and interpolated here (slots of their own), wave 0 does not wait */' % n1o_)
                    B('  citw_search_range<%d, %d, %d, %d>(CITW_TROW, S[%d], lane);' % (max(sr[1] for sr in R0_['searches'][ns_:]), ns_, nso_, R0_['sbase'], R0_['tidx']))
                    B('  citw_lookup1d_range<%d, %d>(CITW_TROW, L[%d][1], g_out0, lane);' % (n1_, n1o_, R0_['tidx']))
            if shared:
                # (the wavefront of the handed-over chain raises ITS flag once, behind the look-up input: calls of another function
                # made later would hide behind a flag that is already up -- r03 sweep 40, `tan` forced onto it: a stale result)
                assert not (b == self.P and not self.spread and any(j not in made for j in self.calls_of.get(b, []))), \
                    'wave %d hands its look-up input over and cannot make further libm calls for the team' % b
                libm_phase_shared()
            else:
                libm_phase(self.have[b])
            if self.spread and b != 0 and any(o == b for o in self.in_owner.values()):
                B('  /* ---- the round-1 look-up inputs this wave computes for wave 0 (their cones), handed over by flag */')
                for k, n in enumerate(self.rounds[0]['ins']):
                    if self.in_owner[n] == b:
                        emit_node(n, self.have[b])
                B('  if (CITW_LANE0) {')
                for k, n in enumerate(self.rounds[0]['ins']):
                    if self.in_owner[n] == b:
                        B('    g_in[0][%d] = %s;' % (k, self.ref(n)))
                B('  }')
                B('  citw_iflag_raise(%d, %s);' % (b, SEQ))
                B('  __builtin_amdgcn_sched_barrier(0);     /* nothing of what follows may be scheduled in front of the hand-over */')
            B('  %s;' % TM(4))
            if b == 0:
                lookup_round(0, self.rounds[0], None)
                done_rounds.add(0)
            if self.chain_round is not None and b == self.EW:
                # the engine chain: its tables of round 1 now, on this wavefront's own (LDS operations of ONE wavefront complete in
                # order, so the later rounds may read these results without waiting for B1)
                lookup_round('c', self.chain_round, None)
                chain_done = True
                if CHAIN_B1 == 'r1':
                    done_rounds.add(0)
                    for r in range(1, self.nrounds):
                        lookup_round(r, self.rounds[r], None)
                        done_rounds.add(r)
                    done_rounds.discard(0)
            def emit_search_share(bb):
                kq = SEARCH_WAVES.index(bb) + 1
                nsq = n_search(self.rounds[0])
                B('#if CITW_SEARCH_SHARE(%d) > %d   /* several episodes per team: search pass %d, beside wave 0 */' % (nsq, kq, kq))
                B('  citw_iflag_wait(7, %s);' % SEQ)
                B('  citw_search_part_c<%d, %d, %d, CITW_SEARCH_SHARE(%d), %d>(0, S[0], lane, HAVE_PC, PC);' % (self.rounds[0]['maxn'], nsq, kq, nsq, self.rounds[0]['sbase']))
                B('  citw_iflag_raise(%d, %s);' % (bb, SEQ))
                B('#endif')
            B('  /* ---- share of this wave in the look-up independent glue */')
            foreign = lambda n: any(m in shared and self.row_slot[m][0] != b for m in self.closure([n], self.S0))
            order = sorted(self.pre_sinks[b], key=lambda n: (foreign(n), self.pre_sinks[b].index(n))) if shared else self.pre_sinks[b]
            search_at = int(len(order) * SEARCH_AT) if (b in SEARCH_WAVES and self.l2_helpers and SHARE_SEARCH and SEARCH_AT < 1.0) else -1
            if shared and TWO_PASS and b != 0:
                # first everything of this wave's share that needs no libm result of ANOTHER wavefront (node by node, not sink by
                # sink: the cones of the later sinks hold plenty of it), then the rest behind the first flag wait
                dep = {}
                for m in self.order:
                    if m in self.have[b] or m in shared:
                        dep[m] = (m in shared and self.row_slot[m][0] != b) or any(dep.get(c, False) for c in build_dag.children(g, m))
                early = [m for m in self.order if m in self.have[b] and not dep[m] and m not in emitted and m not in self.gate
                         and m not in shared and g.nodes[m][0] not in LEAF + LOOKUPS
                         and any(dep.get(u, False) for u in self.users[m] if u in self.have[b])]
                for m in early:       # (only the frontier towards the libm-dependent part: the rest comes with its sinks above)
                    emit_node(m, self.have[b])
            for kk, n in enumerate(order):
                if kk == search_at:
                    emit_search_share(b)
                emit_node(n, self.have[b])
            for n in self.exp[b]:
                emit_node(n, self.have[b])
            emit_node(self.stop)
            B('  STOP = %s;' % self.ref(self.stop))
            # keep the sinks on this side of the barrier (it is no scheduling barrier for plain arithmetic)
            for n in self.pre_sinks[b]:
                if g.ty[n] == 'f' and g.nodes[n][0] not in LEAF and n not in self.stage0[b]:
                    B('  asm volatile("" :: "v"(%s));' % self.ref(n))
            pre_x = [i for i, n in enumerate(self.xdot) if self.xdot_owner[i] == b and n not in self.post]
            for i in pre_x:
                emit_node(self.xdot[i], self.have[b] if b else None)
            if self.exp[b]:
                B('  if (CITW_LANE0) {')
                for n in self.exp[b]:
                    if n not in self.stage0[b]:
                        B('    g_x[%d] = %s;' % (self.xslot[n], ('%s ? 1.0 : 0.0' % self.ref(n)) if g.ty[n] == 'b' else self.ref(n)))
                B('  }')
            # (derivatives that need no look-up are ready here, but they are STORED behind B1: g_f[0][0] of this env step is row 0 of
            # the stage derivatives the slowest wavefront may still be combining for the previous step's last stage -- nothing but
            # timing kept a store in front of B1 behind that read; found by the jitter build, profiles/r04_experiments.md)
            ns0 = n_search(self.rounds[0])
            R0 = self.rounds[0]

            def wait_searches(me):
                """every wave that reads g_sidx waits for the waves that filled it"""
                B('  citw_iflag_wait(0, %s);' % SEQ)
                if self.l2_helpers and SHARE_SEARCH:
                    for kq, wq in enumerate(SEARCH_WAVES[:2]):
                        if me != wq:
                            B('#if CITW_SEARCH_SHARE(%d) > %d' % (ns0, kq + 1))
                            B('  citw_iflag_wait(%d, %s);' % (wq, SEQ))
                            B('#endif')
            if b in SEARCH_WAVES[:2] and self.l2_helpers and SHARE_SEARCH and SEARCH_AT >= 1.0:
                emit_search_share(b)
            if b in self.l2_helpers:
                k = self.l2_helpers.index(b) + 1
                B('#if CITW_L2_SHARE > %d   /* several episodes per team: pass %d (of every CITW_L2_SHARE) of round 1\'s 2-D interpolation, beside wave 0 */' % (k, k))
                wait_searches(b)
                B('  citw_lookup2d_part_c<%d, %d, CITW_L2_SHARE>(0, L[0][0], g_out0, lane, HAVE_PC, PC);' % (len(self.rounds[0]['L2']), k))
                B('#endif')
            if b == SHARE_1D_WAVE and self.h1d is not None and self.l2_helpers and SHARE_1D:
                n1 = n_l1(self.rounds[0])
                B('#if CITW_L1_SHARE(%d) > 1   /* 16 lanes per episode: the second pass of the 1-D interpolation, beside wave 1 */' % n1)
                wait_searches(b)
                B('  citw_lookup1d_part_c<%d, 1, 2>(0, L[0][1], g_out0, lane, HAVE_PC, PC);' % n1)
                B('#endif')
            if b != 0 and b == self.h1d:
                if SPEC_1D and self.spec is not None:
                    B('#if !(CITW_GROUP_LANES == 64 && CITW_SPEC_LOOKUP)   /* (one episode per team: the 1-D tables ride wave 0\'s precomputed lanes) */')
                B('  /* ---- the 1-D interpolation pass of round 1, taken over from wave 0 */')
                R0_ = self.rounds[0]
                wait_searches(b)
                if self.l2_helpers and SHARE_1D:
                    n1 = n_l1(self.rounds[0])
                    B('  citw_lookup1d_part_c<%d, 0, CITW_L1_SHARE(%d)>(0, L[0][1], g_out0, lane, HAVE_PC, PC);   /* (16 lanes per episode: wave 3 takes the second pass) */' % (n1, n1))
                else:
                    B('  citw_lookup1d<%d>(0, L[0][1], g_out0, lane);' % n_l1(self.rounds[0]))
                if SPEC_1D and self.spec is not None:
                    B('#endif')
            B('  %s;' % TM(0))
            B('  CITW_TEAM_BARRIER1();   /* B1: look-up results (g_out0) and exchanged values (g_x) are visible to every wave */')
            after_b1[0] = True
            B('  %s;' % TM(1))
            done_rounds.add(0)
            for n in self.imp[b]:
                import_value(n)
            mine_post = self.phave[b] | (set(n for R in self.rounds[1:] for n in R['ins']) if b == self.EW else set())
            for e in self.rounds[0]['L2'] + self.rounds[0]['L1'] + (self.chain_round['L2'] + self.chain_round['L1'] if self.chain_round else []):
                if e['node'] not in emitted and any(u in mine_post for u in self.users[e['node']]):
                    emitted.add(e['node'])
                    B(self.stmt(e['node']))
            if b == self.EW:
                for r in range(1, self.nrounds):
                    if not (self.chain_round is not None and CHAIN_B1 == 'r1'):
                        lookup_round(r, self.rounds[r], None)
                    done_rounds.add(r)
                    for e in self.rounds[r]['L2'] + self.rounds[r]['L1']:
                        if e['node'] not in emitted:
                            emitted.add(e['node'])
                            B(self.stmt(e['node']))
            if self.tasks is not None and self.task_order[b]:
                B('  /* ---- tasks of this wave in the glue behind the look-ups (values of other waves: g_y, announced by g_pflag) */')
                pwaited = {}
                for t in self.task_order[b]:
                    for d in self.task_deps[t]:
                        if d in emitted:
                            continue
                        assert self.task_wave[d] != b and d in self.pub, (t, d)
                        q, k = self.pub[d]
                        src = 'g_y[%d]' % self.yslot[d]
                        if pwaited.get(q, 0) < k:
                            # flag and value in ONE poll (the value's load is issued behind the flag's: one LDS round trip on a hit)
                            src = 'citw_pflag_wait_load(%d, (%s) * 16u + %du, &g_y[%d])' % (q, SEQ, k, self.yslot[d])
                            pwaited[q] = k
                        if g.ty[d] == 'b':
                            B('  const bool b%d = %s != 0.0;' % (d, src))
                        else:
                            B('  const double v%d = %s;' % (d, src))
                        emitted.add(d)
                    emit_node(t)
                    if t in self.pub:
                        B('  if (CITW_LANE0) g_y[%d] = %s;' % (self.yslot[t], ('%s ? 1.0 : 0.0' % self.ref(t)) if g.ty[t] == 'b' else self.ref(t)))
                        B('  citw_pflag_raise(%d, (%s) * 16u + %du);' % (b, SEQ, self.pub[t][1]))
            elif self.post_sinks[b]:
                B('  /* ---- share of this wave in the glue behind the look-ups */')
                for n in self.post_sinks[b]:
                    emit_node(n)
            post_x = [i for i, n in enumerate(self.xdot) if self.xdot_owner[i] == b and n in self.post]
            if post_x or pre_x:
                B('  if (CITW_LANE0) {')
                for i in pre_x + post_x:
                    B('    g_f[0][stage][%d] = %s;' % (i, self.ref(self.xdot[i])))
                B('  }')
            dws = [k for k, o in sorted(self.dw_owner.items()) if o == b]
            if dws:
                for k in dws:
                    emit_node(self.dw_out[k])
                B('  if (major && CITW_LANE0) {')
                for k in dws:
                    B('    g_dw[0][%d] = %s;' % (k, self.ref(self.dw_out[k])))
                B('  }')
            B('  %s;' % TM(2))
            B('  CITW_TEAM_BARRIER2();   /* B2: all derivatives of this stage are in g_f */')
            B('  %s;' % TM(3))
            B('  return STOP;')
            B('}')
            self.in_override = {}
            text = '\n'.join(body)
            if PRELOAD:
                # states, commands and table constants as locals loaded once at the top (one batch of LDS loads, one wait): an inline
                # g_xs[..][k] is loaded again behind every flag poll / barrier (memory fences) -- up to ten exposed LDS round trips
                # per wave and evaluation.  The rows are not written during an evaluation (the ODE5 combination follows it).
                import re as _re
                head, rest = text.split('\n', 2)[0:2], text.split('\n', 2)[2]
                decl = []
                for arr, pre in (('g_xs', 'xs_'), ('g_cmd', 'cm_')):
                    ks = sorted({int(k) for k in _re.findall(r'\b%s\[wv\]\[(\d+)\]' % arr, rest)})
                    if arr == 'g_xs' and ks:
                        # one episode per team: lane i of XL holds state i (the ODE5 combination every wavefront has just made):
                        # v_readlane instead of the store -> load round trip through g_xs at the top of every evaluation
                        decl.append('#if CITW_GROUP_LANES == 64 && CITW_STATE_BCAST')
                        for k in ks:
                            decl.append('  const double %s%d = citw_bcast(XL, %d);' % (pre, k, k))
                        decl.append('#else')
                    for k in ks:
                        decl.append('  const double %s%d = %s[wv][%d];' % (pre, k, arr, k))
                    if arr == 'g_xs' and ks:
                        decl.append('#endif')
                    rest = _re.sub(r'\b%s\[wv\]\[(\d+)\]' % arr, lambda m: '%s%s' % (pre, m.group(1)), rest)
                ks = sorted({int(k) for k in _re.findall(r'\bg_ro\[(\d+)\]', rest)})
                for k in ks:
                    decl.append('  const double ro_%d = g_ro[%d];' % (k, k))
                rest = _re.sub(r'\bg_ro\[(\d+)\]', lambda m: 'ro_%s' % m.group(1), rest)
                text = '\n'.join(head + decl + [rest])
            # shared blackboards live in row 0; each wave has its own libm / look-up input rows
            text = text.replace('g_in[wv]', 'g_in[%d]' % b).replace('g_m[wv]', 'g_m[%d]' % b)
            text = text.replace('g_inv[wv]', 'g_inv[%d]' % b).replace('g_out%d[wv]' % len(self.rounds), 'g_out%d[%d]' % (len(self.rounds), b))
            text = text.replace('[wv]', '[0]').replace('(wv, ', '(%d, ' % b)
            # rows as macros (citation_wave.h): CITW_TROW = the episode's row of the shared blackboards (0; the lane group in the
            # two-episodes-per-team kernels), CITW_MROW(q) = wave q's libm-result row, CITW_XOFF = base of the episode's g_x slots
            import re
            text = re.sub(r'\b(g_xs|g_out0|g_out1|g_in|g_dw|g_cmd|g_f)\[0\]', r'\1[CITW_TROW]', text)
            text = re.sub(r'\bg_m\[(\d+)\]', r'g_m[CITW_MROW(\1)]', text)
            text = re.sub(r'\bg_x\[(\d+)\]', r'g_x[CITW_XOFF + \1]', text)
            text = re.sub(r'\bg_y\[(\d+)\]', r'g_y[CITW_YOFF + \1]', text)
            text = re.sub(r'\b(citw_spec_pre<[^>]*>|citw_spec_tail<[^>]*>|citw_search<[^>]*>|citw_search_part<[^>]*>|citw_lookup2d<\d+>|citw_lookup2d_part<[^>]*>|citw_lookup1d<\d+>|citw_lookup1d_part<[^>]*>|citw_search_part_c<[^>]*>|citw_lookup2d_part_c<[^>]*>|citw_lookup1d_part_c<[^>]*>)\(0, ', r'\1(CITW_TROW, ', text)
            text = text.replace('const int lane = threadIdx.x & 63;', 'const int lane = CITW_LANE;')
            if FMA:
                text = self.fuse_fma(text)
            if KREGS:
                text = self.assign_kregs(b, text)
            return text

        self.klit = {}
        for b in range(K - 1, -1, -1):
            P(function(b))
        P('/* wave-uniform dispatch: each wavefront of the team executes exactly one of the parts and its two barriers */')
        P('static __device__ __forceinline__ double citw_%s_team_eval(const int wave, const int stage, const double T, const unsigned TICK, const unsigned FSEQ, const double XL, const bool HAVE_K = false, const CitwKRegs &KR = citw_no_kregs, const bool HAVE_SC = false, CitwSpecCache &SC = citw_no_spec_cache, const bool HAVE_PC = false, CitwPassCache &PC = citw_no_pass_cache)' % V)
        P('{')
        for b in range(K - 1):
            P('  if (wave == %d) return citw_%s_team_eval_w%d(stage, T, TICK, FSEQ, XL, HAVE_K, KR, HAVE_SC, SC, HAVE_PC, PC);' % (b, V, b))
        P('  return citw_%s_team_eval_w%d(stage, T, TICK, FSEQ, XL, HAVE_K, KR, HAVE_SC, SC, HAVE_PC, PC);' % (V, K - 1))
        P('}')
        if KREGS:
            nk = max(1, max(len(v) for v in self.klit.values()))
            P('/* the f64 literals behind CITW_K(slot, literal): row b = the register set of role b (citation_wave.h CitwKRegs; staged into LDS by the kernels that use it) */')
            P('enum { citw_%s_team_NKLIT = %d };' % (V, nk))
            P('static __device__ const double citw_%s_team_klit[%d][%d] = {' % (V, K, nk))
            for b in range(K):
                row = [symex.f2b(x) for x in self.klit[b]]
                P('  {%s},' % ', '.join([hexf(v) for v in row] + ['0.0'] * (nk - len(row))))
            P('};')
        ks = sorted(self.kslot.items(), key=lambda kv: kv[1])
        if ks:
            P('#define CITW_TEAM_HAS_K 1')
            P('enum { citw_%s_team_NK = %d };' % (V, len(ks)))
            P('static __device__ const double citw_%s_team_k[%d] = {%s};' % (V, len(ks), ', '.join(hexf(v) for v, _ in ks)))
        self.libm_slot = dict(self.libm_slot_all)
        return '\n'.join(out) + '\n'


def main():
    variants = [a for a in sys.argv[1:] if not a.startswith('--')] or ['nominal']
    waves = next((int(a.split('=', 1)[1]) for a in sys.argv[1:] if a.startswith('--waves=')), None)     # --waves=6 --suffix=6: the six-wavefront team of the streamed-actor kernels (gen/citation_<v>_team6.inc)
    for v in variants:
        gen = TeamGen(v, waves=waves, hoist='--hoist-invariants' in sys.argv, lds_consts=int(os.environ.get('CITW_TEAM_LDS_CONSTS', 0)))
        text = gen.emit_team()
        suffix = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--suffix=')), '')     # experiments: --suffix=_exp7
        if '--lane-groups' in sys.argv and not suffix:
            suffix = 'g'
        path = os.path.join(build_dag.ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s_team%s.inc' % (v, suffix))
        open(path, 'w').write(text)
        print('%s: %d lines; %d waves; pre-barrier glue %s (load estimate %s), post %s (%s), exchanged %d, xdot owners %s'
              % (path, text.count('\n'), gen.K, [len(h) for h in gen.have], gen.load_estimate, [len(h) for h in gen.phave],
                 gen.post_load, len(gen.xslot), gen.xdot_owner))


if __name__ == '__main__':
    main()
