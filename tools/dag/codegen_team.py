#!/usr/bin/env python3
"""Generate the TWO-WAVEFRONT ("team") model evaluation: serl_amd/csrc/gen/citation_<variant>_team.inc.

With one wavefront per SIMD every instruction -- VALU, SALU, LDS, waitcnt -- costs a 4-cycle issue slot, so one model
evaluation (~4 500 instructions) is issue-bound on a single wavefront.  When there are fewer episodes than CUs, a
second wavefront of the same workgroup (another SIMD of the CU) takes half of the work:

  wave 0 (main)    libm calls + glue that the round-1 look-up inputs need  ->  index search, 2-D and 1-D
                   interpolation passes  ->  [barrier B1]  ->  glue that depends on look-up results, later rounds,
                   derivatives  ->  [barrier B2]
  wave 1 (helper)  its own libm calls (sincos of the attitude angles ...) + all glue that does NOT depend on any
                   look-up and is not needed for the look-up inputs (rotation matrices, gravity, engine, kinematic
                   equations ...), exports what wave 0 needs to LDS (g_x)  ->  [B1]  ->  Derivative-block banks
                   ->  [B2]

Light glue shared by both halves is computed twice rather than exchanged.  Both functions execute exactly two
workgroup barriers per evaluation.  The arithmetic (operation order per value) is that of the single-wave code.

Usage: python tools/dag/codegen_team.py [variant ...]
"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_dag, codegen
from codegen import LOOKUPS, hexf

LEAF = ('cf', 'ci', 'in', 'in_i', 'true', 'false')
POST_ROUND2_COST = 300      # instruction estimate of the later look-up rounds wave 0 runs after the first barrier
LOOKUP_PHASES = int(os.environ.get('CITW_TEAM_LOOKUP_COST', 1400))   # instruction estimate of search + 2-D + 1-D passes
LIBM = ('sc_sin', 'sc_cos', 'sin', 'cos', 'tan', 'exp', 'log10', 'log', 'atan', 'pow')


class TeamGen(codegen.Gen):
    split_post = True

    def closure(self, sinks, within):
        out, st = set(), list(sinks)
        while st:
            m = st.pop()
            if m in out or m not in within:
                continue
            out.add(m)
            st.extend(build_dag.children(self.g, m))
        return out

    def plan(self):
        g = self.g
        users = collections.defaultdict(list)
        for n in self.order:
            for c in build_dag.children(g, n):
                users[c].append(n)
        S0 = set(n for n in self.order if self.rnd[n] == 0 and g.nodes[n][0] not in LEAF + LOOKUPS)
        r1_inputs = [n for n in self.rounds[0]['ins']]
        A0 = self.closure([n for n in r1_inputs if n in S0], S0)
        roots = list(self.xdot) + list(self.dw_out.values())
        rootset = set(roots)
        later_use = lambda n: any((u not in S0) for u in users[n])
        sinks = [n for n in self.order if n in S0 and n not in A0 and (later_use(n) or n in rootset)]
        # ---- balance: every sink (with its cone inside S0, shared ancestors recomputed) goes to the wave that ends up
        # with the smaller load.  Loads are instruction estimates: 2 per glue cost unit (constants, moves, waits ride
        # along), the look-up phases of wave 0 ~ 630, a libm function the first time a wave needs it.
        CW = dict(div=11, sqrt=15, sel=3, unord=2, table3=120)
        FN = dict(sc_sin=100, sc_cos=100, sin=100, cos=100, tan=120, exp=40, log10=60, log=60, atan=80, pow=250)
        cost = lambda n: 0 if g.nodes[n][0] in FN else 2 * CW.get(g.nodes[n][0], 1)
        fns = [set(), set()]
        def fn_cost(nodes, b, commit=False):
            c = 0
            for m in nodes:
                f = g.nodes[m][0]
                if f in FN:
                    f2 = {'sc_sin': 'sincos', 'sc_cos': 'sincos', 'sin': 'sincos', 'cos': 'sincos'}.get(f, f)
                    if f2 not in fns[b]:
                        c += FN[f]
                        if commit:
                            fns[b].add(f2)
            return c
        have = [set(A0), set()]
        load = [sum(cost(m) for m in A0) + fn_cost(A0, 0, True) + LOOKUP_PHASES, 0]
        owner = {}
        cones = {n: self.closure([n], S0) for n in sinks}
        for n in sorted(sinks, key=lambda n: -sum(cost(m) for m in cones[n])):
            res = []
            for b in (0, 1):
                add = [m for m in cones[n] if m not in have[b]]
                res.append(load[b] + sum(cost(m) for m in add) + fn_cost(add, b))
            b = 0 if res[0] < res[1] else 1
            add = [m for m in cones[n] if m not in have[b]]
            load[b] = res[b]
            fn_cost(add, b, True)
            have[b].update(add)
            owner[n] = b
        self.load_estimate = load
        A0x = have[0]                      # everything wave 0 computes before the first barrier
        A1 = have[1]
        sinks0 = [n for n in sinks if owner[n] == 0]
        sinks1 = [n for n in sinks if owner[n] == 1]
        exports = [n for n in sinks1 if later_use(n) and n not in have[0]]
        self.users, self.S0, self.A0, self.A0x, self.A1, self.exports, self.sinks0 = users, S0, A0, A0x, A1, exports, sinks0
        self.xslot = {n: k for k, n in enumerate(exports)}
        assert len(exports) <= 256
        # owners of the outputs
        self.own1_xdot = [i for i, n in enumerate(self.xdot) if owner.get(n) == 1]
        self.own0_xdot = [i for i in range(19) if i not in self.own1_xdot]
        self.own1_dw = [k for k, n in sorted(self.dw_out.items()) if (owner.get(n) == 1 or g.nodes[n][0] in LEAF) and g.nodes[n] != ('in', 'DW', k)]
        self.own0_dw = [k for k, n in sorted(self.dw_out.items()) if k not in self.own1_dw and g.nodes[n] != ('in', 'DW', k)]

        # ---- the glue AFTER the look-ups: derivative cones are shared out the same way.  Cones that need a later
        # look-up round stay on wave 0 (it runs those rounds); what a wave needs from the other's pre-barrier values
        # is exchanged through g_x before the first barrier.
        post = set(n for n in self.order if self.rnd[n] >= 1 and g.nodes[n][0] not in LEAF + LOOKUPS)
        psinks = [n for n in dict.fromkeys(list(self.xdot) + list(self.dw_out.values())) if n in post]
        pcones = {n: self.closure([n], post) for n in psinks}
        phave = [set(), set()]
        pload = [POST_ROUND2_COST if self.nrounds > 1 else 0, 0]
        self.powner = {}
        for n in sorted(psinks, key=lambda n: -sum(cost(m) for m in pcones[n])):
            forced0 = self.rnd[n] >= 2 or not self.split_post
            res = [pload[b] + sum(cost(m) for m in pcones[n] if m not in phave[b]) for b in (0, 1)]
            b = 0 if (forced0 or res[0] <= res[1]) else 1
            phave[b].update(pcones[n]); pload[b] = res[b]; self.powner[n] = b
        self.post, self.phave, self.post_load = post, phave, pload
        pre_ok = lambda c: c not in post and g.nodes[c][0] not in LEAF + LOOKUPS
        need = [set(), set()]
        for b in (0, 1):
            for m in phave[b]:
                for c in build_dag.children(g, m):
                    if pre_ok(c):
                        need[b].add(c)
        for R in self.rounds[1:]:                     # inputs of later look-up rounds are computed by wave 0
            for c in R['ins']:
                if pre_ok(c):
                    need[0].add(c)
        exp10 = [n for n in self.order if n in need[0] and n not in A0x]       # wave 1 -> wave 0
        exp01 = [n for n in self.order if n in need[1] and n not in A1]        # wave 0 -> wave 1
        assert all(n in A1 for n in exp10) and all(n in A0x for n in exp01)
        self.exports, self.exports01 = exp10, exp01
        self.xslot = {n: k for k, n in enumerate(exp10)}
        self.xslot01 = {n: 128 + k for k, n in enumerate(exp01)}
        assert len(exp10) <= 128 and len(exp01) <= 128
        for i, n in enumerate(self.xdot):
            if n in post:
                if self.powner[n] == 1:
                    self.own1_xdot.append(i); self.own0_xdot.remove(i)
        for k, n in sorted(self.dw_out.items()):
            if n in post and self.powner[n] == 1:
                self.own1_dw.append(k); self.own0_dw.remove(k)

    # ---- libm phase for an explicit set of nodes (level-1 libm nodes among `needed`)
    def libm_plan(self, needed):
        g = self.g
        calls = {}
        for n in self.order:
            if n not in needed or n not in self.libm_slot_all:
                continue
            key = self.libm_key[n]
            calls.setdefault(key, {})[self.libm_which[n]] = n
        order_fn = ['sincos', 'tan', 'exp', 'log10', 'log', 'atan', 'pow']
        lst = sorted(calls.items(), key=lambda kv: (order_fn.index(kv[0][0]), kv[0][1], kv[0][2]))
        slot = {}
        for j, (key, outs) in enumerate(lst):
            for which, node in outs.items():
                slot[node] = 2 * j + (0 if which == 'r0' else 1)
        return lst, slot

    def emit_team(self):
        g = self.g
        V = self.variant
        self.plan()
        # libm bookkeeping from the base class: node -> (key, which)
        self.libm_key, self.libm_which = {}, {}
        for (key, outs) in self.libm_calls:
            for which, node in outs.items():
                self.libm_key[node] = key
                self.libm_which[node] = which
        self.libm_slot_all = dict(self.libm_slot)
        out = []
        P = out.append
        P('/* GENERATED by tools/dag/codegen_team.py from gen/citation_%s.inc -- do not edit.' % V)
        P(' * Two-wavefront evaluation of the %s model: wave 0 = look-up path + dependent glue (%d glue nodes before the' % (V, len(self.A0x)))
        P(' * first barrier), wave 1 = look-up independent glue (%d nodes, %d of them recomputed on both), %d values exported. */'
          % (len(self.A1), len(self.A0x & self.A1), len(self.exports)))
        P('enum { citw_%s_team_NX = %d };' % (V, len(self.exports)))

        def function(which):
            name = 'citw_%s_team_eval_w%d' % (V, which)
            body = []
            B = body.append
            emitted = set()
            self.libm_slot = {}
            done_rounds = set()

            def emit_node(n, allowed=None):
                stack = [(n, False)]
                while stack:
                    m, done = stack.pop()
                    if m in emitted:
                        continue
                    if done:
                        emitted.add(m)
                        t = g.nodes[m]
                        if t[0] in LOOKUPS:
                            assert which == 0 and self.outslot[m][0] in done_rounds, 'look-up result used before its round'
                        if t[0] in ('sc_sin', 'sc_cos') and m not in self.libm_slot:
                            s_, c_ = g.memo.get(('sc_sin', t[1])), g.memo.get(('sc_cos', t[1]))
                            B('  double v%d, v%d; sincos(%s, &v%d, &v%d);' % (s_, c_, self.ref(t[1]), s_, c_))
                            emitted.add(s_); emitted.add(c_)
                            continue
                        s = self.stmt(m)
                        if s:
                            B(s)
                        continue
                    if allowed is not None and g.nodes[m][0] not in LEAF:
                        assert m in allowed, 'wave %d would compute node %d %s outside its share' % (which, m, g.nodes[m][:1])
                    stack.append((m, True))
                    if g.nodes[m][0] in LOOKUPS or m in self.libm_slot:
                        continue
                    for c in build_dag.children(g, m):
                        if c not in emitted:
                            stack.append((c, False))

            def libm_phase(needed):
                lst, slot = self.libm_plan(needed)
                if not lst:
                    return
                B('  /* ---- libm calls of this wave: one lane per call */')
                for (fn, arg, prm), outs in lst:
                    emit_node(arg)
                self.libm_slot = slot
                B('  if (lane == 0) {')
                for j, ((fn, arg, prm), outs) in enumerate(lst):
                    B('    g_in[wv][%d] = %s;' % (j, self.ref(arg)))
                B('  }')
                B('  {')
                B('    const double a_ = g_in[wv][lane < %d ? lane : 0];' % len(lst))
                B('    double r0_ = 0.0, r1_ = 0.0;')
                j, first = 0, True
                while j < len(lst):
                    fn, prm = lst[j][0][0], lst[j][0][2]
                    k = j
                    while k < len(lst) and lst[k][0][0] == fn and (fn != 'pow' or lst[k][0][2] == prm):
                        k += 1
                    cond = '(lane >= %d && lane < %d)' % (j, k) if k - j > 1 else '(lane == %d)' % j
                    call = {'sincos': 'sincos(a_, &r0_, &r1_)', 'pow': 'r0_ = pow(a_, %s)' % hexf(prm)}.get(fn, 'r0_ = %s(a_)' % fn)
                    B('    %sif %s { %s; }' % ('' if first else 'else ', cond, call))
                    first = False
                    j = k
                B('    if (lane < %d) { g_m[wv][2 * lane] = r0_; g_m[wv][2 * lane + 1] = r1_; }' % len(lst))
                B('  }')
                for (fn, arg, prm), outs in lst:
                    for node in outs.values():
                        emitted.add(node)
                        B(self.stmt(node))

            B('static __device__ CITW_EVAL_INLINE double %s(const int stage, const double T, const unsigned TICK)' % name)
            B('{')
            B('  const CitwSearch (*S)[64] = g_S; const CitwLookup (*L)[2][64] = g_L;')
            B('  const bool major = stage == 0;')
            B('  double STOP = 0.0;')
            B('  const int lane = threadIdx.x & 63;')
            B('  %s;' % ('CITW_U0()' if which == 1 else 'CITW_T0()'))
            if which == 1:
                # Derivative-block bank inputs: read all of them before B1 (either wave rewrites its banks after B1)
                mine, st, seen = set(), [n for n in self.powner if self.powner[n] == 1] + list(self.A1), set()
                while st:
                    m = st.pop()
                    if m in seen:
                        continue
                    seen.add(m)
                    if m in self.xslot01:
                        continue
                    if g.nodes[m][0] == 'in' and g.nodes[m][1] == 'DW':
                        mine.add(m)
                    st.extend(build_dag.children(g, m))
                self.in_override = {}
                for m in sorted(mine):
                    B('  const double dw%d = g_dw[0][%d];' % (g.nodes[m][2], g.nodes[m][2]))
                    self.in_override[m] = 'dw%d' % g.nodes[m][2]
                libm_phase(self.A1)
                B('  /* ---- look-up independent glue */')
                for n in self.exports:
                    emit_node(n, self.A1)
                pre1_xdot = [i for i in self.own1_xdot if self.xdot[i] not in self.post]
                for i in pre1_xdot:
                    emit_node(self.xdot[i], self.A1)
                for k in self.own1_dw:
                    if self.dw_out[k] not in self.post:
                        emit_node(self.dw_out[k], self.A1)
                emit_node(self.stop)
                B('  STOP = %s;' % self.ref(self.stop))
                B('  if (lane == 0) {')
                for n in self.exports:
                    if g.ty[n] == 'b':
                        B('    g_x[%d] = %s ? 1.0 : 0.0;' % (self.xslot[n], self.ref(n)))
                    else:
                        assert g.ty[n] == 'f'
                        B('    g_x[%d] = %s;' % (self.xslot[n], self.ref(n)))
                for i in pre1_xdot:
                    B('    g_f[0][stage][%d] = %s;' % (i, self.ref(self.xdot[i])))
                B('  }')
                B('  CITW_U(10);')
                B('  __syncthreads();   /* B1: exports visible to wave 0; wave 0 has read the Derivative-block banks */')
                B('  CITW_U(11);')
                mine_post = [n for n in self.powner if self.powner[n] == 1]
                if mine_post:
                    B('  /* ---- share of wave 1 in the glue behind the look-ups */')
                    for n in self.exports01:
                        if g.ty[n] == 'b':
                            B('  const bool b%d = g_x[%d] != 0.0;' % (n, self.xslot01[n]))
                        else:
                            B('  const double v%d = g_x[%d];' % (n, self.xslot01[n]))
                        emitted.add(n)
                    lk = [m for m in self.order if g.nodes[m][0] in LOOKUPS and any(u in self.phave[1] for u in self.users[m])]
                    for m in lk:
                        assert self.outslot[m][0] == 0
                        emitted.add(m)
                        B(self.stmt(m))
                    for n in mine_post:
                        emit_node(n)
                    B('  if (lane == 0) {')
                    for i, n in enumerate(self.xdot):
                        if n in self.powner and self.powner[n] == 1:
                            B('    g_f[0][stage][%d] = %s;' % (i, self.ref(n)))
                    B('  }')
                if self.own1_dw:
                    B('  if (major && lane == 0) {')
                    for k in self.own1_dw:
                        B('    g_dw[0][%d] = %s;' % (k, self.ref(self.dw_out[k])))
                    B('  }')
                B('  __syncthreads();   /* B2 */')
                B('  CITW_U(12);')
                self.in_override = {}
            else:
                # Derivative-block bank inputs this wave reads: load them before B1 (wave 1 rewrites the banks after B1)
                mine = set()
                post_roots = [self.xdot[i] for i in self.own0_xdot] + [self.dw_out[k] for k in self.own0_dw]
                st = list(post_roots) + [n for R in self.rounds for n in R['ins']]
                seen = set()
                while st:
                    m = st.pop()
                    if m in seen:
                        continue
                    seen.add(m)
                    if m in self.xslot:
                        continue        # imported, not recomputed
                    if g.nodes[m] [0] == 'in' and g.nodes[m][1] == 'DW':
                        mine.add(m)
                    st.extend(build_dag.children(g, m))
                self.in_override = {}
                for m in sorted(mine):
                    B('  const double dw%d = g_dw[0][%d];' % (g.nodes[m][2], g.nodes[m][2]))
                    self.in_override[m] = 'dw%d' % g.nodes[m][2]
                libm_phase(self.A0x)
                B('  CITW_T(4);')
                for r, R in enumerate(self.rounds):
                    B('  /* ---- look-up round %d */' % (r + 1))
                    for n in R['ins']:
                        emit_node(n, self.A0 if r == 0 else None)
                    B('  if (lane == 0) {')
                    for k, n in enumerate(R['ins']):
                        B('    g_in[wv][%d] = %s;' % (k, self.ref(n)))
                    B('  }')
                    if r == 0:
                        B('  CITW_T(5);')
                    B('  citw_search<%d>(wv, S[%d], lane);' % (R['maxn'], r))
                    if r == 0:
                        B('  CITW_T(6);')
                    if R['L2']:
                        B('  citw_lookup2d(wv, L[%d][0], g_out%d, lane);' % (r, r))
                    if r == 0:
                        B('  CITW_T(7);')
                    if R['L1']:
                        B('  citw_lookup1d(wv, L[%d][1], g_out%d, lane);' % (r, r))
                    if r == 0:
                        B('  CITW_T(8);')
                    done_rounds.add(r)
                    if r == 0:
                        B('  /* ---- share of wave 0 in the look-up independent glue (overlaps the LDS latency of the passes above) */')
                        for n in self.sinks0:
                            emit_node(n, self.A0x)
                        # keep them on this side of the barrier (it is no scheduling barrier for plain arithmetic)
                        for n in self.sinks0:
                            if g.ty[n] == 'f' and g.nodes[n][0] not in LEAF:
                                B('  asm volatile("" :: "v"(%s));' % self.ref(n))
                        if self.exports01:
                            B('  if (lane == 0) {')
                            for n in self.exports01:
                                if g.ty[n] == 'b':
                                    B('    g_x[%d] = %s ? 1.0 : 0.0;' % (self.xslot01[n], self.ref(n)))
                                else:
                                    B('    g_x[%d] = %s;' % (self.xslot01[n], self.ref(n)))
                            B('  }')
                        B('  CITW_T(0);')
                        B('  __syncthreads();   /* B1: the exports of wave 1 are in g_x */')
                        B('  CITW_T(1);')
                        for n in self.exports:
                            if g.ty[n] == 'b':
                                B('  const bool b%d = g_x[%d] != 0.0;' % (n, self.xslot[n]))
                            else:
                                B('  const double v%d = g_x[%d];' % (n, self.xslot[n]))
                            emitted.add(n)
                    mine0 = self.phave[0] | set(n for RR in self.rounds[1:] for n in RR['ins'])
                    for e in R['L2'] + R['L1']:
                        emitted.add(e['node'])
                        if r > 0 or any(u in mine0 for u in self.users[e['node']]):
                            B(self.stmt(e['node']))
                B('  /* ---- derivatives */')
                for i in self.own0_xdot:
                    emit_node(self.xdot[i])
                for k in self.own0_dw:
                    emit_node(self.dw_out[k])
                emit_node(self.stop)
                B('  STOP = %s;' % self.ref(self.stop))
                B('  if (lane == 0) {')
                for i in self.own0_xdot:
                    B('    g_f[0][stage][%d] = %s;' % (i, self.ref(self.xdot[i])))
                B('  }')
                if self.own0_dw:
                    B('  if (major && lane == 0) {')
                    for k in self.own0_dw:
                        B('    g_dw[0][%d] = %s;' % (k, self.ref(self.dw_out[k])))
                    B('  }')
                B('  CITW_T(2);')
                B('  __syncthreads();   /* B2: all derivatives of this stage are in g_f */')
                B('  CITW_T(3);')
                self.in_override = {}
            B('  return STOP;')
            B('}')
            text = '\n'.join(body)
            # shared blackboards live in row 0; each wave has its own libm / look-up input rows
            text = text.replace('g_in[wv]', 'g_in[%d]' % which).replace('g_m[wv]', 'g_m[%d]' % which)
            text = text.replace('[wv]', '[0]').replace('(wv, ', '(0, ')
            return text

        P(function(1))
        P(function(0))
        self.libm_slot = dict(self.libm_slot_all)
        return '\n'.join(out) + '\n'


def main():
    variants = [a for a in sys.argv[1:] if not a.startswith('--')] or ['nominal']
    for v in variants:
        gen = TeamGen(v)
        text = gen.emit_team()
        path = os.path.join(build_dag.ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s_team.inc' % v)
        open(path, 'w').write(text)
        print('%s: %d lines; wave0 pre-barrier glue %d, wave1 glue %d (shared %d), exports %d, xdot owned by wave1 %s, dw by wave1 %d, load estimate %s'
              % (path, text.count('\n'), len(gen.A0x), len(gen.A1), len(gen.A0x & gen.A1), len(gen.exports), gen.own1_xdot,
                 len(gen.own1_dw), gen.load_estimate))


if __name__ == '__main__':
    main()
