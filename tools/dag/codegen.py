#!/usr/bin/env python3
"""Generate the wave-cooperative model evaluation (serl_amd/csrc/gen/citation_<variant>_wave.inc) from the DAG.

One model evaluation = ~1 500 scalar f64 operations of "glue" around ~80 table look-ups (symex.py).  On the GPU
one wavefront owns one episode:

  * the glue is emitted as plain SSA (`const double v123 = v77 * v100;`): every lane computes the same
    values (wave-uniform), the compiler allocates registers and schedules; the operation order of the
    reference binary is kept, nothing is re-associated;
  * the look-ups are grouped into dependency ROUNDS (2 in the nominal model).  In each round the distinct
    look-up inputs are put on a per-wave LDS blackboard, then
        phase A: one lane per distinct (breakpoint vector, input) pair finds the interval index,
        phase B: one lane per look-up interpolates (pass 0: all 2-D tables, pass 1: all 1-D tables),
    and the glue reads the results back as wave-uniform LDS loads.

Usage: python tools/dag/codegen.py [variant ...]      (products are committed)
"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_dag, symex, interp, constdiv
from symex import f2b, b2f

LAZY_LIBM = int(os.environ.get('CITW_LAZY_LIBM', 1))      # 1: libm calls whose result is used on one side of a select only run under its condition
CONST_DIV = int(os.environ.get('CITW_CONST_DIV', 1))      # 1: divisions by literals as reciprocal multiply + fma correction (constdiv.py)

LOOKUPS = ('l2d', 'l1d')

# ---- lazy select operands (find_gates): knobs and the cost table of the estimate
GATES = int(os.environ.get('CITW_TEAM_GATES', os.environ.get('CITW_GATES', 1)))   # 1: cones that only the unselected operands of selects on ONE condition need run under that condition
GATE_MIN = float(os.environ.get('CITW_TEAM_GATE_MIN', 8))                        # ... if they cost at least this much (units)
GATE_CW = dict(div=11, sqrt=15, sel=3, unord=2, table3=120)
GATE_FN = ('sc_sin', 'sc_cos', 'sin', 'cos', 'tan', 'exp', 'log10', 'log', 'atan', 'pow')
GATE_LEAF = ('cf', 'ci', 'in', 'in_i', 'true', 'false')



TRUNC_LITERALS = int(os.environ.get('CITW_TRUNC_LITERALS', 0))      # TIMING EXPERIMENT ONLY (wrong results): f64 literals cut to their high dword (one 32-bit literal operand, no s_mov_b32 pair): what would literals that cost nothing buy?


def hexf(bits):
    if TRUNC_LITERALS:
        bits &= ~0xffffffff
    x = symex.b2f(bits)
    if x != x:
        return '__longlong_as_double(0x%016xLL)' % bits
    if x in (float('inf'), float('-inf')):
        return '(%s__builtin_inf())' % ('-' if x < 0 else '')
    s = x.hex()
    if s.startswith('-'):
        return '(%s)' % s
    return s


class Gen:
    def __init__(self, variant, fast_zero=True, lds_consts=False, lane_libm=True, lazy_loads=False, hoist=False, split_chain=False):
        self.variant = variant
        self.split_chain = split_chain
        self.hoist = hoist
        self.lazy_loads = lazy_loads
        self.lane_libm = lane_libm
        self.lds_consts = lds_consts
        self.kslot = {}
        self.g, self.res, self.datas = build_dag.build(variant, fast_zero=fast_zero)
        inc = open(os.path.join(build_dag.ROOT, 'oracle', 'gen', 'citation_%s.inc' % variant)).read()
        import re
        self.ro_base = int(re.search(r'#define RO_BASE (0x[0-9a-f]+)', inc).group(1), 16)
        self.ro_lo = int(re.search(r'#define RO_USED_LO (0x[0-9a-f]+)', inc).group(1), 16)
        self.ro_hi = int(re.search(r'#define RO_USED_HI (0x[0-9a-f]+)', inc).group(1), 16)
        self.low = self.ro_lo >> 3
        g = self.g
        mj, mn = self.res[1]['outs'], self.res[0]['outs']
        self.xdot = [mj['XDOT%d' % i] for i in range(19)]
        self.single = self.xdot == [mn['XDOT%d' % i] for i in range(19)]
        if not self.single:
            raise NotImplementedError('major / minor derivative DAGs differ for %s' % variant)
        self.dw_out = {int(k[2:]): v for k, v in mj.items() if k.startswith('DW')}
        self.stop = mj['STOP0']
        for i in range(12):
            assert g.nodes[mj['Y%d' % i]] == ('in', 'X', i), 'rtY latch is not a copy of X'
        self.roots = self.xdot + list(self.dw_out.values()) + [self.stop]
        self.order = interp.topo(g, self.roots)
        # ---- per-step invariants: nodes that depend only on the command vector, the two constant states
        # (Parameter_CSTATE / _g: their derivatives are the literal 0) and build constants have the same value in all six
        # evaluations of an env step (actuator saturations, trim / flap / gear tables ...): computed once per step
        # (citw_<v>_step_invariants), the evaluations read the frontier values from LDS (g_inv, g_out<R>).
        # Off by default (--hoist-invariants): measured +1.4 % for the team kernel, -2 % for the one-wave kernel --
        # the 22 extra LDS loads per evaluation cost what the ~80 saturation nodes saved.
        const_states = [i for i, n in enumerate(self.xdot) if g.nodes[n] == ('cf', 0)]
        self.inv = {}
        for n in self.order:
            t = g.nodes[n]
            if t[0] == 'in':
                self.inv[n] = self.hoist and (t[1] in ('CMD', 'RO') or (t[1] == 'X' and t[2] in const_states))
            elif t[0] in ('cf', 'ci', 'true', 'false'):
                self.inv[n] = True
            elif t[0] == 'in_i':
                self.inv[n] = False
            else:
                self.inv[n] = self.hoist and all(self.inv[c] for c in build_dag.children(g, n))
        users = collections.defaultdict(list)
        for n in self.order:
            for c in build_dag.children(g, n):
                users[c].append(n)
        rootset = set(self.roots)
        is_leaf = lambda n: g.nodes[n][0] in ('cf', 'ci', 'true', 'false', 'in', 'in_i')
        self.inv_frontier = [n for n in self.order if self.inv[n] and not is_leaf(n)
                             and (any(not self.inv[u] for u in users[n]) or n in rootset)]
        self.inv_slot = {n: k for k, n in enumerate(m for m in self.inv_frontier if g.nodes[m][0] not in LOOKUPS)}
        assert len(self.inv_slot) <= 128

        def make_round(l2, l1):
            ins, searches = [], []
            def slot(lst, key):
                if key not in lst:
                    lst.append(key)
                return lst.index(key)
            L2, L1 = [], []
            for n in l2:
                t = g.nodes[n]
                i0, i1 = slot(ins, t[6]), slot(ins, t[7])
                sx, sy = slot(searches, (t[1], t[2], i0)), slot(searches, (t[3], t[4], i1))
                L2.append(dict(node=n, xr=t[1], nr=t[2], xc=t[3], nc=t[4], z=t[5], sx=sx, sy=sy, in0=i0, in1=i1))
            for n in l1:
                t = g.nodes[n]
                i0 = slot(ins, t[4])
                sx = slot(searches, (t[1], t[2], i0))
                L1.append(dict(node=n, x=t[1], n=t[2], y=t[3], sx=sx, in0=i0))
            assert len(ins) <= 32 and len(searches) <= 64, (len(ins), len(searches))
            assert len(L2) <= 64 and len(L1) <= 64, (len(L2), len(L1))
            return dict(ins=ins, searches=searches, L2=L2, L1=L1, maxn=max([s[1] for s in searches] or [2]))
        # ---- look-up rounds of one evaluation (invariant look-ups are not part of them)
        self.rnd = {}
        for n in self.order:
            r = max([self.rnd[c] for c in build_dag.children(g, n)], default=0)
            if g.nodes[n][0] in LOOKUPS and not self.inv[n]:
                r += 1
            self.rnd[n] = r
        self.nrounds = max(self.rnd.values())
        self.rounds = []
        for r in range(1, self.nrounds + 1):
            l2 = [n for n in self.order if g.nodes[n][0] == 'l2d' and self.rnd[n] == r and not self.inv[n]]
            l1 = [n for n in self.order if g.nodes[n][0] == 'l1d' and self.rnd[n] == r and not self.inv[n]]
            self.rounds.append(make_round(l2, l1))
        # ---- the look-up CHAIN (team kernels, split_chain): first-round look-ups that only feed derivatives which also need the
        # second round -- the engine model: thrust tables -> (divide by the air-data ratio) -> spool-speed tables -> N1 / N2
        # derivatives.  Nothing else waits for them, so a wavefront of its own walks that chain (chain round -> second round ->
        # engine derivatives) beside the aerodynamic look-ups instead of behind them.  The chain round shares the blackboards of
        # round 1 (g_in, g_sidx, g_out0) with slot ranges of its own (ibase / sbase / obase*).
        self.chain_round = None
        if split_chain and self.nrounds == 2:
            late = [r for r in self.roots if self.rnd[r] >= 2]
            anc_other = set()
            for r in self.roots:
                if self.rnd[r] < 2:
                    anc_other |= self.closure_all(r)
            is_l = lambda n, op: g.nodes[n][0] == op and self.rnd[n] == 1 and not self.inv[n]
            chain = [n for n in self.order if (is_l(n, 'l2d') or is_l(n, 'l1d')) and n not in anc_other]
            if chain and late:
                cs = set(chain)
                self.rounds[0] = make_round([n for n in self.order if is_l(n, 'l2d') and n not in cs],
                                            [n for n in self.order if is_l(n, 'l1d') and n not in cs])
                self.chain_round = make_round([n for n in chain if g.nodes[n][0] == 'l2d'], [n for n in chain if g.nodes[n][0] == 'l1d'])
        for r, R in enumerate(self.rounds):
            R.update(ibase=0, obase2=0, obase1=0, oarr=r)
        if self.chain_round is not None:
            R0, RC = self.rounds[0], self.chain_round
            RC.update(ibase=len(R0['ins']), obase2=len(R0['L2']), obase1=len(R0['L1']), oarr=0)
            self.rounds[1]['ibase'] = len(R0['ins']) + len(RC['ins'])         # the second round runs beside round 1 now
            assert self.rounds[1]['ibase'] + len(self.rounds[1]['ins']) <= 32
            assert RC['obase2'] + len(RC['L2']) <= 64 and RC['obase1'] + len(RC['L1']) <= 64
        inv_l2 = [n for n in self.order if g.nodes[n][0] == 'l2d' and self.inv[n]]
        inv_l1 = [n for n in self.order if g.nodes[n][0] == 'l1d' and self.inv[n]]
        for n in inv_l2 + inv_l1:            # one invariant look-up round: their inputs must not need another look-up
            assert not any(g.nodes[a][0] in LOOKUPS for a in self.closure_all(n) if a != n), 'nested invariant look-ups'
        self.inv_round = make_round(inv_l2, inv_l1) if (inv_l2 or inv_l1) else None
        if self.inv_round:
            self.inv_round.update(ibase=0, obase2=0, obase1=0, oarr=len(self.rounds))
        self.all_rounds = self.rounds + ([self.inv_round] if self.inv_round else []) + ([self.chain_round] if self.chain_round else [])
        for ti, R in enumerate(self.all_rounds):
            R['tidx'] = ti                     # row of the descriptor tables (g_S / g_L)
        # every index search owns a slot of g_sidx for the whole episode (the hinted search of citation_wave.h re-verifies the
        # interval the previous evaluation found): the rounds' slots follow each other
        off = 0
        for R in self.all_rounds:
            R['sbase'] = off
            off += len(R['searches'])
        assert off <= 63, 'index-search slots of all rounds must fit g_sidx[.][64] (slot 63 stays 0 for the filler descriptors)'
        assert len(self.all_rounds) <= 3
        # ---- libm calls that depend on the states only (no look-up / libm ancestor): one lane per call
        LIBM = ('sc_sin', 'sc_cos', 'sin', 'cos', 'tan', 'exp', 'log10', 'log', 'atan', 'pow')
        has_anc = {}
        for n in self.order:
            has_anc[n] = any(has_anc[c] or g.nodes[c][0] in LIBM or g.nodes[c][0] in LOOKUPS
                             for c in build_dag.children(g, n))
        calls = {}      # (fn, arg node, param bits) -> {'sin': node, 'cos': node} / {'out': node}
        for n in self.order:
            t = g.nodes[n]
            if t[0] not in LIBM or not self.lane_libm or self.inv[n]:
                continue
            if has_anc[n] or (t[0] == 'pow' and not g.is_cf(t[2])):
                continue
            if t[0] in ('sc_sin', 'sin'):
                calls.setdefault(('sincos', t[1], 0), {})['r0'] = n
            elif t[0] in ('sc_cos', 'cos'):
                calls.setdefault(('sincos', t[1], 0), {})['r1'] = n
            elif t[0] == 'pow':
                calls.setdefault(('pow', t[1], g.nodes[t[2]][1]), {})['r0'] = n
            else:
                calls.setdefault((t[0], t[1], 0), {})['r0'] = n
        order_fn = ['sincos', 'tan', 'exp', 'log10', 'log', 'atan', 'pow']
        self.libm_calls = sorted(calls.items(), key=lambda kv: (order_fn.index(kv[0][0]), kv[0][1], kv[0][2]))
        assert len(self.libm_calls) <= 32
        self.libm_slot = {}
        for j, (key, outs) in enumerate(self.libm_calls):
            for which, node in outs.items():
                self.libm_slot[node] = 2 * j + (0 if which == 'r0' else 1)
        # ---- libm calls whose result is only ever used on ONE side of a select (the ISA atmosphere evaluates its troposphere
        # power law and its stratosphere exponential and then picks one; log10 of the altitude only counts below 300 m): the
        # call is guarded by the select's condition -- a wave-uniform branch around a 60 .. 330 instruction body.  need[n] is a
        # conservative summary of "when does a root depend on n": True (always), or one literal (condition node, polarity).
        self.call_guard = {}
        if LAZY_LIBM:
            rootset = set(self.roots)
            need = {}
            for n in reversed(self.order):
                acc = True if n in rootset else None           # None: no user seen yet
                for u in users[n]:
                    nu = need.get(u)
                    if nu is None:
                        continue
                    tu = g.nodes[u]
                    via = nu
                    if tu[0] == 'sel' and n != tu[1] and (tu[2] == n) != (tu[3] == n):
                        lit = (tu[1], tu[2] == n)
                        via = lit if nu is True else nu          # (either literal alone is a superset of their conjunction)
                    if acc is None:
                        acc = via
                    elif acc is not True and acc != via:
                        acc = True
                    if acc is True:
                        break
                need[n] = acc
            for j, (key, outs) in enumerate(self.libm_calls):
                lits = {need.get(node) for node in outs.values()}
                if len(lits) == 1:
                    lit = lits.pop()
                    if lit is not None and lit is not True:
                        cone = self.closure_all(lit[0])
                        if not any(g.nodes[m][0] in LIBM or g.nodes[m][0] in LOOKUPS or g.nodes[m][0] == 'table3' for m in cone):
                            self.call_guard[j] = lit
        # distinct breakpoint vectors over all rounds -> rows of g_bp
        self.bpvec = []
        for R in self.all_rounds:
            for (xa, n, slot) in R['searches']:
                if (xa, n) not in self.bpvec:
                    self.bpvec.append((xa, n))
        assert len(self.bpvec) <= 48 and max(n for _, n in self.bpvec) <= 23
        self.outslot = {}
        for R in self.all_rounds:
            for k, e in enumerate(R['L2']):
                self.outslot[e['node']] = (R['oarr'], R['obase2'] + k)
            for k, e in enumerate(R['L1']):
                self.outslot[e['node']] = (R['oarr'], 64 + R['obase1'] + k)
        self.find_gates()

    # ---- lazy select operands ------------------------------------------------------------------------------------------------
    def find_gates(self):
        """The model computes both operands of every Switch block: the landing-gear legs (three 35-node contact models with
        divisions and square roots), the ISA stratosphere branch, icing terms ... -- a fifth of the glue feeds only operands
        that the trimmed flight condition never selects.  For a condition c and a polarity p, ex(c, p) is the largest set of
        nodes ALL of whose users are in the set or are selects on c that take the node as their p-operand: nothing outside
        reads them unless c == p.  Such a set is emitted inside `if (c == p) { ... }` (a wave-uniform branch in the single-
        episode kernels, an exec-masked region with a skip branch in the lane-group kernels); the selects themselves stay.
        Values and operation order are untouched: the skipped operand is the one the select discards."""
        g = self.g
        self.gate, self.gate_nodes, self.cold = {}, {}, set()
        if not GATES or not getattr(self, 'use_gates', True):
            return
        users = collections.defaultdict(list)
        for n in self.order:
            for c in build_dag.children(g, n):
                users[c].append(n)
        rootset = set(self.roots)
        sels = collections.defaultdict(list)
        for n in self.order:
            if g.nodes[n][0] == 'sel':
                sels[g.nodes[n][1]].append(n)
        gateable = lambda m: g.nodes[m][0] not in GATE_LEAF + LOOKUPS and g.nodes[m][0] not in GATE_FN and m not in self.libm_slot
        cands = []
        for c in sels:
            for pol, idx in (('T', 2), ('F', 3)):
                seeds = [g.nodes[s_][idx] for s_ in sels[c]]
                cone, st = set(), list(seeds)
                while st:
                    m = st.pop()
                    if m in cone or not gateable(m):
                        continue
                    cone.add(m)
                    st.extend(build_dag.children(g, m))
                ex = set(cone)

                def ok_user(m, u):
                    if u in ex:
                        return True
                    ku = g.nodes[u]
                    return ku[0] == 'sel' and ku[1] == c and ku[idx] == m and ku[5 - idx] != m
                changed = True
                while changed:
                    changed = False
                    for m in list(ex):
                        if m == c or m in rootset or m == self.stop or not all(ok_user(m, u) for u in users[m]):
                            ex.discard(m); changed = True
                w = sum(2 * GATE_CW.get(g.nodes[m][0], 1) for m in ex)
                if ex and w >= GATE_MIN:
                    cands.append((w, c, pol, ex))
        # Which conditions does the trimmed flight condition leave closed?  Only those gates are worth a branch (the gear-down
        # cone, 243 nodes, is open whenever the gear command is 0: gating it would only cost); nested sets: the heaviest closed
        # one takes the node.  Correctness never depends on this choice.
        try:
            import interp, math
            import numpy as np
            src = interp.pysrc(g, self.res[1]['outs'], 'ev_all').rsplit('\n', 1)[0] + '\n    return locals()'
            ns = dict(math=math, safe=interp.safe, sc_sin=interp.sc_sin, sc_cos=interp.sc_cos, fdiv=interp.fdiv, bitsf=interp.bitsf, fbits=interp.fbits,
                      l2d=interp.l2d, l1d=interp.l1d, table3=interp.table3)
            exec(src, ns)
            data = {'nominal': 'h2000_v90'}.get(self.variant, self.variant)
            z = np.load(os.path.join(build_dag.ROOT, 'serl_amd', 'data', 'citation_%s.npz' % data))
            loc = ns['ev_all']([float(x) for x in z['x0']], [0.0] * 10, [float(x) for x in z['dw0'][:29]], [0.0] * 12, 0.0, 0,
                               [float(x) for x in z['ro']], int(z['ro_base']) >> 3, [float(x) for x in z['t3']])
        except Exception as e:      # (no data file for the variant: no gates)
            print('find_gates: trimmed condition not evaluated (%s)' % e, file=sys.stderr)
            return
        self.trim_loc = loc          # every node's value in trimmed flight (tools/dag/critical_path.py: which guards are closed there)
        for w, c, pol, ex in sorted(cands, key=lambda t: (-t[0], t[1], t[2])):
            if bool(loc['v%d' % c]) == (pol == 'T'):
                continue            # open in trimmed flight
            mine = set(m for m in ex if m not in self.gate)
            if sum(2 * GATE_CW.get(g.nodes[m][0], 1) for m in mine) < GATE_MIN:
                continue
            for m in mine:
                self.gate[m] = (c, pol)
            self.gate_nodes[(c, pol)] = mine
            self.cold |= mine


    def closure_all(self, n):
        out, st = set(), [n]
        while st:
            m = st.pop()
            if m in out:
                continue
            out.add(m)
            st.extend(build_dag.children(self.g, m))
        return out

    @staticmethod
    def two_mov_literal(bits):
        """does materialising this f64 literal (given as its bit pattern) cost two 32-bit moves?  (inline constants and values
        whose low dword is zero -- one v_mov_b64 with a 32-bit literal -- do not)"""
        return (int(bits) & 0xffffffff) != 0 and symex.b2f(bits) not in (0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 4.0, -4.0)

    # ---- expression of a node -----------------------------------------------------------------------
    def ref(self, n):
        g = self.g
        t = g.nodes[n]
        op = t[0]
        if op == 'cf':
            if self.lds_consts and t[1] not in (0,) and (self.lds_consts != 2 or self.two_mov_literal(t[1])):
                if t[1] not in self.kslot:
                    self.kslot[t[1]] = len(self.kslot)
                return 'g_k[%d]' % self.kslot[t[1]]
            return hexf(t[1])
        if op == 'ci':
            return '%dLL' % t[1]
        if op == 'true':
            return 'true'
        if op == 'false':
            return 'false'
        if op == 'in':
            ov = getattr(self, 'in_override', None)
            if ov and n in ov:
                return ov[n]
            if t[1] == 'RO':
                return 'g_ro[%d]' % ((t[2] >> 3) - self.low)
            if t[1] == 'T':
                return 'T'
            if t[1] == 'DW':
                return 'g_dw[wv][%d]' % t[2]
            return {'X': 'g_xs[wv][%d]', 'CMD': 'g_cmd[wv][%d]'}[t[1]] % t[2]
        if op == 'in_i':
            return '(long long)TICK'
        return {'f': 'v%d', 'b': 'b%d', 'i': 'i%d'}[g.ty[n]] % n

    def inv_load(self, n):
        """statement that brings a per-step invariant into an evaluation"""
        g = self.g
        if g.nodes[n][0] in LOOKUPS:
            return self.stmt(n)
        k = self.inv_slot[n]
        if g.ty[n] == 'b':
            return '  const bool b%d = g_inv[wv][%d] != 0.0;' % (n, k)
        assert g.ty[n] == 'f', g.nodes[n]
        return '  const double v%d = g_inv[wv][%d];' % (n, k)

    BIN = dict(add='+', sub='-', mul='*', div='/', gt='>', ge='>=', lt='<', le='<=', eq='==', ne='!=')
    FN1 = dict(sqrt='sqrt', exp='exp', log10='log10', log='log', sin='citw_sin', cos='citw_cos', tan='citw_tan', atan='citw_atan', asin='asin',
               acos='acos', floor='floor', fabs='fabs')

    _cdiv = {}

    def const_div_ok(self, bits):
        if bits not in Gen._cdiv:
            Gen._cdiv[bits] = constdiv.verify(b2f(bits))[0]
        return Gen._cdiv[bits]

    def stmt(self, n):
        g = self.g
        t = g.nodes[n]
        op = t[0]
        R = self.ref
        if op in ('cf', 'ci', 'true', 'false', 'in', 'in_i'):
            return None
        ty = {'f': 'const double', 'b': 'const bool', 'i': 'const long long'}[g.ty[n]]
        name = R(n)
        if n in self.libm_slot:
            return '  %s %s = g_m[wv][%d];' % (ty, name, self.libm_slot[n])
        if op == 'div' and CONST_DIV and getattr(self, 'const_div', True) and g.nodes[t[2]][0] == 'cf' and self.const_div_ok(g.nodes[t[2]][1]):
            # division by a literal: multiply by the correctly rounded reciprocal + one fma correction step, correctly
            # rounded for every dividend (tools/dag/constdiv.py proves it per divisor); 4 instructions instead of 13
            c = b2f(g.nodes[t[2]][1])
            e = 'citw_div_const(%s, %s, %s)' % (R(t[1]), hexf(g.nodes[t[2]][1]), hexf(f2b(constdiv.recip(c))))
        elif op in self.BIN:
            e = '%s %s %s' % (R(t[1]), self.BIN[op], R(t[2]))
        elif op == 'neg':
            e = '-%s' % R(t[1])
        elif op in self.FN1:
            e = '%s(%s)' % (self.FN1[op], R(t[1]))
        elif op in ('pow', 'atan2'):
            e = '%s(%s, %s)' % ({'pow': 'citw_pow'}.get(op, op), R(t[1]), R(t[2]))
        elif op == 'powsnf':
            e = 'citw_powd_snf(%s, %s)' % (R(t[1]), R(t[2]))
        elif op == 'sel':
            e = '%s ? %s : %s' % (R(t[1]), R(t[2]), R(t[3]))
        elif op == 'bnot':
            e = '!%s' % R(t[1])
        elif op == 'band':
            e = '%s && %s' % (R(t[1]), R(t[2]))
        elif op == 'bor':
            e = '%s || %s' % (R(t[1]), R(t[2]))
        elif op == 'unord':
            e = '(%s != %s) || (%s != %s)' % (R(t[1]), R(t[1]), R(t[2]), R(t[2]))
        elif op in LOOKUPS:
            r, k = self.outslot[n]
            e = 'g_out%d[wv][%d]' % (r, k)
        elif op == 'table3':
            e = 'citw_table3(g_t3, %s, %s, %s)' % (R(t[1]), R(t[2]), R(t[3]))
        elif op == 'iadd':
            e = '%s + %s' % (R(t[1]), R(t[2]))
        elif op == 'i2d':
            e = '(double)(%s)' % R(t[1])
        elif op in ('fxor', 'fand', 'for'):
            e = 'citw_u2d(citw_d2u(%s) %s citw_d2u(%s))' % (R(t[1]), {'fxor': '^', 'fand': '&', 'for': '|'}[op], R(t[2]))
        elif op == 'fandn':
            e = 'citw_u2d(~citw_d2u(%s) & citw_d2u(%s))' % (R(t[1]), R(t[2]))
        elif op == 'fmask':
            e = 'citw_u2d(%s ? ~0ULL : 0ULL)' % R(t[1])
        elif op == 'bits':
            e = '(long long)citw_d2u(%s)' % R(t[1])
        elif op in ('sc_sin', 'sc_cos'):
            return None       # emitted as a pair by emit()
        else:
            raise NotImplementedError(op)
        return '  %s %s = %s;' % (ty, name, e)

    def emit_invariants(self):
        """-> lines of citw_<v>_step_invariants(wv): everything that is the same in the six evaluations of an env step"""
        g, V = self.g, self.variant
        out = []
        P = out.append
        P('enum { citw_%s_NINV = %d };' % (V, len(self.inv_slot)))
        P('/* once per env step, after the command vector is in g_cmd[wv]: %d values for g_inv[wv]%s */' %
          (len(self.inv_slot), (' and %d table look-ups for g_out%d[wv]' % (len(self.inv_round['L2']) + len(self.inv_round['L1']), len(self.rounds))) if self.inv_round else ''))
        P('/* wv: LDS rows this wavefront owns (g_in, g_sidx, g_inv, g_out%d); sv: row of the episode state (g_cmd, g_xs) */' % len(self.rounds))
        P('static __device__ __forceinline__ void citw_%s_step_invariants(const int wv, const int sv)' % V)
        P('{')
        P('  const CitwSearch (*S)[64] = g_S; const CitwLookup (*L)[2][64] = g_L;')
        P('  const int lane = CITW_LANE;')
        P('  (void)S; (void)L; (void)lane;')
        emitted = set()
        RI = len(self.rounds)

        def emit_node(n):
            stack = [(n, False)]
            while stack:
                m, done = stack.pop()
                if m in emitted:
                    continue
                if done:
                    emitted.add(m)
                    t = g.nodes[m]
                    assert self.inv[m], 'non-invariant node %s in the invariants function' % (t[:2],)
                    if t[0] in ('sc_sin', 'sc_cos'):
                        s_, c_ = g.memo.get(('sc_sin', t[1])), g.memo.get(('sc_cos', t[1]))
                        P('  double v%d = 0.0, v%d = 0.0; citw_sincos(%s, &v%d, &v%d); (void)v%d; (void)v%d;' % (s_, c_, self.ref(t[1]), s_, c_, s_, c_))
                        emitted.add(s_); emitted.add(c_)
                        continue
                    s = self.stmt(m)
                    if s:
                        P(s)
                    continue
                stack.append((m, True))
                if g.nodes[m][0] in LOOKUPS:
                    continue
                for c in build_dag.children(g, m):
                    if c not in emitted:
                        stack.append((c, False))
        if self.inv_round:
            R = self.inv_round
            for n in R['ins']:
                emit_node(n)
            P('  if (CITW_LANE0) {')
            for k, n in enumerate(R['ins']):
                P('    g_in[wv][%d] = %s;' % (k, self.ref(n)))
            P('  }')
            P('  citw_search<%d, %d, %d>(wv, S[%d], lane);' % (R['maxn'], len(R['searches']), R['sbase'], RI))
            if R['L2']:
                P('  citw_lookup2d<%d>(wv, L[%d][0], g_out%d, lane);' % (len(R['L2']), RI, RI))
            if R['L1']:
                P('  citw_lookup1d<%d>(wv, L[%d][1], g_out%d, lane);' % (len(R['L1']), RI, RI))
            for e in R['L2'] + R['L1']:
                emitted.add(e['node'])
                P(self.stmt(e['node']))
        for n in self.inv_slot:
            emit_node(n)
        if self.inv_slot:
            P('  if (CITW_LANE0) {')
            for n, k in self.inv_slot.items():
                P('    g_inv[wv][%d] = %s;' % (k, ('%s ? 1.0 : 0.0' % self.ref(n)) if g.ty[n] == 'b' else self.ref(n)))
            P('  }')
        P('}')
        return [ln.replace('g_cmd[wv]', 'g_cmd[sv]').replace('g_xs[wv]', 'g_xs[sv]') for ln in out]

    def table_lines(self, pre):
        """descriptor tables of the look-up rounds (index searches, 2-D / 1-D tables, distinct breakpoint vectors) as `pre`_search /
        _lookup / _bpvec / _NBP"""
        out = []
        P = out.append
        lw = self.low
        srows = lambda R: ['{%d, %d, %d, %d}' % (self.bpvec.index((s[0], s[1])), s[1], R['ibase'] + s[2], R['sbase'] + k) for k, s in enumerate(R['searches'])]
        rows2 = lambda R: ['{%d, %d, %d, %d, %d, %d, %d, %d, %d, %d}' % ((e['xr'] >> 3) - lw, e['nr'], (e['xc'] >> 3) - lw, (e['z'] >> 3) - lw, R['sbase'] + e['sx'], R['sbase'] + e['sy'],
                                                                        R['ibase'] + e['in0'], R['ibase'] + e['in1'], R['obase2'] + k, e['nc']) for k, e in enumerate(R['L2'])]
        rows1 = lambda R: ['{%d, %d, 0, %d, %d, 0, %d, 0, %d}' % ((e['x'] >> 3) - lw, e['n'], (e['y'] >> 3) - lw, R['sbase'] + e['sx'], R['ibase'] + e['in0'], 64 + R['obase1'] + k)
                           for k, e in enumerate(R['L1'])]
        # a 1-D table as the first half of a 2-D lane (citw_spec_pre / citw_spec_tail: nr = 0, both axes the same search, p1 = 1)
        rows1as2 = lambda R: ['{%d, 0, %d, %d, %d, %d, %d, %d, %d, %d, 1}' % ((e['x'] >> 3) - lw, (e['x'] >> 3) - lw, (e['y'] >> 3) - lw, R['sbase'] + e['sx'], R['sbase'] + e['sx'],
                                                                            R['ibase'] + e['in0'], R['ibase'] + e['in0'], 64 + R['obase1'] + k, e['n']) for k, e in enumerate(R['L1'])]
        fill_s, fill_2, fill_1 = '{0, 2, 0, 0}', '{0, 2, 0, 0, 63, 63, 0, 0, 127, 2}', '{0, 2, 0, 0, 63, 0, 0, 0, 127}'
        spec = getattr(self, 'spec', None)
        P('static __device__ const CitwSearch %s_search[%d][64] = {' % (pre, len(self.all_rounds) + (1 if spec else 0)))
        for R in self.all_rounds:
            rows = srows(R)
            rows += [fill_s] * (64 - len(rows))
            P('  {' + ', '.join(rows) + '},')
        if spec:
            # the merged row of citw_spec_pre: wave 0's searches of round 1, then the later rounds' (one lane each)
            rows = srows(self.rounds[0])[:spec['ns'][0][1]]
            for R in self.rounds[1:]:
                rows += srows(R)
            assert len(rows) == spec['NS']
            P('  {' + ', '.join(rows + [fill_s] * (64 - len(rows))) + '},')
        P('};')
        P('enum { %s_NBP = %d };' % (pre, len(self.bpvec)))
        P('static __device__ const CitwBpVec %s_bpvec[%d] = {%s};' % (pre, len(self.bpvec), ', '.join('{%d, %d}' % ((a >> 3) - lw, n) for a, n in self.bpvec)))
        P('static __device__ const CitwLookup %s_lookup[%d][2][64] = {' % (pre, len(self.all_rounds) + (1 if spec else 0)))
        for R in self.all_rounds:
            r2, r1 = rows2(R), rows1(R)
            r2 += [fill_2] * (64 - len(r2))
            r1 += [fill_1] * (64 - len(r1))
            P('  {{' + ', '.join(r2) + '},')
            P('   {' + ', '.join(r1) + '}},')
        if spec:
            rows = rows2(self.rounds[0])
            for R in self.rounds[1:]:
                rows += rows2(R) + rows1as2(R)
            assert len(rows) == spec['NT']
            P('  {{' + ', '.join(rows + [fill_2] * (64 - len(rows))) + '},')
            P('   {' + ', '.join([fill_1] * 64) + '}},')
        P('};')
        return out

    def emit(self):
        g = self.g
        V = self.variant
        out = []
        P = out.append
        P('/* GENERATED by tools/dag/codegen.py from gen/citation_%s.inc -- do not edit.' % V)
        P(' * Wave-cooperative evaluation of the %s model: %d live nodes, %d look-up round(s).' % (V, len(self.order), self.nrounds))
        P(' * Operation order of every f64 expression is that of the reference binary (tools/dag/symex.py). */')
        cnt = collections.Counter(g.nodes[n][0] for n in self.order)
        P('/* node census: %s */' % ', '.join('%s %d' % kv for kv in cnt.most_common()))
        P('#define CITW_%s_ROUNDS %d' % (V.upper(), len(self.all_rounds)))
        P('enum { citw_%s_ROUNDS = %d, citw_%s_RO_BASE_W = %d, citw_%s_RO_LO_W = %d, citw_%s_RO_HI_W = %d };  /* f64 word range of .rodata the model reads */'
          % (V, len(self.all_rounds), V, self.ro_base >> 3, V, self.ro_lo >> 3, V, self.ro_hi >> 3))
        for line in self.table_lines('citw_%s' % V):
            P(line)
        for line in self.emit_invariants():
            P(line)
        # ---- the evaluation function
        P('/* state in g_xs[wv][19], command in g_cmd[wv][10] (wave-uniform LDS reads); derivatives -> g_f[wv][stage][19];')
        P(' * major step: returns the solver stop time and updates the Derivative-block banks g_dw[wv] */')
        P('static __device__ CITW_EVAL_INLINE double citw_%s_eval(const int wv, const int stage, const double T, const unsigned TICK)' % V)
        P('{')
        P('  const CitwSearch (*S)[64] = g_S; const CitwLookup (*L)[2][64] = g_L;')
        P('  const bool major = stage == 0;')
        P('  double STOP = 0.0;')
        P('  const int lane = CITW_LANE;')
        P('  CITW_T0();')
        emitted = set()
        done_rounds = set()
        if self.inv_frontier:
            P('  /* ---- per-step invariants (citw_%s_step_invariants) */' % V)
            done_rounds.add(len(self.rounds))
            for n in self.inv_frontier:
                P(self.inv_load(n))
                emitted.add(n)

        def emit_gated(m0):
            """the not yet emitted part of gate (c, pol) that m0 needs, in one conditional block; what it reads from outside first"""
            import re as _re
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
            emit_node(c)
            for r in R:
                for ch in build_dag.children(g, r):
                    if ch not in exk and ch not in emitted and g.nodes[ch][0] not in GATE_LEAF:
                        emit_node(ch)
            body = []
            for r in R:
                mt = _re.match(r'^  const (double|bool|long long) (\w+) = (.*);$', self.stmt(r))
                assert mt, self.stmt(r)
                P('  %s %s = %s;' % (mt.group(1), mt.group(2), {'double': '0.0', 'bool': 'false', 'long long': '0'}[mt.group(1)]))
                body.append('    %s = %s;' % (mt.group(2), mt.group(3)))
                emitted.add(r)
            P('  if (%s%s) {   /* only the selects on this condition read these */' % ('' if pol == 'T' else '!', self.ref(c)))
            for line in body:
                P(line)
            P('  }')

        def emit_node(n):
            # iterative post-order over un-emitted children
            stack = [(n, False)]
            while stack:
                m, done = stack.pop()
                if m in emitted:
                    continue
                if m in self.gate and not done:
                    emit_gated(m)
                    continue
                if done:
                    emitted.add(m)
                    t = g.nodes[m]
                    if t[0] in LOOKUPS:
                        assert self.outslot[m][0] in done_rounds, 'look-up result used before its round'
                    if t[0] in ('sc_sin', 'sc_cos'):
                        s_, c_ = g.memo.get(('sc_sin', t[1])), g.memo.get(('sc_cos', t[1]))
                        P('  double v%d, v%d; citw_sincos(%s, &v%d, &v%d);' % (s_, c_, self.ref(t[1]), s_, c_))
                        emitted.add(s_); emitted.add(c_)
                        continue
                    s = self.stmt(m)
                    if s:
                        P(s)
                    continue
                stack.append((m, True))
                if g.nodes[m][0] in LOOKUPS or m in self.libm_slot:
                    continue          # value comes from LDS; its inputs were consumed by the lane phase
                for c in build_dag.children(g, m):
                    if c not in emitted:
                        stack.append((c, False))

        if self.libm_calls:
            P('  /* ---- libm calls that depend on the states only: one lane per call (the branches of one function run together) */')
            for (fn, arg, prm), outs in self.libm_calls:
                emit_node(arg)
            for j in sorted(self.call_guard):
                emit_node(self.call_guard[j][0])         # the condition under which call j's result is used at all
            P('  if (CITW_LANE0) {')
            for j, ((fn, arg, prm), outs) in enumerate(self.libm_calls):
                P('    g_in[wv][%d] = %s;' % (j, self.ref(arg)))
            P('  }')
            P('  CITW_WAVE_FENCE();')
            P('  {')
            P('    const double a_ = g_in[wv][lane < %d ? lane : 0];' % len(self.libm_calls))
            P('    double r0_ = 0.0, r1_ = 0.0;')
            j = 0
            first = True
            while j < len(self.libm_calls):
                fn, prm = self.libm_calls[j][0][0], self.libm_calls[j][0][2]
                k = j
                while k < len(self.libm_calls) and self.libm_calls[k][0][0] == fn and (fn != 'pow' or self.libm_calls[k][0][2] == prm):
                    k += 1
                cond = '(lane >= %d && lane < %d)' % (j, k) if k - j > 1 else '(lane == %d)' % j
                if k - j == 1 and j in self.call_guard:
                    cond = '(lane == %d && %s%s)' % (j, '' if self.call_guard[j][1] else '!', self.ref(self.call_guard[j][0]))
                call = {'sincos': 'citw_sincos(a_, &r0_, &r1_)', 'pow': 'r0_ = citw_pow(a_, %s)' % hexf(prm), 'tan': 'r0_ = citw_tan(a_)'}.get(fn, 'r0_ = %s(a_)' % {'atan': 'citw_atan'}.get(fn, fn))
                P('    %sif %s { %s; }' % ('' if first else 'else ', cond, call))
                first = False
                j = k
            P('    if (lane < %d) { g_m[wv][2 * lane] = r0_; g_m[wv][2 * lane + 1] = r1_; }' % len(self.libm_calls))
            P('  }')
            P('  CITW_WAVE_FENCE();   /* the lanes that made the calls stored, every lane loads the results */')
            if not self.lazy_loads:
                for (fn, arg, prm), outs in self.libm_calls:
                    for node in outs.values():
                        emitted.add(node)
                        P(self.stmt(node))
        for r, R in enumerate(self.rounds):
            P('  /* ---- look-up round %d: %d inputs, %d index searches, %d 2-D + %d 1-D tables */' %
              (r + 1, len(R['ins']), len(R['searches']), len(R['L2']), len(R['L1'])))
            for n in R['ins']:
                emit_node(n)
            P('  CITW_T(%d);' % (4 * r))
            P('  if (CITW_LANE0) {')
            for k, n in enumerate(R['ins']):
                P('    g_in[wv][%d] = %s;' % (k, self.ref(n)))
            P('  }')
            P('  CITW_WAVE_FENCE();   /* (one lane may have stored the inputs alone: CITW_UNIFORM_STORE) */')
            sa = (R['maxn'], len(R['searches']), R['sbase'])
            P('#if CITW_GROUP_LANES == 64 && CITW_FUSED_LOOKUP   /* the look-up lanes verify the hints of their own index searches */')
            P('  CITW_T(%d);' % (4 * r + 1))
            if R['L2']:
                P('  citw_lookup2d_fused<%d, %d, %d, %d>(wv, S[%d], L[%d][0], g_out%d, lane);' % ((len(R['L2']),) + sa + (r, r, r)))
            P('  CITW_T(%d);' % (4 * r + 2))
            if R['L1']:
                P('  citw_lookup1d_fused<%d, %d, %d, %d>(wv, S[%d], L[%d][1], g_out%d, lane);' % ((len(R['L1']),) + sa + (r, r, r)))
            P('  CITW_T(%d);' % (4 * r + 3))
            P('#else')
            P('  citw_search<%d, %d, %d>(wv, S[%d], lane);' % (R['maxn'], len(R['searches']), R['sbase'], r))
            P('  CITW_T(%d);' % (4 * r + 1))
            if R['L2']:
                P('  citw_lookup2d<%d>(wv, L[%d][0], g_out%d, lane);' % (len(R['L2']), r, r))
            P('  CITW_T(%d);' % (4 * r + 2))
            if R['L1']:
                P('  citw_lookup1d<%d>(wv, L[%d][1], g_out%d, lane);' % (len(R['L1']), r, r))
            P('  CITW_T(%d);' % (4 * r + 3))
            P('#endif')
            done_rounds.add(r)
            if not self.lazy_loads:
                for e in R['L2'] + R['L1']:
                    emitted.add(e['node'])
                    P(self.stmt(e['node']))
        P('  /* ---- derivatives */')
        for i, n in enumerate(self.xdot):
            emit_node(n)
        P('  if (CITW_LANE0) {')
        for i, n in enumerate(self.xdot):
            P('    g_f[wv][stage][%d] = %s;' % (i, self.ref(n)))
        P('  }')
        P('  CITW_T(%d);' % (4 * self.nrounds))
        P('  if (major) {   /* major step only: solver stop time, Derivative-block banks */')
        emit_node(self.stop)
        P('    STOP = %s;' % self.ref(self.stop))
        for k, n in sorted(self.dw_out.items()):
            emit_node(n)
        P('    if (CITW_LANE0) {')
        for k, n in sorted(self.dw_out.items()):
            if g.nodes[n] != ('in', 'DW', k):
                P('      g_dw[wv][%d] = %s;' % (k, self.ref(n)))
        P('    }')
        P('  }')
        P('  return STOP;')
        P('}')
        ks = sorted(self.kslot.items(), key=lambda kv: kv[1])
        P('/* f64 literals of the model, staged into LDS (g_k) so that they are loaded where they are used instead of')
        P(' * being materialised into scalar registers and kept live across the whole stage loop */')
        P('enum { citw_%s_NK = %d };' % (V, max(len(ks), 1)))
        P('static __device__ const double citw_%s_k[%d] = {' % (V, max(len(ks), 1)))
        P('  ' + ', '.join(hexf(b) for b, _ in ks) if ks else '  0.0')
        P('};')
        return '\n'.join(out) + '\n'


def main():
    variants = [a for a in sys.argv[1:] if not a.startswith('--')] or ['nominal']
    for v in variants:
        gen = Gen(v, fast_zero='--exact-zero' not in sys.argv, lds_consts='--lds-consts' in sys.argv,
                  lane_libm='--uniform-libm' not in sys.argv, lazy_loads='--lazy-loads' in sys.argv, hoist='--hoist-invariants' in sys.argv)
        text = gen.emit()
        path = os.path.join(build_dag.ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s_wave.inc' % v)
        open(path, 'w').write(text)
        print('%s: %d lines, rounds %s' % (path, text.count('\n'),
                                            [(len(R['ins']), len(R['searches']), len(R['L2']), len(R['L1'])) for R in gen.rounds]))


if __name__ == '__main__':
    main()
