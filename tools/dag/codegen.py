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
import build_dag, symex, interp

LOOKUPS = ('l2d', 'l1d')


def hexf(bits):
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
    def __init__(self, variant, fast_zero=True, lds_consts=False, lane_libm=True, lazy_loads=False):
        self.variant = variant
        self.lazy_loads = lazy_loads
        self.lane_libm = lane_libm
        self.lds_consts = lds_consts
        self.kslot = {}
        self.g, self.res, self.datas = build_dag.build(variant, fast_zero=fast_zero)
        inc = open(os.path.join(build_dag.ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s.inc' % variant)).read()
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
        # ---- look-up rounds
        self.rnd = {}
        for n in self.order:
            r = max([self.rnd[c] for c in build_dag.children(g, n)], default=0)
            if g.nodes[n][0] in LOOKUPS:
                r += 1
            self.rnd[n] = r
        self.nrounds = max(self.rnd.values())
        self.rounds = []
        for r in range(1, self.nrounds + 1):
            l2 = [n for n in self.order if g.nodes[n][0] == 'l2d' and self.rnd[n] == r]
            l1 = [n for n in self.order if g.nodes[n][0] == 'l1d' and self.rnd[n] == r]
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
            self.rounds.append(dict(ins=ins, searches=searches, L2=L2, L1=L1,
                                    maxn=max(s[1] for s in searches)))
        # ---- libm calls that depend on the states only (no look-up / libm ancestor): one lane per call
        LIBM = ('sc_sin', 'sc_cos', 'sin', 'cos', 'tan', 'exp', 'log10', 'log', 'atan', 'pow')
        has_anc = {}
        for n in self.order:
            has_anc[n] = any(has_anc[c] or g.nodes[c][0] in LIBM or g.nodes[c][0] in LOOKUPS
                             for c in build_dag.children(g, n))
        calls = {}      # (fn, arg node, param bits) -> {'sin': node, 'cos': node} / {'out': node}
        for n in self.order:
            t = g.nodes[n]
            if t[0] not in LIBM or not self.lane_libm:
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
        # distinct breakpoint vectors over all rounds -> rows of g_bp
        self.bpvec = []
        for R in self.rounds:
            for (xa, n, slot) in R['searches']:
                if (xa, n) not in self.bpvec:
                    self.bpvec.append((xa, n))
        assert len(self.bpvec) <= 48 and max(n for _, n in self.bpvec) <= 23
        self.outslot = {}
        for r, R in enumerate(self.rounds):
            for k, e in enumerate(R['L2']):
                self.outslot[e['node']] = (r, k)
            for k, e in enumerate(R['L1']):
                self.outslot[e['node']] = (r, 64 + k)

    # ---- expression of a node -----------------------------------------------------------------------
    def ref(self, n):
        g = self.g
        t = g.nodes[n]
        op = t[0]
        if op == 'cf':
            if self.lds_consts and t[1] not in (0,):
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

    BIN = dict(add='+', sub='-', mul='*', div='/', gt='>', ge='>=', lt='<', le='<=', eq='==', ne='!=')
    FN1 = dict(sqrt='sqrt', exp='exp', log10='log10', log='log', sin='sin', cos='cos', tan='tan', atan='atan', asin='asin',
               acos='acos', floor='floor', fabs='fabs')

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
        if op in self.BIN:
            e = '%s %s %s' % (R(t[1]), self.BIN[op], R(t[2]))
        elif op == 'neg':
            e = '-%s' % R(t[1])
        elif op in self.FN1:
            e = '%s(%s)' % (self.FN1[op], R(t[1]))
        elif op in ('pow', 'atan2'):
            e = '%s(%s, %s)' % (op, R(t[1]), R(t[2]))
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
        P('#define CITW_%s_ROUNDS %d' % (V.upper(), self.nrounds))
        P('enum { citw_%s_ROUNDS = %d, citw_%s_RO_BASE_W = %d, citw_%s_RO_LO_W = %d, citw_%s_RO_HI_W = %d };  /* f64 word range of .rodata the model reads */'
          % (V, self.nrounds, V, self.ro_base >> 3, V, self.ro_lo >> 3, V, self.ro_hi >> 3))
        lw = self.low
        # ---- descriptor tables
        P('static __device__ const CitwSearch citw_%s_search[%d][64] = {' % (V, self.nrounds))
        for R in self.rounds:
            rows = ['{%d, %d, %d, 0}' % (self.bpvec.index((s[0], s[1])), s[1], s[2]) for s in R['searches']]
            rows += ['{0, 2, 0, 0}'] * (64 - len(rows))
            P('  {' + ', '.join(rows) + '},')
        P('};')
        P('enum { citw_%s_NBP = %d };' % (V, len(self.bpvec)))
        P('static __device__ const CitwBpVec citw_%s_bpvec[%d] = {%s};' % (V, len(self.bpvec), ', '.join('{%d, %d}' % ((a >> 3) - lw, n) for a, n in self.bpvec)))
        P('static __device__ const CitwLookup citw_%s_lookup[%d][2][64] = {' % (V, self.nrounds))
        for R in self.rounds:
            rows2 = ['{%d, %d, %d, %d, %d, %d, %d, %d, %d}' % ((e['xr'] >> 3) - lw, e['nr'], (e['xc'] >> 3) - lw, (e['z'] >> 3) - lw, e['sx'], e['sy'],
                                                               e['in0'], e['in1'], k) for k, e in enumerate(R['L2'])]
            rows2 += ['{0, 2, 0, 0, 63, 63, 0, 0, 127}'] * (64 - len(rows2))
            rows1 = ['{%d, %d, 0, %d, %d, 0, %d, 0, %d}' % ((e['x'] >> 3) - lw, e['n'], (e['y'] >> 3) - lw, e['sx'], e['in0'], 64 + k)
                     for k, e in enumerate(R['L1'])]
            rows1 += ['{0, 2, 0, 0, 63, 0, 0, 0, 127}'] * (64 - len(rows1))
            P('  {{' + ', '.join(rows2) + '},')
            P('   {' + ', '.join(rows1) + '}},')
        P('};')
        # ---- the evaluation function
        P('/* state in g_xs[wv][19], command in g_cmd[wv][10] (wave-uniform LDS reads); derivatives -> g_f[wv][stage][19];')
        P(' * major step: returns the solver stop time and updates the Derivative-block banks g_dw[wv] */')
        P('static __device__ CITW_EVAL_INLINE double citw_%s_eval(const int wv, const int stage, const double T, const unsigned TICK)' % V)
        P('{')
        P('  const CitwSearch (*S)[64] = g_S; const CitwLookup (*L)[2][64] = g_L;')
        P('  const bool major = stage == 0;')
        P('  double STOP = 0.0;')
        P('  const int lane = threadIdx.x & 63;')
        P('  CITW_T0();')
        emitted = set()
        done_rounds = set()

        def emit_node(n):
            # iterative post-order over un-emitted children
            stack = [(n, False)]
            while stack:
                m, done = stack.pop()
                if m in emitted:
                    continue
                if done:
                    emitted.add(m)
                    t = g.nodes[m]
                    if t[0] in LOOKUPS:
                        assert self.outslot[m][0] in done_rounds, 'look-up result used before its round'
                    if t[0] in ('sc_sin', 'sc_cos'):
                        s_, c_ = g.memo.get(('sc_sin', t[1])), g.memo.get(('sc_cos', t[1]))
                        P('  double v%d, v%d; sincos(%s, &v%d, &v%d);' % (s_, c_, self.ref(t[1]), s_, c_))
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
            P('  if (lane == 0) {')
            for j, ((fn, arg, prm), outs) in enumerate(self.libm_calls):
                P('    g_in[wv][%d] = %s;' % (j, self.ref(arg)))
            P('  }')
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
                call = {'sincos': 'sincos(a_, &r0_, &r1_)', 'pow': 'r0_ = pow(a_, %s)' % hexf(prm)}.get(fn, 'r0_ = %s(a_)' % fn)
                P('    %sif %s { %s; }' % ('' if first else 'else ', cond, call))
                first = False
                j = k
            P('    if (lane < %d) { g_m[wv][2 * lane] = r0_; g_m[wv][2 * lane + 1] = r1_; }' % len(self.libm_calls))
            P('  }')
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
            P('  if (lane == 0) {')
            for k, n in enumerate(R['ins']):
                P('    g_in[wv][%d] = %s;' % (k, self.ref(n)))
            P('  }')
            P('  citw_search<%d>(wv, S[%d], lane);' % (R['maxn'], r))
            P('  CITW_T(%d);' % (4 * r + 1))
            if R['L2']:
                P('  citw_lookup2d(wv, L[%d][0], g_out%d, lane);' % (r, r))
            P('  CITW_T(%d);' % (4 * r + 2))
            if R['L1']:
                P('  citw_lookup1d(wv, L[%d][1], g_out%d, lane);' % (r, r))
            P('  CITW_T(%d);' % (4 * r + 3))
            done_rounds.add(r)
            if not self.lazy_loads:
                for e in R['L2'] + R['L1']:
                    emitted.add(e['node'])
                    P(self.stmt(e['node']))
        P('  /* ---- derivatives */')
        for i, n in enumerate(self.xdot):
            emit_node(n)
        P('  if (lane == 0) {')
        for i, n in enumerate(self.xdot):
            P('    g_f[wv][stage][%d] = %s;' % (i, self.ref(n)))
        P('  }')
        P('  CITW_T(%d);' % (4 * self.nrounds))
        P('  if (major) {   /* major step only: solver stop time, Derivative-block banks */')
        emit_node(self.stop)
        P('    STOP = %s;' % self.ref(self.stop))
        for k, n in sorted(self.dw_out.items()):
            emit_node(n)
        P('    if (lane == 0) {')
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
                  lane_libm='--uniform-libm' not in sys.argv, lazy_loads='--lazy-loads' in sys.argv)
        text = gen.emit()
        path = os.path.join(build_dag.ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s_wave.inc' % v)
        open(path, 'w').write(text)
        print('%s: %d lines, rounds %s' % (path, text.count('\n'),
                                            [(len(R['ins']), len(R['searches']), len(R['L2']), len(R['L1'])) for R in gen.rounds]))


if __name__ == '__main__':
    main()
