"""Symbolic execution of the lifted Citation model (gen/citation_<variant>.inc) into a dataflow graph.

The lifted code (tools/lift) is a faithful but shapeless restatement: ~5 k statements, one per x86-64
instruction, with every memory reference already resolved to a named cell.  For the GPU we want the
*structure* of one model evaluation -- which f64 operations depend on which -- so that the table
look-ups can be spread over the lanes of a wavefront and the rest emitted as clean SSA (codegen.py).
This module executes the statement list symbolically:

  * values are hash-consed nodes of a DAG (constants folded in IEEE-754 double arithmetic, which
    Python floats implement exactly like SSE2 scalar code: round-to-nearest-even, no contraction);
  * the control-flow graph (acyclic after the lifter's unrolling) is if-converted: every block gets a
    path predicate, states are merged at joins with select nodes;
  * S-function calls are expanded: ac_atmos by executing its lifted body, ac_axes by the reference's
    operation order (sincos of the five angles -> three rotation matrices -> matmultiply chain,
    @0x103e0/0x103a0), table3 / rt_powd_snf / libm calls become opaque leaf nodes;
  * the operation ORDER of every f64 expression is preserved -- nothing is re-associated.

Inputs  : X[19], CMD[10], DW[29], Y[12], T (model time), TICK, RO[...] (build tables), STOP
Outputs : XDOT[19] always; in the major evaluation also Y[12], DW[26] (Derivative-block banks), STOP.
"""
import math, re, struct, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cparse

M64 = (1 << 64) - 1
SIGN = 0x8000000000000000
ABSM = 0x7fffffffffffffff


def f2b(x):
    return struct.unpack('<Q', struct.pack('<d', x))[0]


def b2f(b):
    return struct.unpack('<d', struct.pack('<Q', b & M64))[0]


class Dag:
    def __init__(self):
        self.nodes = []      # tuples (op, *args)
        self.ty = []         # 'f' | 'i' | 'b'
        self.memo = {}
        self.TRUE = self.mk('true')
        self.FALSE = self.mk('false')

    def _new(self, tup, ty):
        k = self.memo.get(tup)
        if k is None:
            k = len(self.nodes)
            self.nodes.append(tup)
            self.ty.append(ty)
            self.memo[tup] = k
        return k

    def op(self, n):
        return self.nodes[n][0]

    # ---- constants
    def cf(self, x):
        return self._new(('cf', f2b(x) if isinstance(x, float) else x), 'f')

    def cfbits(self, b):
        return self._new(('cf', b & M64), 'f')

    def ci(self, v):
        return self._new(('ci', v), 'i')

    def is_cf(self, n):
        return self.nodes[n][0] == 'cf'

    def is_ci(self, n):
        return self.nodes[n][0] == 'ci'

    def fval(self, n):
        return b2f(self.nodes[n][1])

    def ival(self, n):
        return self.nodes[n][1]

    TYPES = dict(true='b', false='b', gt='b', ge='b', lt='b', le='b', eq='b', ne='b', unord='b', bnot='b', band='b',
                 bor='b', ieq='b', ine='b', bits='i', bitsnot='i', in_i='i', iadd='i', isel='i', b2i='i')

    def mk(self, op, *a):
        N = self.nodes
        # ---------------- folding / simplification (value preserving)
        if op in ('add', 'sub', 'mul', 'div') and self.is_cf(a[0]) and self.is_cf(a[1]):
            x, y = self.fval(a[0]), self.fval(a[1])
            try:
                r = {'add': x + y, 'sub': x - y, 'mul': x * y}[op] if op != 'div' else x / y
                return self.cf(r)
            except (ZeroDivisionError, OverflowError):
                pass
        if op == 'sqrt' and self.is_cf(a[0]) and self.fval(a[0]) >= 0:
            return self.cf(math.sqrt(self.fval(a[0])))
        if op == 'i2d' and self.is_ci(a[0]):
            return self.cf(float(self.ival(a[0])))
        if op == 'i2d' and N[a[0]][0] == 'b2i':
            return self.mk('sel', N[a[0]][1], self.cf(1.0), self.cf(0.0))
        if op in ('fxor', 'fand', 'for', 'fandn') and self.is_cf(a[0]) and self.is_cf(a[1]):
            x, y = N[a[0]][1], N[a[1]][1]
            return self.cfbits({'fxor': x ^ y, 'fand': x & y, 'for': x | y, 'fandn': (~x) & y}[op])
        if op == 'fxor':
            for u, v in ((a[0], a[1]), (a[1], a[0])):
                if self.is_cf(v) and N[v][1] == SIGN:
                    return self.mk('neg', u)
        if op == 'fand':
            for u, v in ((a[0], a[1]), (a[1], a[0])):
                if self.is_cf(v) and N[v][1] == ABSM:
                    return self.mk('fabs', u)
                if N[v][0] == 'fmask':
                    return self.mk('sel', N[v][1], u, self.cfbits(0))
        if op == 'fandn':
            if self.is_cf(a[0]) and N[a[0]][1] == SIGN:
                return self.mk('fabs', a[1])
            if N[a[0]][0] == 'fmask':
                return self.mk('sel', N[a[0]][1], self.cfbits(0), a[1])
        if op == 'for':
            x, y = N[a[0]], N[a[1]]
            if x[0] == 'sel' and y[0] == 'sel' and x[1] == y[1]:
                z = self.cfbits(0)
                if x[3] == z and y[2] == z:
                    return self.mk('sel', x[1], x[2], y[3])
                if x[2] == z and y[3] == z:
                    return self.mk('sel', x[1], y[2], x[3])
        if op == 'neg' and self.is_cf(a[0]):
            return self.cfbits(N[a[0]][1] ^ SIGN)
        if op == 'fabs' and self.is_cf(a[0]):
            return self.cfbits(N[a[0]][1] & ABSM)
        if op == 'sel':
            c, x, y = a
            if c == self.TRUE:
                return x
            if c == self.FALSE:
                return y
            if x == y:
                return x
            if N[c][0] == 'bnot':
                return self.mk('sel', N[c][1], y, x)
        if op in ('gt', 'ge', 'lt', 'le', 'eq', 'ne') and self.is_cf(a[0]) and self.is_cf(a[1]):
            x, y = self.fval(a[0]), self.fval(a[1])
            r = {'gt': x > y, 'ge': x >= y, 'lt': x < y, 'le': x <= y, 'eq': x == y, 'ne': x != y}[op]
            return self.TRUE if r else self.FALSE
        if op == 'unord':
            if self.is_cf(a[0]) and self.is_cf(a[1]):
                x, y = self.fval(a[0]), self.fval(a[1])
                return self.TRUE if (x != x or y != y) else self.FALSE
        if op == 'bnot':
            if a[0] == self.TRUE:
                return self.FALSE
            if a[0] == self.FALSE:
                return self.TRUE
            if N[a[0]][0] == 'bnot':
                return N[a[0]][1]
        if op == 'band':
            if a[0] == self.FALSE or a[1] == self.FALSE:
                return self.FALSE
            if a[0] == self.TRUE:
                return a[1]
            if a[1] == self.TRUE or a[0] == a[1]:
                return a[0]
        if op == 'bor':
            if a[0] == self.TRUE or a[1] == self.TRUE:
                return self.TRUE
            if a[0] == self.FALSE:
                return a[1]
            if a[1] == self.FALSE or a[0] == a[1]:
                return a[0]
        if op == 'fmask':
            if a[0] == self.TRUE:
                return self.cfbits(M64)
            if a[0] == self.FALSE:
                return self.cfbits(0)
        if op == 'bits' and self.is_cf(a[0]):
            return self.ci(N[a[0]][1])
        if op == 'powsnf' and self.is_cf(a[1]) and self.fval(a[1]) == 2.0:
            return self.mk('mul', a[0], a[0])       # rt_powd_snf(u, 2.0) = u*u (NaN in -> NaN out either way)
        if op == 'iadd' and self.is_ci(a[0]) and self.is_ci(a[1]):
            return self.ci(self.ival(a[0]) + self.ival(a[1]))
        return self._new((op,) + a, self.TYPES.get(op, 'f'))


# ---------------------------------------------------------------------------------------------------
# path predicates: DNF = frozenset of conjunctions; conjunction = frozenset of (cond node, polarity)
# ---------------------------------------------------------------------------------------------------
PTRUE = frozenset([frozenset()])
PFALSE = frozenset()


def p_simplify(p):
    p = set(p)
    changed = True
    while changed:
        changed = False
        lst = list(p)
        for i in range(len(lst)):
            for j in range(i + 1, len(lst)):
                a, b = lst[i], lst[j]
                if a <= b:
                    p.discard(b); changed = True; break
                if b <= a:
                    p.discard(a); changed = True; break
                d = a ^ b
                if len(d) == 2:
                    (c1, s1), (c2, s2) = tuple(d)
                    if c1 == c2 and s1 != s2 and len(a) == len(b):
                        p.discard(a); p.discard(b); p.add(a & b); changed = True; break
            if changed:
                break
    return frozenset(p)


def p_and_lit(p, cond, pol):
    out = set()
    for c in p:
        if (cond, not pol) in c:
            continue
        out.add(c | {(cond, pol)})
    return frozenset(out)


def p_or(p, q):
    return p_simplify(p | q)


def p_split_literal(p, q):
    """a literal (cond, pol) present in every conjunction of p whose negation is in every conjunction of q"""
    if not p or not q:
        return None
    common = frozenset.intersection(*p)
    for (c, s) in sorted(common):
        if all((c, not s) in cj for cj in q):
            return (c, s)
    return None


class Undefined(Exception):
    pass


class SymEx:
    """Executes functions of one lifted .inc file."""

    LOCAL_F = re.compile(r'^(x\d+h?|sd_0x[0-9a-f]+|fda|fdb)$')

    def __init__(self, inc_text, prefix, dag, ro_const=None, major=1, fast_zero=False, log=None):
        self.g = dag
        self.prefix = prefix                       # e.g. 'cit_nominal_'
        self.header, self.fns = cparse.split_functions(inc_text)
        self.parsed = {}
        self.ro_const = ro_const or {}             # byte address -> f64 bit pattern for words equal in all builds
        self.major = major
        self.fast_zero = fast_zero
        self.sfun = {}
        for ln in self.header:
            m = re.match(r'^#define SFUN_CALL_(\d+)\(\) (.*)$', ln)
            if m:
                self.sfun[int(m.group(1))] = cparse.parse_expr(m.group(2))
        self.warn = []
        self.nsel = 0
        self.lifted_axes = True

    # ---- helpers --------------------------------------------------------------------------------
    def fn_stmts(self, name):
        if name not in self.parsed:
            self.parsed[name] = cparse.parse_statements(self.fns[name])
        return self.parsed[name]

    def read(self, st, key):
        if isinstance(key, tuple) and key[0] == 'ISEL':
            return self.g.mk('sel', key[1], self.read(st, key[2]), self.read(st, key[3]))
        if key in st:
            v = st[key]
            if isinstance(v, tuple):
                raise Undefined('read of path-ambiguous %s: %s' % (key, v))
            return v
        g = self.g
        if isinstance(key, tuple):
            reg, off = key
            if reg == 'RO':
                if off in self.ro_const:
                    return g.cfbits(self.ro_const[off])
                return g.mk('in', 'RO', off)
            if reg in ('X', 'CMD', 'DW', 'Y', 'T', 'STOP'):
                return g.mk('in', reg, off >> 3)
            if reg in ('STK', 'STKI', 'STKW'):
                raise Undefined('read of unset stack cell %s' % (key,))
            if reg == 'M' and off == 0xba20:
                return g.cf(0.01)                  # stepSize0: 0.01 in every build (serl_build_desc.dt)
            if reg == 'B':
                self.warn.append('read of unset B cell 0x%x' % off)
                return g.mk('undef', 'B', off)
            raise Undefined(str(key))
        if self.LOCAL_F.match(key):
            return g.mk('undef', key, 0)
        return g.ci(0)

    def fmul(self, a, b):
        g = self.g
        if self.fast_zero:
            for u, v in ((a, b), (b, a)):
                if g.is_cf(u):
                    if g.nodes[u][1] == 0:
                        return g.cfbits(0)
                    if g.fval(u) == 1.0:
                        return v
        return g.mk('mul', a, b)

    def fadd(self, a, b):
        g = self.g
        if self.fast_zero:
            for u, v in ((a, b), (b, a)):
                if g.is_cf(u) and g.nodes[u][1] == 0:
                    return v
        return g.mk('add', a, b)

    # ---- expression evaluation -------------------------------------------------------------------
    MEM = {'B_D': 'B', 'X_D': 'X', 'DW_D': 'DW', 'Y_D': 'Y', 'CMD_D': 'CMD', 'OUT_D': 'OUT', 'TPTR_D': 'T',
           'XDOT_D': 'XDOT', 'RO_D': 'RO'}

    def memkey(self, name, args, st, fr):
        g = self.g
        off = self.ev(args[0], st, fr)
        if g.op(off) == 'isel' and name in self.MEM:
            n = g.nodes[off]
            return ('ISEL', n[1], (self.MEM[name], g.ival(n[2])), (self.MEM[name], g.ival(n[3])))
        if not g.is_ci(off):
            raise Undefined('non-constant address in %s' % name)
        off = g.ival(off) & M64
        if off >= 1 << 63:
            off -= 1 << 64
        if name in ('STK_D', 'STK_I', 'STK_W'):
            return ({'STK_D': 'STK', 'STK_I': 'STKI', 'STK_W': 'STKW'}[name], off)
        if name in ('P_rdi_D', 'P_rsi_D', 'P_rdx_D', 'P_r8_D'):
            return ('STK', fr[name[:-2].lower().replace('p_', 'p_')] + off)
        if name == 'SU_D':
            return ('B', fr['su'] + off)
        if name == 'SY_D':
            return ('B', fr['sy'] + off)
        if name == 'M_D':
            return {0x40: ('STOP', 0)}.get(off, ('M', off))
        return (self.MEM[name], off)

    def as_f(self, n):
        """int node -> the f64 with the same bits"""
        g = self.g
        if g.ty[n] == 'f':
            return n
        if g.is_ci(n):
            return g.cfbits(g.ival(n))
        if g.op(n) == 'bits':
            return g.nodes[n][1]
        raise Undefined('cannot view %s as f64' % (g.nodes[n],))

    def as_i(self, n):
        g = self.g
        if g.ty[n] == 'i':
            return n
        if g.ty[n] == 'f':
            return g.mk('bits', n)
        raise Undefined('bool as int')

    def ev(self, e, st, fr):
        g = self.g
        k = e[0]
        if k == 'num':
            return g.cf(e[1]) if isinstance(e[1], float) else g.ci(e[1])
        if k == 'id':
            if e[1] == 'NULL':
                return g.ci(0)
            return self.read(st, e[1])
        if k == 'cast':
            v = self.ev(e[2], st, fr)
            t = e[1]
            if t == 'double':
                if g.ty[v] == 'f':
                    return v
                return g.mk('i2d', v)
            if g.ty[v] == 'b':
                raise Undefined('cast of bool')
            if g.ty[v] == 'f':
                if g.is_cf(v):
                    return g.ci(int(g.fval(v)))
                raise Undefined('float->int cast')
            if g.is_ci(v):
                x = g.ival(v)
                bits = int(re.search(r'\d+', t).group())
                x &= (1 << bits) - 1
                if t.startswith('int') and x >> (bits - 1):
                    x -= 1 << bits
                return g.ci(x)
            return v          # symbolic ints are 64-bit patterns / the tick counter: casts are value preserving here
        if k == 'un':
            v = self.ev(e[2], st, fr)
            if e[1] == '!':
                return g.mk('bnot', self.as_b(v))
            if e[1] == '~':
                if g.is_ci(v):
                    return g.ci((~g.ival(v)) & M64)
                return g.mk('bitsnot', self.as_f(v))
            if e[1] == '-':
                if g.is_ci(v):
                    return g.ci(-g.ival(v))
                return g.mk('neg', v)
        if k == 'tern':
            c = self.as_b(self.ev(e[1], st, fr))
            a, b = self.ev(e[2], st, fr), self.ev(e[3], st, fr)
            if g.is_ci(a) and g.is_ci(b):
                if (g.ival(a) & M64) == M64 and g.ival(b) == 0:
                    return g.mk('bits', g.mk('fmask', c))
                if g.ival(a) == 1 and g.ival(b) == 0:
                    return g.mk('b2i', c)                       # setcc
                if c == g.TRUE:
                    return a
                if c == g.FALSE:
                    return b
                raise Undefined('int ternary')
            return g.mk('sel', c, a, b)
        if k == 'bin':
            o = e[1]
            if o in ('||', '&&'):
                a, b = self.as_b(self.ev(e[2], st, fr)), self.as_b(self.ev(e[3], st, fr))
                return g.mk('bor' if o == '||' else 'band', a, b)
            a, b = self.ev(e[2], st, fr), self.ev(e[3], st, fr)
            ta, tb = g.ty[a], g.ty[b]
            if ta == 'f' and tb == 'f':
                if o == '+':
                    return self.fadd(a, b)
                if o == '*':
                    return self.fmul(a, b)
                if o in ('-', '/'):
                    return g.mk({'-': 'sub', '/': 'div'}[o], a, b)
                cmpop = {'>': 'gt', '>=': 'ge', '<': 'lt', '<=': 'le', '==': 'eq', '!=': 'ne'}[o]
                if cmpop == 'ne' and a == b:
                    return g.mk('unord', a, a)
                return g.mk(cmpop, a, b)
            if ta == 'i' and tb == 'i':
                if g.is_ci(a) and g.is_ci(b):
                    x, y = g.ival(a), g.ival(b)
                    if o in ('+', '-', '*', '&', '|', '^'):
                        return g.ci({'+': x + y, '-': x - y, '*': x * y, '&': x & y, '|': x | y, '^': x ^ y}[o])
                    if o == '<<':
                        return g.ci((x << y) & M64)
                    if o == '>>':
                        return g.ci(x >> y)
                    r = {'>': x > y, '>=': x >= y, '<': x < y, '<=': x <= y, '==': x == y, '!=': x != y}[o]
                    return g.TRUE if r else g.FALSE
                if o in ('&', '|'):
                    for u, v in ((a, b), (b, a)):
                        if g.op(u) == 'b2i' and g.is_ci(v):
                            if o == '&':
                                return u if (g.ival(v) & 1) else g.ci(0)
                            if g.ival(v) == 0:
                                return u
                if o in ('^', '&', '|'):
                    na, nb = g.nodes[a], g.nodes[b]
                    if o == '&' and na[0] == 'bitsnot':
                        return g.mk('bits', g.mk('fandn', na[1], self.as_f(b)))
                    if o == '&' and nb[0] == 'bitsnot':
                        return g.mk('bits', g.mk('fandn', nb[1], self.as_f(a)))
                    if o == '&' and a == b:
                        return a
                    return g.mk('bits', g.mk({'^': 'fxor', '&': 'fand', '|': 'for'}[o], self.as_f(a), self.as_f(b)))
                if o == '+':
                    for u, v in ((a, b), (b, a)):
                        if g.op(u) == 'isel' and g.is_ci(v):
                            nu = g.nodes[u]
                            return g.mk('isel', nu[1], g.ci(g.ival(nu[2]) + g.ival(v)), g.ci(g.ival(nu[3]) + g.ival(v)))
                    return g.mk('iadd', a, b)
                if o == '-' and g.is_ci(b):
                    return g.mk('iadd', a, g.ci(-g.ival(b)))
                if o in ('==', '!='):
                    r = g.mk('ieq', a, b)
                    return r if o == '==' else g.mk('bnot', r)
            raise Undefined('binary %s on %s,%s: %s' % (o, ta, tb, e))
        if k == 'call':
            return self.call(e, st, fr)
        raise Undefined(str(e))

    def as_b(self, n):
        g = self.g
        if g.ty[n] == 'b':
            return n
        if g.is_ci(n):
            return g.TRUE if g.ival(n) != 0 else g.FALSE
        raise Undefined('non-bool condition %s' % (g.nodes[n],))

    UNARY = {'LIFT_SQRT': 'sqrt', 'LIFT_EXP': 'exp', 'LIFT_LOG10': 'log10', 'LIFT_LOG': 'log', 'LIFT_SIN': 'sin',
             'LIFT_COS': 'cos', 'LIFT_TAN': 'tan', 'LIFT_ATAN': 'atan', 'LIFT_ASIN': 'asin', 'LIFT_ACOS': 'acos',
             'LIFT_FLOOR': 'floor'}

    def call(self, e, st, fr):
        g = self.g
        name, args = e[1], e[2]
        if name in self.MEM or name in ('SU_D', 'SY_D', 'M_D', 'STK_D', 'STK_I', 'STK_W', 'P_rdi_D', 'P_rsi_D', 'P_rdx_D', 'P_r8_D'):
            return self.read(st, self.memkey(name, args, st, fr))
        if name == 'MX0_D':
            return g.cf(float(fr['mode']))
        if name == 'M_I32':
            off = args[0][1]
            if off == 0xba48:
                return g.ci(self.major)
            if off == 0xba18:
                return g.mk('in_i', 'TICK')
            raise Undefined('M_I32 %x' % off)
        if name == 'RTINF_D':
            return g.cf(float('inf'))
        if name == 'RTMINF_D':
            return g.cf(float('-inf'))
        if name == 'RTNAN_D':
            return g.cfbits(0x7ff8000000000000)
        if name == 'u2d':
            return self.as_f(self.ev(args[0], st, fr))
        if name == 'd2u':
            return self.as_i(self.ev(args[0], st, fr))
        if name in self.UNARY:
            return g.mk(self.UNARY[name], self.ev(args[0], st, fr))
        if name in ('LIFT_POW', 'LIFT_ATAN2'):
            return g.mk('pow' if name == 'LIFT_POW' else 'atan2', self.ev(args[0], st, fr), self.ev(args[1], st, fr))
        if name == 'LIFT_ISNAN':
            v = self.ev(args[0], st, fr)
            return g.mk('unord', v, v)
        if name == 'LIFT_L2D':
            c = [a[1] for a in args[1:6]]
            return g.mk('l2d', c[0], c[1], c[2], c[3], c[4], self.ev(args[6], st, fr), self.ev(args[7], st, fr))
        if name == 'LIFT_L1D':
            return g.mk('l1d', args[1][1], args[2][1], args[4][1], self.ev(args[3], st, fr))
        if name.endswith('rt_powd_snf'):
            return g.mk('powsnf', self.ev(args[1], st, fr), self.ev(args[2], st, fr))
        raise Undefined('call %s' % name)

    # ---- statements with side effects -------------------------------------------------------------
    def lkey(self, lhs, st, fr):
        if lhs[0] == 'id':
            return lhs[1]
        if lhs[0] == 'call':
            return self.memkey(lhs[1], lhs[2], st, fr)
        raise Undefined('lvalue %s' % (lhs,))

    def coerce(self, key, v):
        g = self.g
        isf = (isinstance(key, tuple) and key[0] not in ('STKI', 'STKW')) or (not isinstance(key, tuple) and bool(self.LOCAL_F.match(key)))
        if isf and g.ty[v] == 'i':
            if g.is_ci(v):
                return g.cf(float(g.ival(v)))       # `x = 0;` style initialisers only
            raise Undefined('symbolic int stored to f64 %s' % (key,))
        return v

    def assign(self, lhs, v, st, fr):
        key = self.lkey(lhs, st, fr)
        if isinstance(key, tuple) and key[0] == 'ISEL':
            v = self.coerce(key[2], v)
            st[key[2]] = self.g.mk('sel', key[1], v, self.read(st, key[2]))
            st[key[3]] = self.g.mk('sel', key[1], self.read(st, key[3]), v)
            return
        st[key] = self.coerce(key, v)

    def bcell(self, e, st, fr):
        """&B_D(off) -> byte offset"""
        assert e[0] == 'addr' and e[1][0] == 'call' and e[1][1] == 'B_D', e
        return self.g.ival(self.ev(e[1][2][0], st, fr))

    def do_sfun(self, k, st, fr):
        e = self.sfun[k]
        name, args = e[1], e[2]
        if name.endswith('ac_atmos'):
            su, sy = self.bcell(args[1], st, fr), self.bcell(args[2], st, fr)
            mem = {kk: vv for kk, vv in st.items() if isinstance(kk, tuple)}
            out = self.run('ac_atmos', mem, dict(su=su, sy=sy))
            for kk, vv in out.items():
                if isinstance(kk, tuple):
                    st[kk] = vv
        elif name.endswith('ac_axes'):
            su, sy, mode = self.bcell(args[1], st, fr), self.bcell(args[2], st, fr), args[3][1]
            if self.lifted_axes:
                mem = {kk: vv for kk, vv in st.items() if isinstance(kk, tuple) and kk[0] == 'B'}
                out = self.run_concrete('ac_axes', mem, dict(su=su, sy=sy, mode=mode))
                for kk, vv in out.items():
                    if isinstance(kk, tuple) and kk[0] == 'B':
                        st[kk] = vv
            else:
                self.axes(st, su, sy, mode)
        elif name == 'cit_table3':
            u = [self.read(st, ('B', self.bcell(a, st, fr))) for a in args[1:4]]
            st[('B', self.bcell(args[4], st, fr))] = self.g.mk('table3', *u)
        else:
            raise Undefined('sfun %s' % name)

    def axes(self, st, su, sy, mode):
        """ac_axes mdlOutputs @0x103e0 (+ matmultiply @0x103a0), in the reference's operation order"""
        g = self.g
        u = [self.read(st, ('B', su + 8 * i)) for i in range(15)]
        s = {}; c = {}
        for i in range(4, 9):
            s[i], c[i] = g.mk('sc_sin', u[i]), g.mk('sc_cos', u[i])
        Z, ONE = g.cfbits(0), g.cf(1.0)
        neg = lambda x: g.mk('neg', x)
        mul, add, sub = self.fmul, self.fadd, lambda a, b: g.mk('sub', a, b)
        M1 = [c[5], neg(s[5]), Z, s[5], c[5], Z, Z, Z, ONE]
        M2 = [c[4], Z, neg(s[4]), Z, ONE, Z, s[4], Z, c[4]]
        s6s7, c6s7 = mul(s[6], s[7]), mul(c[6], s[7])
        M3 = [mul(c[7], c[8]), sub(mul(s6s7, c[8]), mul(c[6], s[8])), add(mul(c6s7, c[8]), mul(s[6], s[8])),
              mul(c[7], s[8]), add(mul(s6s7, s[8]), mul(c[6], c[8])), sub(mul(c6s7, s[8]), mul(c[8], s[6])),
              neg(s[7]), mul(s[6], c[7]), mul(c[6], c[7])]

        def mv(A, b, T=False):
            out = []
            for r in range(3):
                acc = Z
                for k in range(3):
                    acc = add(acc, mul(A[k * 3 + r] if T else A[r * 3 + k], b[k]))
                out.append(acc)
            return out
        v = u[12:15]
        if mode == 0:
            f0 = v; f1 = mv(M1, f0); f2 = mv(M2, f1); f3 = mv(M3, f2)
        elif mode == 1:
            f1 = v; f0 = mv(M1, f1, True); f2 = mv(M2, f1); f3 = mv(M3, f2)
        elif mode == 2:
            f2 = v; f1 = mv(M2, f2, True); f3 = mv(M3, f2); f0 = mv(M1, f1, True)
        else:
            f3 = v; f2 = mv(M3, f3, True); f1 = mv(M2, f2, True); f0 = mv(M1, f1, True)
        for i in range(3):
            st[('B', sy + 8 * i)] = f0[i]; st[('B', sy + 8 * (3 + i))] = f1[i]
            st[('B', sy + 8 * (6 + i))] = f2[i]; st[('B', sy + 8 * (9 + i))] = f3[i]

    def exec_stmt(self, s, st, fr):
        g = self.g
        k = s[0]
        if k == 'assign':
            self.assign(s[1], self.ev(s[2], st, fr), st, fr)
        elif k == 'addassign':
            key = self.lkey(s[1], st, fr)
            cur, v = self.read(st, key), self.ev(s[2], st, fr)
            st[key] = g.ci(g.ival(cur) + g.ival(v))
        elif k == 'expr':
            e = s[1]
            if e[0] == 'call' and e[1].startswith('SFUN_CALL_'):
                self.do_sfun(int(e[1].split('_')[-1]), st, fr)
            elif e[0] == 'call' and e[1] == 'LIFT_SINCOS':
                x = self.ev(e[2][0], st, fr)
                st[self.lkey(e[2][1][1], st, fr)] = g.mk('sc_sin', x)     # sincos(): glibc's differs from sin()/cos() by an ulp at times
                st[self.lkey(e[2][2][1], st, fr)] = g.mk('sc_cos', x)
            elif e[0] == 'call' and e[1].endswith('matmultiply'):
                base = []
                for a in e[2][1:4]:
                    assert a[0] == 'addr' and a[1][0] == 'call' and a[1][1] == 'STK_D', a
                    base.append(g.ival(self.ev(a[1][2][0], st, fr)))
                mem = {kk: vv for kk, vv in st.items() if isinstance(kk, tuple)}
                out = self.run_concrete('matmultiply', mem, dict(p_rdi=base[0], p_rsi=base[1], p_rdx=base[2]))
                for kk, vv in out.items():
                    if isinstance(kk, tuple):
                        st[kk] = vv
            elif e[0] in ('id', 'num'):
                pass
            else:
                raise Undefined('expr statement %s' % (e,))
        else:
            raise Undefined('stmt %s' % (s,))

    # ---- control flow ----------------------------------------------------------------------------
    def run(self, fname, state, fr):
        """symbolically execute lifted function `fname` from `state`; returns the merged exit state"""
        g = self.g
        stmts = self.fn_stmts(fname)
        label_of = {}
        blocks, names = [], []
        cur, curname = [], None
        started = False
        for s in stmts:
            if s[0] == 'label':
                if started:
                    blocks.append(cur); names.append(curname)
                cur, curname, started = [], s[1], True
                continue
            if not started:
                started = True
            cur.append(s)
            if s[0] in ('goto', 'if', 'return'):
                blocks.append(cur); names.append(curname)
                cur, curname, started = [], None, False
        if started:
            blocks.append(cur); names.append(curname)
        for i, n in enumerate(names):
            if n is not None:
                label_of[n] = i
        nb = len(blocks)
        succ = []
        for i, b in enumerate(blocks):
            t = b[-1] if b else None
            if t and t[0] == 'goto':
                succ.append([label_of[t[1]]])
            elif t and t[0] == 'if':
                succ.append([label_of[t[2]], i + 1])
            elif t and t[0] == 'return':
                succ.append([])
            else:
                succ.append([i + 1] if i + 1 < nb else [])
        # topological order (Kahn) over the blocks reachable from the entry; the CFG must be acyclic
        reach, work = {0}, [0]
        while work:
            for n in succ[work.pop()]:
                if n not in reach:
                    reach.add(n); work.append(n)
        indeg = {b: 0 for b in reach}
        for b in reach:
            for n in succ[b]:
                indeg[n] += 1
        order, ready = [], [0]
        while ready:
            ready.sort(reverse=True)
            b = ready.pop()
            order.append(b)
            for n in succ[b]:
                indeg[n] -= 1
                if indeg[n] == 0:
                    ready.append(n)
        if len(order) != len(reach):
            done = set(order)
            left = [b for b in reach if b not in done]
            # walk predecessors inside the leftover set until a block repeats -> one concrete cycle
            pred = {b: [p for p in left if b in succ[p]] for b in left}
            cur, seen = left[0], []
            while cur not in seen:
                seen.append(cur)
                cur = pred[cur][0] if pred[cur] else cur
                if not pred[seen[-1]]:
                    break
            cyc = seen[seen.index(cur):] if cur in seen else seen
            raise Undefined('cycle in CFG of %s: %s' % (fname, ' <- '.join(str(names[b]) + '#%d' % b for b in cyc[:12])))
        incoming = {0: [(PTRUE, state)]}
        exits = []
        for b in order:
            inc = incoming.pop(b, [])
            if not inc:
                continue
            pred, st = self.merge(inc)
            blk = blocks[b]
            term = blk[-1] if blk and blk[-1][0] in ('goto', 'if', 'return') else None
            body = blk[:-1] if term else blk
            for s in body:
                self.exec_stmt(s, st, fr)
            if term is None:
                if succ[b]:
                    incoming.setdefault(succ[b][0], []).append((pred, st))
                else:
                    exits.append((pred, st))
            elif term[0] == 'goto':
                incoming.setdefault(succ[b][0], []).append((pred, st))
            elif term[0] == 'return':
                exits.append((pred, st))
            else:
                c = self.as_b(self.ev(term[1], st, fr))
                if c == g.TRUE:
                    incoming.setdefault(succ[b][0], []).append((pred, st))
                elif c == g.FALSE:
                    incoming.setdefault(succ[b][1], []).append((pred, st))
                else:
                    # normalise to a positive literal
                    pol = True
                    while g.op(c) == 'bnot':
                        c = g.nodes[c][1]; pol = not pol
                    pt, pf = p_and_lit(pred, c, pol), p_and_lit(pred, c, not pol)
                    if pt:
                        incoming.setdefault(succ[b][0], []).append((pt, dict(st)))
                    if pf:
                        incoming.setdefault(succ[b][1], []).append((pf, st))
        pred, st = self.merge(exits)
        if pred != PTRUE:
            self.warn.append('%s: exit predicate does not simplify to TRUE (%d conjunctions)' % (fname, len(pred)))
        return st

    def run_concrete(self, fname, st, fr):
        """execute a lifted function whose every branch condition folds to a constant (loops allowed)"""
        g = self.g
        stmts = self.fn_stmts(fname)
        labels = {s[1]: i for i, s in enumerate(stmts) if s[0] == 'label'}
        pc, n = 0, 0
        while pc < len(stmts):
            s = stmts[pc]
            n += 1
            if n > 200000:
                raise Undefined('runaway loop in %s' % fname)
            k = s[0]
            if k == 'label':
                pc += 1
            elif k == 'goto':
                pc = labels[s[1]]
            elif k == 'if':
                c = self.as_b(self.ev(s[1], st, fr))
                if c == g.TRUE:
                    pc = labels[s[2]]
                elif c == g.FALSE:
                    pc += 1
                else:
                    raise Undefined('symbolic branch in %s' % fname)
            elif k == 'return':
                break
            else:
                self.exec_stmt(s, st, fr)
                pc += 1
        return st

    def pred_node(self, p):
        g = self.g
        acc = g.FALSE
        for cj in sorted(p, key=lambda c: sorted(c)):
            t = g.TRUE
            for (c, s) in sorted(cj):
                t = g.mk('band', t, c if s else g.mk('bnot', c))
            acc = g.mk('bor', acc, t)
        return acc

    def merge(self, inc):
        g = self.g
        inc = list(inc)
        while len(inc) > 1:
            # prefer a pair whose union simplifies
            best = None
            for i in range(len(inc)):
                for j in range(i + 1, len(inc)):
                    u = p_or(inc[i][0], inc[j][0])
                    gain = len(inc[i][0]) + len(inc[j][0]) - len(u)
                    if best is None or gain > best[0]:
                        best = (gain, i, j, u)
            _, i, j, u = best
            (p1, s1), (p2, s2) = inc[i], inc[j]
            lit = p_split_literal(p1, p2)
            if lit is not None:
                cn = lit[0] if lit[1] else g.mk('bnot', lit[0])
            else:
                lit2 = p_split_literal(p2, p1)
                if lit2 is not None:
                    cn = g.mk('bnot', lit2[0]) if lit2[1] else lit2[0]
                else:
                    cn = self.pred_node(p1)
            out = {}
            for key in sorted(set(s1) | set(s2), key=str):      # deterministic node numbering
                a, b = s1.get(key), s2.get(key)
                if a is None or b is None:
                    if isinstance(key, tuple):
                        # a cell written on one path only: on the other path it keeps its incoming value
                        other = self.read({}, key) if key[0] != 'B' else None
                        if other is None:
                            out[key] = a if a is not None else b
                            continue
                        a = other if a is None else a
                        b = other if b is None else b
                    else:
                        out[key] = a if a is not None else b
                        continue
                if a == b:
                    out[key] = a
                elif isinstance(a, tuple) or isinstance(b, tuple):
                    out[key] = ('ambiguous', a, b)
                elif g.ty[a] == g.ty[b] == 'f':
                    out[key] = g.mk('sel', cn, a, b)
                    self.nsel += 1
                elif g.ty[a] == g.ty[b] == 'i':
                    if g.is_ci(a) and g.is_ci(b):
                        out[key] = g.mk('isel', cn, a, b)      # e.g. a pointer to one of the two Derivative-block banks
                    else:
                        try:
                            out[key] = g.mk('bits', g.mk('sel', cn, self.as_f(a), self.as_f(b)))
                        except Undefined:
                            out[key] = ('ambiguous', a, b)
                else:
                    out[key] = ('ambiguous', a, b)
            inc = [x for k2, x in enumerate(inc) if k2 not in (i, j)] + [(u, out)]
        return inc[0]
