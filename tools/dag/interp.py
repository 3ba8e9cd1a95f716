"""CPU evaluation of the model DAG (validation of tools/dag against the oracle; build tooling).

The DAG is turned into straight-line Python (one assignment per node, IEEE-754 doubles, libm through
`math` = glibc, i.e. the same library the reference binary calls), wrapped in the reference's ODE5
macro step.  `check()` replays tests/golden/dyn_open_loop.npz (states returned by the reference's own
shared object) and expects bit-identical states.
"""
import math, os, sys, struct
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import symex, build_dag

INF = float('inf')


def lookup_index(x, n, u):
    lt = le = 0
    for i in range(n):
        v = x[i]
        lt += v < u
        le += v <= u
    idx = (le if u < 0.0 else lt) - 1
    return min(max(idx, 0), n - 2)


def l1d(ro, w0, xa, n, ya, u):
    x = ro[(xa >> 3) - w0:]; y = ro[(ya >> 3) - w0:]
    i = lookup_index(x, n, u)
    r = y[i + 1] - y[i]
    r = r / (x[i + 1] - x[i])
    r = r * (u - x[i])
    return r + y[i]


def l2d(ro, w0, xra, nr, xca, nc, za, u0, u1):
    xr = ro[(xra >> 3) - w0:]; xc = ro[(xca >> 3) - w0:]; z = ro[(za >> 3) - w0:]
    ix, iy = lookup_index(xr, nr, u0), lookup_index(xc, nc, u1)
    x0, x1 = xr[ix], xr[ix + 1]
    dx, wx = x1 - x0, u0 - x0
    z00, z10 = z[ix + nr * iy], z[ix + 1 + nr * iy]
    z01, z11 = z[ix + nr * (iy + 1)], z[ix + 1 + nr * (iy + 1)]
    a = z10 - z00; a = a / dx; a = a * wx; a = a + z00
    b = z11 - z01; b = b / dx; b = b * wx; b = b + z01
    y0 = xc[iy]
    dy = xc[iy + 1] - y0
    r = b - a; r = r / dy; r = r * (u1 - y0)
    return r + a


def t3_interval(tab, n, x):
    """interval the reference's cached linear walk ends on: clamp(max{i: tab[i] < x}, 0, n-2)"""
    i = -1
    for k in range(n):
        if tab[k] < x:
            i = k
    return min(max(i, 0), n - 2)


def table2(xt, yt, ix, iy, tab, M, x, y):
    rows = []
    for j in range(2):
        v0, v1 = tab[(ix + 0) * M + (iy + j)], tab[(ix + 1) * M + (iy + j)]
        x1, x0 = xt[ix + 1], xt[ix]
        if x == x1:
            rows.append(v1)
        else:
            d = v1 - v0; d = d * (x - x0); d = d / (x1 - x0); rows.append(d + v0)
    y1 = yt[iy + 1]
    if y == y1:
        return rows[1]
    y0 = yt[iy]
    d = rows[1] - rows[0]; d = d * (y - y0); d = d / (y1 - y0)
    return d + rows[0]


def table3(t3, u0, u1, u2):
    P1, P2, P3, P4 = t3[0:3], t3[3:7], t3[7:10], t3[10:46]
    i0, i1, i2 = t3_interval(P1, 3, u0), t3_interval(P2, 4, u1), t3_interval(P3, 3, u2)
    a = table2(P1, P2, i0, i1, P4[i2 * 12:], 4, u0, u1)
    b = table2(P1, P2, i0, i1, P4[i2 * 12 + 12:], 4, u0, u1)
    z1, z0 = P3[i2 + 1], P3[i2]
    if u2 == z1:
        return b
    d = b - a; d = d * (u2 - z0); d = d / (z1 - z0)
    return d + a


def fdiv(a, b):
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return float('nan')
        return math.copysign(INF, a) * math.copysign(1.0, b)


def safe(f, *a):
    """libm semantics for arguments a speculated (if-converted) path may see: NaN instead of an exception"""
    try:
        return f(*a)
    except (ValueError, OverflowError):
        return float('nan')


_libm = None


def _sincos(x):
    global _libm
    import ctypes
    if _libm is None:
        _libm = ctypes.CDLL('libm.so.6')
        _libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    s, c = ctypes.c_double(), ctypes.c_double()
    _libm.sincos(x, ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value


def sc_sin(x):
    return _sincos(x)[0]


def sc_cos(x):
    return _sincos(x)[1]


def fbits(x):
    return struct.unpack('<Q', struct.pack('<d', x))[0]


def bitsf(b):
    return struct.unpack('<d', struct.pack('<Q', b & symex.M64))[0]


def topo(g, roots):
    order, seen = [], set()
    stack = [(r, 0) for r in roots]
    for r in roots:
        stack = [(r, False)]
        while stack:
            n, done = stack.pop()
            if done:
                order.append(n); continue
            if n in seen:
                continue
            seen.add(n)
            stack.append((n, True))
            for c in build_dag.children(g, n):
                if c not in seen:
                    stack.append((c, False))
    return order


BIN = dict(add='+', sub='-', mul='*', gt='>', ge='>=', lt='<', le='<=', eq='==', ne='!=')


def pysrc(g, outs, fname):
    roots = list(outs.values())
    lines = ['def %s(X, CMD, DW, Y, T, TICK, RO, W0, T3):' % fname]
    for n in topo(g, roots):
        t = g.nodes[n]
        op = t[0]
        v = lambda k: 'v%d' % k
        if op == 'cf':
            e = 'bitsf(%d)' % t[1]
        elif op == 'ci':
            e = repr(t[1])
        elif op == 'in':
            if t[1] == 'RO':
                e = 'RO[%d - W0]' % (t[2] >> 3)
            else:
                e = {'X': 'X[%d]', 'CMD': 'CMD[%d]', 'DW': 'DW[%d]', 'Y': 'Y[%d]', 'T': 'T', 'STOP': 'STOP'}[t[1]]
                e = e % t[2] if '%' in e else e
        elif op == 'in_i':
            e = 'TICK'
        elif op in BIN:
            e = '(%s %s %s)' % (v(t[1]), BIN[op], v(t[2]))
        elif op == 'div':
            e = 'fdiv(%s, %s)' % (v(t[1]), v(t[2]))
        elif op == 'neg':
            e = '(-%s)' % v(t[1])
        elif op == 'fabs':
            e = 'abs(%s)' % v(t[1])
        elif op in ('sqrt', 'sin', 'cos', 'tan', 'exp', 'log10', 'log', 'atan', 'asin', 'acos', 'floor'):
            e = 'safe(math.%s, %s)' % (op, v(t[1]))
        elif op in ('sc_sin', 'sc_cos'):
            e = '%s(%s)' % (op, v(t[1]))
        elif op in ('pow', 'atan2'):
            e = 'safe(math.%s, %s, %s)' % (op, v(t[1]), v(t[2]))
        elif op == 'sel':
            e = '(%s if %s else %s)' % (v(t[2]), v(t[1]), v(t[3]))
        elif op == 'true':
            e = 'True'
        elif op == 'false':
            e = 'False'
        elif op == 'bnot':
            e = '(not %s)' % v(t[1])
        elif op == 'band':
            e = '(%s and %s)' % (v(t[1]), v(t[2]))
        elif op == 'bor':
            e = '(%s or %s)' % (v(t[1]), v(t[2]))
        elif op == 'unord':
            e = '(%s != %s or %s != %s)' % (v(t[1]), v(t[1]), v(t[2]), v(t[2]))
        elif op == 'l2d':
            e = 'l2d(RO, W0, %d, %d, %d, %d, %d, %s, %s)' % (t[1], t[2], t[3], t[4], t[5], v(t[6]), v(t[7]))
        elif op == 'l1d':
            e = 'l1d(RO, W0, %d, %d, %d, %s)' % (t[1], t[2], t[3], v(t[4]))
        elif op == 'table3':
            e = 'table3(T3, %s, %s, %s)' % (v(t[1]), v(t[2]), v(t[3]))
        elif op == 'iadd':
            e = '(%s + %s)' % (v(t[1]), v(t[2]))
        elif op == 'i2d':
            e = 'float(%s)' % v(t[1])
        elif op in ('fxor', 'fand', 'for'):
            e = 'bitsf(fbits(%s) %s fbits(%s))' % (v(t[1]), {'fxor': '^', 'fand': '&', 'for': '|'}[op], v(t[2]))
        elif op == 'fandn':
            e = 'bitsf((~fbits(%s)) & fbits(%s))' % (v(t[1]), v(t[2]))
        elif op == 'fmask':
            e = 'bitsf(%d if %s else 0)' % (symex.M64, v(t[1]))
        else:
            raise NotImplementedError(op)
        lines.append('    v%d = %s' % (n, e))
    lines.append('    return {%s}' % ', '.join('%r: v%d' % (k, n) for k, n in outs.items()))
    return '\n'.join(lines)


A = [0.2, 0.3, 0.8, 0.8888888888888888, 1.0, 1.0]
Bt = [[0.2], [0.075, 0.225], [0.9777777777777777, -3.7333333333333334, 3.5555555555555554],
      [2.9525986892242035, -11.595793324188385, 9.822892851699436, -0.2908093278463649],
      [2.8462752525252526, -10.757575757575758, 8.906422717743473, 0.2784090909090909, -0.2735313036020583],
      [0.09114583333333333, 0.0, 0.44923629829290207, 0.6510416666666666, -0.322376179245283, 0.13095238095238096]]


class Sim:
    def __init__(self, variant, build, fast_zero=False):
        g, res, _ = build_dag.build(variant, fast_zero=fast_zero)
        ns = dict(math=math, safe=safe, sc_sin=sc_sin, sc_cos=sc_cos, fdiv=fdiv, bitsf=bitsf, fbits=fbits, l2d=l2d, l1d=l1d, table3=table3)
        exec(pysrc(g, res[1]['outs'], 'ev_major'), ns)
        exec(pysrc(g, res[0]['outs'], 'ev_minor'), ns)
        self.major, self.minor = ns['ev_major'], ns['ev_minor']
        root = build_dag.ROOT
        z = np.load(os.path.join(root, 'serl_amd', 'data', 'citation_%s.npz' % build))
        self.ro = [float(x) for x in z['ro']]
        self.w0 = int(z['ro_base']) >> 3
        self.t3 = [float(x) for x in z['t3']]
        self.X = [float(x) for x in z['x0']]
        self.DW = [float(x) for x in z['dw0'][:29]]
        self.Y = [0.0] * 12
        self.t, self.tick, self.h = 0.0, 0, 0.01

    def step(self, cmd):
        h, t0 = self.h, self.t
        y = list(self.X)
        f = []
        o = self.major(self.X, cmd, self.DW, self.Y, self.t, self.tick, self.ro, self.w0, self.t3)
        out = [o['Y%d' % i] for i in range(12)]
        stop = o['STOP0']
        DWn = [o.get('DW%d' % i, self.DW[i]) for i in range(29)]
        f.append([o['XDOT%d' % i] for i in range(19)])
        for s in range(1, 6):
            hB = [b * h for b in Bt[s - 1]]
            X = []
            for i in range(19):
                acc = f[0][i] * hB[0]
                for j in range(1, s):
                    acc = acc + f[j][i] * hB[j]
                X.append(acc + y[i])
            t = stop if s == 5 else (hB[0] + t0 if s == 1 else h * A[s - 1] + t0)
            o = self.minor(X, cmd, DWn, self.Y, t, self.tick, self.ro, self.w0, self.t3)
            f.append([o['XDOT%d' % i] for i in range(19)])
        hB = [b * h for b in Bt[5]]
        for i in range(19):
            acc = f[0][i] * hB[0]
            for j in range(1, 6):
                acc = acc + f[j][i] * hB[j]
            self.X[i] = acc + y[i]
        self.DW, self.Y = DWn, out
        self.tick += 1
        self.t = stop
        return out


def check(variant, build, nsteps=3000, fast_zero=False):
    root = build_dag.ROOT
    gd = np.load(os.path.join(root, 'tests', 'golden', 'dyn_open_loop.npz'))
    cmds, xs = gd[build + '_cmd'], gd[build + '_x']
    sim = Sim(variant, build, fast_zero)
    worst = 0.0
    nbad = 0
    for k in range(min(nsteps, len(cmds))):
        out = sim.step([float(c) for c in cmds[k]])
        if k % 10 == 0:
            ref = xs[k // 10]
            if not np.array_equal(np.array(out), ref):
                nbad += 1
                d = np.max(np.abs(np.array(out) - ref) / (np.abs(ref) + 1e-300))
                worst = max(worst, d)
    print('%s/%s: %d checked rows, %d not bit-identical, worst rel %.3g' % (variant, build, min(nsteps, len(cmds)) // 10, nbad, worst))
    return nbad


if __name__ == '__main__':
    v = sys.argv[1] if len(sys.argv) > 1 else 'nominal'
    b = sys.argv[2] if len(sys.argv) > 2 else 'h2000_v90'
    check(v, b, int(sys.argv[3]) if len(sys.argv) > 3 else 3000, '--fast-zero' in sys.argv)
