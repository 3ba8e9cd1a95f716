"""Per-evaluation comparison of the model DAG with the oracle's lifted C: derivative vector and every block
signal B, on the states the ODE5 stages actually visit along a golden command sequence (build tooling)."""
import sys, os, math, ctypes
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools', 'dag'))
import interp, build_dag, symex
sys.path.insert(0, build_dag.ROOT)
from oracle import dynamics

D = ctypes.POINTER(ctypes.c_double)


def main(variant='nominal', build='h2000_v90', nsteps=700):
    text = open(os.path.join(build_dag.ROOT, 'serl_amd/csrc/gen/citation_%s.inc' % variant)).read()
    roc, _ = build_dag.ro_constants(variant)
    g = symex.Dag()
    sx = symex.SymEx(text, 'cit_%s_' % variant, g, ro_const=roc, major=0)
    st = sx.run('model', {}, {})
    bmap = {('B%d' % (k[1] >> 3)): v for k, v in st.items() if isinstance(k, tuple) and k[0] == 'B' and not isinstance(v, tuple)}
    ns = dict(math=math, safe=interp.safe, sc_sin=interp.sc_sin, sc_cos=interp.sc_cos, fdiv=interp.fdiv, bitsf=interp.bitsf, fbits=interp.fbits, l2d=interp.l2d,
              l1d=interp.l1d, table3=interp.table3)
    exec(interp.pysrc(g, bmap, 'ev_b'), ns)
    gd = np.load(os.path.join(build_dag.ROOT, 'tests/golden/dyn_open_loop.npz'))
    cmds = gd[build + '_cmd']
    sim = interp.Sim(variant, build)
    o = dynamics.CitationDynamics(build)
    o2 = dynamics.CitationDynamics(build)
    L = o.L
    L.cit_eval.argtypes = [ctypes.c_void_p, D, D, ctypes.c_double, ctypes.c_int, D]
    calls = []
    orig = sim.minor
    def wrap(*a):
        r = orig(*a); calls.append((a, r)); return r
    sim.minor = wrap
    nb = int(np.load(os.path.join(build_dag.ROOT, 'serl_amd/data/citation_%s.npz' % dynamics.build_index()[build]['data']))['nB'])
    for k in range(nsteps):
        c = [float(x) for x in cmds[k]]
        calls.clear()
        a = np.array(sim.step(c)); b = o.step(np.array(c))
        for si, (args, r) in enumerate(calls):
            X = np.array(args[0]); cm = np.array(args[1]); xd = np.zeros(19)
            L.cit_eval(o2.buf, X.ctypes.data_as(D), cm.ctypes.data_as(D), float(args[4]), 0, xd.ctypes.data_as(D))
            mine = np.array([r['XDOT%d' % i] for i in range(19)])
            if not np.array_equal(mine, xd):
                print('step', k, 'stage', si + 1, 'xdot differs at', np.nonzero(mine != xd)[0])
                Bo = np.ctypeslib.as_array(L.cit_B(o2.buf), shape=(nb,)).copy()
                Bd = ns['ev_b'](*args)
                bad = [(i, Bo[i], Bd['B%d' % i]) for i in range(nb) if 'B%d' % i in Bd and not (Bo[i] == Bd['B%d' % i] or (Bo[i] != Bo[i] and Bd['B%d' % i] != Bd['B%d' % i]))]
                for x in bad[:60]:
                    print('  B[%d] @0x%x oracle %r dag %r' % (x[0], x[0] * 8, x[1], x[2]))
                return 1
        if not np.array_equal(a, b):
            print('state mismatch at step', k, 'without a per-stage xdot mismatch'); return 1
    print('%s/%s: %d steps, all stage derivatives bit-identical' % (variant, build, nsteps))
    return 0


if __name__ == '__main__':
    sys.exit(main(*sys.argv[1:3], *(int(x) for x in sys.argv[3:4])))
