"""Drive symex over one code variant: model + derivatives -> DAG with named outputs (major and minor)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import symex

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ro_constants(variant):
    """RO words that are bit-identical in every data build sharing this code variant -> compile-time constants"""
    builds = json.load(open(os.path.join(ROOT, 'serl_amd', 'data', 'builds.json')))
    datas = sorted({v['data'] for v in builds.values() if v['code'] == variant})
    arrs, base = [], None
    for d in datas:
        z = np.load(os.path.join(ROOT, 'serl_amd', 'data', 'citation_%s.npz' % d))
        arrs.append(z['ro'].view(np.uint64)); base = int(z['ro_base'])
    same = np.ones(len(arrs[0]), bool)
    for a in arrs[1:]:
        same &= (a == arrs[0])
    return {base + 8 * i: int(arrs[0][i]) for i in range(len(same)) if same[i]}, datas


def build(variant, fast_zero=False, fold_ro=True):
    text = open(os.path.join(ROOT, 'oracle', 'gen', 'citation_%s.inc' % variant)).read()
    roc, datas = ro_constants(variant)
    g = symex.Dag()
    res = {}
    for major in (1, 0):
        sx = symex.SymEx(text, 'cit_%s_' % variant, g, ro_const=roc if fold_ro else {}, major=major, fast_zero=fast_zero)
        st = sx.run('model', {}, {})
        mem = {k: v for k, v in st.items() if isinstance(k, tuple)}
        st2 = sx.run('derivatives', dict(mem), {})
        outs = {}
        for i in range(19):
            outs['XDOT%d' % i] = st2[('XDOT', 8 * i)]
        if major:
            for k, v in st.items():
                if isinstance(k, tuple) and k[0] in ('Y', 'DW', 'STOP', 'OUT', 'M', 'T'):
                    outs['%s%d' % (k[0], k[1] >> 3)] = v
        res[major] = dict(outs=outs, warn=sx.warn, nsel=sx.nsel)
    return g, res, datas


def reach(g, roots):
    seen, stack = set(), list(roots)
    while stack:
        n = stack.pop()
        if n in seen:
            continue
        seen.add(n)
        for a in g.nodes[n][1:]:
            if isinstance(a, int) and not isinstance(a, bool) and g.nodes[n][0] not in ('cf', 'ci', 'in', 'in_i', 'undef') \
                    and not (g.nodes[n][0] in ('l2d',) and False):
                stack.append(a)
    return seen


def children(g, n):
    t = g.nodes[n]
    op = t[0]
    if op in ('cf', 'ci', 'in', 'in_i', 'undef', 'true', 'false'):
        return []
    if op == 'l2d':
        return [t[6], t[7]]
    if op == 'l1d':
        return [t[4]]
    return [a for a in t[1:]]


if __name__ == '__main__':
    import collections
    variant = sys.argv[1] if len(sys.argv) > 1 else 'nominal'
    g, res, datas = build(variant, fast_zero='--fast-zero' in sys.argv)
    print('data builds:', datas, 'nodes created:', len(g.nodes))
    for major in (1, 0):
        r = res[major]
        print('major=%d outputs=%d selects=%d warnings=%d' % (major, len(r['outs']), r['nsel'], len(r['warn'])))
        for w in r['warn'][:10]:
            print('   ', w)
        roots = list(r['outs'].values())
        seen, stack = set(), list(roots)
        while stack:
            n = stack.pop()
            if n in seen: continue
            seen.add(n)
            stack.extend(children(g, n))
        cnt = collections.Counter(g.nodes[n][0] for n in seen)
        print('   live nodes', len(seen), dict(cnt.most_common()))
        # depth
        depth = {}
        W = dict(add=1, sub=1, mul=1, div=4, sqrt=4, sin=15, cos=15, tan=25, pow=40, exp=20, log10=25, powsnf=50, l2d=0, l1d=0, table3=30, sel=1)
        def dep(n):
            stack = [n]
            while stack:
                m = stack[-1]
                if m in depth: stack.pop(); continue
                ch = children(g, m)
                miss = [c for c in ch if c not in depth]
                if miss: stack.extend(miss); continue
                depth[m] = W.get(g.nodes[m][0], 0 if g.nodes[m][0] in ('cf','ci','in','in_i','true','false') else 1) + max([depth[c] for c in ch], default=0)
                stack.pop()
            return depth[n]
        print('   weighted critical path', max(dep(n) for n in roots))
