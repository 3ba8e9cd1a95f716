#!/usr/bin/env python3
"""Lane-level SLP potential of the model DAG (analysis only).

  python tools/dag/slp_shapes.py [variant]        -> one JSON line + a short table

Every glue instruction of the team kernels computes ONE scalar in all 64 lanes.  If the DAG holds groups of ISOMORPHIC
sub-expressions (the three rows of a rotation, body-axis triples, the six derivative cones), a group of width w could be
computed by one instruction stream with its w instances in w lanes -- operands routed by DPP / v_readlane, literals that differ
between the instances as per-lane registers.  This tool measures how much of the TRIMMED evaluation (what the benchmark's flight
condition executes: the closed gates of codegen.find_gates left out) sits in such groups:

  shape(n)   the expression tree of node n with its leaves abstracted -- operation names and operand ORDER only; a leaf is a
             literal, an input (state / command / table constant), a look-up result, a libm result, or a node with MORE THAN ONE
             user (a value that has to exist anyway: the cut keeps the instances of a group lane-local -- nothing inside a
             shape is read from outside it)
  group      the maximal single-user trees with the same shape; width = number of trees, size = operations per tree

Reported: the histogram of (width, size) weighted by operations, the share of the trimmed glue in groups of width >= 3 and
size >= 2 (a lone isomorphic operation gains nothing: the routing costs more than it saves), and the largest groups.
"""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_dag, codegen

LEAF = ('cf', 'ci', 'in', 'in_i', 'true', 'false', 'undef')
CUT = ('l2d', 'l1d', 'table3', 'sc_sin', 'sc_cos', 'sin', 'cos', 'tan', 'exp', 'log10', 'log', 'atan', 'pow', 'powsnf')


def main():
    variant = next((a for a in sys.argv[1:] if not a.startswith('--')), 'nominal')
    gen = codegen.Gen(variant)
    g, res = gen.g, gen.res
    roots = list(res[1]['outs'].values()) + list(res[0]['outs'].values())
    live, stack = set(), list(roots)
    while stack:
        n = stack.pop()
        if n in live:
            continue
        live.add(n)
        stack.extend(build_dag.children(g, n))
    gated = set()
    for key, nodes in getattr(gen, 'gate_nodes', {}).items():
        gated |= set(nodes)
    glue = [n for n in live if g.nodes[n][0] not in LEAF and g.nodes[n][0] not in CUT and n not in gated]
    users = collections.Counter()
    for n in live:
        for c in build_dag.children(g, n):
            users[c] += 1
    for r in roots:
        users[r] += 1
    glue_set = set(glue)

    def is_leaf(n, top):
        return n not in glue_set or (not top and users[n] > 1)

    memo = {}

    def shape(n, top=True):
        """(shape string, operations) of the single-user tree rooted at n"""
        if is_leaf(n, top):
            return '_', 0
        if (n, top) in memo:
            return memo[(n, top)]
        t = g.nodes[n]
        parts, ops = [], 1
        for c in build_dag.children(g, n):
            s, k = shape(c, False)
            parts.append(s)
            ops += k
        memo[(n, top)] = ('%s(%s)' % (t[0], ','.join(parts)), ops)
        return memo[(n, top)]

    # tree roots: glue nodes that are leaves for their users (several users, a root, or a user outside the glue)
    in_tree_of_other = set()
    for n in glue:
        for c in build_dag.children(g, n):
            if c in glue_set and users[c] == 1:
                in_tree_of_other.add(c)
    tree_roots = [n for n in glue if n not in in_tree_of_other]
    groups = collections.defaultdict(list)
    for n in tree_roots:
        s, k = shape(n)
        groups[s].append((n, k))
    total_ops = sum(k for v in groups.values() for _, k in v)
    hist = collections.Counter()
    wide = 0
    big = []
    for s, v in groups.items():
        w, k = len(v), v[0][1]
        hist[(min(w, 8), min(k, 8))] += w * k
        if w >= 3 and k >= 2:
            wide += w * k
            big.append((w * k, w, k, s[:100]))
    big.sort(reverse=True)
    out = dict(variant=variant, live_nodes=len(live), trimmed_glue_nodes=len(glue), single_user_trees=len(tree_roots), shapes=len(groups),
               operations_in_trees=total_ops, operations_in_groups_width_ge3_size_ge2=wide,
               share_width_ge3_size_ge2=round(wide / max(total_ops, 1), 4),
               share_width_ge2_size_ge2=round(sum(len(v) * v[0][1] for v in groups.values() if len(v) >= 2 and v[0][1] >= 2) / max(total_ops, 1), 4),
               histogram_ops_by_width_and_size={'w%d%s_k%d%s' % (w, '+' if w == 8 else '', k, '+' if k == 8 else ''): c for (w, k), c in sorted(hist.items())},
               largest_groups=[dict(ops=o, width=w, size=k, shape=s) for o, w, k, s in big[:12]])
    print(json.dumps(out))
    if '--table' in sys.argv:
        print('\nwidth x size -> operations')
        for (w, k), c in sorted(hist.items()):
            print('  width %d%s size %d%s: %d' % (w, '+' if w == 8 else '', k, '+' if k == 8 else '', c))


if __name__ == '__main__':
    main()
