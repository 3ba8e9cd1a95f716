#!/usr/bin/env python3
"""Issue floor and dependency floor of ONE model evaluation, from the DAG the kernels are generated from.

  python tools/dag/critical_path.py [variant] [--latency gpurun_out/<tag>/valu_latency.json]     -> one JSON line

Two lower bounds on the time of an evaluation on a team of four wavefronts, one per SIMD (what bench.py prints next to
the measured 23 us per env step; six evaluations + one ODE5 combination make an env step):

  issue floor       every live node costs at least one VALU instruction, a wave64 VALU instruction occupies its SIMD for
                    4 cycles whatever its lanes hold, and four SIMDs share the work:
                        sum over nodes of instr(op) x 4 cycles / 4 SIMDs
                    with instr(op) = 1 for + - x, compares, logic; 2 for an f64 select (two v_cndmask); the measured
                    instruction count of the compiler's IEEE division / sqrt expansion; the lane-parallel phases (index
                    search, interpolation passes, the libm bodies, table3) at their INSTRUCTION counts (PHASE_INSTR below:
                    counted in the ISA of the shipped kernel, one pass serves all tables of a round) x 4 cycles.
                    A division by a literal that tools/dag/constdiv.py proves counts 4 instructions (citw_div_const),
                    any other the 13 of the IEEE expansion.  Reported twice: for the FULL DAG (every node, every libm
                    body -- what an evaluation costs when every Switch block of the model is open), and for the TRIMMED
                    flight condition (`issue_floor_trimmed_*`: without the nodes the lazy select operands skip there --
                    codegen.find_gates -- and without the libm bodies whose guard is closed: exp, log10 at 2 000 m) --
                    the floor of what the benchmark's episodes actually execute, and the one `issue_floor_frac` of the
                    bench line is taken against.
  dependency floor  the longest chain of dependent operations, each at its measured dependent-issue latency
                    (tools/valu_latency.hip): no partition of the DAG over wavefronts can finish sooner.  A table look-up
                    counts with ITS OWN dependent chain -- input to the lanes, breakpoints, table values: three LDS round
                    trips, then (z1-z0)/(x1-x0)*(u-x0)+z0 along x and again along y: two divisions and six additions /
                    multiplications (one division and three for a 1-D table) -- not with the measured time of the current
                    implementation's look-up phase (round 2 did that, which made the floor self-referential).
"""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_dag, codegen

LEAF = ('cf', 'ci', 'in', 'in_i', 'true', 'false', 'undef')
# instructions per node (VALU issue slots)
INSTR = dict(add=1, sub=1, mul=1, neg=1, fabs=1, sel=2, gt=1, ge=1, lt=1, le=1, eq=1, ne=1, band=1, bor=1, bnot=1, unord=1,
             div=13, sqrt=15)
# lane-parallel phases of the wave-cooperative kernels: instructions per evaluation on the wave that runs them, counted in the
# ISA of the shipped team kernel (hipcc -S): index search over 22-entry rows / 13-entry rows, one 2-D pass, one 1-D pass per
# round; the ocml bodies the evaluation calls once each (sincos, tan, exp, log10, pow: the calls of one function share a body);
# table3.  A dependent chain through a libm body: ~40 dependent f64 operations (polynomial + reconstruction), pow twice that.
# Round 4: sincos / tan / pow are the short bodies of serl_amd/csrc/citation_libm.h (hipcc -S of one call each: 70 / sincos + one
# division / 130 instructions; ocml's: 189 / 209 / 247 in the same harness), exp / log10 ocml's.
PHASE_INSTR = dict(search_r1=70, l2d_r1=77, l1d_r1=32, search_r2=45, l2d_r2=77, l1d_r2=32, sincos=70, tan=13, exp=60, log10=120,
                   pow=130, table3=186)
LIBM_CHAIN_OPS = dict(pow=80, powsnf=80)
LIBM = ('sc_sin', 'sc_cos', 'sin', 'cos', 'tan', 'exp', 'log10', 'log', 'atan', 'pow', 'powsnf')


def main():
    variant = next((a for a in sys.argv[1:] if not a.startswith('--')), 'nominal')
    lat = dict(dep_add_f64=8.0, dep_mul_f64=8.0, dep_div_f64=110.0, dep_sqrt_plus_add=120.0, cmp_select_add=20.0,
               lds_roundtrip_plus_add=93.0)
    if '--latency' in sys.argv:
        m = json.load(open(sys.argv[sys.argv.index('--latency') + 1]))
        lat.update({k: (v[1] if isinstance(v, list) else v) for k, v in m.items() if k != 'what'})
    gen = codegen.Gen(variant)
    g, res = gen.g, gen.res
    roots = list(res[1]['outs'].values())
    seen, stack = set(), list(roots)
    while stack:
        n = stack.pop()
        if n in seen:
            continue
        seen.add(n)
        stack.extend(build_dag.children(g, n))
    census = collections.Counter(g.nodes[n][0] for n in seen if g.nodes[n][0] not in LEAF)

    def instr(n):
        t = g.nodes[n]
        if t[0] == 'div' and g.nodes[t[2]][0] == 'cf' and gen.const_div_ok(g.nodes[t[2]][1]):
            return 4          # reciprocal multiply + two fma + div_fixup (citw_div_const), proved per divisor
        return INSTR.get(t[0], 1)
    glue = [n for n in seen if g.nodes[n][0] not in LEAF + LIBM + ('l2d', 'l1d', 'table3')]
    const_divs = sum(1 for n in glue if g.nodes[n][0] == 'div' and instr(n) == 4)
    glue_instr = sum(instr(n) for n in glue)
    issue_cycles = glue_instr * 4 + 4 * sum(PHASE_INSTR.values())
    # ... and what the trimmed flight condition executes: gated nodes are skipped, guarded libm calls whose condition is closed too
    cold = set(gen.cold) & set(glue)
    phases_trim = dict(PHASE_INSTR)
    closed = []
    loc = getattr(gen, 'trim_loc', None)
    for j, (cnode, pol) in gen.call_guard.items():
        fn = gen.libm_calls[j][0][0]
        if loc is not None and bool(loc['v%d' % cnode]) != bool(pol) and fn in phases_trim:
            others = [k for k in range(len(gen.libm_calls)) if k != j and gen.libm_calls[k][0][0] == fn
                      and not (k in gen.call_guard and bool(loc['v%d' % gen.call_guard[k][0]]) != bool(gen.call_guard[k][1]))]
            if not others:          # (the calls of one function share a body: it is skipped only if every call of it is)
                closed.append(fn)
                phases_trim.pop(fn)
    glue_instr_trim = sum(instr(n) for n in glue if n not in cold)
    issue_cycles_trim = glue_instr_trim * 4 + 4 * sum(phases_trim.values())
    # dependent-latency weights (cycles)
    L_add, L_mul = lat['dep_add_f64'], lat['dep_mul_f64']
    W = dict(add=L_add, sub=L_add, mul=L_mul, neg=4, fabs=4, div=lat['dep_div_f64'], sqrt=lat['dep_sqrt_plus_add'] - L_add,
             sel=lat['cmp_select_add'] - L_add)
    L_lds = lat['lds_roundtrip_plus_add'] - L_add
    W['l2d'] = 3 * L_lds + 2 * lat['dep_div_f64'] + 6 * L_add
    W['l1d'] = 3 * L_lds + 1 * lat['dep_div_f64'] + 3 * L_add
    W['table3'] = 2 * L_lds + 3 * lat['dep_div_f64'] + 12 * L_add
    for f in LIBM:
        W[f] = LIBM_CHAIN_OPS.get(f, 40) * L_add
    depth, via = {}, {}

    def dep(n):
        st = [n]
        while st:
            m = st[-1]
            if m in depth:
                st.pop(); continue
            ch = build_dag.children(g, m)
            miss = [c for c in ch if c not in depth]
            if miss:
                st.extend(miss); continue
            op = g.nodes[m][0]
            best = max(ch, key=lambda c: depth[c], default=None)
            depth[m] = (0 if op in LEAF else W.get(op, 4)) + (depth[best] if best is not None else 0)
            via[m] = best
            st.pop()
        return depth[n]
    end = max(roots, key=dep)
    chain = collections.Counter()
    n = end
    while n is not None:
        if g.nodes[n][0] not in LEAF:
            chain[g.nodes[n][0]] += 1
        n = via.get(n)
    out = dict(variant=variant, live_nodes=sum(census.values()), census=dict(census.most_common()),
               glue_instructions_min=glue_instr, issue_floor_cycles_per_eval_4_simds=issue_cycles / 4.0,
               divisions_by_proved_literals=const_divs, glue_instructions_min_trimmed=glue_instr_trim, gated_nodes_skipped_in_trim=len(cold),
               libm_bodies_closed_in_trim=sorted(closed), issue_floor_trimmed_cycles_per_eval_4_simds=issue_cycles_trim / 4.0,
               dependency_floor_cycles_per_eval=depth[end], critical_chain=dict(chain.most_common()),
               latencies_used={k: lat[k] for k in ('dep_add_f64', 'dep_mul_f64', 'dep_div_f64', 'dep_sqrt_plus_add', 'cmp_select_add',
                                                   'lds_roundtrip_plus_add')},
               chain_weights={k: round(W[k], 1) for k in ('l2d', 'l1d', 'table3', 'pow', 'sc_sin')},
               phase_instructions=PHASE_INSTR,
               clock_ghz=2.4)
    per_step = lambda c: (6 * c) / 2.4e3
    out['issue_floor_us_per_env_step'] = per_step(out['issue_floor_cycles_per_eval_4_simds'])
    out['issue_floor_trimmed_us_per_env_step'] = per_step(out['issue_floor_trimmed_cycles_per_eval_4_simds'])
    out['dependency_floor_us_per_env_step'] = per_step(out['dependency_floor_cycles_per_eval'])
    print(json.dumps(out))


if __name__ == '__main__':
    main()
