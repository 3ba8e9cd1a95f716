"""The model DAG that the wave-cooperative HIP kernel is generated from (tools/dag: symbolic execution of the lifted
reference code, if-conversion, x*0 / x+0 folding) evaluated on the CPU and compared bit-for-bit with
  * the states the REFERENCE'S OWN shared object returned (tests/golden/dyn_open_loop.npz), and
  * the oracle's C restatement, every step,
for every dynamics code variant.  Also pins the committed generated sources to the generator."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools', 'dag'))

CASES = [('nominal', 'h2000_v90'), ('nominal', 'cg'), ('ice', 'ice'), ('cg_timed', 'cg_timed'), ('gust', 'gust'), ('test', 'test')]


@pytest.mark.parametrize('variant,build', CASES)
def test_dag_bit_identical_to_reference_library_and_oracle(golden, variant, build):
    import interp
    from oracle import dynamics
    g = golden('dyn_open_loop')
    cmds, xs = g[build + '_cmd'], g[build + '_x']
    sim = interp.Sim(variant, build, fast_zero=True)
    o = dynamics.CitationDynamics(build)
    for k in range(121):
        c = [float(v) for v in cmds[k]]
        a = np.array(sim.step(c))
        b = o.step(np.array(c))
        np.testing.assert_array_equal(a, b, err_msg='%s/%s step %d vs oracle' % (variant, build, k))
        if k % 10 == 0:
            np.testing.assert_array_equal(a, xs[k // 10], err_msg='%s/%s step %d vs reference .so' % (variant, build, k))


def test_dag_large_commands_and_clock_offset():
    """full-deflection random commands (look-up extrapolation, saturations) and a model clock that starts late
    (the reference's initialize() leaves it running): still bit-identical to the oracle"""
    import interp
    from oracle import dynamics
    rng = np.random.default_rng(0)
    for variant, build, tick0 in (('nominal', 'h2000_v90', 0), ('cg_timed', 'cg_timed', 1990)):
        sim = interp.Sim(variant, build, fast_zero=True)
        sim.tick, sim.t = tick0, float(tick0) * 0.01
        o = dynamics.CitationDynamics(build)
        o.L.cit_set_clock(o.buf, tick0)
        for k in range(60):
            c = np.zeros(10); c[:3] = rng.uniform(-0.1745, 0.1745, 3)
            np.testing.assert_array_equal(np.array(sim.step(list(c))), o.step(c), err_msg='%s step %d' % (variant, k))


@pytest.mark.parametrize('variant', ['nominal', 'ice'])
def test_generated_sources_are_current(variant):
    import codegen
    text = codegen.Gen(variant).emit()
    have = open(os.path.join(ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s_wave.inc' % variant)).read()
    assert text == have, 'serl_amd/csrc/gen/citation_%s_wave.inc is stale: run python tools/dag/codegen.py' % variant


@pytest.mark.parametrize('variant', ['nominal', 'gust'])
def test_generated_team_sources_are_current(variant):
    import codegen_team
    text = codegen_team.TeamGen(variant).emit_team()
    have = open(os.path.join(ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s_team.inc' % variant)).read()
    assert text == have, 'serl_amd/csrc/gen/citation_%s_team.inc is stale: run python tools/dag/codegen_team.py' % variant


@pytest.mark.parametrize('variant', ['nominal', 'ice', 'cg_timed', 'gust', 'test'])
def test_generated_lane_sources_are_current(variant):
    import codegen_lane
    text = codegen_lane.LaneGen(variant).emit_lane()
    have = open(os.path.join(ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s_lane.inc' % variant)).read()
    assert text == have, 'serl_amd/csrc/gen/citation_%s_lane.inc is stale: run python tools/dag/codegen_lane.py nominal ice cg_timed gust test' % variant


@pytest.mark.parametrize('variant', ['nominal', 'ice', 'cg_timed', 'gust', 'test'])
def test_constant_divisions_are_proved_correctly_rounded(variant):
    """Every literal divisor the generated kernels divide by with `citw_div_const` (reciprocal multiply + fma correction,
    serl_amd/csrc/citation_wave.h) passes the per-divisor proof of tools/dag/constdiv.py: the x whose quotient lies within
    64 units of a rounding boundary are enumerated and the three-operation sequence is evaluated on them in exact rational
    arithmetic; a random spot check exercises the emulation itself against IEEE division."""
    import struct
    import build_dag, constdiv
    g, res, _ = build_dag.build(variant, fast_zero=True)
    consts = sorted({g.nodes[g.nodes[n][2]][1] for n in range(len(g.nodes)) if g.nodes[n][0] == 'div' and g.nodes[g.nodes[n][2]][0] == 'cf'})
    assert consts
    for bits in consts:
        c = struct.unpack('<d', struct.pack('<Q', bits))[0]
        ok, ncand = constdiv.verify(c)
        assert ok, 'divisor %r must keep the IEEE division' % c
        assert constdiv.spot_check(c, 3000)
    # the enumeration finds the near-midpoint dividends: for 288.15 the quotient of the first candidate is within 1e-15 of a midpoint
    X, d = constdiv.candidates(288.15, 8)[0]
    import math
    from fractions import Fraction
    m, e = math.frexp(288.15)
    C = int(m * (1 << 53))
    q = Fraction(X, C) * (1 << (52 if X >= C else 53))
    assert abs((q - int(q)) - Fraction(1, 2)) < Fraction(1, 10 ** 15)


@pytest.mark.parametrize('variant', ['nominal', 'ice', 'gust'])
def test_gated_cones_are_read_only_through_their_selects(variant):
    """Lazy select operands (tools/dag/codegen.py find_gates): a node emitted under `if (c == p)` may be read ONLY by nodes of the
    same gate or by a select on c that takes it as its p-operand (and not as the other one) -- checked here on the DAG,
    independently of the analysis that built the sets.  Gates never hold a root, a look-up, a libm result or their own
    condition."""
    import codegen, build_dag
    gen = codegen.Gen(variant)
    g = gen.g
    assert gen.gate_nodes, 'no gates found for %s' % variant
    users = {}
    for n in gen.order:
        for c in build_dag.children(g, n):
            users.setdefault(c, []).append(n)
    roots = set(gen.roots) | {gen.stop}
    total = 0
    for (c, pol), nodes in gen.gate_nodes.items():
        idx = 2 if pol == 'T' else 3
        assert c not in nodes
        for m in nodes:
            total += 1
            assert gen.gate[m] == (c, pol)
            assert m not in roots and g.nodes[m][0] not in ('l2d', 'l1d') + codegen.GATE_FN + codegen.GATE_LEAF and m not in gen.libm_slot
            for u in users.get(m, []):
                if u in nodes:
                    continue
                t = g.nodes[u]
                assert t[0] == 'sel' and t[1] == c and t[idx] == m and t[5 - idx] != m, (variant, c, pol, m, u, t[:4])
    assert total >= 100            # (nominal: 188 of 1 086 nodes)


@pytest.mark.parametrize('variant,suffix', [('nominal', ''), ('ice', ''), ('cg_timed', ''), ('gust', ''), ('test', ''), ('nominal', '6'), ('gust', '6')])
def test_every_cross_wavefront_read_of_the_generated_team_code_is_covered_by_its_flag(variant, suffix):
    """The hand-over protocol of gen/citation_<v>_team.inc, checked on the EMITTED TEXT, independently of the generator's own bookkeeping
    (a consumer that reads a slot before, or without, its producer's store would return a plausible stale value):
      * in front of barrier B1 a wavefront may read another wavefront's libm results g_m[CITW_MROW(q)][s] only behind a wait on the flag
        that the producer raises BEHIND its store of slot s -- g_flag[8 + q] for the results stored in front of its early flag,
        g_flag[q] (which also covers the early ones) for the rest;
      * the producer's stores of its result slots precede the matching raise, and a wavefront-scope fence stands between the stores and
        the wavefront's own loads of them (the lanes that made the calls are not the lanes that read);
      * behind B1 a value of the task graph g_y[CITW_YOFF + s] is read from another wavefront only through citw_pflag_wait_load on the
        producer's flag with the publication count the producer raises behind its store of that slot;
      * every derivative is stored behind B1 (row 0 of g_f is still being combined by a late wavefront in front of it);
      * behind B1, and only there, results of other wavefronts are read without a wait (the barrier orders them)."""
    import re
    text = open(os.path.join(ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s_team%s.inc' % (variant, suffix))).read()
    funcs = re.split(r'(?=static __device__ CITW_EVAL_INLINE double citw_\w+_team_eval_w\d+\()', text)[1:]
    assert len(funcs) == (6 if suffix == '6' else 7)
    # ---- producers: which flag covers which slot of g_m, which publication count covers which slot of g_y
    m_flag, y_pub = {}, {}
    for f in funcs:
        b = int(re.match(r'static __device__ CITW_EVAL_INLINE double citw_\w+_team_eval_w(\d+)\(', f).group(1))
        pre = f.split('CITW_TEAM_BARRIER1()')[0]
        stored = []
        for line in pre.splitlines():
            st = re.search(r'if \(l_ >= (\d+) && l_ < (\d+)\) \{ g_m\[CITW_MROW\((\d+)\)\]\[2 \* lane\] = r0_;', line)
            if st:
                assert int(st.group(3)) == b
                stored += [(b, 2 * j) for j in range(int(st.group(1)), int(st.group(2)))] + [(b, 2 * j + 1) for j in range(int(st.group(1)), int(st.group(2)))]
            sc = re.search(r'g_m\[CITW_MROW\((\d+)\)\]\[(\d+)\] = v\d+; g_m\[CITW_MROW\((\d+)\)\]\[(\d+)\] = 0\.0;', line)
            if sc:      # (round 5) a guarded call made wave-uniformly: the result and its unused second slot stored by the wavefront itself
                assert int(sc.group(1)) == b and int(sc.group(3)) == b and int(sc.group(4)) == int(sc.group(2)) + 1
                stored += [(b, int(sc.group(2))), (b, int(sc.group(4)))]
            rs = re.search(r'citw_flag_raise\((\d+),', line)
            if rs:
                fl = int(rs.group(1))
                assert fl in (b, 8 + b), 'wave %d raises flag %d' % (b, fl)
                for slot in stored:
                    m_flag.setdefault(slot, fl)          # the FIRST raise behind the store covers the slot
        post = f.split('CITW_TEAM_BARRIER1()')[1]
        last_y = None
        for line in post.splitlines():
            sy = re.search(r'g_y\[CITW_YOFF \+ (\d+)\] = ', line)
            if sy:
                last_y = int(sy.group(1))
            rp = re.search(r'citw_pflag_raise\((\d+), \(.*\) \* 16u \+ (\d+)u\)', line)
            if rp:
                assert int(rp.group(1)) == b and last_y is not None
                y_pub[last_y] = (b, int(rp.group(2)))
                last_y = None
    assert m_flag, 'no shared libm results found'
    # ---- consumers
    for f in funcs:
        b = int(re.match(r'static __device__ CITW_EVAL_INLINE double citw_\w+_team_eval_w(\d+)\(', f).group(1))
        pre, post = f.split('CITW_TEAM_BARRIER1()')
        post = post.split('CITW_TEAM_BARRIER2()')[0]
        assert 'g_f[' not in pre, 'wave %d stores a derivative in front of B1' % b
        waited = set()
        fenced = True
        for line in pre.splitlines():
            if re.search(r'g_m\[CITW_MROW\(%d\)\]\[2 \* lane\] = ' % b, line) or re.search(r'g_m\[CITW_MROW\(%d\)\]\[\d+\] = ' % b, line):
                fenced = False
            if 'CITW_WAVE_FENCE()' in line:
                fenced = True
            for w in re.finditer(r'citw_flag_wait(?:_load)?\((\d+),', line):
                waited.add(int(w.group(1)))
            for r in re.finditer(r'(?<![\w.])(?:&)?g_m\[CITW_MROW\((\d+)\)\]\[(\d+)\](?!\s*=[^=])', line):
                q, sl = int(r.group(1)), int(r.group(2))
                if sl >= 48:
                    continue                      # (argument slots: written and read by their own wavefront)
                if q == b:
                    assert fenced, 'wave %d reads its own libm result %d without a wavefront fence behind the store' % (b, sl)
                    continue
                fl = m_flag[(q, sl)]
                assert fl in waited or (fl == 8 + q and q in waited), 'wave %d reads g_m[%d][%d] in front of B1 without a wait on flag %d (waited: %s)' % (b, q, sl, fl, sorted(waited))
        pwaited = {}
        for line in post.splitlines():
            for w in re.finditer(r'citw_pflag_wait_load\((\d+), \(.*?\) \* 16u \+ (\d+)u, &g_y\[CITW_YOFF \+ (\d+)\]\)', line):
                q, k, sl = int(w.group(1)), int(w.group(2)), int(w.group(3))
                assert y_pub[sl][0] == q and k >= y_pub[sl][1], 'wave %d waits for g_y[%d] on (%d, %d), published as %s' % (b, sl, q, k, y_pub[sl])
                pwaited[q] = max(pwaited.get(q, 0), k)
            for r in re.finditer(r'(?<![&\w])g_y\[CITW_YOFF \+ (\d+)\](?!\s*=[^=])', line):
                sl = int(r.group(1))
                q, k = y_pub[sl]
                assert q == b or pwaited.get(q, 0) >= k, 'wave %d reads g_y[%d] behind B1 without the producer\'s flag (%d, %d)' % (b, sl, q, k)


def test_interpolation_over_precomputed_quotients_equals_the_reference_order(tmp_path):
    """Round 6 (lane-per-episode kernels): citation_leaves.h cit_lookup2d_at_s / cit_lookup1d_at_s read the x-direction quotient of a table interval from g_sl, where
    rollout_variant.inc left it when it staged the tables -- (z[ix + 1] - z[ix]) / (x[ix + 1] - x[ix]), the interpolation's own first two operations (rt_Lookup2D_Normal
    of the reference's library).  On the host, random tables and inputs (inside, outside, on both sides of zero): bit for bit against cit_lookup2d_at /
    cit_lookup1d_at; and the generated files' plan fits the LDS next to the tables."""
    import subprocess, re
    exe = str(tmp_path / 'lsh')
    subprocess.run(['g++', '-O2', '-ffp-contract=off', '-D_GNU_SOURCE', '-I', os.path.join(ROOT, 'serl_amd', 'csrc'),
                    os.path.join(ROOT, 'tests', 'tools', 'leaves_slopes_host.cpp'), '-o', exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    bad, n = (int(x) for x in r.stdout.split())
    assert r.returncode == 0 and bad == 0 and n == 200000, r.stdout
    for v in ('nominal', 'ice', 'cg_timed', 'gust', 'test'):
        text = open(os.path.join(ROOT, 'serl_amd', 'csrc', 'gen', 'citation_%s_lane.inc' % v)).read()
        words, tables = (int(x) for x in re.search(r'cit_%s_NSLOPE = (\d+), cit_%s_NSLOPE_TABLES = (\d+)' % (v, v), text).groups())
        rows = re.findall(r'^  \{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},$', text, re.M)
        assert len(rows) == tables and tables >= 60
        end = 0
        for kind, zw, nr, nc, xw, off in (tuple(int(x) for x in row) for row in rows):      # the windows of g_sl follow each other without gaps or overlaps
            assert off == end and kind in (1, 2)
            end += (nr - 1) * (nc if kind == 2 else 1)
        assert end == words and 8 * (12040 + words) <= 160 * 1024
        assert text.count('cit_lookup2d_at_s(') + text.count('cit_lookup1d_at_s(') >= 70
