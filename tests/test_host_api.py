"""Host-side mirror of the reference interface (serl_amd/*.py) and the C-ABI surface, without a GPU:
  * the library loads and exports every entry point include/serl_amd.h declares (no compute calls);
  * the product refuses to run without a GPU instead of falling back to a CPU path;
  * Actor / NetSpec packing is the layout the kernel and the oracle read (shipped checkpoints);
  * metrics (calc_smoothness on torch.fft, calc_nMAE) against the reference's numbers in the golden files;
  * reference-signal tabulation against the shipped trajectories; PH-LAB mode names."""
import os, re, ctypes
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from serl_amd import build, _capi
    lib_path = build.build()
    hdr = open(os.path.join(ROOT, 'include', 'serl_amd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = sorted(set(re.findall(r'\b(serl_[a-z0-9_]+)\s*\(', hdr)))
    assert len(names) >= 12, names
    L = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(L, n), 'libserl_amd.so does not export %s' % n
    assert set(_capi.EXPORTS) <= set(names)
    L.serl_abi_version.restype = ctypes.c_int
    assert L.serl_abi_version() == int(re.search(r'#define SERL_ABI_VERSION (\d+)', hdr).group(1))
    L.serl_param_count.restype = ctypes.c_int
    assert L.serl_param_count(7, 32, 3, 3) == 3715 and L.serl_param_count(7, 72, 3, 3) == 16995


def test_documented_raw_binding_asserts_the_current_abi_version():
    """INTEGRATION.md section 2 is a binding a maintainer copies: the version literal it asserts must be the header's (round 5 shipped a v7 library
    with a document that asserted 6), and the family names of the Python binding must cover enum serl_kernel_family."""
    from serl_amd import _capi
    hdr = open(os.path.join(ROOT, 'include', 'serl_amd.h')).read()
    ver = int(re.search(r'#define SERL_ABI_VERSION (\d+)', hdr).group(1))
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    lits = [int(v) for v in re.findall(r'serl_abi_version\(\) == (\d+)', doc)]
    assert lits and all(v == ver for v in lits), (lits, ver)
    assert _capi.ABI_VERSION == ver
    fam = re.search(r'enum serl_kernel_family \{(.*?)\};', hdr, re.S).group(1)
    fam = re.sub(r'/\*.*?\*/', '', fam, flags=re.S)
    vals = {int(v): n for n, v in re.findall(r'SERL_FAMILY_([A-Z0-9_]+) = (\d+)', fam)}
    assert sorted(vals) == sorted(_capi.FAMILIES), (vals, _capi.FAMILIES)
    for v, n in vals.items():
        assert (_capi.FAMILIES[v] or 'none').upper() == n, (v, n, _capi.FAMILIES[v])


def test_no_cpu_fallback_in_the_product():
    import serl_amd
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError):
        serl_amd.RolloutEngine(0)
    src = ''.join(open(os.path.join(ROOT, 'serl_amd', f)).read() for f in os.listdir(os.path.join(ROOT, 'serl_amd'))
                  if f.endswith('.py'))
    assert 'import oracle' not in src and 'from oracle' not in src, 'the product must not touch the oracle'


def test_actor_packing_matches_the_oracle_layout(golden):
    import serl_amd
    from serl_amd.actor import NetSpec
    spec = NetSpec(7, 3, 32, 3, 'tanh')
    assert spec.param_count == 3715 and spec.row_stride % 4 == 0
    row = golden('actors')['serl50'][18]
    import argparse
    args = argparse.Namespace(hidden_size=32, num_layers=3, activation_actor='tanh', state_dim=7, action_dim=3,
                              device=torch.device('cpu'))
    a = serl_amd.Actor(args)
    sd, off = {}, 0
    for name, o, shape in spec.param_layout():
        n = int(np.prod(shape))
        sd[name] = torch.from_numpy(row[o:o + n].reshape(shape).copy())
    a.load_state_dict(sd)
    packed = serl_amd.pack_population([a])
    np.testing.assert_array_equal(packed[0, :3715].numpy(), row)
    # forward of the torch interface module == golden samples of the reference's own Actor
    g = golden('pop_serl50')
    obs = torch.from_numpy(g['obs_samples'].astype(np.float32))
    with torch.no_grad():
        out = a(obs).numpy()
    np.testing.assert_allclose(out, g['act_samples'][18], atol=5e-6)
    # GA genome = 2-D weights only (genetic_agent.py:131-141)
    assert sum(n for _, n in spec.genome_segments()) == 7 * 32 + 3 * 32 * 32 + 3 * 32


def test_metrics_against_reference_numbers(golden):
    from serl_amd import metrics
    from oracle import rollout as R
    NET = dict(state_dim=7, action_dim=3, hidden=32, num_layers=3, activation='tanh')
    w = golden('actors')['serl50'][[18, 0]]
    ref = golden('ref_base')['ref']
    o = R.rollout(w, NET, [0, 1], ref, t_max=80, traces=True)
    sm = metrics.calc_smoothness(torch.from_numpy(o['actions']), torch.from_numpy(o['length_steps'])).numpy()
    g = golden('pop_serl50')
    np.testing.assert_allclose(sm, g['smoothness'][[18, 0]], rtol=1e-4)      # the reference's scipy FFT numbers
    # calc_nMAE: closed form on a synthetic error history (base/core/utils.py:39-58)
    err = np.stack([np.full(100, 0.01), np.full(100, -0.02), np.linspace(-0.001, 0.001, 100)], 1)
    want = np.mean(np.abs(err).mean(0) / np.array([np.deg2rad(20), np.deg2rad(20), 3.14159 / 180])) * 100
    assert abs(metrics.calc_nMAE(err) - want) < 1e-12


def test_smoothness_full_pass_and_speculation():
    """The lean full-length pass (weighted 2-norm of the half spectrum) against the oracle's numpy restatement of base/core/utils.py:82-120,
    and the speculative variant: a hit returns the same numbers and a true flag, a miss is remembered per batch shape and forgotten after a hit."""
    from serl_amd import metrics
    from oracle.smoothness import calc_smoothness as ref_smoothness
    rng = np.random.default_rng(3)
    for N in (8001, 2001, 800, 57):
        a = (rng.normal(size=(3, N, 3)).cumsum(1) * 0.01)
        got = metrics.calc_smoothness(torch.from_numpy(a)).numpy()
        want = np.array([ref_smoothness(a[e]) for e in range(3)])
        np.testing.assert_allclose(got, want, rtol=1e-10)
    metrics._SPEC_MISS.clear()
    a = torch.from_numpy(rng.normal(size=(4, 301, 3)).cumsum(1) * 0.01)
    full = torch.full((4,), 301, dtype=torch.int32)
    sm, flag = metrics.calc_smoothness_speculative(a, full)
    assert metrics.smoothness_speculation_result(a, flag)
    np.testing.assert_allclose(sm.numpy(), metrics.calc_smoothness(a, full).numpy(), rtol=1e-13)
    short = torch.tensor([301, 120, 301, -301], dtype=torch.int32)      # (a negative length: the table ran out -- still the whole table)
    sm, flag = metrics.calc_smoothness_speculative(a, short)
    assert not metrics.smoothness_speculation_result(a, flag)            # the caller takes the general path ...
    assert metrics.calc_smoothness_speculative(a, full) is None          # ... and the next batch of this shape does not guess
    want = np.array([ref_smoothness(a[e, :n].numpy()) for e, n in enumerate([301, 120, 301, 301])])
    np.testing.assert_allclose(metrics.calc_smoothness(a, short).numpy(), want, rtol=1e-10)
    metrics._SPEC_MISS.clear()


def test_role_maps_and_pairing_sweep():
    """The role <-> wavefront maps of the team kernels are permutations of the seven roles with the actor on wavefront 7, and
    tools/sweep_roles.py enumerates every pairing of the roles on the four SIMDs exactly once (105), the shipped map first."""
    import re, subprocess, sys, glob
    for f in glob.glob(os.path.join(ROOT, 'serl_amd', 'csrc', 'rollout_team_*.hip')):
        t = open(f).read()
        for name in ('SERL_TEAM_ROLES', 'SERL_TEAMS_ROLES'):
            m = re.search(r'#define %s \{([0-9, ]+)\}' % name, t)
            if m is None:
                continue
            r = [int(v) for v in m.group(1).split(',')]
            assert len(r) == 16 and sorted(r[:7]) == list(range(7)) and r[7] == 7, (f, name, r)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'sweep_roles.py')], capture_output=True, text=True, check=True).stdout.split()
    assert len(out) == 105 and len(set(out)) == 105
    pair_sets = set()
    for m in out:
        v = int(m, 16)
        w = [(v >> (4 * i)) & 15 for i in range(8)]
        assert sorted(w) == list(range(8)) and w[7] == 7
        pair_sets.add(frozenset(frozenset((w[i], w[i + 4])) for i in range(4)))
    assert len(pair_sets) == 105


def test_reference_tables_and_mode_names(golden):
    from serl_amd import refsignals, builds
    ref = refsignals.tabulate(*refsignals.base_reference(80), 80)
    np.testing.assert_allclose(ref, golden('ref_base')['ref'], atol=2e-15)
    assert ref.shape == (8001, 3)
    t = refsignals.env_times(8001)
    assert t[0] == 0.0 and t[-1] >= 80.0 and t[-2] < 80.0          # accumulated t += 0.01 reaches t_max at k = 8000
    for mode, build in [('nominal', 'h2000_v90'), ('PHlab_attitude_be', 'h2000_v90'), ('ice', 'ice'), ('cg', 'cg'),
                        ('cg-for', 'cg_for'), ('cg-shift', 'cg_timed'), ('high-q', 'h2000_v150'), ('low-q', 'h10000_v90')]:
        assert builds.resolve_mode(mode)[0] == build, mode
    assert builds.resolve_mode('be')[1][0] == 0.3                      # envs/be/citation.py:73  cmd[0] *= 0.3
    with pytest.raises(ValueError):
        builds.resolve_mode('no-such-mode')


def test_batched_reference_tabulation_matches_the_host_tables():
    from serl_amd import refsignals
    rng = np.random.default_rng(5)
    tt = np.tile(np.linspace(0.0, 80, 6), (4, 1))
    a_th = rng.choice(np.linspace(-12, 12, 6), size=(4, 6)); a_th[:, 0] = 0.0
    a_ph = rng.choice(np.linspace(-10, 10, 6), size=(4, 6))
    got = refsignals.tabulate_batch_device(tt, a_th, a_ph, 8, 80).numpy()
    for e in range(4):
        want = refsignals.tabulate(refsignals.SmoothedStepSequence(tt[e], a_th[e], 8),
                                   refsignals.SmoothedStepSequence(tt[e], a_ph[e], 8), 80)
        np.testing.assert_allclose(got[e], want, rtol=0, atol=1e-15)
    assert got.shape == (4, 8001, 3) and got[0, -1, 0] != got[0, -2, 0]     # the trim drops out at the terminal sample


def test_store_transitions_follows_agent_evaluate():
    """base/core/agent.py:101-125: every stored step goes to the shared buffer and the agent's own buffer, cost-flagged
    steps also to its critical buffer; num_frames / gen_frames advance per step, num_episodes per stored episode."""
    from serl_amd.generation import store_transitions

    class Buf(list):
        def add(self, *t):
            self.append(t)

    class Agent:
        def __init__(self):
            self.buffer, self.critical_buffer = Buf(), Buf()
    rng = np.random.default_rng(0)
    rows = rng.normal(size=(5, 20)).astype(np.float32)
    rows[:, 18] = [0, 0, 0, 0, 1]          # done
    rows[:, 19] = [0, 1, 0, 1, 0]          # cost
    shared, ag, counters = Buf(), Agent(), {'num_frames': 10, 'num_episodes': 2}
    store_transitions(rows, ag, shared, counters)
    assert len(shared) == 5 and len(ag.buffer) == 5 and len(ag.critical_buffer) == 2
    s, a, ns, r, d = shared[4]
    assert s.dtype == np.float64 and s.shape == (7,) and a.shape == (3,) and ns.shape == (7,) and d == 1.0
    np.testing.assert_array_equal(s, rows[4, 0:7].astype(np.float64))
    np.testing.assert_array_equal(ag.critical_buffer[1][1], rows[3, 7:10])
    assert counters == {'num_frames': 15, 'gen_frames': 5, 'num_episodes': 3}
    store_transitions(rows[:2], Agent(), None, None)      # no shared buffer, no counters: still fine


def test_sensor_noise_table_draw_order():
    """envs/noise/citation.py:71-82 draws randn(3), randn(1), randn(1), randn(2) per step() from the legacy generator; one
    randn(n, 7) block of the same generator must yield the same stream (the Box-Muller cache lives in the generator)."""
    from serl_amd import builds
    rs = np.random.RandomState(42)
    rows = []
    for _ in range(6):
        a, b, c, d = rs.randn(3), rs.randn(1), rs.randn(1), rs.randn(2)
        rows.append(np.concatenate([3.0 * 10**(-5) + 6.3 * 10**(-4) * a, 4.0 * 10**(-10) * b,
                                    1.8 * 10**(-3) + 2.7 * 10**(-4) * c, 4.0 * 10**(-3) + 3.2 * 10**(-5) * d]))
    np.testing.assert_array_equal(builds.sensor_noise_table(5, np.random.RandomState(42)), np.array(rows))
    assert builds.has_sensor_noise('PHlab_attitude_noise') and builds.has_sensor_noise('gust') and not builds.has_sensor_noise('ice')


def test_gen_refs_draw_order_matches_reference_evaluate(golden):
    """refsignals.gen_refs against base/evaluation_utils.py:23-55 run in evaluate.py main()'s order with its seed 7
    (tests/golden/make_evalpop_golden.py): same amplitudes and step times for the theta and the phi reference."""
    import numpy as np
    from serl_amd import refsignals
    g = golden('evalpop')
    tt = np.linspace(0., 80, 6)
    np.random.seed(7)
    th = refsignals.gen_refs(80, tt, 12.0, num_trails=1)
    ph = refsignals.gen_refs(80, tt, 10.0, num_trails=1)
    np.testing.assert_array_equal(th[0].times, g['times'][0]); np.testing.assert_array_equal(th[0].amps, g['amps_theta'][0])
    np.testing.assert_array_equal(ph[0].times, g['times_phi'][0]); np.testing.assert_array_equal(ph[0].amps, g['amps_phi'][0])
    assert th[0].w == 8.0


class _ListBuf(list):
    def add(self, *t):
        self.append(t)


def run_sequence(golden, name, engine, mode):
    """Drive serl_amd.make_evaluate through one plan of tests/golden/make_seq_golden.py and compare with what the
    REFERENCE'S OWN Agent.evaluate produced on one env / one Agent: returns, lengths, the carried error, counters, buffer
    fills, the stored tuples, and the position of the np.random stream afterwards."""
    import types
    import numpy as np, torch
    import serl_amd
    from serl_amd import refsignals, actor as A
    g = golden('sequence')
    w = golden('actors')['serl50']
    args = types.SimpleNamespace(state_dim=7, action_dim=3, hidden_size=32, num_layers=3, activation_actor='tanh',
                                 smooth_fitness=False, noise_sd=0.2962183114680794, noise_clip=0.5)
    shared, counters = _ListBuf(), {}
    ref = refsignals.tabulate(*refsignals.base_reference(20), 20)
    evaluate = serl_amd.make_evaluate(args, mode=mode, t_max=20, ref_fn=lambda: ref, engine=engine, replay_buffer=shared,
                                      counters=counters)
    agents = {}
    for idx in set(int(p[0]) for p in g[name + '_plan']):
        ag = serl_amd.GeneticAgent(args, buffer=_ListBuf(), critical_buffer=_ListBuf())
        A.unpack_into(ag.actor, torch.from_numpy(w[idx]))
        agents[idx] = ag
    np.random.seed(int(g[name + '_seed']))
    for j, (idx, noisy, store) in enumerate(g[name + '_plan']):
        ep = evaluate(agents[int(idx)], bool(noisy), bool(store))
        fit, length, sm, n = g[name + '_ret'][j]
        assert len(ep.reward_lst) == int(n) and ep.length == length, (name, j)
        np.testing.assert_allclose(ep.fitness, fit, rtol=1e-5, err_msg='%s episode %d' % (name, j))
        np.testing.assert_allclose(ep.smoothness, sm, rtol=2e-3)
        assert (len(ep.state_history) == 0) == bool(store)           # agent.py:113-115: states kept for unstored episodes only
        cn = g[name + '_counters'][j]
        assert [counters.get('num_frames', 0), counters.get('gen_frames', 0), counters.get('num_episodes', 0)] == list(cn)
        nb = g[name + '_nbuf'][j]
        assert [len(shared), len(agents[int(idx)].buffer), len(agents[int(idx)].critical_buffer)] == list(nb), (name, j)
    rows = np.stack([np.concatenate([np.ravel(np.asarray(x, np.float64)) for x in t]) for t in shared])
    np.testing.assert_allclose(rows[:30], g[name + '_shared_head'], atol=5e-6)
    np.testing.assert_allclose(rows[-30:], g[name + '_shared_tail'], atol=2e-4)
    # an episode consumed exactly the reference's number of np.random draws
    np.testing.assert_array_equal(np.random.randn(4), g[name + '_after'])


import pytest as _pytest


@_pytest.mark.parametrize('name,mode', [('nominal', 'nominal'), ('gust', 'gust')])
def test_make_evaluate_sequence_vs_reference_agent_evaluate(golden, oracle_engine, name, mode):
    """(host logic of make_evaluate on the CPU: the rollout itself is served by the oracle through tests/conftest.py's
    OracleEngine; the GPU suite runs the same sequences through the HIP kernel)"""
    run_sequence(golden, name, oracle_engine, mode)


CONFIG_SEEDS = {'sym': 5101, 'sym_soft': 5111, 'sym_inc_n': 5102, 'full': 5103, 'att_inc': 5104, 'att_inc_soft': 5114,
                'full_inc_n': 5105, 'att_inc_n': 5106, 'full_be': 5107, 'sym_ice': 5108}


def run_config_episode(golden, name, engine):
    """serl_amd.make_evaluate on an env NAME of another configuration ('PHlab_symmetric_incremental', ...) against the
    reference's own Agent.evaluate in TRAINING mode (tests/golden/make_config_golden.py): the env draws its references itself
    (refsignals.training_references; one theta sequence for the symmetric configuration), exploration noise is randn(A) per
    step, the stored tuples are (obs S, action A, next_obs S, reward, done)."""
    import types
    import numpy as np, torch
    import serl_amd
    from serl_amd import actor as A
    g = golden('config')
    cfg, incr, S, Ad, H, L = (int(v) for v in g[name + '_cfg'])
    args = types.SimpleNamespace(state_dim=S, action_dim=Ad, hidden_size=H, num_layers=L, activation_actor=str(g[name + '_act']),
                                 smooth_fitness=False, noise_sd=0.2962183114680794, noise_clip=0.5)
    mode = 'PHlab_%s_%s' % ({0: 'attitude', 1: 'symmetric', 2: 'full'}[cfg], str(g[name + '_mode']))
    shared, counters = _ListBuf(), {}
    evaluate = serl_amd.make_evaluate(args, mode=mode, t_max=20, engine=engine, replay_buffer=shared, counters=counters)
    ag = serl_amd.GeneticAgent(args, buffer=_ListBuf(), critical_buffer=_ListBuf())
    A.unpack_into(ag.actor, torch.from_numpy(g[name + '_w']))
    np.random.seed(CONFIG_SEEDS[name])
    ep = evaluate(ag, bool(np.any(g[name + '_noise'])), True)
    fit, length, n, sm = g[name + '_ret']
    assert len(ep.reward_lst) == int(n) and ep.length == length
    np.testing.assert_allclose(ep.fitness, fit, rtol=1e-5)
    if abs(sm) > 1e-6:
        np.testing.assert_allclose(ep.smoothness, sm, rtol=2e-3)
    assert ep.actions.shape == (int(n), Ad)
    np.testing.assert_allclose(ep.actions, g[name + '_u'], rtol=1e-5, atol=1e-7)
    assert counters == {'num_frames': int(n), 'gen_frames': int(n), 'num_episodes': 1}
    assert len(shared) == len(ag.buffer) == int(n) and len(ag.critical_buffer) == int(g[name + '_cost'].sum())
    rows = np.stack([np.concatenate([np.ravel(np.asarray(x, np.float64)) for x in t]) for t in shared])
    np.testing.assert_allclose(rows, g[name + '_rows'], rtol=2e-4, atol=2e-5)


@_pytest.mark.parametrize('name', sorted(CONFIG_SEEDS))
def test_make_evaluate_other_env_configurations(golden, oracle_engine, name):
    run_config_episode(golden, name, oracle_engine)
