"""The oracle's episode / GA-evaluate restatement (oracle/rollout_ref.c) pinned against golden vectors that the
REFERENCE'S OWN Python produced (tests/golden/make_golden.py: unmodified Agent.evaluate + CitationEnv + Actor +
the reference shared object) and against the trajectories the reference ships in logs/wandb.

Tolerances: episode lengths / termination step / ranking indices exact; episodic return 1e-5 relative (the only
difference is the f32 summation order of the actor: torch's CPU GEMV vs the sequential restatement).
"""
import numpy as np
import pytest
from conftest import assert_fitness, assert_same_ranking

NET = {'serl50': dict(state_dim=7, action_dim=3, hidden=32, num_layers=3, activation='tanh'),
       'serl10': dict(state_dim=7, action_dim=3, hidden=72, num_layers=3, activation='tanh'),
       'td3': dict(state_dim=7, action_dim=3, hidden=96, num_layers=3, activation='relu')}
RTOL = 1e-5


@pytest.mark.parametrize('tag', ['serl50', 'serl10', 'td3'])
def test_population_fitness_vs_reference_python(golden, tolerances, tag):
    from oracle import rollout as R
    g = golden('pop_' + tag if tag != 'td3' else 'td3')
    w = golden('actors')[tag]
    ref = golden('ref_base')['ref']
    o = R.rollout(w, NET[tag], np.arange(len(w)), ref, t_max=80, threads=8)
    rtol, stable = tolerances.pop(tag)
    assert_fitness(o['fitness'], g['fitness'], rtol, tag)
    np.testing.assert_array_equal(o['length_t'], g['length'])
    assert (o['length_steps'] == 8001).all()
    assert_same_ranking(o['fitness'], g['fitness'], stable)
    assert int(np.argmax(o['fitness'])) == int(np.argmax(g['fitness']))


def test_known_answers_from_the_survey(golden):
    """sum of rewards over all 8 001 steps, base reference, nominal build."""
    from oracle import rollout as R
    w = golden('actors')['serl50'][[18, 0, 7]]
    o = R.rollout(w, NET['serl50'], [0, 1, 2], golden('ref_base')['ref'], t_max=80)
    np.testing.assert_allclose(o['fitness'], [-82.3058562477, -116.4606332304, -117.5455179587], rtol=RTOL)


@pytest.mark.parametrize('tag,idx', [('serl50', 18), ('serl10', 0), ('td3', 0)])
def test_trajectory_vs_reference_python(golden, tag, idx):
    from oracle import rollout as R
    from oracle.smoothness import calc_smoothness
    t = golden('traj')
    w = golden('actors')[tag][[idx]]
    o = R.rollout(w, NET[tag], [0], golden('ref_base')['ref'], t_max=80, traces=True)
    # trajectory-level agreement is limited by how strongly the closed loop amplifies f32 rounding of the actor
    # (tests/golden/make_sensitivity.py): the h=32 champion damps it, the TD3 actor (LeakyReLU, chattering
    # aileron command) carries ~1e-3 rad of it -- its episodic return still agrees to 2e-6.
    atol_u = {'serl50': 2e-6, 'serl10': 1e-5, 'td3': 5e-3}[tag]
    np.testing.assert_allclose(o['actions'][0], t['%s_%d_actions' % (tag, idx)], atol=atol_u)
    np.testing.assert_allclose(o['rewards'][0], t['%s_%d_rewards' % (tag, idx)], atol=10 * atol_u)
    # (states: 20 x the action tolerance -- one f32 ulp in an action of the oscillating SERL10 actor shows as 1.6e-4 in a rate)
    np.testing.assert_allclose(o['states'][0][::25], t['%s_%d_states25' % (tag, idx)], rtol=2e-4, atol=20 * atol_u)
    g = golden('pop_' + tag if tag != 'td3' else 'td3')
    np.testing.assert_allclose(calc_smoothness(o['actions'][0]), g['smoothness'][idx], rtol=1e-4)


@pytest.mark.parametrize('tag', ['serl50', 'td3'])
def test_shipped_closed_loop_trajectories(golden, tag):
    """logs/wandb/*/figures/nominal/nominal_trajectory.csv (de-filtered): what the authors' machine produced with
    torch 1.10 -- return to 1e-6 relative, states/actions to plotting accuracy."""
    from oracle import rollout as R
    s = golden('shipped_csv')
    w = golden('actors')[tag][[18 if tag == 'serl50' else 0]]
    ref = golden('ref_base')['ref']
    np.testing.assert_allclose(ref[::25], s[tag + '_ref'], atol=2e-15)
    o = R.rollout(w, NET[tag], [0], ref, t_max=80, traces=True)
    n = int(s[tag + '_n'])
    assert n == 8001
    # de-filtered row k of the shipped file = (ref(t_k), last_u, x before step k, reward of step k)
    ours = o['rewards'][0]
    shipped = s[tag + '_reward']
    rtol = {'serl50': 1e-6, 'td3': 1e-5}[tag]       # the TD3 actor amplifies f32 rounding more (make_sensitivity.py: 2e-6)
    np.testing.assert_allclose(ours.sum(), shipped.sum(), rtol=rtol)
    np.testing.assert_allclose(ours[:n - 1].sum(), shipped[:n - 1].sum(), rtol=rtol)
    # shipped row k holds x BEFORE step k = our state after step k-1; the fixture keeps every 25th row
    np.testing.assert_allclose(o['states'][0][24::25], s[tag + '_x'][1:], rtol=1e-3, atol={'serl50': 1e-4, 'td3': 2e-2}[tag])


def test_faults_and_trims_vs_reference_python(golden, tolerances):
    from oracle import rollout as R
    from serl_amd import builds
    g = golden('faults')
    w = golden('actors')['serl50'][[18, 0, 7]]
    ref = golden('ref_base')['ref']
    # 'gust' / 'noise' draw np.random sensor noise in their wrappers: test_sensor_noise_wrappers_vs_reference_python
    for mode in ['be', 'jr', 'sa', 'se', 'ice', 'cg', 'cg-for', 'high-q', 'low-q', 'cg-shift']:
        build, row = builds.resolve_mode(mode)
        # the golden episodes ran one after the other on ONE env object per mode, and the reference's initialize()
        # leaves the model clock running: episode j starts at tick = steps simulated before it (1 reset step + 8001)
        tick0 = [int(sum(g['%s_%d' % (mode, i)][3] + 1 for i in (18, 0, 7)[:j])) for j in range(3)]
        o = R.rollout(w, NET['serl50'], [0, 1, 2], ref, build=build, faults=[row] * 3, tick0=tick0, t_max=80, threads=3)
        for j, i in enumerate((18, 0, 7)):
            ref_fit, ref_len, ref_sm, ref_n = g['%s_%d' % (mode, i)]
            assert o['length_steps'][j] == int(ref_n), (mode, i)
            assert o['length_t'][j] == ref_len, (mode, i)
            np.testing.assert_allclose(o['fitness'][j], ref_fit, rtol=tolerances.fault(mode, i, ref_fit), err_msg='%s %d' % (mode, i))


def test_error_carried_into_next_episode(golden):
    """envs/phlabenv.py:401-428 never clears self.error: obs0 of an episode carries the previous episode's last
    tracking error when the same env object is reused (the sequential reference loop)."""
    from oracle import rollout as R
    from serl_amd import refsignals
    c = golden('carry')
    w = golden('actors')['serl50'][[18, 0, 7]]
    ref = refsignals.tabulate(*refsignals.base_reference(20), 20)
    o = R.rollout(w, NET['serl50'], [0, 1, 2], ref, err0=c['err0'], t_max=20, traces=True)
    np.testing.assert_allclose(o['fitness'], c['fitness'], rtol=RTOL)
    np.testing.assert_array_equal(o['length_t'], c['length'])
    # and the carried value IS the last error of the previous episode
    for j in (0, 1):
        n = o['length_steps'][j]
        np.testing.assert_allclose(ref[n - 1] - o['states'][j][n - 1][[7, 6, 5]], c['err0'][j + 1], atol=1e-6)


def test_actor_forward_samples_vs_torch_reference(golden):
    """The f32 MLP (custom LayerNorm: unbiased std, eps on std; 'relu' = LeakyReLU) against the reference's
    torch Actor on 64 random observations for every shipped actor -- via a 1-step open rollout is not possible,
    so the numpy restatement below mirrors oracle/rollout_ref.c:actor_forward and is itself checked end-to-end
    by the population tests above."""
    for tag in ('serl50', 'serl10', 'td3'):
        g = golden('pop_' + tag if tag != 'td3' else 'td3')
        w = golden('actors')[tag]
        net = NET[tag]
        H, L = net['hidden'], net['num_layers']
        obs = g['obs_samples'].astype(np.float32)
        for m in range(len(w)):
            p = w[m]; off = 0

            def take(n, shape):
                nonlocal off
                v = p[off:off + n].reshape(shape); off += n
                return v
            act = {'tanh': np.tanh, 'relu': lambda v: np.where(v > 0, v, np.float32(0.01) * v)}[net['activation']]
            h = act(obs @ take(H * 7, (H, 7)).T + take(H, (H,)))
            for _ in range(L):
                z = h @ take(H * H, (H, H)).T + take(H, (H,))
                gm, bt = take(H, (H,)), take(H, (H,))
                mean = z.mean(-1, keepdims=True)
                std = z.std(-1, ddof=1, keepdims=True)
                h = act(gm * (z - mean) / (std + np.float32(1e-6)) + bt)
            a = np.tanh(h @ take(3 * H, (3, H)).T + take(3, (3,)))
            np.testing.assert_allclose(a, g['act_samples'][m], atol=5e-6)


def test_early_termination_penalty_and_length():
    """Diverging actors: done at the first step out of bounds, penalty -(1/dt)*(t_max - t)*2 added to the last
    reward (envs/phlabenv.py:391-399)."""
    from oracle import rollout as R
    from serl_amd import refsignals
    rng = np.random.default_rng(3)
    w = rng.normal(0, 0.3, (6, 3715)).astype(np.float32)
    ref = refsignals.tabulate(*refsignals.base_reference(20), 20)
    o = R.rollout(w, NET['serl50'], np.arange(6), ref, t_max=20, traces=True)
    early = o['length_steps'] < 2001
    assert early.any()
    for e in np.nonzero(early)[0]:
        n = o['length_steps'][e]
        x = o['states'][e][n - 1]
        assert abs(x[7]) > np.deg2rad(60) or abs(x[6]) > np.deg2rad(75) or x[9] < 50
        t_last = refsignals.env_times(n)[-1]
        pen = -1.0 / 0.01 * (20.0 - t_last) * 2.0
        assert o['rewards'][e][n - 1] < pen + 1e-9 and o['rewards'][e][n - 1] >= pen - 1.0
        np.testing.assert_allclose(o['fitness'][e], o['rewards'][e][:n].sum(), rtol=1e-12)


def test_action_noise_path_and_table_exhaustion():
    from oracle import rollout as R
    from serl_amd import refsignals
    w = np.load(__import__('os').path.join(__import__('os').path.dirname(__file__), 'golden', 'actors.npz'))['serl50'][[18]]
    ref = refsignals.tabulate(*refsignals.base_reference(20), 20)
    rng = np.random.default_rng(0)
    noise = np.clip(0.2 * rng.standard_normal((1, len(ref), 3)), -0.5, 0.5)
    a = R.rollout(w, NET['serl50'], [0], ref, t_max=20)
    b = R.rollout(w, NET['serl50'], [0], ref, t_max=20, action_noise=noise)
    c = R.rollout(w, NET['serl50'], [0], ref, t_max=20, action_noise=np.zeros_like(noise))
    assert b['fitness'][0] != a['fitness'][0]
    np.testing.assert_allclose(c['fitness'], a['fitness'], rtol=1e-6)   # f64 vs f32 scaling of the same action
    with pytest.raises(RuntimeError):
        R.rollout(w, NET['serl50'], [0], ref[:100], t_max=20)
    # per-episode noise rows (a population evaluation and the RL actor's exploration episode in one call)
    m = R.rollout(w, NET['serl50'], [0, 0, 0], ref, t_max=20, action_noise=noise, noise_row=[-1, 0, -1])
    assert m['fitness'][0] == a['fitness'][0] == m['fitness'][2] and m['fitness'][1] == b['fitness'][0]
    assert (m['length_steps'] == [a['length_steps'][0], b['length_steps'][0], a['length_steps'][0]]).all()


def _sensor_case(golden, mode):
    from serl_amd import builds
    g = golden('sensor_noise')
    ref = golden('ref_base')['ref']
    build, row = builds.resolve_mode(mode)
    # the wrapper draws randn(3), randn(1), randn(1), randn(2) per step() from the seeded legacy generator
    sn = np.stack([builds.sensor_noise_table(len(ref), np.random.RandomState(int(g['seed_%s_%d' % (mode, i)])))
                   for i in (18, 0, 7)])
    # the episodes ran one after the other on one library instance: the model clock keeps running (gust is time-switched)
    tick0 = [int(sum(g['%s_%d' % (mode, i)][3] + 1 for i in (18, 0, 7)[:j])) for j in range(3)]
    return g, ref, build, sn, tick0


@pytest.mark.parametrize('mode', ['noise', 'gust'])
def test_sensor_noise_wrappers_vs_reference_python(golden, mode):
    """envs/noise/citation.py:71-82 and envs/gust/citation.py:72-86 add a sensor model (bias + sd * np.random.randn) to
    what step() returns.  Golden: the reference's own CitationEnv / Agent.evaluate / wrappers with np.random seeded
    before every episode (tests/golden/make_sensor_golden.py); here the same draws arrive as a pre-drawn table."""
    from oracle import rollout as R
    g, ref, build, sn, tick0 = _sensor_case(golden, mode)
    w = golden('actors')['serl50'][[18, 0, 7]]
    o = R.rollout(w, NET['serl50'], [0, 1, 2], ref, build=build, sensor_noise=sn, tick0=tick0, t_max=80, threads=3)
    clean = R.rollout(w, NET['serl50'], [0, 1, 2], ref, build=build, tick0=tick0, t_max=80, threads=3)
    for j, i in enumerate((18, 0, 7)):
        ref_fit, ref_len, ref_sm, ref_n = g['%s_%d' % (mode, i)]
        assert o['length_steps'][j] == int(ref_n) and o['length_t'][j] == ref_len
        np.testing.assert_allclose(o['fitness'][j], ref_fit, rtol=RTOL, err_msg='%s %d' % (mode, i))
        assert abs(clean['fitness'][j] - ref_fit) > 1e-3 * abs(ref_fit)        # the sensor model is not a no-op
    # per-episode rows: an episode without a row is the clean episode
    m = R.rollout(w, NET['serl50'], [0, 1, 2], ref, build=build, sensor_noise=sn[[1]], sensor_row=[-1, 0, -1], tick0=tick0,
                  t_max=80, threads=3)
    assert m['fitness'][0] == clean['fitness'][0] and m['fitness'][1] == o['fitness'][1] and m['fitness'][2] == clean['fitness'][2]


def det_expm1f_neg(xf):
    """expm1 of a non-positive f32 as include/serl_amd.h specifies it (the ELU branch): f64 with + - * / only, one rounding"""
    import math
    x = float(xf)
    if x != x:
        return np.float32(xf)
    if x < -104.0:
        return np.float32(-1.0)
    v = x * 1.4426950408889634
    k = -int(0.5 - v) if v < 0.0 else int(v + 0.5)
    r = (x - k * 0.6931471803691238) - k * 1.9082149292705877e-10
    r2 = r * r; r4 = r2 * r2; r8 = r4 * r4
    b = [0.5 + 0.16666666666666666 * r, 0.041666666666666664 + 0.008333333333333333 * r,
         0.001388888888888889 + 0.0001984126984126984 * r, 2.48015873015873e-05 + 2.7557319223985893e-06 * r,
         2.755731922398589e-07 + 2.505210838544172e-08 * r, 2.08767569878681e-09 + 1.6059043836821613e-10 * r]
    c = [b[0] + b[1] * r2, b[2] + b[3] * r2, b[4] + b[5] * r2]
    q = r + r2 * ((c[0] + c[1] * r4) + c[2] * r8)
    return np.float32(q if k == 0 else math.ldexp(1.0, k) * (q + 1.0) - 1.0)


def _np_actor(w, net, obs):
    """The actor's f32 arithmetic as include/serl_amd.h specifies it, restated with numpy scalars: dot products as four
    interleaved fma partial sums, LayerNorm sums as pairwise trees per 16 rows, tanh through the f64 det_tanhf."""
    import math
    f32 = np.float32
    S, H, L, A = net['state_dim'], net['hidden'], net['num_layers'], net['action_dim']

    def fma(a, b, c):            # exact product + one rounding: f32 operands are exact in f64 and the product fits 48 bits
        return f32(np.float64(a) * np.float64(b) + np.float64(c))

    def dot4(wr, h, bias):
        p = [f32(0)] * 4
        for j in range(len(h)):
            p[j & 3] = fma(wr[j], h[j], p[j & 3])
        return f32(bias + f32(f32(p[0] + p[1]) + f32(p[2] + p[3])))

    def tree16(x):
        t = [f32(v) for v in x] + [f32(0)] * (16 - len(x))
        s = 1
        while s < 16:
            for i in range(0, 16, 2 * s):
                t[i] = f32(t[i] + t[i + s])
            s *= 2
        return t[0]

    def tree_sum(x):
        s = tree16(x[:16])
        for b in range(16, len(x), 16):
            s = f32(s + tree16(x[b:b + 16]))
        return s

    def det_tanhf(xf):
        x = float(xf); ax = abs(x)
        if ax > 20.0:
            t = 1.0
        else:
            z = ax + ax
            v = z * 1.4426950408889634
            k = int(v + 0.5)
            r = (z - k * 0.6931471803691238) - k * 1.9082149292705877e-10
            r2 = r * r; r4 = r2 * r2; r8 = r4 * r4
            b = [0.5 + 0.16666666666666666 * r, 0.041666666666666664 + 0.008333333333333333 * r,
                 0.001388888888888889 + 0.0001984126984126984 * r, 2.48015873015873e-05 + 2.7557319223985893e-06 * r,
                 2.755731922398589e-07 + 2.505210838544172e-08 * r, 2.08767569878681e-09 + 1.6059043836821613e-10 * r]
            c = [b[0] + b[1] * r2, b[2] + b[3] * r2, b[4] + b[5] * r2]
            q = r + r2 * ((c[0] + c[1] * r4) + c[2] * r8)
            t = q / (q + 2.0) if k == 0 else 1.0 - 2.0 / (math.ldexp(1.0, k) * (q + 1.0) + 1.0)
        return f32(-t if x < 0 else t)
    # hidden activation: tanh, 'relu' = LeakyReLU(0.01), or ELU (base/core/mod_utils.py:14-18); the output layer is always tanh
    act = {'tanh': det_tanhf, 'relu': (lambda v: v if v > 0 else f32(f32(0.01) * v)),
           'elu': (lambda v: v if v > 0 else det_expm1f_neg(v))}[net['activation']]
    o = 0
    W0 = w[o:o + H * S].reshape(H, S); o += H * S
    b0 = w[o:o + H]; o += H
    h = [act(dot4(W0[i], obs, b0[i])) for i in range(H)]
    for _ in range(L):
        W = w[o:o + H * H].reshape(H, H); o += H * H
        bl, g, be = w[o:o + H], w[o + H:o + 2 * H], w[o + 2 * H:o + 3 * H]; o += 3 * H
        y = [dot4(W[i], h, bl[i]) for i in range(H)]
        mean = f32(tree_sum(y) / f32(H))
        d = [f32(v - mean) for v in y]
        var = tree_sum([f32(v * v) for v in d])
        den = f32(np.sqrt(f32(var / f32(H - 1))) + f32(1e-6))
        h = [act(f32(f32(f32(g[i] * d[i]) / den) + be[i])) for i in range(H)]
    Wo = w[o:o + A * H].reshape(A, H); o += A * H
    bo = w[o:o + A]
    return np.array([det_tanhf(dot4(Wo[i], h, bo[i])) for i in range(A)], dtype=np.float32)


@pytest.mark.parametrize('tag', ['serl50', 'serl10', 'td3'])
def test_actor_arithmetic_is_the_specified_one(golden, tag):
    """First action of an episode (obs0 = carried error, initial p q r alpha) against the numpy restatement of the
    arithmetic the C ABI specifies -- bit for bit."""
    from oracle import rollout as R
    from serl_amd import builds, refsignals
    net = NET[tag]
    w = golden('actors')[tag]
    w = w[[0, 1 % len(w), 2 % len(w)]]
    ref = refsignals.tabulate(*refsignals.base_reference(20), 20)
    rng = np.random.default_rng(5)
    err0 = rng.normal(0, 0.05, (3, 3))
    o = R.rollout(w, net, [0, 1, 2], ref, err0=err0, t_max=20, traces=True)
    x0 = builds.load('h2000_v90')[0]['x0']
    bound = 10.0 * (3.14159265358979323846 / 180.0)
    for e in range(3):
        obs = np.array([err0[e, 0], err0[e, 1], err0[e, 2], x0[0], x0[1], x0[2], x0[4]], dtype=np.float64).astype(np.float32)
        a = _np_actor(w[e], net, obs)
        u = [-bound + float(np.float32(0.5) * (a[i] + np.float32(1.0))) * (bound - (-bound)) for i in range(3)]
        np.testing.assert_array_equal(o['actions'][e][0], np.array(u))


NOISE_SD, NOISE_CLIP = 0.2962183114680794, 0.5      # base/parameters.py:49,76
NOISE_CASES = [('serl50_18_n', 'serl50', 18, 20, True), ('td3_0_n', 'td3', 0, 20, True),
               ('serl50_7_n80', 'serl50', 7, 80, True), ('serl50_0_clean', 'serl50', 0, 20, False)]


def noise_case_inputs(golden, name, tag, idx, t_max, noisy):
    """(weights, reference table, pre-drawn clipped noise table) of one tests/golden/make_noise_golden.py case: the
    reference draws `noise_sd * np.random.randn(3)` once per step (agent.py:91) -- the same legacy stream in one block."""
    from oracle import signals as S
    g = golden('noise_path')
    T = int(g[name + '_ret'][2])
    tt = np.linspace(0., t_max, 6)
    ref = S.tabulate_refs(S.SmoothedStepSequence(tt, [0, 12, 3, -4, -8, 2], smooth_width=t_max // 10),
                          S.SmoothedStepSequence(tt, [2, -2, 2, 10, 2, -6], smooth_width=t_max // 10), t_max)
    noise = None
    if noisy:
        noise = np.clip(NOISE_SD * np.random.RandomState(int(g[name + '_seed'])).randn(T, 3), -NOISE_CLIP, NOISE_CLIP)[None]
    return golden('actors')[tag][[idx]], ref, noise, T


def check_noise_case(g, name, fitness, length_t, length_steps, cost_steps, tr, rtol=RTOL):
    """One episode's outputs against the reference's own Agent.evaluate(is_action_noise, store_transition=True):
    return <= 1e-5, every cost flag and the critical-buffer routing exact, the stored tuples to f32 (trajectory-level
    agreement is bounded by the closed loop's amplification of the actor's f32 rounding, as in test_trajectory_*)."""
    ret = g[name + '_ret']
    T = int(ret[2])
    assert int(length_steps) == T and length_t == ret[1]
    np.testing.assert_allclose(fitness, ret[0], rtol=rtol)
    np.testing.assert_array_equal(tr[:T, 19].astype(np.int8), g[name + '_cost'])          # every info['cost']
    assert int(cost_steps) == int(g[name + '_ncrit'])
    np.testing.assert_array_equal(np.nonzero(tr[:T, 19])[0], g[name + '_crit_idx'])
    # agent.py:93,103: the stored action is the clipped NOISY one (the un-noised actor output is off by up to noise_clip = 0.5)
    atol = 1e-3 if T > 4000 else 5e-5
    np.testing.assert_allclose(tr[:T, 7:10], g[name + '_action'], atol=atol)
    np.testing.assert_allclose(tr[:50, :19], g[name + '_head'], atol=5e-6)
    np.testing.assert_allclose(tr[T - 50:T, :19], g[name + '_tail'], atol=atol)
    assert tr[T - 1, 18] == 1.0 and not tr[:T - 1, 18].any()


@pytest.mark.parametrize('name,tag,idx,t_max,noisy', NOISE_CASES)
def test_exploration_noise_and_stored_transitions_vs_reference_python(golden, name, tag, idx, t_max, noisy):
    from oracle import rollout as R
    from oracle.smoothness import calc_smoothness
    w, ref, noise, T = noise_case_inputs(golden, name, tag, idx, t_max, noisy)
    o = R.rollout(w, NET[tag], [0], ref, t_max=t_max, action_noise=noise, traces=True, transitions=True)
    g = golden('noise_path')
    check_noise_case(g, name, o['fitness'][0], o['length_t'][0], o['length_steps'][0], o['cost_steps'][0], o['transitions'][0])
    np.testing.assert_allclose(calc_smoothness(o['actions'][0][:T]), float(g[name + '_smoothness']), rtol=1e-4)
    assert int(g[name + '_ret'][3]) == T and int(g[name + '_ret'][4]) == 1      # frames / episodes the reference counted


def test_generated_references_equal_their_table():
    """serl_ref_spec in the oracle: the generator inside the episode loop == the same arithmetic tabulated on the host
    (bit for bit), and within 2 ulp of the libm-cosine table of the `signals` restatement that the shipped trajectories pin."""
    from oracle import rollout as R
    from serl_amd import refsignals as rs
    import os
    w = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'actors.npz'))['serl50'][[18, 0, 7]]
    th, ph = rs.training_references(2, 20, np.random.RandomState(1))
    th0, ph0 = rs.base_reference(20)
    ths, phs, trims = [th0] + th, [ph0] + ph, [0.22, 0.2106, 0.2106]
    specs = rs.ref_specs(ths, phs, trims)
    table = rs.tabulate_specs(specs, 20)
    libm = np.stack([rs.tabulate(a, b, 20, theta_trim_deg=t) for a, b, t in zip(ths, phs, trims)])
    assert 0 < np.abs(table).max() and np.abs(table - libm).max() < 5e-16
    a = R.rollout(w, NET['serl50'], [0, 1, 2], specs, t_max=20, traces=True)
    b = R.rollout(w, NET['serl50'], [0, 1, 2], table, t_max=20, traces=True)
    c = R.rollout(w, NET['serl50'], [0, 1, 2], libm, t_max=20)
    np.testing.assert_array_equal(a['fitness'], b['fitness']); np.testing.assert_array_equal(a['states'], b['states'])
    np.testing.assert_allclose(a['fitness'], c['fitness'], rtol=1e-10)
    x = np.linspace(0, 1, 200001)
    assert np.abs(rs.det_cospi(x) - np.cos(np.pi * x)).max() < 4e-16


ELU_NET = dict(state_dim=7, action_dim=3, hidden=32, num_layers=3, activation='elu')


def test_elu_activation_spec_and_closed_loop_vs_reference(golden):
    """ELU actors (mod_utils.py:14-18: nn.ELU), which no shipped checkpoint uses: (1) det_expm1f_neg -- the libm-free
    arithmetic the C ABI specifies for the ELU branch -- is the correctly rounded f32 expm1 on 1e5 samples and equals
    torch.nn.ELU there; (2) the oracle's first action equals the numpy restatement of the specified arithmetic bit for bit;
    (3) shipped SERL50 weights flown as ELU actors by the REFERENCE'S OWN Actor / Agent.evaluate
    (tests/golden/make_elu_golden.py; two of the three crash within 2 s: early termination and penalty included):
    forward samples to 5e-6, episode length exact, return to 1e-5."""
    import torch
    from oracle import rollout as R
    from serl_amd import builds, refsignals
    rs = np.random.RandomState(0)
    x = np.concatenate([-np.abs(rs.randn(50000) * 3), -rs.rand(30000) * 1e-3, -rs.rand(20000) * 110]).astype(np.float32)
    got = np.array([det_expm1f_neg(v) for v in x], dtype=np.float32)
    exact = np.expm1(x.astype(np.float64))
    np.testing.assert_array_equal(got, exact.astype(np.float32))          # round-to-nearest of the exact value
    np.testing.assert_allclose(got, torch.nn.ELU()(torch.from_numpy(x)).numpy(), rtol=3e-7, atol=1e-38)
    g = golden('elu')
    w = golden('actors')['serl50'][g['actors']]
    ref = refsignals.tabulate(*refsignals.base_reference(20), 20)
    o = R.rollout(w, ELU_NET, [0, 1, 2], ref, t_max=20, traces=True)
    x0 = builds.load('h2000_v90')[0]['x0']
    bound = 10.0 * (3.14159265358979323846 / 180.0)
    obs0 = np.array([0, 0, 0, x0[0], x0[1], x0[2], x0[4]], dtype=np.float64).astype(np.float32)
    for e in range(3):
        a = _np_actor(w[e], ELU_NET, obs0)
        u = [-bound + float(np.float32(0.5) * (a[i] + np.float32(1.0))) * (bound - (-bound)) for i in range(3)]
        np.testing.assert_array_equal(o['actions'][e][0], np.array(u))
        for j in range(0, 64, 7):
            np.testing.assert_allclose(_np_actor(w[e], ELU_NET, g['obs_samples'][j].astype(np.float32)), g['act_samples'][e][j], atol=5e-6)
        fit, length, sm, n = g['ret'][e]
        assert int(o['length_steps'][e]) == int(n) and o['length_t'][e] == length
        np.testing.assert_allclose(o['fitness'][e], fit, rtol=RTOL)
        np.testing.assert_allclose(o['actions'][e][:int(n)], g['actions_%d' % g['actors'][e]], atol=5e-3)     # (a saturating, oscillating controller: f32 rounding of the actor shows at 3e-3 rad late in the episode; the return agrees to 1e-5)
    assert (g['ret'][:, 3] < 2001).sum() == 2


CONFIG_CASES = ['sym', 'sym_soft', 'sym_inc_n', 'full', 'att_inc', 'att_inc_soft', 'full_inc_n', 'att_inc_n', 'full_be', 'sym_ice']


def config_case(g, name):
    """inputs of one case of config.npz (tests/golden/make_config_golden.py) in the layouts of serl_rollout_desc"""
    cfg, incr, S, A, H, L = (int(v) for v in g[name + '_cfg'])
    net = dict(state_dim=S, action_dim=A, hidden=H, num_layers=L, activation=str(g[name + '_act']))
    rows = g[name + '_rows']
    T = len(rows)
    ref = np.zeros((T + 1, 3)); ref[:T, :A] = g[name + '_ref']
    noise = np.zeros((1, T + 1, 3)); noise[0, :T, :A] = g[name + '_noise']
    return cfg, incr, net, rows, ref, (noise if np.any(noise) else None)


def config_build(g, name):
    """(build, fault rows) of a case's env mode: the fault wrappers act on the padded command whatever the configuration"""
    from serl_amd import builds
    build, row = builds.resolve_mode(str(g[name + '_mode']))
    return build, (None if row == builds.NOMINAL_ROW else [list(row)])


@pytest.mark.parametrize('name', CONFIG_CASES)
def test_env_configurations_vs_reference_python(golden, name):
    """symmetric / full observation sets and incremental (rate) control, envs/phlabenv.py:84-97,174-176,205-220,377-380:
    every stored transition of the reference's own Agent.evaluate (obs S, action A, next_obs S, reward, done), every cost
    flag, env.last_u of every step, return and length -- random actors built by the reference's own Actor class."""
    from oracle import rollout as R
    g = golden('config')
    cfg, incr, net, rows, ref, noise = config_case(g, name)
    S, A, T = net['state_dim'], net['action_dim'], len(rows)
    build, faults = config_build(g, name)
    o = R.rollout(g[name + '_w'][None], net, [0], ref, t_max=20, action_noise=noise, traces=True, transitions=True,
                  env_config=cfg, incremental=incr, build=build, faults=faults)
    ret = g[name + '_ret']
    assert int(o['length_steps'][0]) == T == int(ret[2])
    assert o['length_t'][0] == ret[1]
    np.testing.assert_allclose(o['fitness'][0], ret[0], rtol=RTOL)
    tr = o['transitions'][0, :T].astype(np.float64)
    assert tr.shape[1] == 2 * S + A + 3
    # the f32 rows against the reference's f64 tuples: what the learners build FloatTensors from
    np.testing.assert_allclose(tr[:, :2 * S + A + 2], rows, rtol=2e-4, atol=2e-5)
    np.testing.assert_array_equal(tr[:, 2 * S + A + 1], rows[:, -1])                        # done flags
    np.testing.assert_array_equal(tr[:, 2 * S + A + 2].astype(np.int8), g[name + '_cost'])  # info['cost'] of every step
    np.testing.assert_allclose(o['actions'][0, :T, :A], g[name + '_u'], rtol=1e-5, atol=1e-7)
    assert not o['actions'][0, :T, A:].any()


def test_short_libm_flavour_of_the_oracle_stays_within_libm_accuracy_of_the_pinned_one(golden):
    """oracle/_build/libcitation_oracle_shortlibm.so (oracle/citation_rt.h CIT_SHORT_LIBM) is the C restatement with sin / cos / tan / pow
    taken from the CPU build of the product's serl_amd/csrc/citation_libm.h -- the flavour the GPU parity tests compare with at ZERO
    tolerance.  The flavour pinned to the reference binary is the glibc one; this test bounds the distance between the two: open loop
    (3 000 steps of the native step(), every build whose code differs) <= 1e-9 relative per state, the episodic returns of the stable
    shipped actors <= 1e-7, lengths and cost counts equal (/root/reference/envs/h2000_v90/citation.py:65-72, base/core/agent.py:63-138)."""
    from oracle.dynamics import CitationDynamics
    from oracle import rollout as R
    rng = np.random.default_rng(5)
    for build in ('h2000_v90', 'ice', 'cg_timed', 'gust', 'test'):
        a, b = CitationDynamics(build), CitationDynamics(build, short_libm=True)
        cmd = np.zeros(10)
        worst = 0.0
        for k in range(3000):
            cmd[0] = 0.01 * np.sin(0.01 * k) + 0.002 * rng.standard_normal()
            cmd[1] = 0.01 * np.cos(0.013 * k)
            cmd[2] = 0.005 * np.sin(0.007 * k)
            xa, xb = a.step(cmd), b.step(cmd)
            worst = max(worst, float(np.max(np.abs(xa - xb) / (np.abs(xa) + 1e-3))))
        assert worst <= 1e-9, (build, worst)
    from serl_amd import refsignals
    w = golden('actors')['serl50']
    ref = refsignals.tabulate(*refsignals.base_reference(80), 80)
    moe = list(range(0, len(w), 3))
    a = R.rollout(w, NET['serl50'], moe, ref, t_max=80, threads=8)
    b = R.rollout(w, NET['serl50'], moe, ref, t_max=80, threads=8, short_libm=True)
    np.testing.assert_array_equal(a['length_steps'], b['length_steps'])
    np.testing.assert_array_equal(a['cost_steps'], b['cost_steps'])
    assert (np.abs(a['fitness'] - b['fitness']) <= 1e-7 * np.abs(a['fitness'])).all(), np.abs(a['fitness'] / b['fitness'] - 1).max()
