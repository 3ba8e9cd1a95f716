#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the build container only).

  python tests/golden/make_golden.py            (cwd anywhere; ~6 min)

What is produced (all small, committed):
  actors.npz          packed f32 weights of every shipped actor (SERL50 x50, SERL10 x10, TD3 x1)
  ref_base.npz        the fixed evaluation reference of base/evaluate.py:167-180 tabulated at the
                      env's accumulated step times (radians)
  pop_serl50.npz      reference Agent.evaluate on all 50 SERL50 actors, base ref, nominal, t_max=80:
                      fitness, length, smoothness, + actor forward samples
  pop_serl10.npz      same for the 10 SERL10 actors (h=72)
  td3.npz             same for the TD3 actor (h=96, LeakyReLU)
  faults.npz          SERL50 actors {18,0,7} under be/jr/sa/se/ice/cg/cg-for/high-q/low-q
  traj.npz            full-resolution trajectories (actions, states every 25th step, rewards) for 3 episodes
  shipped_csv.npz     de-filtered shipped closed-loop trajectories (logs/wandb/*/figures/nominal/
                      nominal_trajectory.csv): episodic return + subsampled states
  dyn_open_loop.npz   open-loop dynamics KATs from the reference .so for every build: seeded command
                      sequence -> state samples (pins the C restatement where /root/reference is absent)
  ga_ops.npz          SSNE operator KATs (reference mod_neuro_evo.py run on shipped actors, fixed seeds)
"""
import os, sys, glob, time, io, contextlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim
os.chdir('/tmp')
refshim.install()
import torch
torch.set_num_threads(1)
from oracle import signals as S, rollout as R
from oracle.refso import RefCitation

REF = refshim.REF
RUNS = {'serl50': ('run-20220924_144643-1xzaqiba_SERL50', 32, 'tanh'),
        'serl10': ('run-20220913_165505-12zowviu_SERL10', 72, 'tanh'),
        'td3': ('run-20221102_144601-1dixcrrl_TD3', 96, 'relu')}


def load_pop(tag):
    run, h, act = RUNS[tag]
    f = os.path.join(REF, 'logs', 'wandb', run, 'files', 'rl_net.pkl' if tag == 'td3' else 'evo_nets.pkl')
    sd = torch.load(f, weights_only=True, map_location='cpu')
    if tag == 'td3':
        return [sd], h, act
    return [sd['actor_%d' % i] for i in range(len(sd))], h, act


def base_refs(t_max=80):
    tt = np.linspace(0., t_max, 6)
    th = S.SmoothedStepSequence(tt, [0, 12, 3, -4, -8, 2], smooth_width=t_max // 10)
    ph = S.SmoothedStepSequence(tt, [2, -2, 2, 10, 2, -6], smooth_width=t_max // 10)
    return th, ph


def run_ref(env, actor, th, ph, clear=True):
    """The env object never clears self.error in reset() (envs/phlabenv.py:401-428), so obs0 of an episode
    carries the last tracking error of the previous episode on the same env.  clear=True zeroes it first
    (== a fresh env per episode, which is what the batched evaluator models); the `carry` golden keeps it."""
    if clear:
        env.error = np.zeros(3)
    with contextlib.redirect_stdout(io.StringIO()):
        ep = refshim.reference_evaluate(env, actor, user_refs={'theta_ref': th, 'phi_ref': ph})
    return ep


def main():
    out = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--out=')), HERE)
    os.makedirs(out, exist_ok=True)
    t0 = time.time()
    parts = set(a for a in sys.argv[1:] if not a.startswith('--')) or {'pop', 'faults', 'carry', 'shipped', 'kat'}
    th, ph = base_refs()
    ref = S.tabulate_refs(th, ph, 80)
    np.savez_compressed(os.path.join(out, 'ref_base.npz'), ref=ref)
    packs = {}
    for tag in RUNS:
        sds, h, act = load_pop(tag)
        packs[tag] = np.stack([R.pack_state_dict(sd) for sd in sds])
    np.savez_compressed(os.path.join(out, 'actors.npz'), **packs)

    env = refshim.make_env('nominal', 80)
    rng = np.random.default_rng(0)
    if 'pop' not in parts:
        RUNS_ = []
    else:
        RUNS_ = ['serl50', 'serl10', 'td3']
    obs_samples = np.concatenate([rng.normal(0, 0.05, (64, 3)), rng.normal(0, 0.1, (64, 4))], axis=1)
    traj = {}
    for tag in RUNS_:
        sds, h, act = load_pop(tag)
        fit, length, sm, acts = [], [], [], []
        for i, sd in enumerate(sds):
            actor = refshim.make_actor(sd, h, 3, act)
            ep = run_ref(env, actor, th, ph)
            fit.append(ep.fitness); length.append(ep.length); sm.append(ep.smoothness)
            acts.append(np.stack([actor.select_action(o) for o in obs_samples]))
            if (tag, i) in (('serl50', 18), ('serl10', 0), ('td3', 0)):
                traj['%s_%d_actions' % (tag, i)] = np.asarray(ep.actions)
                traj['%s_%d_rewards' % (tag, i)] = np.asarray(ep.reward_lst)
                traj['%s_%d_states25' % (tag, i)] = np.asarray(ep.state_history)[::25]
            print(tag, i, ep.fitness, ep.length, '%.0fs' % (time.time() - t0), flush=True)
        np.savez_compressed(os.path.join(out, 'pop_%s.npz' % tag if tag != 'td3' else os.path.join(out, 'td3.npz')),
                            fitness=np.array(fit), length=np.array(length), smoothness=np.array(sm),
                            obs_samples=obs_samples, act_samples=np.stack(acts), hidden=h,
                            activation=act)
    if 'pop' in parts:
        np.savez_compressed(os.path.join(out, 'traj.npz'), **traj)

    # fault / trim modes
    sds, h, act = load_pop('serl50')
    modes = [] if 'faults' not in parts else ['be', 'jr', 'sa', 'se', 'ice', 'cg', 'cg-for', 'high-q', 'low-q', 'cg-shift', 'gust']
    res = {}
    for m in modes:
        envm = refshim.make_env(m, 80)
        for i in (18, 0, 7):
            ep = run_ref(envm, refshim.make_actor(sds[i], h, 3, act), th, ph)
            res['%s_%d' % (m, i)] = np.array([ep.fitness, ep.length, ep.smoothness, len(ep.reward_lst)])
            print(m, i, ep.fitness, ep.length, flush=True)
    if 'faults' in parts:
        np.savez_compressed(os.path.join(out, 'faults.npz'), **res)

    if 'carry' in parts:   # three episodes on ONE env object, error carried into the next obs0
        envc = refshim.make_env('nominal', 20)
        th20, ph20 = base_refs(20)
        sds, h, act = load_pop('serl50')
        car = {'err0': [], 'fitness': [], 'length': []}
        for i in (18, 0, 7):
            car['err0'].append(np.array(envc.error, dtype=np.float64).copy())
            ep = run_ref(envc, refshim.make_actor(sds[i], h, 3, act), th20, ph20, clear=False)
            car['fitness'].append(ep.fitness); car['length'].append(ep.length)
        np.savez_compressed(os.path.join(out, 'carry.npz'), **{k: np.array(v) for k, v in car.items()})

    # shipped closed-loop CSVs (each row is the 2-tap causal average; raw[k] = 2*row[k] - raw[k-1])
    ship = {}
    for tag in (('serl50', 'td3') if 'shipped' in parts else ()):
        f = os.path.join(REF, 'logs', 'wandb', RUNS[tag][0], 'figures', 'nominal', 'nominal_trajectory.csv')
        rows = np.loadtxt(f)
        raw = np.zeros_like(rows)
        prev = np.zeros(rows.shape[1])
        for k in range(len(rows)):
            raw[k] = 2 * rows[k] - prev
            prev = raw[k]
        ship[tag + '_return'] = raw[:, -1].sum()
        ship[tag + '_ref'] = raw[::25, 0:3]
        ship[tag + '_u'] = raw[::25, 3:6]
        ship[tag + '_x'] = raw[::25, 6:18]
        ship[tag + '_reward'] = raw[:, -1]
        ship[tag + '_n'] = len(raw)
    if 'shipped' in parts:
        np.savez_compressed(os.path.join(out, 'shipped_csv.npz'), **ship)

    # open-loop dynamics KATs, every build
    kat = {}
    for b in (['h2000_v90', 'h2000_v150', 'h10000_v90', 'cg', 'cg_for', 'cg_timed', 'ice', 'gust', 'test'] if 'kat' in parts else []):
        sim = RefCitation(b)
        r = np.random.default_rng(11)
        cmds = np.zeros((3000, 10))
        tt = np.arange(3000) * 0.01
        cmds[:, 0] = np.deg2rad(1.5 * np.sin(0.9 * tt) + 0.2 * r.normal(size=3000))
        cmds[:, 1] = np.deg2rad(2.0 * np.sin(0.5 * tt + 1) + 0.2 * r.normal(size=3000))
        cmds[:, 2] = np.deg2rad(1.0 * np.sin(0.3 * tt))
        cmds[1500:, 8:10] = 0.05
        xs = np.stack([sim.step(c) for c in cmds])
        kat[b + '_cmd'] = cmds.astype(np.float64)
        kat[b + '_x'] = xs[::10]
        kat[b + '_xlast'] = xs[-1]
        assert np.isfinite(xs).all(), b
    if 'kat' in parts:
        np.savez_compressed(os.path.join(out, 'dyn_open_loop.npz'), **kat)
    print('done in %.0f s' % (time.time() - t0))


if __name__ == '__main__':
    main()
