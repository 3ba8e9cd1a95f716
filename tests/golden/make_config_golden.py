#!/usr/bin/env python3
"""Golden vectors for the env configurations outside BASELINE's (`CitationEnv(configuration, mode)`, envs/phlabenv.py:84-97,
174-176, 205-220, 377-380): 'symmetric' (1 action, obs = [e_theta, q]), 'full' (obs = [e(3), x[0:10]]) and the
'incremental' control mode (actuator RATE commands, bound 25 deg/s, u = last_u + scaled * dt, obs + last_u), produced by the
reference's OWN, unmodified `Agent.evaluate(agent, is_action_noise, store_transition=True)` (base/core/agent.py:63-138) under
refshim.py with the reference's own `Actor` (random initialisation, torch seed below) in training mode: the env draws its
reference signals itself (`init_ref`, phlabenv.py:303-345; the symmetric configuration ignores user references).

  config.npz  per case <c>:
      <c>_cfg      [config id (0 attitude, 1 symmetric, 2 full), incremental, state_dim, action_dim, hidden, num_layers]
      <c>_w        the actor's packed f32 parameter row (state_dict order)
      <c>_ref      info['ref'] of every step: the reference values the env tracked, rad, f64 [T, A]
      <c>_noise    the clipped exploration noise added to the action at every step, f64 [T, A] (zeros without noise)
      <c>_rows     every stored transition (obs S, action A, next_obs S, reward, done) f64 [T, 2S + A + 2]
      <c>_cost     info['cost'] of every step, int8 [T]
      <c>_u        env.last_u after every step (what Agent.evaluate collects for the smoothness), f64 [T, A]
      <c>_ret      [fitness, length (= info['t']), steps, smoothness]
      <c>_mode     the env's mode string ('nominal', 'incremental', 'be', 'ice')

Run in the build container (needs /root/reference):  python tests/golden/make_config_golden.py
"""
import os, sys, io, contextlib, argparse
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refshim
refshim.install()

NOISE_SD, NOISE_CLIP = 0.2962183114680794, 0.5      # base/parameters.py:49,76
CONFIG_ID = {'attitude': 0, 'symmetric': 1, 'full': 2}

CASES = [  # name, configuration, mode, hidden, layers, activation, torch seed, numpy seed, noise?, output-layer scale
         # (a freshly initialised actor saturates its tanh outputs and crashes within seconds: two cases get a timid output layer and fly the 20 s)
    ('sym', 'symmetric', 'nominal', 32, 3, 'tanh', 11, 5101, False, 1.0),
    ('sym_soft', 'symmetric', 'nominal', 32, 3, 'tanh', 21, 5111, False, 0.02),
    ('sym_inc_n', 'symmetric', 'incremental', 32, 3, 'tanh', 12, 5102, True, 1.0),
    ('full', 'full', 'nominal', 32, 3, 'tanh', 13, 5103, False, 1.0),
    ('att_inc', 'attitude', 'incremental', 32, 3, 'tanh', 14, 5104, False, 1.0),
    ('att_inc_soft', 'attitude', 'incremental', 32, 3, 'tanh', 24, 5114, True, 0.02),
    ('full_inc_n', 'full', 'incremental', 72, 3, 'relu', 15, 5105, True, 1.0),
    ('att_inc_n', 'attitude', 'incremental', 32, 3, 'elu', 16, 5106, True, 1.0),
    # the other configurations on fault / off-nominal builds (the wrappers of envs/{be,ice}/citation.py see the padded command)
    ('full_be', 'full', 'be', 32, 3, 'tanh', 17, 5107, False, 0.05),
    ('sym_ice', 'symmetric', 'ice', 32, 3, 'tanh', 18, 5108, True, 0.05),
]


class Recorder:
    def __init__(self):
        self.rows = []

    def add(self, obs, action, next_obs, reward, done):
        self.rows.append(np.concatenate([np.asarray(obs, np.float64), np.asarray(action, np.float64).reshape(-1),
                                         np.asarray(next_obs, np.float64), [float(reward)], [float(done)]]))


def run_case(cfg, mode, hidden, layers, activation, tseed, nseed, noisy, out_scale):
    import torch
    from core.agent import Agent
    from core.genetic_agent import Actor
    from envs.phlabenv import CitationEnv
    from serl_amd.actor import NetSpec
    with contextlib.redirect_stdout(io.StringIO()):
        env = CitationEnv(configuration=cfg, mode=mode)
    S, A = env.n_obs, env.n_actions
    torch.manual_seed(tseed)
    args = argparse.Namespace(hidden_size=hidden, num_layers=layers, activation_actor=activation, state_dim=S,
                              action_dim=A, device=torch.device('cpu'))
    actor = Actor(args)
    with torch.no_grad():
        actor.net[-2].weight.mul_(out_scale)
        actor.net[-2].bias.mul_(out_scale)
    actor.eval()
    spec = NetSpec(S, A, hidden, layers, activation)
    w = np.zeros(spec.row_stride, np.float32)
    sd = actor.state_dict()
    for name, off, shape in spec.param_layout():
        w[off:off + int(np.prod(shape))] = sd[name].numpy().reshape(-1)
    costs, last_u, refs, draws = [], [], [], []
    real_randn = np.random.randn

    def randn(*shape):            # agent.py:91: the only randn consumer after reset()
        v = real_randn(*shape)
        draws.append(np.array(v, dtype=np.float64).reshape(-1))
        return v

    class _Env:
        def __init__(self, e):
            self.__dict__['_e'] = e

        def reset(self):
            o = self._e.reset()
            np.random.randn = randn
            return o

        def step(self, a):
            r = self._e.step(a)
            costs.append(int(r[3]['cost']))
            refs.append(np.array(r[3]['ref'], dtype=np.float64).reshape(-1).copy())
            last_u.append(np.array(self._e.last_u, dtype=np.float64).reshape(-1).copy())
            return r

        def __getattr__(self, k):
            return getattr(self._e, k)

    fake = argparse.Namespace()
    fake.args = argparse.Namespace(smooth_fitness=False, noise_sd=NOISE_SD, noise_clip=NOISE_CLIP)
    fake.env = _Env(env)
    fake.replay_buffer = Recorder()
    fake.num_frames, fake.gen_frames, fake.num_episodes = 0, 0, 0
    agent = argparse.Namespace(actor=actor, buffer=Recorder(), critical_buffer=Recorder())
    np.random.seed(nseed)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ep = Agent.evaluate(fake, agent, noisy, True)
    finally:
        np.random.randn = real_randn
    rows = np.stack(fake.replay_buffer.rows)
    T = len(rows)
    assert rows.shape[1] == 2 * S + A + 2 and len(costs) == T
    noise = np.zeros((T, A))
    if noisy:
        assert len(draws) == T
        noise = np.clip(NOISE_SD * np.stack(draws), -NOISE_CLIP, NOISE_CLIP)
    return dict(cfg=np.array([CONFIG_ID[cfg], int(mode == 'incremental'), S, A, hidden, layers], np.int32), w=w,
                ref=np.stack(refs), noise=noise, rows=rows, cost=np.asarray(costs, np.int8), u=np.stack(last_u),
                ret=np.array([ep.fitness, ep.length, T, ep.smoothness], np.float64)), activation


def main():
    res = {}
    for name, cfg, mode, hidden, layers, activation, tseed, nseed, noisy, out_scale in CASES:
        r, act = run_case(cfg, mode, hidden, layers, activation, tseed, nseed, noisy, out_scale)
        for k, v in r.items():
            res['%s_%s' % (name, k)] = v
        res['%s_act' % name] = np.array(act)
        res['%s_mode' % name] = np.array(mode)
        print(name, r['cfg'], r['ret'], 'cost steps', int(r['cost'].sum()), flush=True)
    np.savez_compressed(os.path.join(HERE, 'config.npz'), **res)


if __name__ == '__main__':
    main()
