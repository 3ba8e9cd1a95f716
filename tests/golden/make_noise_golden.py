#!/usr/bin/env python3
"""Golden vectors for the exploration-noise / stored-transition path, produced by the reference's OWN, unmodified
`Agent.evaluate(agent, is_action_noise=True, store_transition=True)` (base/core/agent.py:63-138) under refshim.py,
with recording stand-ins for the three replay buffers (base/core/replay_memory.py:21-31 `add(*transition)`).

  noise_path.npz  per case <c>:
      <c>_seed       np.random seed set right before the episode (the only consumer of the stream in a nominal-build
                     episode is agent.py:91 `noise_sd * np.random.randn(3)`, one call per step)
      <c>_ret        [fitness, length (= info['t']), steps, frames counted, episodes counted]
      <c>_cost       info['cost'] of EVERY step (phlabenv.py:369-375), int8[T]           -> pins SURVEY a7
      <c>_action     the `action` element of EVERY stored transition, f64[T,3]          -> pins agent.py:93,103
      <c>_head/_tail first / last 50 stored transitions (obs7, action3, next_obs7, reward, done) f64[50,19]
      <c>_ncrit      number of tuples that went to agent.critical_buffer; <c>_crit_idx their step indices
      <c>_smoothness Episode.smoothness (pins env.last_u, the smoothness input of agent.py:98, through the FFT)

Run in the build container (needs /root/reference):  python tests/golden/make_noise_golden.py
"""
import os, sys, io, contextlib, argparse
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refshim
refshim.install()
import make_golden as MG

NOISE_SD, NOISE_CLIP = 0.2962183114680794, 0.5      # base/parameters.py:49,76

CASES = [  # name, population tag, actor index, t_max, seed, noise on?
    ('serl50_18_n', 'serl50', 18, 20, 4101, True),
    ('td3_0_n', 'td3', 0, 20, 4102, True),           # the RL actor's exploration episode of agent.py:269
    ('serl50_7_n80', 'serl50', 7, 80, 4103, True),
    ('serl50_0_clean', 'serl50', 0, 20, 4104, False),  # store_transition without noise (the GA members' stored episodes)
]


class Recorder:
    def __init__(self):
        self.rows = []

    def add(self, obs, action, next_obs, reward, done):
        self.rows.append(np.concatenate([np.asarray(obs, np.float64), np.asarray(action, np.float64).reshape(-1),
                                         np.asarray(next_obs, np.float64), [float(reward)], [float(done)]]))


def run_case(tag, idx, t_max, seed, noisy):
    from core.agent import Agent
    sds, h, act = MG.load_pop(tag)
    actor = refshim.make_actor(sds[idx], h, 3, act)
    env = refshim.make_env('nominal', t_max)
    th, ph = MG.base_refs(t_max)
    costs, last_u = [], []

    class _Env:    # forwards reset() with the user refs and records info['cost'] / last_u of every step
        def __init__(self, e):
            self.__dict__['_e'] = e

        def reset(self):
            return self._e.reset(user_refs={'theta_ref': th, 'phi_ref': ph})

        def step(self, a):
            r = self._e.step(a)
            costs.append(int(r[3]['cost']))
            last_u.append(np.array(self._e.last_u, dtype=np.float64).copy())
            return r

        def __getattr__(self, k):
            return getattr(self._e, k)

    fake = argparse.Namespace()
    fake.args = argparse.Namespace(smooth_fitness=False, noise_sd=NOISE_SD, noise_clip=NOISE_CLIP)
    fake.env = _Env(env)
    fake.replay_buffer = Recorder()
    fake.num_frames, fake.gen_frames, fake.num_episodes = 0, 0, 0
    agent = argparse.Namespace(actor=actor, buffer=Recorder(), critical_buffer=Recorder())
    np.random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        ep = Agent.evaluate(fake, agent, noisy, True)
    rows = np.stack(fake.replay_buffer.rows)
    assert len(agent.buffer.rows) == len(rows) == len(costs)
    crit = np.stack(agent.critical_buffer.rows) if agent.critical_buffer.rows else np.zeros((0, 19))
    crit_idx = np.nonzero(np.asarray(costs))[0]
    assert len(crit) == len(crit_idx) and (len(crit) == 0 or np.array_equal(crit, rows[crit_idx]))
    return dict(seed=np.array(seed),
                ret=np.array([ep.fitness, ep.length, len(rows), fake.num_frames, fake.num_episodes], np.float64),
                cost=np.asarray(costs, np.int8), action=rows[:, 7:10].copy(), head=rows[:50].copy(), tail=rows[-50:].copy(),
                ncrit=np.array(len(crit)), crit_idx=crit_idx.astype(np.int32),
                smoothness=np.array(ep.smoothness))


def main():
    res = {}
    for name, tag, idx, t_max, seed, noisy in CASES:
        r = run_case(tag, idx, t_max, seed, noisy)
        for k, v in r.items():
            res['%s_%s' % (name, k)] = v
        print(name, r['ret'], 'cost steps', int(r['cost'].sum()), flush=True)
    np.savez_compressed(os.path.join(HERE, 'noise_path.npz'), **res)


if __name__ == '__main__':
    main()
