#!/usr/bin/env python3
"""Golden vectors for the literal drop-in of `Agent.evaluate` (serl_amd.make_evaluate): SEVERAL CONSECUTIVE episodes of the
REFERENCE'S OWN Agent.evaluate on ONE env object and ONE Agent (build container only) -> tests/golden/sequence.npz.

What a sequence exercises that single episodes do not: the tracking error carried over reset() into the next obs0
(envs/phlabenv.py:401-428), the model clock that initialize() does not reset (time-switched `gust` build), the shared /
per-agent / critical buffers and the num_frames / gen_frames / num_episodes counters across stored and unstored
episodes (agent.py:101-125), and the position of the np.random stream after episodes with exploration noise and
with the sensor model of the `gust` wrapper (draws interleaved per step).

  <seq>_plan      rows (actor index, is_action_noise, store_transition)
  <seq>_ret       per episode [fitness, length, smoothness, steps]
  <seq>_err0      env.error when each episode's reset() ran
  <seq>_counters  per episode [num_frames, gen_frames, num_episodes] afterwards
  <seq>_nbuf      per episode [len(shared), len(agent.buffer), len(agent.critical_buffer)] afterwards
  <seq>_shared_head / _tail   first / last 30 tuples of the shared buffer at the end (obs7, a3, next_obs7, r, done)
  <seq>_after     np.random.randn(4) drawn right after the sequence (pins the stream position)
"""
import os, sys, io, contextlib, argparse
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim
os.chdir('/tmp')
refshim.install()
import make_golden as MG
from make_noise_golden import Recorder, NOISE_SD, NOISE_CLIP

SEQS = {'nominal': ('nominal', 4242, [(18, 0, 0), (0, 1, 1), (7, 0, 1), (33, 1, 0)]),
        'gust': ('gust', 4343, [(18, 1, 1), (0, 0, 0), (7, 1, 1)])}


def main():
    from core.agent import Agent
    sds, h, act = MG.load_pop('serl50')
    out = {}
    for name, (mode, seed, plan) in SEQS.items():
        env = refshim.make_env(mode, 20)
        th, ph = MG.base_refs(20)

        class _Env:
            def __init__(self, e):
                self.__dict__['_e'] = e

            def reset(self):
                return self._e.reset(user_refs={'theta_ref': th, 'phi_ref': ph})

            def __getattr__(self, k):
                return getattr(self._e, k)
        fake = argparse.Namespace()
        fake.args = argparse.Namespace(smooth_fitness=False, noise_sd=NOISE_SD, noise_clip=NOISE_CLIP)
        fake.env = _Env(env)
        fake.replay_buffer = Recorder()
        fake.num_frames, fake.gen_frames, fake.num_episodes = 0, 0, 0
        agents = {i: argparse.Namespace(actor=refshim.make_actor(sds[i], h, 3, act), buffer=Recorder(), critical_buffer=Recorder())
                  for i in {p[0] for p in plan}}
        ret, err0, counters, nbuf = [], [], [], []
        np.random.seed(seed)
        for idx, noisy, store in plan:
            err0.append(np.array(env.error, dtype=np.float64).copy() if getattr(env, 'error', None) is not None else np.zeros(3))
            with contextlib.redirect_stdout(io.StringIO()):
                ep = Agent.evaluate(fake, agents[idx], bool(noisy), bool(store))
            ret.append([ep.fitness, ep.length, ep.smoothness, len(ep.reward_lst)])
            counters.append([fake.num_frames, fake.gen_frames, fake.num_episodes])
            nbuf.append([len(fake.replay_buffer.rows), len(agents[idx].buffer.rows), len(agents[idx].critical_buffer.rows)])
            print(name, idx, noisy, store, ret[-1], flush=True)
        out[name + '_after'] = np.random.randn(4)
        out[name + '_seed'] = np.array(seed)
        out[name + '_plan'] = np.array(plan)
        out[name + '_ret'] = np.array(ret)
        out[name + '_err0'] = np.array(err0)
        out[name + '_counters'] = np.array(counters)
        out[name + '_nbuf'] = np.array(nbuf)
        rows = np.stack(fake.replay_buffer.rows)
        out[name + '_shared_head'], out[name + '_shared_tail'] = rows[:30], rows[-30:]
    np.savez_compressed(os.path.join(HERE, 'sequence.npz'), **out)


if __name__ == '__main__':
    main()
