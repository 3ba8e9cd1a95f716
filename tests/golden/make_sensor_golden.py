#!/usr/bin/env python3
"""Golden vectors for the sensor-noise wrappers (envs/noise/citation.py:71-82, envs/gust/citation.py:72-86), produced by
the reference's OWN Python (CitationEnv + Agent.evaluate + the SWIG wrappers with their np.random draws) under the shims
of refshim.py:

  sensor_noise.npz   <mode>_<actor> = [fitness, length, smoothness, steps], seed_<mode>_<actor> = the np.random seed
                     set right before the episode (the wrapper draws randn(3), randn(1), randn(1), randn(2) per step())

Run in the build container (needs /root/reference):  python tests/golden/make_sensor_golden.py
"""
import os, sys, io, contextlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refshim
refshim.install()
import make_golden as MG


def main():
    th, ph = MG.base_refs()
    sds, h, act = MG.load_pop('serl50')
    res = {}
    for mode in ('noise', 'gust'):
        env = refshim.make_env(mode, 80)
        for j, i in enumerate((18, 0, 7)):
            seed = 1000 + 10 * ('noise', 'gust').index(mode) + j
            np.random.seed(seed)
            ep = MG.run_ref(env, refshim.make_actor(sds[i], h, 3, act), th, ph)
            res['%s_%d' % (mode, i)] = np.array([ep.fitness, ep.length, ep.smoothness, len(ep.reward_lst)])
            res['seed_%s_%d' % (mode, i)] = np.array(seed)
            print(mode, i, ep.fitness, ep.length, len(ep.reward_lst), flush=True)
    np.savez_compressed(os.path.join(HERE, 'sensor_noise.npz'), **res)


if __name__ == '__main__':
    main()
