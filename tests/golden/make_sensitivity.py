#!/usr/bin/env python3
"""Rounding sensitivity of the golden episodes (run in the build container; ~5 min).

The reference computes the actor in f32 with torch's CPU kernels, whose summation order (and tanh) differ from any
other f32 implementation by an ulp here and there.  Most shipped actors damp such perturbations (the episodic
return moves by ~1e-8 relative); a few poorly trained ones oscillate and amplify them to 1e-3.  This script runs
the REFERENCE'S OWN evaluation a second time with the actor's f32 output nudged by one ulp (seeded, +-1 ulp on a
random channel each step) and stores the alternative returns: |alt - fitness| is the precision to which the
reference itself defines each number, and the parity tests widen their 1e-5 tolerance for exactly those episodes.

  sensitivity.npz   <tag>_alt : f64[n_actors]  for tag in serl50, serl10, td3  (base reference, nominal, t_max = 80)
                    fault_<mode>_<actor> : f64   for the episodes of faults.npz
"""
import os, sys, io, contextlib, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim
os.chdir('/tmp')
refshim.install()
import torch
torch.set_num_threads(1)
import make_golden as MG


class Nudged:
    """actor whose select_action output is moved by one f32 ulp on one channel per call"""
    def __init__(self, actor, seed):
        self.actor, self.rng = actor, np.random.default_rng(seed)

    def eval(self):
        self.actor.eval()

    def select_action(self, obs):
        a = np.array(self.actor.select_action(obs), dtype=np.float32)
        i = self.rng.integers(0, 3)
        a[i] = np.nextafter(a[i], np.float32(2.0 if self.rng.integers(0, 2) else -2.0), dtype=np.float32)
        return a


def main():
    t0 = time.time()
    th, ph = MG.base_refs()
    env = refshim.make_env('nominal', 80)
    out = {}
    prev = os.path.join(HERE, 'sensitivity.npz')
    have = dict(np.load(prev)) if os.path.exists(prev) and '--all' not in sys.argv else {}
    out.update(have)
    for tag in ('serl50', 'serl10', 'td3'):
        if tag + '_alt' in out:
            continue
        sds, h, act = MG.load_pop(tag)
        alt = []
        for i, sd in enumerate(sds):
            ep = MG.run_ref(env, argparse_ns(Nudged(refshim.make_actor(sd, h, 3, act), 1000 + i)), th, ph)
            alt.append(ep.fitness)
            print(tag, i, ep.fitness, '%.0fs' % (time.time() - t0), flush=True)
        out[tag + '_alt'] = np.array(alt)
    sds, h, act = MG.load_pop('serl50')
    for m in ['be', 'jr', 'sa', 'se', 'ice', 'cg', 'cg-for', 'high-q', 'low-q', 'cg-shift', 'gust']:
        envm = refshim.make_env(m, 80)
        for i in (18, 0, 7):
            ep = MG.run_ref(envm, Nudged(refshim.make_actor(sds[i], h, 3, act), 2000 + i), th, ph)
            out['fault_%s_%d' % (m, i)] = np.array(ep.fitness)
            print(m, i, ep.fitness, '%.0fs' % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, 'sensitivity.npz'), **out)


def argparse_ns(actor):
    return actor


if __name__ == '__main__':
    main()
