#!/usr/bin/env python3
"""Index decisions of the REFERENCE'S OWN SSNE.epoch (base/core/mod_neuro_evo.py:447-543), recorded with its tensor
operators replaced by recorders (build container only) -> tests/golden/ssne_epoch.npz.

For a few (seed, fitness vector) cases: the sequence of clone / crossover_inplace / mutate calls epoch() makes and its
return value.  Seeds on which the reference trips over its own inclusive-randint IndexError are skipped."""
import os, sys, types, random, json
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim
os.chdir('/tmp')
refshim.install()
from core.mod_neuro_evo import SSNE


def run(seed, fitness, num_elit, mutation_prob):
    ops = []
    pop = list(range(len(fitness)))          # "agents" are their indices
    fake = types.SimpleNamespace()
    fake.num_elitists, fake.population_size, fake.rl_policy = num_elit, len(fitness), None
    fake.args = types.SimpleNamespace(distil_crossover=False, crossover_prob=0.0, mutation_prob=mutation_prob, mutation_mag=0.05)
    fake.selection_tournament = types.MethodType(SSNE.selection_tournament, fake)
    fake.clone = lambda master, replacee: ops.append((0, master, replacee))
    fake.crossover_inplace = lambda a, b: ops.append((1, a, b))
    fake.mutate = lambda g, mag: ops.append((2, g, -1))
    fake.stats = types.SimpleNamespace(reset=lambda: None)
    random.seed(seed); np.random.seed(seed)
    ret = SSNE.epoch(fake, pop, fitness)
    return np.array(ops, dtype=np.int64).reshape(-1, 3), int(ret)


def main():
    rng = np.random.default_rng(0)
    out, cases = {}, []
    for pop, elit in ((10, 1), (50, 5), (50, 10), (7, 2)):
        fit = rng.normal(-200, 80, pop)
        n = 0
        for seed in range(100):
            try:
                ops, ret = run(seed, fit, elit, 0.9)
            except IndexError:
                continue
            key = 'p%d_e%d_s%d' % (pop, elit, seed)
            out[key + '_ops'], out[key + '_ret'], out[key + '_fit'] = ops, np.array(ret), fit
            cases.append(key); n += 1
            if n == 3:
                break
    np.savez_compressed(os.path.join(HERE, 'ssne_epoch.npz'), **out)
    print(cases)


if __name__ == '__main__':
    main()
