#!/usr/bin/env python3
"""SSNE operator known-answer vectors from the REFERENCE'S OWN base/core/mod_neuro_evo.py (build container only).

  ga_ops.npz   for a few (seed, actor pair) cases: the packed f32 weights after the reference's
               crossover_inplace(gene1, gene2) / mutate_inplace(gene, mag) / clone(master, replacee), with python
               `random` and `numpy.random` seeded beforehand.  Seeds are chosen so that the reference does not hit
               its own inclusive-randint IndexError (mod_neuro_evo.py:76,89,357-358) in these cases.
"""
import os, sys, types, random
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim
os.chdir('/tmp')
refshim.install()
import torch
import make_golden as MG
from oracle import rollout as R
from core.mod_neuro_evo import SSNE


def gene(sd, h, act):
    return types.SimpleNamespace(actor=refshim.make_actor({k: v.clone() for k, v in sd.items()}, h, 3, act))


def main():
    sds, h, act = MG.load_pop('serl50')
    fake = types.SimpleNamespace(regularize_weight=lambda w, mag: torch.clamp(w, -mag, mag))
    out = {}
    n_ok = 0
    for seed in range(200):
        g1, g2 = gene(sds[2], h, act), gene(sds[3], h, act)
        random.seed(seed); np.random.seed(seed)
        try:
            SSNE.crossover_inplace(fake, g1, g2)
        except IndexError:
            continue
        out['cross_seed%d_a' % seed] = R.pack_state_dict(g1.actor.state_dict())
        out['cross_seed%d_b' % seed] = R.pack_state_dict(g2.actor.state_dict())
        n_ok += 1
        if n_ok == 3:
            break
    n_ok = 0
    for seed in range(200):
        g = gene(sds[5], h, act)
        random.seed(seed); np.random.seed(seed)
        try:
            SSNE.mutate_inplace(fake, g, 0.05)
        except IndexError:
            continue
        out['mut_seed%d' % seed] = R.pack_state_dict(g.actor.state_dict())
        n_ok += 1
        if n_ok == 3:
            break
    out['base'] = np.stack([R.pack_state_dict(sds[i]) for i in (2, 3, 5)])
    np.savez_compressed(os.path.join(HERE, 'ga_ops.npz'), **out)
    print(sorted(out))


if __name__ == '__main__':
    main()
