#!/usr/bin/env python3
"""Which pairs of launches of a mixed-fault sweep cost more side by side than alone?  python tools/mixed_pairs.py [out.json]
768 episodes x 8 001 steps; the sweep restricted to two fault modes at a time (be = nominal code on the h2000_v90 tables, cg = the SAME code on
other tables, ice = another kernel): kernel time of the pair against the longer of its two launches alone."""
import os, sys, json, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, serl_amd
from serl_amd import refsignals
pop, ne, T = 256, 3, 8001
E = pop * ne
eng = serl_amd.RolloutEngine(0)
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
w = bench.make_population(pop, 0).cuda()
ref = torch.from_numpy(refsignals.synthetic_reference_tables(E, ne, 80, seed=7)).cuda()
res = {}


def run(modes):
    for _ in range(2):
        serl_amd.evaluate_pop(w, mode=modes, num_evals=ne, refs=ref, t_max=80, spec=spec, engine=eng)
    return eng.last_kernel_ms


for name, share in (('be+cg (same code, two table sets)', {'be': 4, 'cg': 2}), ('be+ice (two kernels)', {'be': 4, 'ice': 2}), ('ice+cg', {'ice': 3, 'cg': 3}),
                    ('be+ice+cg', {'be': 4, 'ice': 1, 'cg': 1}), ('be only', {'be': 6}), ('ice only', {'ice': 6}), ('cg only', {'cg': 6})):
    pat = [m for m, k in share.items() for _ in range(k)]
    modes = [pat[e % 6] for e in range(E)]
    res[name] = {'episodes': {m: modes.count(m) for m in share}, 'kernel_ms': round(run(modes), 2)}
    print(name, res[name], flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)
