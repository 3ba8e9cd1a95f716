#!/usr/bin/env python3
"""Run-to-run reproducibility on the GPU box: python tools/repeat_check.py [runs] [E]   (SERL_LIB selects the .so)
the same rollout `runs` times; every fitness / length must be bit-identical between runs (a race shows up as a difference)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serl_amd
from serl_amd import refsignals
eng = serl_amd.RolloutEngine(0)
TAG = os.environ.get('AB_ACTORS', 'serl50')          # serl50 (H = 32) | serl10 (H = 72: the actor streams its weights) | td3 (H = 96)
w = torch.from_numpy(np.load('tests/golden/actors.npz')[TAG])
spec = {'serl50': serl_amd.NetSpec(7, 3, 32, 3, 'tanh'), 'serl10': serl_amd.NetSpec(7, 3, 72, 3, 'tanh'),
        'td3': serl_amd.NetSpec(7, 3, 96, 3, 'relu')}[TAG]
ref = refsignals.tabulate(*refsignals.base_reference(20), 20)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
E = int(sys.argv[2]) if len(sys.argv) > 2 else 150
moe = np.arange(E) % len(w)
base, bad = None, 0
for r in range(runs):
    out = eng.rollout(w, spec, moe, ref, t_max=20, build=os.environ.get('AB_BUILD', 'h2000_v90'))
    fit = out['fitness'].cpu().numpy().copy(); ln = out['length_steps'].cpu().numpy().copy()
    if base is None:
        base = (fit, ln)
    elif not (np.array_equal(fit, base[0], equal_nan=False) and np.array_equal(ln, base[1])):
        bad += 1
        d = np.flatnonzero(~((fit == base[0]) & (ln == base[1])))
        print('run', r, 'differs in episodes', d[:10].tolist(), fit[d[:3]].tolist(), base[0][d[:3]].tolist())
if os.environ.get('REPEAT_OUT'):      # the FIRST launch's results, for the caller to compare with the oracle (tests/test_gpu_rollout.py)
    np.savez(os.environ['REPEAT_OUT'], fitness=base[0], length_steps=base[1], moe=moe)
print(os.environ.get('SERL_LIB', 'default'), json.dumps({'build': os.environ.get('AB_BUILD', 'h2000_v90'), 'runs': runs, 'episodes': E, 'runs_that_differ': bad, 'nan_in_first_run': int(np.isnan(base[0]).sum())}))
