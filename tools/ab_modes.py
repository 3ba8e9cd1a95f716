#!/usr/bin/env python3
"""Per-mode timing of the mixed-fault workload on the GPU box: python tools/ab_modes.py [E]   (SERL_LIB selects the .so)
one evaluate_pop per fault mode / dynamics build with E episodes (t_max = 20 s), kernel ms and us per env step of a team."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serl_amd
eng = serl_amd.RolloutEngine(0)
w = torch.from_numpy(np.load('tests/golden/actors.npz')['serl50'])
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
res = {}
for mode in ['nominal', 'be', 'jr', 'sa', 'se', 'ice', 'cg']:
    wd = w[np.arange(E) % len(w)].contiguous().cuda()
    for _ in range(2):
        r = serl_amd.evaluate_pop(wd, mode=mode, num_evals=1, t_max=20, spec=spec, engine=eng)
    steps = float(np.sum(np.abs(r.length_steps)))
    res[str(mode)] = [round(eng.last_kernel_ms, 2), int(steps / E), round(float(np.mean(r.returns)), 6)]
print(os.environ.get('SERL_LIB', 'default'), json.dumps(res))
