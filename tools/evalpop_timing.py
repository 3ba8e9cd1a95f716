#!/usr/bin/env python3
"""Wall time of the reference's `-eval_pop` loop through serl_amd.validate_pop (base/evaluate.py:236-256: every actor of the
population flies num_trails + 1 references, nMAE / smoothness / Stats per actor, champion) against the kernel time inside it.
    python tools/evalpop_timing.py [repeats]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import serl_amd
from serl_amd import refsignals

R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
engine = serl_amd.RolloutEngine(0)
w = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'actors.npz'))['serl50'])
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
refs = refsignals.synthetic_reference_tables(6, 6, 80, seed=3)
out = []
for carry in (False, True):
    serl_amd.validate_pop(w, refs, spec=spec, engine=engine, t_max=80, carry_error=carry)
    ts, ks = [], []
    for _ in range(R):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = serl_amd.validate_pop(w, refs, spec=spec, engine=engine, t_max=80, carry_error=carry)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); ks.append(engine.last_kernel_ms)
    out.append(dict(carry_error=carry, wall_ms=round(1e3 * float(np.mean(ts)), 1), last_kernel_ms=round(float(np.mean(ks)), 1),
                    champion=int(res['champion'])))
print(json.dumps(dict(what='validate_pop: 50 actors x 6 references x 8 001 steps (300 episodes)', runs=out)))
