#!/usr/bin/env python3
"""BASELINE config 5 on one GPU's share: PH-LAB mixed-fault sweep, fault mode per episode = e mod 6 over
{be, jr, sa, se, ice, cg} -- three dynamics builds, hence three launches per evaluation, side by side on streams of
their own (evaluate_pop) or one after the other (--sequential).   python tools/bench_mixed.py [--pop 256] [--steps 3]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serl_amd
from serl_amd import refsignals
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--pop', type=int, default=256)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--sequential', action='store_true')
a = ap.parse_args()
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
ne = 3
E = a.pop * ne
w = bench.make_population(a.pop, 0).cuda()
refs = torch.from_numpy(refsignals.synthetic_reference_tables(E, ne, 80, seed=7)).cuda()
modes = [['be', 'jr', 'sa', 'se', 'ice', 'cg'][e % 6] for e in range(E)]
eng = serl_amd.RolloutEngine(0)
res = serl_amd.evaluate_pop(w, mode=modes, num_evals=ne, refs=refs, t_max=80, spec=spec, engine=eng, concurrent=not a.sequential)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    res = serl_amd.evaluate_pop(w, mode=modes, num_evals=ne, refs=refs, t_max=80, spec=spec, engine=eng,
                                concurrent=not a.sequential)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
steps = int(np.abs(res.length_steps).sum())
print(json.dumps({'workload': 'PH-LAB mixed-fault sweep (be/jr/sa/se/ice/cg by episode), pop=%d x 3 evals x 8001 steps on one GPU' % a.pop,
                  'launches': 'sequential' if a.sequential else 'concurrent streams', 'episodes': E, 'env_steps': steps,
                  'ms_per_evaluation': dt * 1e3, 'kernel_ms': res.kernel_ms, 'env_steps_per_s': steps / dt,
                  'fitness_checksum': float(res.returns.sum())}))
