#!/usr/bin/env python3
"""BASELINE config 1: PH-LAB nominal h2000_v90, pop=10 (SERL10: actor 7-72x4-3 tanh), 3 evals x 80 s episodes, one GPU.
The H = 72 actor does not fit the LDS staging of the team kernels (SERL50 shape only): its rows stream from L2.
   python tools/bench_serl10.py [--steps 3]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serl_amd
from serl_amd import refsignals, metrics

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=3)
a = ap.parse_args()
spec = serl_amd.NetSpec(7, 3, 72, 3, 'tanh')
w = torch.from_numpy(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'actors.npz'))['serl10']).cuda()
pop, ne = w.shape[0], 3
E = pop * ne
moe = np.repeat(np.arange(pop, dtype=np.int32), ne)
ref = torch.from_numpy(refsignals.synthetic_reference_tables(E, ne, 80, seed=7)).cuda()
eng = serl_amd.RolloutEngine(0)


def one():
    out = eng.rollout(w, spec, moe, ref, t_max=80.0, traces='actions', sync=False)
    sm = metrics.calc_smoothness(out['actions'], out['length_steps'])
    return out, sm
one(); torch.cuda.synchronize()
t0 = time.perf_counter()
km = []
for _ in range(a.steps):
    out, sm = one()
    km.append(eng.kernel_ms())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
steps = int(out['length_steps'].abs().sum())
print(json.dumps({'workload': 'PH-LAB nominal, pop=10 (SERL10 actors 7-72x4-3 tanh) x 3 evals x 8001 steps, 1 GPU', 'episodes': E,
                  'env_steps': steps, 'ms_per_evaluation': dt * 1e3, 'kernel_ms': float(np.mean(km)), 'env_steps_per_s': steps / dt,
                  't_step_us': float(np.mean(km)) * 1e3 / ref.shape[1], 'fitness_mean_per_member': out['fitness'].view(pop, ne).mean(1).tolist()}))
