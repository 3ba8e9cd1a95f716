#!/usr/bin/env python3
"""ONE launch of the lane-per-episode kernels for counter passes: python tools/lane_rollout.py [episodes=16384] [t_max=5] [lanes=64]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, serl_amd
from serl_amd import refsignals
E = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
t_max = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 64
eng = serl_amd.RolloutEngine(0)
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
w = bench.make_population(2048, 0, tag='serl50').to(eng.device)
ref = refsignals.tabulate(*refsignals.base_reference(20), 20)[:refsignals.n_steps_for(t_max)]
out = eng.rollout(w, spec, (np.arange(E) % 2048).astype(np.int32), ref, t_max=t_max, lanes_per_wave=lanes)
print(json.dumps(dict(episodes=E, steps=int(out['length_steps'].abs().sum()), kernel_ms=eng.last_kernel_ms, **eng.last_rollout_info())))
