#!/usr/bin/env python3
"""ONE rollout launch for counter passes: python tools/one_rollout.py [episodes=150] [t_max=20] [actors=serl50]   (SERL_LIB selects the library).
150 episodes of the SERL50 shape x 2 001 env steps on the base reference; prints the kernel time and the family that ran."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import serl_amd
from serl_amd import refsignals
E = int(sys.argv[1]) if len(sys.argv) > 1 else 150
t_max = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
tag = sys.argv[3] if len(sys.argv) > 3 else 'serl50'
eng = serl_amd.RolloutEngine(0)
w = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'actors.npz'))[tag])
spec = {'serl50': serl_amd.NetSpec(7, 3, 32, 3, 'tanh'), 'serl10': serl_amd.NetSpec(7, 3, 72, 3, 'tanh'), 'td3': serl_amd.NetSpec(7, 3, 96, 3, 'relu')}[tag]
ref = refsignals.tabulate(*refsignals.base_reference(t_max), t_max)
out = eng.rollout(w, spec, np.arange(E) % len(w), ref, t_max=t_max)
print(json.dumps(dict(lib=os.environ.get('SERL_LIB', 'product'), episodes=E, steps=int(out['length_steps'].abs().sum()), kernel_ms=eng.last_kernel_ms, **eng.last_rollout_info())))
