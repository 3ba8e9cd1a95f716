#!/usr/bin/env python3
"""GPU micro-benchmarks of the rollout kernel pieces (run on the GPU box via gpurun)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serl_amd
from serl_amd import refsignals

eng = serl_amd.RolloutEngine(0)
res = {}
T = 400
for E, lanes in [(1, 0), (150, 0), (192, 0), (768, 0), (1024, 0), (2048, 0), (150, 1), (4096, 64)]:
    cmds = np.zeros((E, T, 10)); cmds[:, :, 0] = 0.01 * np.sin(np.arange(T) * 0.01)[None]
    eng.dynamics_open_loop(cmds, lanes_per_wave=lanes)
    eng.dynamics_open_loop(cmds, lanes_per_wave=lanes)
    ms = eng.last_kernel_ms
    res['dyn_E%d_L%d' % (E, lanes)] = dict(ms=ms, us_per_step=ms * 1e3 / T, steps_per_s=E * T / (ms * 1e-3))
w = torch.from_numpy(np.load('tests/golden/actors.npz')['serl50'])
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
ref = refsignals.tabulate(*refsignals.base_reference(4), 4)
n = ref.shape[0]
for E, lanes in [(1, 0), (150, 0), (192, 0), (768, 0), (1024, 0), (2048, 0), (4096, 0), (150, 1), (4096, 64)]:
    moe = np.arange(E) % 50
    eng.rollout(w, spec, moe, ref, t_max=4, lanes_per_wave=lanes)
    out = eng.rollout(w, spec, moe, ref, t_max=4, lanes_per_wave=lanes)
    ms = eng.last_kernel_ms
    res['loop_E%d_L%d' % (E, lanes)] = dict(ms=ms, us_per_step=ms * 1e3 / n, steps_per_s=E * n / (ms * 1e-3))
    if os.environ.get('SERL_PROFILE'):
        import ctypes
        buf = (ctypes.c_ulonglong * 32)()
        eng.lib.serl_debug_profile(eng.ctx, buf)
        st = max(buf[3], 1)
        res['loop_E%d_L%d' % (E, lanes)]['cycles_per_step'] = dict(actor=buf[0] / st, dyn=buf[1] / st, env=buf[2] / st)
for k, v in res.items():
    print(k, json.dumps(v))
