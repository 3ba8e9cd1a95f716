#!/usr/bin/env python3
"""A/B timing of library builds on the GPU box: python tools/ab.py [E ...]  (SERL_LIB selects the .so)."""
import os, sys, json, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serl_amd
from serl_amd import refsignals
eng = serl_amd.RolloutEngine(0)
TAG = os.environ.get('AB_ACTORS', 'serl50')          # serl50 (H = 32) | serl10 (H = 72) | td3 (H = 96, LeakyReLU)
w = torch.from_numpy(np.load('tests/golden/actors.npz')[TAG])
spec = {'serl50': serl_amd.NetSpec(7, 3, 32, 3, 'tanh'), 'serl10': serl_amd.NetSpec(7, 3, 72, 3, 'tanh'),
        'td3': serl_amd.NetSpec(7, 3, 96, 3, 'relu')}[TAG]
ref = refsignals.tabulate(*refsignals.base_reference(20), 20)
n = ref.shape[0]
T = 400
res = {}
for E in [int(x) for x in sys.argv[1:]] or [150, 1024]:
    cmds = np.zeros((E, T, 10)); cmds[:, :, 0] = 0.01 * np.sin(np.arange(T) * 0.01)[None]
    eng.dynamics_open_loop(cmds); eng.dynamics_open_loop(cmds)
    res['dyn_E%d' % E] = round(eng.last_kernel_ms * 1e3 / T, 2)
    moe = np.arange(E) % len(w)
    eng.rollout(w, spec, moe, ref, t_max=20)
    out = eng.rollout(w, spec, moe, ref, t_max=20)
    res['loop_E%d' % E] = round(eng.last_kernel_ms * 1e3 / n, 2)
    if os.environ.get('SERL_PROFILE'):
        buf = (ctypes.c_ulonglong * 32)()
        eng.lib.serl_debug_profile(eng.ctx, buf)
        st = max(buf[3], 1)
        res['cyc_E%d' % E] = [int(buf[0] / st), int(buf[1] / st), int(buf[2] / st)]
        res['simd_E%d' % E] = [int(v) - 100 for v in buf[16:25] if v]      # SIMD of every wavefront of workgroup 0
        res['actor_busy_E%d' % E] = int(buf[31] / st)
        res['split_actor_E%d' % E] = dict(forward=[int(buf[26] / st), int(buf[27] / st)], partner_wait=[int(buf[28] / st), int(buf[29] / st)], obs_wait=[int(buf[30] / st), int(buf[31] / st)])                      # cycles per env step the actor wavefront spends on a forward pass
        if any(buf[4:16]):
            res['phase_E%d' % E] = [int(v / st) for v in buf[4:32]]
    res['fit0_E%d' % E] = float(out['fitness'][0])
print(os.environ.get('SERL_LIB', 'default'), json.dumps(res))
