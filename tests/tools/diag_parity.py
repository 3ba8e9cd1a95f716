#!/usr/bin/env python3
"""GPU-vs-oracle divergence diagnosis on the BASELINE pop=50 workload (run on the GPU box)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import serl_amd
from serl_amd import refsignals
from oracle import rollout as R
NET32 = dict(state_dim=7, action_dim=3, hidden=32, num_layers=3, activation='tanh')
w = np.load('tests/golden/actors.npz')['serl50']
moe = np.repeat(np.arange(50, dtype=np.int32), 3)
ref = refsignals.synthetic_reference_tables(150, 3, 80, seed=7)
o = R.rollout(w, NET32, moe, ref, t_max=80, threads=16, traces=True)
eng = serl_amd.RolloutEngine(0)
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
for lanes in (0, 1):
    out = eng.rollout(torch.from_numpy(w), spec, moe, ref, t_max=80, lanes_per_wave=lanes, traces=True)
    fit = out['fitness'].cpu().numpy()
    rel = np.abs(fit - o['fitness']) / np.abs(o['fitness'])
    order = np.argsort(rel)[::-1]
    print('lanes', lanes, 'kernel ms', eng.last_kernel_ms, 'max rel', rel.max(), 'episodes > 1e-5:', int((rel > 1e-5).sum()),
          'lengths equal', bool((out['length_steps'].cpu().numpy() == o['length_steps']).all()))
    A, S, Rw = out['actions'].cpu().numpy(), out['states'].cpu().numpy(), out['rewards'].cpu().numpy()
    for e in order[:3]:
        n = int(o['length_steps'][e])
        da = np.abs(A[e, :n] - o['actions'][e, :n]).max(1)
        ds = np.abs(S[e, :n] - o['states'][e, :n]) / (np.abs(o['states'][e, :n]) + 1e-3)
        dsm = ds.max(1)
        dr = np.abs(Rw[e, :n] - o['rewards'][e, :n])
        first = lambda d, thr: int(np.argmax(d > thr)) if (d > thr).any() else -1
        print(' episode', e, 'member', moe[e], 'rel', rel[e], 'fit', fit[e], o['fitness'][e])
        print('   first step action diff >1e-9/1e-7/1e-5:', first(da, 1e-9), first(da, 1e-7), first(da, 1e-5), 'max', da.max(), 'at', int(da.argmax()))
        print('   first step state rel >1e-12/1e-9/1e-6:', first(dsm, 1e-12), first(dsm, 1e-9), first(dsm, 1e-6), 'max', dsm.max(), 'at', int(dsm.argmax()), 'state idx', int(ds[int(dsm.argmax())].argmax()))
        print('   reward abs diff max', dr.max(), 'at', int(dr.argmax()), 'sum diff', (Rw[e, :n] - o['rewards'][e, :n]).sum())
        for k in (1, 10, 100, 1000, 4000, n - 1):
            print('   k=%d da=%.3g ds=%.3g' % (k, da[k], dsm[k]))
