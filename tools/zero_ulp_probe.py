#!/usr/bin/env python3
"""GPU vs the two flavours of the CPU oracle (glibc / the product's short libm, oracle/citation_rt.h), episode by episode:
how many episodes are bit-identical, the largest relative difference of the episodic return.  Development aid for the zero-tolerance
parity tests (tests/test_gpu_rollout.py); run on the GPU box:  python tools/zero_ulp_probe.py"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
import serl_amd
from serl_amd import refsignals
from oracle import rollout as R
import handover_stress as hs

eng = serl_amd.RolloutEngine(0)
cus = torch.cuda.get_device_properties(0).multi_processor_count
A = np.load(os.path.join(ROOT, 'tests', 'golden', 'actors.npz'))
NET = dict(hs.NET, td3=dict(state_dim=7, action_dim=3, hidden=96, num_layers=3, activation='relu'))


def spec_of(n):
    return serl_amd.NetSpec(n['state_dim'], n['action_dim'], n['hidden'], n['num_layers'], n['activation'])


def compare(name, w, net, moe, ref, build='h2000_v90', t_max=80, tick0=None, hint=None):
    eng.kernel_hint = hint
    r = eng.rollout(torch.from_numpy(np.ascontiguousarray(w)), spec_of(net), moe, ref, build=build, t_max=t_max, tick0=tick0)
    f, ln = r['fitness'].cpu().numpy(), r['length_steps'].cpu().numpy()
    out = dict(case=name, build=build, episodes=len(moe))
    for flavour in (False, True):
        o = R.rollout(w, net, moe, ref, build=build, t_max=t_max, tick0=tick0, threads=16, short_libm=flavour)
        rel = np.abs(f - o['fitness']) / np.abs(o['fitness'])
        out['short' if flavour else 'glibc'] = dict(identical=int((f == o['fitness']).sum()), max_rel=float(rel.max()), lengths_equal=bool((ln == o['length_steps']).all()),
                                                    worst=[int(i) for i in np.argsort(-rel)[:3] if rel[i] > 0])
    print(json.dumps(out), flush=True)


ref80 = refsignals.tabulate(*refsignals.base_reference(80), 80)
for tag in ('serl50', 'serl10', 'td3'):
    w = A[tag]
    w = w if w.ndim == 2 else w[None]
    compare('shipped_' + tag, w, NET[tag], np.arange(len(w)), ref80)
for build in ('h2000_v90', 'ice', 'cg_timed', 'gust', 'test'):
    for name, case in hs.cases(cus).items():
        if name in ('teams_split', 'queue') and build != 'h2000_v90':
            continue
        w, moe, ref, tick0 = hs.inputs(name, case, build)
        compare(name, w, hs.NET[case[0]], moe, ref, build=build, t_max=hs.T_MAX, tick0=tick0, hint=case[1])
