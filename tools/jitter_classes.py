#!/usr/bin/env python3
"""Which hand-over is only right because its producer is usually early?  (diagnosis on the GPU box, jitter library)

  SERL_LIB=serl_amd/csrc/libserl_amd_jitter.so python tools/jitter_classes.py [build] [case ...]

Runs the cases of tests/tools/handover_stress.py with pauses at ONE class of sites at a time (SERL_JITTER_SITES = 1 << class,
serl_amd/csrc/citation_wave.h: citw_jitter_class_) and prints, per class, how many episodes differ from the run without pauses.
A class whose pauses change a result marks a read that is ordered by timing, not by a flag or a barrier."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
import torch, serl_amd
import handover_stress as hs

CLASSES = ['flag raise', 'flag wait', 'iflag raise', 'iflag wait', 'pflag raise', 'pflag wait', 'value poll', 'actor: before its flag',
           'team: behind the action wait', 'team: before the observation flag', 'behind the ODE5 combination', 'behind a lane-group step',
           'actor barriers', 'before B1', 'behind B1', 'before B2', 'behind B2']
build = sys.argv[1] if len(sys.argv) > 1 else 'h2000_v90'
names = sys.argv[2:] or ['team', 'teams', 'team4']
cus = torch.cuda.get_device_properties(0).multi_processor_count
cases = hs.cases(cus)


def run(seed, sites):
    os.environ['SERL_JITTER_SEED'] = str(seed); os.environ['SERL_JITTER_SITES'] = hex(sites)
    eng = serl_amd.RolloutEngine(0)
    out = {}
    for name in names:
        tag, kern, n = cases[name]
        w, moe, ref, tick0 = hs.inputs(name, cases[name], build)
        n_ = hs.NET[tag]
        eng.kernel_hint = kern
        r = eng.rollout(torch.from_numpy(w), serl_amd.NetSpec(n_['state_dim'], n_['action_dim'], n_['hidden'], n_['num_layers'], n_['activation']),
                        moe, ref, build=build, t_max=hs.T_MAX, tick0=tick0)
        out[name] = (r['fitness'].cpu().numpy().copy(), r['length_steps'].cpu().numpy().copy())
    eng.close()
    return out


base = run(0, 0xffffffff)
res = {}
for c, label in enumerate(CLASSES):
    row = {}
    for seed in (1, 2):
        o = run(seed, 1 << c)
        for name in names:
            d = np.flatnonzero((o[name][0] != base[name][0]) | (o[name][1] != base[name][1]))
            rel = float(np.nanmax(np.abs(o[name][0][d] / base[name][0][d] - 1))) if len(d) else 0.0
            k = row.setdefault(name, [0, 0.0])
            k[0] += len(d); k[1] = max(k[1], rel)
    res[label] = row
    print('%2d %-36s' % (c, label), ' '.join('%s: %d differ (%.1e)' % (n, row[n][0], row[n][1]) for n in names), flush=True)
print(json.dumps(res))
