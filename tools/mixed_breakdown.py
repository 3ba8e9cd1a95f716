#!/usr/bin/env python3
"""Where the mixed-fault sweep's time goes (BASELINE config 5): python tools/mixed_breakdown.py [pop] [out.json]
pop members x 3 evals x 8 001 steps, fault mode per episode = e mod 6 (bench.py --workload mixed): every dynamics build's launch ALONE
(sized as in the sweep: concurrent_episodes tells it the others), the three launches side by side, the same number of nominal episodes
in one launch, and the wall time of evaluate_pop around the kernels."""
import os, sys, json, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, serl_amd
from serl_amd import refsignals, builds
pop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ne, T = 3, 8001
E = pop * ne
eng = serl_amd.RolloutEngine(0)
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
w = bench.make_population(pop, 0).cuda()
ref = torch.from_numpy(refsignals.synthetic_reference_tables(E, ne, 80, seed=7)).cuda()
moe = np.repeat(np.arange(pop, dtype=np.int32), ne)
modes = [bench.MIXED_MODES[e % 6] for e in range(E)]
res = {'pop': pop, 'episodes': E}
rs = [builds.resolve_mode(m) for m in modes]
groups = {}
for e, (b, _) in enumerate(rs):
    groups.setdefault(b, []).append(e)
for b, idx in groups.items():
    idx = np.asarray(idx)
    rows = [rs[e][1] for e in idx]
    faults = np.array(rows, dtype=np.float64) if any(r != builds.NOMINAL_ROW for r in rows) else None
    for alone in (False, True):
        for _ in range(2):
            eng.rollout(w, spec, moe[idx], ref[torch.as_tensor(idx).cuda()], build=b, faults=faults, t_max=80.0, traces='actions', sync=False,
                        concurrent_episodes=0 if alone else E - len(idx))
        # (a launch that shares the GPU is not timed by the context's events: time it from the host)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.rollout(w, spec, moe[idx], ref[torch.as_tensor(idx).cuda()], build=b, faults=faults, t_max=80.0, traces='actions', sync=False,
                    concurrent_episodes=0 if alone else E - len(idx))
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        res['%s_%s' % (b, 'alone_on_the_gpu' if alone else 'sized_as_in_the_sweep')] = {'episodes': len(idx), 'ms': round(ms, 2), 'us_per_env_step_of_a_round': round(ms * 1e3 / T, 2)}
for _ in range(2):
    r = serl_amd.evaluate_pop(w, mode=modes, num_evals=ne, refs=ref, t_max=80, spec=spec, engine=eng)
torch.cuda.synchronize(); t0 = time.perf_counter()
r = serl_amd.evaluate_pop(w, mode=modes, num_evals=ne, refs=ref, t_max=80, spec=spec, engine=eng)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
res['sweep_side_by_side'] = {'kernel_ms': round(eng.last_kernel_ms, 2), 'evaluate_pop_wall_ms': round(wall, 2), 'env_steps_per_s_kernel': round(E * T / eng.last_kernel_ms * 1e3)}
for _ in range(2):
    eng.rollout(w, spec, moe, ref, t_max=80.0, traces='actions')
res['nominal_same_size'] = {'kernel_ms': round(eng.last_kernel_ms, 2), 'env_steps_per_s_kernel': round(E * T / eng.last_kernel_ms * 1e3)}
print(json.dumps(res))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], 'w'), indent=1)
