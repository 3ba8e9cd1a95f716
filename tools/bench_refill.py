#!/usr/bin/env python3
"""Training-shaped generation on the GPU box: pop = 512 members x 3 evals, t_max = 20 s, half of the actors untrained (they
crash within seconds, envs/phlabenv.py:391-399) -- 1 536 episodes of very different lengths on one GPU.

  python tools/bench_refill.py        -> one JSON line

Measures the work-queue launch (lane groups take the next episode when theirs ends) against the ideal
    sum of env steps / (lane groups x env steps per second of one lane group)
with the per-step time of a full four-episode team measured in the same run (equal-length episodes, one per lane group), and
against the same population run as rounds of one-episode teams (kernel_hint TEAM: every round waits for its longest episode)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serl_amd
from serl_amd import refsignals

eng = serl_amd.RolloutEngine(0)
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
cus = torch.cuda.get_device_properties(0).multi_processor_count
rng = np.random.default_rng(11)
base = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'actors.npz'))['serl50']
pop, ne = 512, 3
w = base[rng.integers(0, 50, pop)].copy()
bad = rng.random(pop) < 0.5
r = rng.normal(0, 0.3, w.shape).astype(np.float32)
for name, off, shape in spec.param_layout():
    if name.endswith('gamma'):
        r[:, off:off + shape[0]] = 1.0
w[bad] = r[bad]
moe = np.repeat(np.arange(pop, dtype=np.int32), ne)
E = pop * ne
ref = torch.from_numpy(refsignals.synthetic_reference_tables(E, ne, 20, seed=7)).cuda()
wd = torch.from_numpy(w).cuda()


def run(kernel, reps=3):
    ms = []
    for _ in range(reps):
        out = eng.rollout(wd, spec, moe, ref, t_max=20, kernel=kernel)
        ms.append(eng.last_kernel_ms)
    return out, float(np.median(ms))


q, ms_q = run(None)                      # automatic: one launch, four per team, work queue
t, ms_t = run('team')                    # rounds of one-episode teams
ls = q['length_steps'].cpu().numpy()
assert (ls == t['length_steps'].cpu().numpy()).all() and torch.equal(q['fitness'], t['fitness'])
steps = int(ls.sum())
# per-step time of a full team of four: equal-length episodes of shipped actors, one per lane group, no queue
n4 = 4 * cus
full = eng.rollout(torch.from_numpy(base).cuda(), spec, np.arange(n4) % 50, ref[:n4], t_max=20, kernel='team4')
us4 = eng.last_kernel_ms * 1e3 / 2001
ideal_ms = steps / (4 * cus) * us4 * 1e-3
print(json.dumps({'what': 'pop=512 x 3 evals, t_max=20 s, half the actors untrained: 1 536 episodes of unequal length, one GPU',
                  'episodes': E, 'env_steps': steps, 'crashed_early': int((ls < 2001).sum()), 'median_length': int(np.median(ls)),
                  'queue_ms': ms_q, 'queue_env_steps_per_s': steps / ms_q * 1e3,
                  'rounds_of_one_episode_teams_ms': ms_t,
                  'us_per_env_step_full_team_of_four': us4, 'ideal_ms': ideal_ms, 'queue_over_ideal': ms_q / ideal_ms,
                  'bit_identical_to_one_episode_per_team': True}))
