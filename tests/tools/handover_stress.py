#!/usr/bin/env python3
"""Hand-over stress run of the team kernel families (TEST TOOL; tests/test_gpu_rollout.py starts it in a process of its own).

  python tests/tools/handover_stress.py <build> <out.npz> [seed ...]        with SERL_LIB=serl_amd/csrc/libserl_amd_jitter.so

The jitter library (serl_amd/build.py build_jitter: -DCITW_POISON=1 -DCITW_JITTER=1, serl_amd/csrc/citation_wave.h) fills every LDS
blackboard the wavefronts of a team exchange values through with slot-naming signalling NaNs and pauses pseudo-randomly (seed =
SERL_JITTER_SEED, read when the context is made) around every hand-over flag and barrier.  One context per seed; per seed and
case the fitness / length / cost count of every episode go to <out.npz> as `<case>_<seed>_<key>`.  cases() is shared with the
test, which evaluates the same inputs on the CPU oracle.
"""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
T_MAX = 10
NET = {'serl50': dict(state_dim=7, action_dim=3, hidden=32, num_layers=3, activation='tanh'),
       'serl10': dict(state_dim=7, action_dim=3, hidden=72, num_layers=3, activation='tanh')}


def population(tag, n_members, seed):
    """shipped actors, every third one perturbed so that it leaves the flight envelope within seconds (episodes of unequal length:
    a finished lane group idles or refills while its neighbours fly on)"""
    rng = np.random.default_rng(seed)
    base = np.load(os.path.join(ROOT, 'tests', 'golden', 'actors.npz'))[tag]
    w = base[rng.integers(0, len(base), n_members)].copy()
    bad = np.arange(n_members) % 3 == 2
    w[bad] += rng.normal(0, 0.25, w[bad].shape).astype(np.float32)
    return w


def cases(cus):
    """name -> (actor tag, kernel hint, episodes): every kernel family of rollout_team.inc / rollout_team_half.inc"""
    return {
        'team': ('serl50', 'team', 12),                # seven team wavefronts + the actor wavefront, weights in LDS
        'teamr': ('serl10', 'team', 6),                # ... the streamed actor in a workgroup of its own on another CU: mailboxes in global memory, a courier wavefront (rollout_teamr_<v>.hip)
        'teams': ('serl10', 'team', 6),                # ... the actor streams its weights on the team's CU (SERL_REMOTE_ACTOR=0)
        'teams_split': ('serl10', 'team', 6),          # ... on TWO actor wavefronts that share the forward pass (SERL_SPLIT_ACTOR=1, rollout_teams2_<v>.hip)
        'team2': ('serl50', 'team2', 21),              # two episodes per team (an odd count: one lane group stays empty)
        'team2s': ('serl10', 'team2', 11),             # six team wavefronts + two streaming actor wavefronts
        'team4': ('serl50', 'team4', 35),              # four episodes per team
        'queue': ('serl50', None, 4 * cus + 61),       # one launch, four per team, the rest through the work queue
    }


def inputs(name, case, build):
    from serl_amd import refsignals
    tag, kern, n = case
    seed = sum(map(ord, name))
    w = population(tag, 9 if n < 100 else 61, seed)
    moe = (np.arange(n) * 7) % len(w)
    ref = refsignals.synthetic_reference_tables(n, 3, T_MAX, seed=seed)
    tick0 = ((np.arange(n) * 37) % 4100).astype(np.int32) if build == 'cg_timed' else None      # per-episode model clocks
    return w, moe, ref, tick0


def main():
    import torch, serl_amd
    build, out_path = sys.argv[1], sys.argv[2]
    seeds = [int(s) for s in sys.argv[3:]] or [0, 1, 2, 3]
    assert 'jitter' in os.environ.get('SERL_LIB', ''), 'run with SERL_LIB=.../libserl_amd_jitter.so'
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    out = {}
    for seed in seeds:
        os.environ['SERL_JITTER_SEED'] = str(seed)          # read by serl_ctx_create
        os.environ['SERL_SPLIT_ACTOR'] = '0'
        os.environ['SERL_REMOTE_ACTOR'] = '1'
        eng = serl_amd.RolloutEngine(0)
        os.environ['SERL_REMOTE_ACTOR'] = '0'
        eng_same = serl_amd.RolloutEngine(0)
        os.environ['SERL_SPLIT_ACTOR'] = '1'
        eng_split = serl_amd.RolloutEngine(0)
        os.environ['SERL_REMOTE_ACTOR'] = '1'
        engines = (eng, eng_split, eng_same)
        want = {'team': 'team', 'teamr': 'teamr', 'teams': 'teams', 'teams_split': 'teams2', 'team2': 'team2', 'team2s': 'team2s', 'team4': 'team4', 'queue': 'team4'}
        for name, case in cases(cus).items():
            tag, kern, n = case
            eng = engines[1] if name == 'teams_split' else engines[2] if name == 'teams' else engines[0]
            w, moe, ref, tick0 = inputs(name, case, build)
            n_ = NET[tag]
            spec = serl_amd.NetSpec(n_['state_dim'], n_['action_dim'], n_['hidden'], n_['num_layers'], n_['activation'])
            eng.kernel_hint = kern
            r = eng.rollout(torch.from_numpy(w), spec, moe, ref, build=build, t_max=T_MAX, tick0=tick0)
            fam = eng.last_rollout_info()['family']
            assert fam == want[name], 'case %s should run kernel family %s, the library launched %s' % (name, want[name], fam)
            for key in ('fitness', 'length_steps', 'cost_steps'):
                out['%s_%d_%s' % (name, seed, key)] = r[key].cpu().numpy()
            print(build, name, 'seed', seed, 'kernel ms %.1f' % eng.last_kernel_ms, 'nan', int(np.isnan(out['%s_%d_fitness' % (name, seed)]).sum()),
                  'payloads', sorted(set('%#x' % v for v in out['%s_%d_fitness' % (name, seed)][np.isnan(out['%s_%d_fitness' % (name, seed)])].view(np.uint64)))[:6], flush=True)
        for e_ in engines:
            e_.close()
    np.savez(out_path, **out)


if __name__ == '__main__':
    main()
