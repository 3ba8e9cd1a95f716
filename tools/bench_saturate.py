#!/usr/bin/env python3
"""SURVEY 8d's SATURATING configuration: many more episodes than the GPU has lane groups, where throughput -- not the latency of one env step --
is the regime and the fp64 fraction means something.

    python tools/bench_saturate.py [--episodes 16384,65536] [--t-max 20] [--members 2048] [--modes lane64,auto] [--reps 2]

For every episode count and mode one JSON line: env-steps/s of the rollout kernel (HIP events around the launch), the kernel family the library
launched (serl_last_rollout_info), the fraction of the MEASURED fp64 peak (profiles/valu_latency_current.json) by SURVEY 8d's F_alg = 24 059 f64
operations per env step, and -- for the first 64 episodes -- whether the results equal the one-episode-per-team kernel's bit for bit.
Modes: laneN = the lane-per-episode DAG kernels with N episodes per wavefront (lanes_per_wave = N, rollout_variant.inc); auto = what serl_rollout
chooses (four episodes per team + work queue beyond 4 x CUs episodes); half = two episodes per wavefront.
Workload: the shipped SERL50 actors tiled with seeded noise to `members`, member = episode mod members, the base reference of
/root/reference/base/evaluate.py:173-180 shared by all episodes, nominal build, t_max seconds (2 001 env steps at 20 s)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, serl_amd
from serl_amd import refsignals

ap = argparse.ArgumentParser()
ap.add_argument('--episodes', default='16384,65536')
ap.add_argument('--t-max', type=float, default=20.0)
ap.add_argument('--members', type=int, default=2048)
ap.add_argument('--modes', default='lane64,auto')
ap.add_argument('--reps', type=int, default=2)
ap.add_argument('--activation', default='tanh', help="tanh (the shipped SERL50 actors) | relu | elu: what the per-lane actor's 96 hidden activations cost (the weights stay the tanh actors': the flights differ)")
ap.add_argument('--order', choices=['interleaved', 'member-major'], default='interleaved',
                help='interleaved: member = episode mod members (64 different members per wavefront: the worst case for the per-lane actor); member-major: member = episode // (episodes / members), the order evaluate_pop produces (agent.py:234-256: all episodes of a member one after the other)')
ap.add_argument('--profile', action='store_true', help='with SERL_PROFILE=1 in the environment: cycles per env step wavefront 0 of workgroup 0 spent in the actor / dynamics / env bookkeeping (lane-per-episode kernels)')
a = ap.parse_args()
eng = serl_amd.RolloutEngine(0)
spec = serl_amd.NetSpec(7, 3, 32, 3, a.activation)
w = bench.make_population(a.members, 0, tag='serl50').to(eng.device)
ref = refsignals.tabulate(*refsignals.base_reference(a.t_max), a.t_max)
T = ref.shape[0]
try:
    peak = float(json.load(open(os.path.join(ROOT, 'profiles', 'valu_latency_current.json')))['fp64_fma_peak_tflops_measured'])
except Exception:
    peak = None
base = None
for E in [int(x) for x in a.episodes.split(',')]:
    moe = ((np.arange(E) % a.members) if a.order == 'interleaved' else (np.arange(E) // max(E // a.members, 1)) % a.members).astype(np.int32)
    for mode in a.modes.split(','):
        kw = dict(lanes_per_wave=int(mode[4:])) if mode.startswith('lane') else dict(kernel=None if mode == 'auto' else mode)
        ms = []
        try:
            for _ in range(a.reps):
                out = eng.rollout(w, spec, moe, ref, t_max=a.t_max, **kw)
                ms.append(eng.last_kernel_ms)
        except Exception as ex:
            print(json.dumps(dict(episodes=E, mode=mode, error=repr(ex)[:300])), flush=True)
            continue
        steps = int(out['length_steps'].abs().sum())
        info = eng.last_rollout_info()
        k_ms = min(ms)
        rate = steps / (k_ms * 1e-3)
        extra = {}
        if a.profile and os.environ.get('SERL_PROFILE'):
            import ctypes
            buf = (ctypes.c_ulonglong * 32)()
            eng.lib.serl_debug_profile(eng.ctx, buf)
            st = max(int(buf[3]), 1)
            extra['cycles_per_env_step_wave0'] = dict(actor=int(buf[0] / st), dynamics=int(buf[1] / st), env=int(buf[2] / st), steps=int(buf[3]))
        chk = eng.rollout(w, spec, moe[:64], ref, t_max=a.t_max, kernel='team')
        same = bool(torch.equal(chk['fitness'], out['fitness'][:64]) and torch.equal(chk['length_steps'], out['length_steps'][:64]))
        print(json.dumps(dict(extra, what='saturating configuration (SURVEY 8d)', order=a.order, episodes=E, steps_per_episode=T, members=a.members, mode=mode, family=info['family'],
                              workgroups=info['workgroups'], episodes_per_team_or_wave=info['episodes_per_team'], work_queue=info['work_queue'],
                              kernel_ms=[round(v, 2) for v in ms], env_steps=steps, env_steps_per_s=rate,
                              us_per_env_step_and_wavefront_or_team=k_ms * 1e3 / T,
                              fp64_tflops_by_F_alg=rate * bench.F_ALG / 1e12, fp64_frac_of_measured_peak=(rate * bench.F_ALG / 1e12 / peak) if peak else None,
                              fp64_peak_measured_tflops=peak, first_64_episodes_equal_one_per_team=same)), flush=True)
