#!/usr/bin/env python3
"""Time the reference's OWN Python rollout (CitationEnv + Agent.evaluate + torch Actor + the shipped shared object): the CPU
path the GPU evaluator replaces (base/core/agent.py:63-138 driven like the loop of :234-241) -- one process
per host core (the reference has no parallel path of its own; each process owns a private copy of the singleton dynamics
library), torch 1 thread each, one short warm-up episode, then `episodes` full 80 s episodes of the bench workload's shape
(SERL50 actors, the fixed smoothed-step reference of base/evaluate.py:173-180, nominal build).

  python tests/tools/time_reference.py [--procs N] [--episodes K] [--out FILE]      -> one JSON line

The reference is taken from $SERL_REFERENCE, /root/reference (build container) or the archive `oracle/build.py stage_ref()`
packs into the git-ignored oracle/_ref/ (that is what travels to the GPU box); actors come from the committed fixture
tests/golden/actors.npz (the shipped evo_nets.pkl, packed).  bench.py runs this in its cpu_baseline leg
(kind = "reference-python").
"""
import os, sys, time, json, argparse
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'golden'))
sys.path.insert(0, ROOT)


def _state_dict(row, spec):
    import torch
    return {name: torch.from_numpy(row[off:off + int(__import__('numpy').prod(shape))].reshape(shape).copy())
            for name, off, shape in spec.param_layout()}


def worker(rank, episodes, q, go):
    import contextlib, io
    import numpy as np
    os.chdir('/tmp')
    import refshim
    refshim.install()
    import torch
    torch.set_num_threads(1)
    from oracle import signals as S
    from serl_amd.actor import NetSpec
    spec = NetSpec(7, 3, 32, 3, 'tanh')
    rows = np.load(os.path.join(ROOT, 'tests', 'golden', 'actors.npz'))['serl50']
    tt = np.linspace(0., 80, 6)
    th = S.SmoothedStepSequence(tt, [0, 12, 3, -4, -8, 2], smooth_width=8)
    ph = S.SmoothedStepSequence(tt, [2, -2, 2, 10, 2, -6], smooth_width=8)

    def run(env, i):
        actor = refshim.make_actor(_state_dict(rows[i % len(rows)], spec), 32, 3, 'tanh')
        env.error = np.zeros(3)
        with contextlib.redirect_stdout(io.StringIO()):
            return refshim.reference_evaluate(env, actor, user_refs={'theta_ref': th, 'phi_ref': ph})

    warm = refshim.make_env('nominal', 5)                 # warm-up: a 5 s episode (imports, first-touch, torch dispatch caches)
    run(warm, 18)
    env = refshim.make_env('nominal', 80)
    q.put(('ready', rank))
    go.wait()
    t0 = time.perf_counter()
    steps = 0
    for i in range(episodes):
        ep = run(env, rank * episodes + i)
        steps += len(ep.reward_lst)
    q.put(('done', rank, steps, time.perf_counter() - t0))


def host_cores():
    """CPU cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU box shows 256 logical
    CPUs to a container that is allowed 16 cores' worth of time: 256 processes there only add start-up time)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = max(1, min(n, int(int(q) / int(p) + 0.5)))
    except Exception:
        pass
    return n


def measure(procs=None, episodes=1):
    import multiprocessing as mp
    procs = procs or host_cores()
    ctx = mp.get_context('spawn')
    q, go = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=worker, args=(r, episodes, q, go)) for r in range(procs)]
    t_spawn = time.perf_counter()
    for p in ps:
        p.start()
    for _ in ps:
        assert q.get()[0] == 'ready'
    t0 = time.perf_counter()
    go.set()
    res = [q.get() for _ in ps]
    wall = time.perf_counter() - t0
    for p in ps:
        p.join()
    steps = sum(r[2] for r in res)
    from oracle import refso
    return {'what': 'reference Python rollout (unmodified Agent.evaluate + CitationEnv + torch Actor + the shipped _citation '
                    'library via ctypes; PH-LAB nominal, SERL50 actors, 80 s episodes), one process per core, torch 1 thread each',
            'reference': 'archive oracle/_ref' if os.path.isfile(refso.REF) else refso.REF,
            'procs': procs, 'host_cores': os.cpu_count(), 'usable_cores': host_cores(), 'episodes_per_proc': episodes, 'env_steps': steps,
            'seconds': round(wall, 2), 'startup_seconds': round(t0 - t_spawn, 1), 'env_steps_per_s': round(steps / wall, 1),
            'env_steps_per_s_per_core': round(steps / wall / procs, 1)}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--procs', type=int, default=0)
    ap.add_argument('--episodes', type=int, default=1)
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    r = measure(a.procs or None, a.episodes)
    s = json.dumps(r)
    if a.out:
        open(a.out, 'w').write(s + '\n')
    print(s)
