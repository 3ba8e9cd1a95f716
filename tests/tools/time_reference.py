#!/usr/bin/env python3
"""Time the reference's OWN Python rollout (CitationEnv + Agent.evaluate + torch Actor + the shipped shared object): the CPU
path the GPU evaluator replaces -- one process per host core (the reference has no parallel path of its own;
each process owns a private instance of the singleton dynamics library), torch 1 thread each, one warm-up episode, then
`episodes` full 80 s episodes of the bench workload's shape (SERL50 actors, smoothed-step references, nominal build).

  python tests/tools/time_reference.py [--procs N] [--episodes K] [--out FILE]      -> one JSON line

Needs /root/reference, which does not exist on the GPU box: bench.py runs this in its cpu_baseline leg where the reference is
present (kind = "reference-python") and otherwise embeds the committed build-container measurement
(profiles/r02_reference_cpu.json) next to the C restatement it can time there (kind = "port").
"""
import os, sys, time, json, argparse
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'golden'))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def worker(rank, episodes, q, go):
    import refshim
    refshim.install()
    import torch
    import make_golden as MG
    torch.set_num_threads(1)
    th, ph = MG.base_refs()
    sds, h, act = MG.load_pop('serl50')
    env = refshim.make_env('nominal', 80)
    MG.run_ref(env, refshim.make_actor(sds[18], h, 3, act), th, ph)                      # warm-up episode
    q.put(('ready', rank))
    go.wait()
    t0 = time.perf_counter()
    steps = 0
    for i in range(episodes):
        ep = MG.run_ref(env, refshim.make_actor(sds[(rank * episodes + i) % len(sds)], h, 3, act), th, ph)
        steps += len(ep.reward_lst)
    q.put(('done', rank, steps, time.perf_counter() - t0))


def measure(procs=None, episodes=1):
    import multiprocessing as mp
    procs = procs or os.cpu_count() or 1
    ctx = mp.get_context('spawn')
    q, go = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=worker, args=(r, episodes, q, go)) for r in range(procs)]
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
    return {'what': 'reference Python rollout (unmodified Agent.evaluate + CitationEnv + torch Actor + the shipped _citation '
                    'library via ctypes; PH-LAB nominal, SERL50 actors, 80 s episodes), one process per core, torch 1 thread each',
            'procs': procs, 'host_cores': os.cpu_count(), 'episodes_per_proc': episodes, 'env_steps': steps,
            'seconds': round(wall, 2), 'env_steps_per_s': round(steps / wall, 1),
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
