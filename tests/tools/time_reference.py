#!/usr/bin/env python3
"""Time the reference's OWN Python rollout (CitationEnv + Agent.evaluate + torch Actor + the shipped shared object) in the build
container: the CPU path the GPU evaluator replaces.  /root/reference does not exist on the GPU box, so this
number cannot be taken in bench.py's run; bench.py times the C restatement there (cpu_baseline.kind = "port").

  python tests/tools/time_reference.py [episodes]      -> one JSON line
"""
import os, sys, time, json
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'golden'))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refshim
refshim.install()
import torch
import make_golden as MG

torch.set_num_threads(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
th, ph = MG.base_refs()
sds, h, act = MG.load_pop('serl50')
env = refshim.make_env('nominal', 80)
actor = refshim.make_actor(sds[18], h, 3, act)
MG.run_ref(env, actor, th, ph)                      # warm-up episode
t0 = time.perf_counter()
steps = 0
for i in range(n):
    ep = MG.run_ref(env, refshim.make_actor(sds[i], h, 3, act), th, ph)
    steps += len(ep.reward_lst)
dt = time.perf_counter() - t0
print(json.dumps({'what': 'reference Python rollout (Agent.evaluate, PH-LAB nominal, SERL50 actors, 80 s episodes), one core, torch 1 thread',
                  'episodes': n, 'env_steps': steps, 'seconds': round(dt, 2), 'env_steps_per_s_per_core': round(steps / dt, 1),
                  'host_cores': os.cpu_count()}))
