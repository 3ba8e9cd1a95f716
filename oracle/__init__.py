"""oracle/ -- CPU restatement of the SERL reference's population-rollout path.

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import anything from here; the product (``serl_amd``) never does and fails
loudly when its HIP extension is missing.

Contents
  citation_ref.c / citation_rt.h / citation_step.h / gen/*.inc
      C restatement of the reference's source-less dynamics library
      (envs/<build>/_citation.cpython-38-x86_64-linux-gnu.so), bit-exact against it on x86-64
  dynamics.py   ctypes binding of that restatement  (class CitationDynamics)
  refso.py      ctypes binding of the *reference's own* shared object (only where /root/reference exists)
  env.py        restatement of envs/phlabenv.py CitationEnv (step / reward / bounds / cost / reset)
  actor.py      restatement of base/core/genetic_agent.py Actor forward in numpy f32
  rollout.py    restatement of base/core/agent.py Agent.evaluate + the GA evaluate loop
  signals.py    restatement of the two pinned classes of the un-vendored `signals==0.0.1` package
  smoothness.py restatement of base/core/utils.py calc_smoothness / calc_nMAE
  ga_ops.py     restatement of the SSNE weight-tensor edits of base/core/mod_neuro_evo.py

Parity status: dynamics, env, actor, rollout pinned against the reference binary, the reference's
own Python code run under shims (tests/golden/make_golden.py) and the shipped wandb trajectories.
"""
