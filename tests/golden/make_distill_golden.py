#!/usr/bin/env python3
"""Golden vectors for the distillation crossover, produced by the REFERENCE'S OWN `SSNE.distilation_crossover`
(base/core/mod_neuro_evo.py:131-147) + `GeneticAgent.update_parameters` (base/core/genetic_agent.py:22-59) +
`ReplayMemory` (build container only) -> tests/golden/distill.npz

Parents: shipped SERL50 actors whose personal buffers the reference's own Agent.evaluate filled (the rows are the
buf_serl50_<i> entries of proximal.npz, regenerated here by the same code).  Critic: a small twin-Q torch module (the
reference calls `critic(state, action) -> (q1, q2)` and nothing else of it); its parameters are stored.

  <c>_parents      [first, second] shipped-actor indices (gene1, gene2)
  <c>_seed         python `random` and torch are seeded with it right before the call
  <c>_bs           args.individual_bs (child buffer = the latest bs/2 transitions of each parent, shuffled)
  <c>_child        the child's parameters after the 12 x (len // 128) Adam steps, packed in state_dict order, f32
  <c>_states       the child's buffer after add_latest_from x 2 + shuffle: states f32 [n, 7] (pins the buffer plumbing)
  <c>_first_batch  slots of the first random.sample(memory, 128) in the shuffled buffer
  critic_*         state_dict of the critic
"""
import os, sys, random, types
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim
os.chdir('/tmp')
refshim.install()
import torch
import make_proximal_golden as MP
from core.mod_neuro_evo import SSNE

CASES = [('d_18_0', 18, 0, 1000, 71), ('d_7_33', 7, 33, 600, 72), ('d_0_18', 0, 18, 200, 73)]


class Critic(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.q1 = torch.nn.Sequential(torch.nn.Linear(10, 32), torch.nn.ELU(), torch.nn.Linear(32, 1))
        self.q2 = torch.nn.Sequential(torch.nn.Linear(10, 32), torch.nn.ELU(), torch.nn.Linear(32, 1))

    def forward(self, s, a):
        x = torch.cat([s, a], -1)
        return self.q1(x), self.q2(x)


def main():
    torch.manual_seed(5)
    critic = Critic()
    res = {'critic_' + k.replace('.', '_'): v.numpy().copy() for k, v in critic.state_dict().items()}
    agents = {}
    for name, i1, i2, bs, seed in CASES:
        for i in (i1, i2):
            if i not in agents:
                agents[i] = MP.make_agent('serl50', i)
        (g1, args, _), (g2, _, _) = agents[i1], agents[i2]
        args = types.SimpleNamespace(**vars(args))
        args.individual_bs = bs
        fake = types.SimpleNamespace(args=args, critic=critic)
        first = []
        real_sample = random.sample

        def rec_sample(pop, k):
            out = real_sample(pop, k)
            if not first:
                ids = {id(t): j for j, t in enumerate(pop)}
                first.append(np.array([ids[id(t)] for t in out], np.int32))
            return out
        random.seed(seed); torch.manual_seed(seed)
        random.sample = rec_sample
        try:
            child = SSNE.distilation_crossover(fake, g1, g2)
        finally:
            random.sample = real_sample
        sd = child.actor.state_dict()
        res[name + '_parents'] = np.array([i1, i2])
        res[name + '_seed'] = np.array(seed)
        res[name + '_bs'] = np.array(bs)
        res[name + '_child'] = np.concatenate([v.numpy().reshape(-1) for v in sd.values()]).astype(np.float32)
        res[name + '_states'] = np.stack([np.asarray(t.state, np.float32).reshape(-1) for t in child.buffer.memory])
        res[name + '_first_batch'] = first[0]
        print(name, len(child.buffer), res[name + '_child'][:4], flush=True)
    np.savez_compressed(os.path.join(HERE, 'distill.npz'), **res)


if __name__ == '__main__':
    main()
