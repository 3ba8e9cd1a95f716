#!/usr/bin/env python3
"""Golden vectors for the SSNE operators that need the actor itself, produced by the REFERENCE'S OWN
base/core/mod_neuro_evo.py / genetic_agent.py / replay_memory.py (build container only) -> tests/golden/proximal.npz

  buf_<tag>_<i>        the personal replay buffer of agent i as f32 rows [n, 20] (obs7, a3, next_obs7, r, done, cost), filled
                       by the reference's own Agent.evaluate(store_transition=True) on a 10 s episode; crit_<tag>_<i> the
                       slots of the rows that also went to the critical buffer
  prox_<tag>_<i>       genome (extract_parameters) after the reference's SSNE.proximal_mutate(gene, mag) with python
                       `random` and torch seeded with prox_seed_*; prox_scaling_* the clamped sensitivity the same autograd
                       recipe yields for the batch the seeded `random.sample` picked (diagnostic)
  safe_<tag>_<i>       the same for SSNE.safe_mutate (batch from the critical buffer)
  dist_groups          SSNE.sort_groups_by_distance([0,1,2,3], pop) -> rows (second, first, distance), python `random`
                       seeded with dist_seed
  epoch_*              index decisions of SSNE.epoch with distil_crossover=True (distil_type 'fitness', and 'distance' with
                       bcs_evals given -- the only way the reference gets through that branch, mod_neuro_evo.py:500-505),
                       operators replaced by recorders: rows (kind, a, b): 0 clone(master a -> replacee b),
                       3 distilation_crossover(first a, second b) [its child is the master of the next clone], 2 mutate(a)
"""
import os, sys, types, random, argparse, io, contextlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim
os.chdir('/tmp')
refshim.install()
import torch
import make_golden as MG
from core.mod_neuro_evo import SSNE
from core.genetic_agent import GeneticAgent
from core.agent import Agent

MAG = 0.0247682869654          # base/parameters.py:107
MBS = 86                       # mutation_batch_size = batch_size (parameters.py:44,109)


def make_args(h, act, **kw):
    a = argparse.Namespace(hidden_size=h, num_layers=3, activation_actor=act, state_dim=7, action_dim=3, device=torch.device('cpu'),
                           individual_bs=10_000, mutation_batch_size=MBS, test_ea=False, _verbose_mut=False, _verbose_crossover=False,
                           smooth_fitness=False, noise_sd=0.0, noise_clip=0.0)
    a.__dict__.update(kw)
    return a


def make_agent(tag, idx, t_max=10):
    """reference GeneticAgent holding a shipped actor, its buffers filled by the reference's own stored episode"""
    sds, h, act = MG.load_pop(tag)
    args = make_args(h, act)
    g = GeneticAgent(args)
    g.actor.load_state_dict(sds[idx])
    env = refshim.make_env('nominal', t_max)
    th, ph = MG.base_refs(t_max)
    costs = []

    class _Env:
        def __init__(self, e):
            self.__dict__['_e'] = e

        def reset(self):
            return self._e.reset(user_refs={'theta_ref': th, 'phi_ref': ph})

        def step(self, a):
            r = self._e.step(a)
            costs.append(int(r[3]['cost']))
            return r

        def __getattr__(self, k):
            return getattr(self._e, k)
    fake = argparse.Namespace(args=args, env=_Env(env), num_frames=0, gen_frames=0, num_episodes=0,
                              replay_buffer=types.SimpleNamespace(add=lambda *t: None))
    with contextlib.redirect_stdout(io.StringIO()):
        Agent.evaluate(fake, g, False, True)
    return g, args, np.asarray(costs)


def rows_of(buf, costs):
    out = np.zeros((len(buf), 20), np.float32)
    for i, t in enumerate(buf.memory):
        out[i, 0:7], out[i, 7:10], out[i, 10:17] = t.state.reshape(-1), t.action.reshape(-1), t.next_state.reshape(-1)
        out[i, 17], out[i, 18], out[i, 19] = float(t.reward.reshape(-1)[0]), float(t.done.reshape(-1)[0]), float(costs[i])
    return out


def scaling_of(actor, states):
    """the autograd recipe of mod_neuro_evo.py:188-217 for a given batch (diagnostic copy of what the operator computes)"""
    out = actor(states)
    jac = []
    for i in range(out.shape[1]):
        actor.zero_grad()
        go = torch.zeros_like(out); go[:, i] = 1.0
        out.backward(go, retain_graph=True)
        jac.append(actor.extract_grad())
    s = torch.sqrt((torch.stack(jac) ** 2).sum(0))
    s[s == 0] = 1.0
    s[s < 0.01] = 0.01
    return s.numpy()


def recorded_epoch(seed, fitness, num_elit, distil_type, bcs=None):
    ops = []
    n = len(fitness)
    pop = list(range(n))
    fake = types.SimpleNamespace()
    fake.num_elitists, fake.population_size, fake.rl_policy = num_elit, n, None
    fake.args = types.SimpleNamespace(distil_crossover=True, distil_type=distil_type, crossover_prob=0.0, mutation_prob=0.9, mutation_mag=MAG)
    fake.selection_tournament = types.MethodType(SSNE.selection_tournament, fake)
    fake.clone = lambda master, replacee: ops.append((0, -1 if master == 'child' else master, replacee))
    fake.distilation_crossover = lambda a, b: (ops.append((3, a, b)), 'child')[1]
    fake.mutate = lambda g, mag: ops.append((2, g, -1))
    fake.stats = types.SimpleNamespace(reset=lambda: None)
    random.seed(seed); np.random.seed(seed)
    ret = SSNE.epoch(fake, pop, fitness, bcs)
    return np.array(ops, dtype=np.int64).reshape(-1, 3), int(ret)


def main():
    out = {}
    fake = types.SimpleNamespace()
    # ---- proximal / safe mutation -------------------------------------------------------------------------------
    for tag, idx, seed in (('serl50', 18, 501), ('serl50', 0, 502), ('td3', 0, 503)):
        g, args, costs = make_agent(tag, idx)
        key = '%s_%d' % (tag, idx)
        out['buf_' + key] = rows_of(g.buffer, costs)
        out['crit_' + key] = np.nonzero(costs)[0].astype(np.int32)
        assert len(g.critical_buffer) == int(costs.sum())
        fake.args = args
        for op in (('prox', 'safe') if tag != 'td3' else ('prox',)):      # (h = 96: 28 608 genome entries per vector -- one case)
            sd0 = {k: v.clone() for k, v in g.actor.state_dict().items()}
            random.seed(seed); torch.manual_seed(seed)
            src = g.buffer if op == 'prox' or len(g.critical_buffer) <= 1 else g.critical_buffer
            pick = random.sample(range(len(src)), min(MBS, len(src)))
            states = torch.FloatTensor(np.concatenate([src.memory[i].state for i in pick]))
            if tag != 'td3':
                out['%s_scaling_%s' % (op, key)] = scaling_of(g.actor, states)
            random.seed(seed); torch.manual_seed(seed)
            (SSNE.proximal_mutate if op == 'prox' else SSNE.safe_mutate)(fake, g, MAG)
            out['%s_%s' % (op, key)] = g.actor.extract_parameters().numpy()
            out['%s_seed_%s' % (op, key)] = np.array(seed)
            out['%s_pick_%s' % (op, key)] = np.asarray(pick, np.int32)
            g.actor.load_state_dict(sd0)
        print(key, len(g.buffer), len(g.critical_buffer), flush=True)
    # ---- distance-sorted parent groups ----------------------------------------------------------------------------
    pop = []
    for idx in (18, 0, 7, 33):
        g, args, costs = make_agent('serl50', idx)
        pop.append(g)
        key = 'serl50_%d' % idx
        if 'buf_' + key not in out:
            out['buf_' + key] = rows_of(g.buffer, costs)
            out['crit_' + key] = np.nonzero(costs)[0].astype(np.int32)
    random.seed(77)
    groups = SSNE.sort_groups_by_distance([0, 1, 2, 3], pop)
    out['dist_groups'] = np.array([[a, b, d] for a, b, d in groups], dtype=np.float64)
    out['dist_seed'] = np.array(77)
    print(groups)
    # ---- epoch index decisions with distillation crossover --------------------------------------------------------
    rng = np.random.default_rng(5)
    cases = []
    for n, elit, dtype in ((10, 2, 'fitness'), (50, 10, 'fitness'), (10, 2, 'distance')):
        fit = rng.normal(-200, 80, n)
        bcs = rng.normal(0, 1, (n, 2)) if dtype == 'distance' else None
        k = 0
        for seed in range(100):
            try:
                if dtype == 'distance':      # the distance sort itself needs real agents: patch it with the fitness sort's stand-in
                    saved = SSNE.sort_groups_by_distance
                    SSNE.sort_groups_by_distance = staticmethod(lambda genomes, pop: [])
                    try:
                        ops, ret = recorded_epoch(seed, fit, elit, dtype, bcs)
                    finally:
                        SSNE.sort_groups_by_distance = saved
                else:
                    ops, ret = recorded_epoch(seed, fit, elit, dtype)
            except IndexError:
                continue
            key = 'epoch_%s_p%d_e%d_s%d' % (dtype, n, elit, seed)
            out[key + '_ops'], out[key + '_ret'], out[key + '_fit'] = ops, np.array(ret), fit
            if bcs is not None:
                out[key + '_bcs'] = bcs
            cases.append(key); k += 1
            if k == 2:
                break
    print(cases)
    np.savez_compressed(os.path.join(HERE, 'proximal.npz'), **out)


if __name__ == '__main__':
    main()
