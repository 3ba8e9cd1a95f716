"""SSNE generation update (base/core/mod_neuro_evo.py:447-543) for a device-resident population.

`plan_epoch` makes every index decision of `SSNE.epoch` on the host, consuming python `random` / `numpy.random` in
the reference's order, and returns the operation list; `SSNE.epoch` applies it to the packed weight tensor with the
`serl_ga_*` kernels (ga.py).  Covered: elitism (`clone`), the non-distillation branch (`clone` + `crossover_inplace`
on pairs of unselected members) and `mutate_inplace`.  `distilation_crossover`, `proximal_mutate` and `safe_mutate`
need the critic / autograd and stay in the reference's own PyTorch class (their elementwise update is
`ga.scaled_perturb`).

Reference quirks kept: `random.randint(0, len(x))` is inclusive (mod_neuro_evo.py:51,517), so the reference raises
IndexError with probability 1/(len+1) at those two places -- so does `plan_epoch`, after consuming the same draws.
"""
import random
import numpy as np
from . import ga


def selection_tournament(index_rank, num_offsprings, tournament_size, rng=random, nprng=np.random):
    """mod_neuro_evo.py:40-52"""
    total = len(index_rank)
    offsprings = []
    for _ in range(num_offsprings):
        winner = np.min(nprng.randint(total, size=tournament_size))
        offsprings.append(index_rank[winner])
    offsprings = list(set(offsprings))
    if len(offsprings) % 2 != 0:
        offsprings.append(offsprings[rng.randint(0, len(offsprings))])      # inclusive: may raise IndexError
    return offsprings


def plan_epoch(fitness, num_elitists, mutation_prob, rng=random, nprng=np.random):
    """-> (ops, new_elitist0); ops = [('clone', master, replacee) | ('crossover', i, j) | ('mutate', i)] in the
    reference's order (distil_crossover False, crossover_prob <= 0.01)."""
    pop_size = len(fitness)
    index_rank = np.argsort(fitness)[::-1]
    elitist_index = index_rank[:num_elitists]
    offsprings = selection_tournament(index_rank, len(index_rank) - num_elitists, 3, rng, nprng)
    unselects = [i for i in range(pop_size) if i not in offsprings and i not in elitist_index]
    rng.shuffle(unselects)
    ops, new_elitists = [], []
    for i in elitist_index:
        replacee = unselects.pop(0) if unselects else offsprings.pop(0)
        new_elitists.append(replacee)
        ops.append(('clone', int(i), int(replacee)))
    if len(unselects) % 2 != 0:
        unselects.append(unselects[rng.randint(0, len(unselects))])         # inclusive: may raise IndexError
    for i, j in zip(unselects[0::2], unselects[1::2]):
        off_i = rng.choice(new_elitists)
        off_j = rng.choice(offsprings)
        ops.append(('clone', int(off_i), int(i)))
        ops.append(('clone', int(off_j), int(j)))
        ops.append(('crossover', int(i), int(j)))
    for i in index_rank[num_elitists:]:
        if rng.random() < mutation_prob:
            ops.append(('mutate', int(i)))
    return ops, int(new_elitists[0])


class SSNE:
    """Device-side counterpart of the reference class for the operators that are pure tensor edits."""

    def __init__(self, args, engine, spec):
        self.args, self.engine, self.spec = args, engine, spec
        self.population_size = args.pop_size
        self.num_elitists = max(int(args.elite_fraction * args.pop_size), 1)
        if getattr(args, 'distil_crossover', False):
            raise NotImplementedError('distilation_crossover needs the critic and Adam: use the reference SSNE for it')
        if getattr(args, 'mut_type', 'normal') not in ('normal', 'inplace'):
            raise NotImplementedError('proximal / safe mutation need autograd; only their update is a kernel (ga.scaled_perturb)')

    def epoch(self, weights, fitness_evals):
        """weights: f32 [pop, stride] device tensor, edited in place; returns new_elitists[0] like the reference.
        The crossover / mutation edit lists are drawn when their turn comes, so the RNG streams interleave exactly
        like the reference's (which draws inside crossover_inplace / mutate_inplace)."""
        return self._apply(weights, fitness_evals)

    def _apply(self, weights, fitness):
        e, spec, rng, nprng = self.engine, self.spec, random, np.random
        pop_size = len(fitness)
        index_rank = np.argsort(fitness)[::-1]
        elitist_index = index_rank[:self.num_elitists]
        offsprings = selection_tournament(index_rank, len(index_rank) - self.num_elitists, 3, rng, nprng)
        unselects = [i for i in range(pop_size) if i not in offsprings and i not in elitist_index]
        rng.shuffle(unselects)
        new_elitists = []
        for i in elitist_index:
            replacee = unselects.pop(0) if unselects else offsprings.pop(0)
            new_elitists.append(replacee)
            ga.clone(e, weights, [int(i)], [int(replacee)], spec)
        if len(unselects) % 2 != 0:
            unselects.append(unselects[rng.randint(0, len(unselects))])
        for i, j in zip(unselects[0::2], unselects[1::2]):
            off_i = rng.choice(new_elitists)
            off_j = rng.choice(offsprings)
            ga.clone(e, weights, [int(off_i)], [int(i)], spec)
            ga.clone(e, weights, [int(off_j)], [int(j)], spec)
            ga.crossover_inplace(e, weights, int(i), int(j), spec, rng)
        for i in index_rank[self.num_elitists:]:
            if rng.random() < self.args.mutation_prob:
                ga.mutate_inplace(e, weights, int(i), spec, self.args.mutation_mag, rng, nprng)
        return int(new_elitists[0])
