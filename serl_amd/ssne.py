"""SSNE generation update (base/core/mod_neuro_evo.py:447-543) for a device-resident population.

The population is one f32 tensor `weights[pop, stride]` in HBM plus, per member, two device replay rings
(`replay.DeviceReplay`: the GeneticAgent's `buffer` and `critical_buffer`).  Every index decision and every random draw
is made on the host from the reference's own streams in the reference's order (python `random`, `numpy.random`, torch's
CPU generator), so selection, pairing and mutation targets are the reference's; the tensor work runs on the GPU:

  clone                 row copy (`serl_ga_clone`) + ring copy                                  mod_neuro_evo.py:371-382
  crossover_inplace     row swaps (`serl_ga_crossover`)                                         :61-93
  mutate_inplace        sparse edits (`serl_ga_mutate`)                                         :329-369
  proximal / safe       sensitivity by analytic backward passes (`serl_ga_sensitivity`) + update (`serl_ga_scaled_perturb`)
                                                                                                :183-223, :254-298
  sort_groups_by_distance   all parent pairs' novelty batches in one launch (`serl_ga_novelty`)  :411-445
  distilation_crossover     behaviour cloning with Adam and the critic's Q-filter -- gradient code, PyTorch on the GPU
                            (`distill.py`), fed from the rings                                  :131-181

The reference's defaults (base/parameters.py:110-115: mut_type 'proximal', distil_crossover True, distil_type 'distance')
are accepted.  Two reference defects on this path, and what happens here:
  * `random.randint(0, len(x))` is inclusive (mod_neuro_evo.py:51,517): with probability 1/(len+1) the reference raises
    IndexError.  The draw is consumed (stream parity) and the last element taken instead; `InclusiveRandint.count` -- a
    class-level counter shared by every SSNE instance and every `plan_epoch` caller of the process -- counts it.
  * with distil_type 'distance' the stray line :505 overwrites the distance-sorted groups by
    `sort_groups_by_novelty(..., bcs_evals)`, which raises TypeError when `bcs_evals` is None (what Agent.train passes).
    Here: bcs_evals None -> the distance-sorted groups (the evident intent); bcs_evals given -> the novelty-sorted groups,
    like the reference (the distance draws are still consumed first).
Two more deliberate deviations from the reference's behaviour in corner cases:
  * `clone(master, replacee)` with master == replacee (an elite that is its own replacement slot) is a no-op here; the
    reference's clone (:371-382) wipes the replacee's buffer and refills it from the master's -- which is the same, emptied,
    object -- so there the elite loses its buffer.
  * an Adam step of the fused distillation kernel whose Q-filter keeps no state of the minibatch is skipped; the reference
    takes the mean over an empty selection (genetic_agent.py:44-59) and its child's weights become NaN.
"""
import random
import numpy as np
import torch
from . import ga


class InclusiveRandint:
    """`x[random.randint(0, len(x))]` of the reference (mod_neuro_evo.py:51,517): the draw, clamped to the last index"""
    count = 0

    @classmethod
    def pick(cls, seq, rng=random):
        k = rng.randint(0, len(seq))
        if k >= len(seq):
            cls.count += 1
            k = len(seq) - 1
        return seq[k]


def selection_tournament(index_rank, num_offsprings, tournament_size, rng=random, nprng=np.random, strict=False):
    """mod_neuro_evo.py:40-52"""
    total = len(index_rank)
    offsprings = []
    for _ in range(num_offsprings):
        winner = np.min(nprng.randint(total, size=tournament_size))
        offsprings.append(index_rank[winner])
    offsprings = list(set(offsprings))
    if len(offsprings) % 2 != 0:
        offsprings.append(offsprings[rng.randint(0, len(offsprings))] if strict else InclusiveRandint.pick(offsprings, rng))
    return offsprings


def sort_groups_by_fitness(genomes, fitness):
    """mod_neuro_evo.py:388-397"""
    groups = []
    for i, first in enumerate(genomes):
        for second in genomes[i + 1:]:
            if fitness[first] < fitness[second]:
                groups.append((second, first, fitness[first] + fitness[second]))
            else:
                groups.append((first, second, fitness[first] + fitness[second]))
    return sorted(groups, key=lambda g: g[2], reverse=True)


def sort_groups_by_novelty(genomes, bcs):
    """mod_neuro_evo.py:399-409"""
    groups = []
    for i, first in enumerate(genomes):
        for second in genomes[i + 1:]:
            groups.append((second, first, np.linalg.norm(bcs[first, :] - bcs[second, :], axis=-1, ord=2)))
    return sorted(groups, key=lambda g: g[2], reverse=True)


def plan_epoch(fitness, num_elitists, mutation_prob, rng=random, nprng=np.random, strict=True):
    """-> (ops, new_elitist0); ops = [('clone', master, replacee) | ('crossover', i, j) | ('mutate', i)] in the
    reference's order (distil_crossover False, crossover_prob <= 0.01).  strict=True keeps the reference's IndexError on an
    out-of-range inclusive draw (used to pick golden seeds); False clamps it."""
    pick = (lambda s: s[rng.randint(0, len(s))]) if strict else (lambda s: InclusiveRandint.pick(s, rng))
    pop_size = len(fitness)
    index_rank = np.argsort(fitness)[::-1]
    elitist_index = index_rank[:num_elitists]
    offsprings = selection_tournament(index_rank, len(index_rank) - num_elitists, 3, rng, nprng, strict)
    unselects = [i for i in range(pop_size) if i not in offsprings and i not in elitist_index]
    rng.shuffle(unselects)
    ops, new_elitists = [], []
    for i in elitist_index:
        replacee = unselects.pop(0) if unselects else offsprings.pop(0)
        new_elitists.append(replacee)
        ops.append(('clone', int(i), int(replacee)))
    if len(unselects) % 2 != 0:
        unselects.append(pick(unselects))
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
    """Device-side counterpart of the reference class.

    args     : the reference's Parameters fields (pop_size, elite_fraction, mutation_prob, mutation_mag, mut_type,
               distil_crossover, distil_type, crossover_prob, mutation_batch_size, individual_bs)
    critic   : callable (state, action) -> (q1, q2) on the device (TD3's critic); needed for distillation only
    record   : optional list receiving the operator calls (kind, a, b) -- the index decisions the tests pin"""

    def __init__(self, args, engine, spec, critic=None, record=None):
        self.args, self.engine, self.spec, self.critic, self.record = args, engine, spec, critic, record
        self.population_size = args.pop_size
        self.num_elitists = max(int(args.elite_fraction * args.pop_size), 1)
        self.rl_policy = None
        self.selection_stats = {'elite': 0, 'selected': 0, 'discarded': 0, 'total': 0.0000001}
        self.mut_type = getattr(args, 'mut_type', 'normal')
        if self.mut_type not in ('normal', 'inplace', 'proximal', 'safe'):
            raise ValueError('Mutation type is unknown!')                               # mod_neuro_evo.py:30-38
        self.distil = bool(getattr(args, 'distil_crossover', False))
        self.distil_type = str(getattr(args, 'distil_type', 'distance')).lower()
        if self.distil and not ('fitness' in self.distil_type or 'dist' in self.distil_type):
            raise NotImplementedError('Unknown distilation type')

    # ---- operators ---------------------------------------------------------------------------------------------
    def _rec(self, *op):
        if self.record is not None:
            self.record.append(op)

    def clone(self, weights, master, replacee, buffers=None, critical=None, master_rings=None):
        """SSNE.clone: parameters, buffer and critical buffer (mod_neuro_evo.py:371-382).  master = member index, or -1
        with master_rings = (row tensor, buffer, critical_buffer) of a distillation child."""
        self._rec(0, int(master), int(replacee))
        if master_rings is None:
            ga.clone(self.engine, weights, [int(master)], [int(replacee)], self.spec)
            mb, mc = (buffers[master], critical[master]) if buffers is not None else (None, None)
        else:
            row, mb, mc = master_rings
            weights[int(replacee), :row.numel()] = row.to(weights.device)
        if buffers is not None:
            if mb is not buffers[replacee]:
                buffers[replacee].reset(); buffers[replacee].add_content_of(mb)
            if mc is not critical[replacee]:
                critical[replacee].reset(); critical[replacee].add_content_of(mc)

    def mutate(self, weights, i, buffers=None, critical=None):
        self._rec(2, int(i), -1)
        a, e, spec = self.args, self.engine, self.spec
        if self.mut_type in ('normal', 'inplace'):
            ga.mutate_inplace(e, weights, int(i), spec, a.mutation_mag, random, np.random)
        elif self.mut_type == 'proximal':
            ga.proximal_mutate(e, weights, int(i), spec, a.mutation_mag, buffers[i], a.mutation_batch_size, random)
        else:
            ga.safe_mutate(e, weights, int(i), spec, a.mutation_mag, buffers[i], critical[i], a.mutation_batch_size, random)

    def distilation_crossover(self, weights, first, second, buffers):
        self._rec(3, int(first), int(second))
        from . import distill
        return distill.distilation_crossover(self.args, self.engine, self.spec, weights, int(first), int(second), buffers, self.critic)

    def distilation_crossovers(self, weights, pairs, buffers):
        """independent distillations [(first, second)] -> children, trained together (distill.distil_batch)"""
        from . import distill
        return distill.distil_batch(self.args, self.engine, self.spec, weights, pairs, buffers, self.critic)

    # ---- the generation update ------------------------------------------------------------------------------------
    def epoch(self, weights, fitness_evals, bcs_evals=None, buffers=None, critical=None):
        """weights: f32 [pop, stride] device tensor, edited in place; buffers / critical: per-member DeviceReplay lists
        (required for proximal / safe mutation and distillation; optional otherwise).  Returns new_elitists[0] like the
        reference.  Draws interleave exactly like the reference's (each operator draws when its turn comes)."""
        rng, nprng = random, np.random
        fitness = fitness_evals
        if (self.distil or self.mut_type in ('proximal', 'safe')) and (buffers is None or critical is None):
            raise ValueError('SSNE.epoch: this configuration samples the members\' replay rings: pass buffers= and critical=')
        index_rank = np.argsort(fitness)[::-1]
        elitist_index = index_rank[:self.num_elitists]
        offsprings = selection_tournament(index_rank, len(index_rank) - self.num_elitists, 3, rng, nprng)
        unselects = [i for i in range(self.population_size) if i not in offsprings and i not in elitist_index]
        rng.shuffle(unselects)
        if self.rl_policy is not None:                       # RL-selection statistics, mod_neuro_evo.py:479-486
            self.selection_stats['total'] += 1.0
            if self.rl_policy in elitist_index: self.selection_stats['elite'] += 1.0
            elif self.rl_policy in offsprings: self.selection_stats['selected'] += 1.0
            elif self.rl_policy in unselects: self.selection_stats['discarded'] += 1.0
            self.rl_policy = None
        new_elitists = []
        for i in elitist_index:
            replacee = unselects.pop(0) if unselects else offsprings.pop(0)
            new_elitists.append(replacee)
            self.clone(weights, int(i), int(replacee), buffers, critical)
        if self.distil:
            parents = new_elitists + offsprings
            if 'fitness' in self.distil_type:
                groups = sort_groups_by_fitness(parents, fitness)
            else:
                groups = ga.sort_groups_by_distance(self.engine, weights, parents, buffers, self.spec, rng)
                if bcs_evals is not None:                   # mod_neuro_evo.py:505 (see the module docstring)
                    groups = sort_groups_by_novelty(parents, np.asarray(bcs_evals))
            # the children only replace unselected members, which are nobody's parents: the distillations of this loop do
            # not depend on each other and train in one launch (distill.distil_batch); the host draws keep the reference's
            # order (pair by pair: buffer shuffle, initialisation, minibatches), a clone draws nothing
            pairs = []
            for k, unselected in enumerate(unselects):
                first, second, _ = groups[k % len(groups)]
                if fitness[first] < fitness[second]:
                    first, second = second, first
                pairs.append((int(first), int(second)))
            children = self.distilation_crossovers(weights, pairs, buffers)
            for (first, second), unselected, child in zip(pairs, unselects, children):
                self._rec(3, first, second)
                self.clone(weights, -1, int(unselected), buffers, critical, master_rings=child)
        else:
            if len(unselects) % 2 != 0:
                unselects.append(InclusiveRandint.pick(unselects, rng))
            for i, j in zip(unselects[0::2], unselects[1::2]):
                off_i = rng.choice(new_elitists)
                off_j = rng.choice(offsprings)
                self.clone(weights, int(off_i), int(i), buffers, critical)
                self.clone(weights, int(off_j), int(j), buffers, critical)
                self._rec(1, int(i), int(j))
                ga.crossover_inplace(self.engine, weights, int(i), int(j), self.spec, rng)
        if getattr(self.args, 'crossover_prob', 0.0) > 0.01:           # mod_neuro_evo.py:526-532 ("so far this is not called")
            if buffers is None:
                raise ValueError('SSNE.epoch: crossover_prob > 0.01 runs distilation_crossover: pass buffers= and critical=')
            for i in offsprings:
                if rng.random() < self.args.mutation_prob:
                    others = offsprings.copy()
                    others.remove(i)
                    off_j = rng.choice(others)
                    child = self.distilation_crossover(weights, int(i), int(off_j), buffers)
                    self.clone(weights, -1, int(i), buffers, critical, master_rings=child)
        # mutations: a member's draws happen when its turn comes (the reference's order); the Jacobian-based ones are applied
        # together afterwards (ga.ProximalBatch: one sensitivity launch for all of them)
        batch = None
        if self.mut_type in ('proximal', 'safe') and weights is not None:
            batch = ga.ProximalBatch(self.engine, weights, self.spec, self.args.mutation_mag, self.args.mutation_batch_size, random)
        for i in index_rank[self.num_elitists:]:
            if rng.random() < self.args.mutation_prob:
                if batch is None:
                    self.mutate(weights, int(i), buffers, critical)
                else:
                    self._rec(2, int(i), -1)
                    batch.add(int(i), buffers[i], critical[i] if self.mut_type == 'safe' else None)
        if batch is not None:
            batch.apply()
        return int(new_elitists[0])
