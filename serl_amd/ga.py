"""SSNE weight-tensor edits (base/core/mod_neuro_evo.py) on a device-resident population tensor.

The population lives as one f32 tensor `weights[pop, P]` (packed rows, actor.NetSpec.param_layout).
Selection and every random draw stay on the host and consume the *same* RNG streams in the *same*
order as the reference (python `random`, `numpy.random`), so index decisions are identical; only the
tensor edits run as HIP kernels (C ABI `serl_ga_*`).

Reference quirks kept: `random.randint(0, n)` is inclusive, so the reference can index
one past the end (`:76,89,357-358`) and raise IndexError mid-operator; here such a draw is consumed
(RNG parity) and the edit skipped.
"""
import ctypes, math, random
import numpy as np
import torch
from . import _capi
from .actor import NetSpec


def _i32(dev, a):
    return torch.as_tensor(np.asarray(a, dtype=np.int32)).to(dev)


def _f32(dev, a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(dev)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def clone(engine, weights, src, dst, spec: NetSpec):
    """weights[dst[i]] <- weights[src[i]]  (SSNE.clone, mod_neuro_evo.py:371-382; parameter part)."""
    src, dst = np.atleast_1d(src), np.atleast_1d(dst)
    s, d = _i32(weights.device, src), _i32(weights.device, dst)
    _capi.check(engine.lib.serl_ga_clone(engine.ctx, weights.data_ptr(), weights.stride(0), spec.param_count,
                                         s.data_ptr(), d.data_ptr(), len(src), _stream(weights.device)), 'serl_ga_clone')


def plan_crossover(spec: NetSpec, rng=random):
    """Draw the op list of crossover_inplace (mod_neuro_evo.py:61-93) -> int32 [n, 3] (offset, length, dir)
    dir 0: gene1 <- gene2, dir 1: gene2 <- gene1."""
    ops = []
    for name, off, shape in spec.param_layout():
        rows = shape[0]
        if len(shape) == 2:
            n = rng.randint(0, rows * 2)
            width = shape[1]
        else:
            n = rng.randint(0, rows)
            width = 1
        for _ in range(n):
            receiver = rng.random()
            ind = rng.randint(0, rows)
            if ind >= rows:
                continue          # the reference raises IndexError here
            ops.append((off + ind * width, width, 0 if receiver < 0.5 else 1))
    return np.asarray(ops, dtype=np.int32).reshape(-1, 3)


def crossover_inplace(engine, weights, member_a, member_b, spec: NetSpec, rng=random, ops=None):
    ops = plan_crossover(spec, rng) if ops is None else ops
    if len(ops):
        o = _i32(weights.device, ops)
        _capi.check(engine.lib.serl_ga_crossover(engine.ctx, weights.data_ptr(), weights.stride(0), int(member_a),
                                                 int(member_b), o.data_ptr(), len(ops), _stream(weights.device)),
                    'serl_ga_crossover')
    return ops


def plan_mutation(spec: NetSpec, mag, rng=random, nprng=np.random):
    """Draw the edit list of mutate_inplace (mod_neuro_evo.py:329-369) -> (idx, kind, z, strength)."""
    num_mutation_frac, super_mut_strength, super_mut_prob = 0.1, 10 * mag, 0.05
    reset_prob = super_mut_prob + 0.05
    layout = spec.param_layout()
    probs = nprng.uniform(0, 1, len(layout)) * 2
    idx, kind, z, strength = [], [], [], []
    for i, (name, off, shape) in enumerate(layout):
        if len(shape) != 2:
            continue
        num_weights = shape[0] * shape[1]
        if rng.random() < probs[i]:
            for _ in range(rng.randint(0, int(math.ceil(num_mutation_frac * num_weights)))):
                d1 = rng.randint(0, shape[0])
                d2 = rng.randint(0, shape[-1])
                r = rng.random()
                g = rng.gauss(0, 1.0)           # random.gauss(0, sigma) == sigma * this draw
                if d1 >= shape[0] or d2 >= shape[1]:
                    continue                    # the reference raises IndexError here
                idx.append(off + d1 * shape[1] + d2)
                if r < super_mut_prob:
                    kind.append(0); strength.append(super_mut_strength)
                elif r < reset_prob:
                    kind.append(1); strength.append(0.0)
                else:
                    kind.append(0); strength.append(mag)
                z.append(g)
    return (np.asarray(idx, np.int32), np.asarray(kind, np.int32), np.asarray(z, np.float32),
            np.asarray(strength, np.float32))


def mutate_inplace(engine, weights, member, spec: NetSpec, mag, rng=random, nprng=np.random, plan=None):
    idx, kind, z, strength = plan_mutation(spec, mag, rng, nprng) if plan is None else plan
    if len(idx):
        dev = weights.device
        a, b, c, d = _i32(dev, idx), _i32(dev, kind), _f32(dev, z), _f32(dev, strength)
        _capi.check(engine.lib.serl_ga_mutate(engine.ctx, weights.data_ptr(), weights.stride(0), int(member),
                                              a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), len(idx),
                                              _stream(dev)), 'serl_ga_mutate')
    return idx, kind, z, strength


def scaled_perturb(engine, weights, member, spec: NetSpec, delta, scaling):
    """theta <- theta + delta / scaling over the 2-D-weights genome (proximal_mutate / safe_mutate update,
    mod_neuro_evo.py:183-223, 254-298; the Jacobian-based `scaling` comes from torch.autograd)."""
    segs = spec.genome_segments()
    dev = weights.device
    so, sl = _i32(dev, [s[0] for s in segs]), _i32(dev, [s[1] for s in segs])
    dl = torch.as_tensor(delta, dtype=torch.float32).to(dev).contiguous()
    sc = torch.as_tensor(scaling, dtype=torch.float32).to(dev).contiguous()
    assert dl.numel() == sc.numel() == sum(s[1] for s in segs)
    _capi.check(engine.lib.serl_ga_scaled_perturb(engine.ctx, weights.data_ptr(), weights.stride(0), int(member),
                                                  so.data_ptr(), sl.data_ptr(), len(segs), dl.data_ptr(), sc.data_ptr(),
                                                  _stream(dev)), 'serl_ga_scaled_perturb')


# ---- operators that need the actor itself (kernels in csrc/serl_ga.hip) ----------------------------------------------------
def _net_args(spec: NetSpec):
    return (spec.state_dim, spec.hidden, spec.num_layers, spec.action_dim, spec.activation_id)


def genome_size(spec: NetSpec):
    return sum(n for _, n in spec.genome_segments())


def sensitivity(engine, weights, members, spec: NetSpec, states):
    """The clamped output sensitivity of proximal_mutate / safe_mutate (mod_neuro_evo.py:188-217) for several members at
    once: states f32 [n, B, state_dim] (device) -> scaling f32 [n, G] (device)."""
    dev = weights.device
    members = np.atleast_1d(np.asarray(members, dtype=np.int32))
    states = torch.as_tensor(states, dtype=torch.float32).to(dev).contiguous()
    n, B = states.shape[0], states.shape[1]
    assert n == len(members) and states.shape[2] == spec.state_dim
    G = genome_size(spec)
    if B == 0:            # empty buffer: every gradient is zero -> scaling[scaling == 0] = 1
        return torch.ones(n, G, dtype=torch.float32, device=dev)
    out = torch.empty(n, G, dtype=torch.float32, device=dev)
    m = _i32(dev, members)
    _capi.check(engine.lib.serl_ga_sensitivity(engine.ctx, weights.data_ptr(), weights.stride(0), *_net_args(spec), m.data_ptr(), n,
                                               states.data_ptr(), B, out.data_ptr(), _stream(dev)), 'serl_ga_sensitivity')
    return out


def draw_delta(spec: NetSpec, mag):
    """the initial perturbation of mod_neuro_evo.py:195-197, drawn from torch's global CPU generator like the reference
    (whose `params` live on the CPU): Normal(zeros(G), ones(G) * mag).sample()"""
    import torch.distributions as dist
    G = genome_size(spec)
    return dist.Normal(torch.zeros(G), torch.ones(G) * mag).sample()


def proximal_mutate(engine, weights, member, spec: NetSpec, mag, buffer, batch_size, rng=random, delta=None):
    """SSNE.proximal_mutate (mod_neuro_evo.py:183-223) on the device-resident population: the batch is what the
    reference's `gene.buffer.sample(min(mutation_batch_size, len(buffer)))` picks (same python `random` draws), the
    sensitivity comes from serl_ga_sensitivity, delta from torch's generator, the update is serl_ga_scaled_perturb.
    `buffer` is a replay.DeviceReplay.  safe_mutate (:254-298) is the same operator fed from the critical buffer."""
    states = buffer.sample(min(int(batch_size), len(buffer)), rng)[0]
    scaling = sensitivity(engine, weights, [int(member)], spec, states[None])[0]
    delta = draw_delta(spec, mag) if delta is None else delta
    scaled_perturb(engine, weights, int(member), spec, delta, scaling)
    return scaling


class ProximalBatch:
    """Several proximal / safe mutations of one epoch applied together: `add` makes a member's host draws right away, in the
    reference's order (its minibatch from python `random`, its delta from torch's generator); `apply` computes all
    sensitivities in ONE serl_ga_sensitivity launch per batch size (one workgroup per member, side by side instead of one
    after the other) and perturbs the rows.  A mutation only reads and writes its own member."""

    def __init__(self, engine, weights, spec: NetSpec, mag, batch_size, rng=random):
        self.engine, self.weights, self.spec, self.mag, self.batch_size, self.rng = engine, weights, spec, mag, int(batch_size), rng
        self.items = []

    def add(self, member, buffer, critical_buffer=None):
        src = buffer
        if critical_buffer is not None and len(critical_buffer) > 1:           # safe_mutate, mod_neuro_evo.py:258-261
            src = critical_buffer
        states = src.sample(min(self.batch_size, len(src)), self.rng)[0]
        self.items.append((int(member), states, draw_delta(self.spec, self.mag)))

    def apply(self):
        if not self.items:
            return
        dev = self.weights.device
        by_size = {}
        for k, (_, st, _) in enumerate(self.items):
            by_size.setdefault(st.shape[0], []).append(k)
        scal = [None] * len(self.items)
        for B, ks in by_size.items():
            sc = sensitivity(self.engine, self.weights, [self.items[k][0] for k in ks], self.spec, torch.stack([self.items[k][1] for k in ks]))
            for j, k in enumerate(ks):
                scal[k] = sc[j]
        deltas = torch.stack([d for _, _, d in self.items]).to(dev)
        segs = self.spec.genome_segments()
        so, sl = _i32(dev, [s_[0] for s_ in segs]), _i32(dev, [s_[1] for s_ in segs])
        for k, (member, _, _) in enumerate(self.items):
            _capi.check(self.engine.lib.serl_ga_scaled_perturb(self.engine.ctx, self.weights.data_ptr(), self.weights.stride(0), member,
                                                               so.data_ptr(), sl.data_ptr(), len(segs), deltas[k].data_ptr(),
                                                               scal[k].contiguous().data_ptr(), _stream(dev)), 'serl_ga_scaled_perturb')
        self.items = []


def safe_mutate(engine, weights, member, spec: NetSpec, mag, buffer, critical_buffer, batch_size, rng=random, delta=None):
    src = critical_buffer if len(critical_buffer) > 1 else buffer          # mod_neuro_evo.py:258-261
    return proximal_mutate(engine, weights, member, spec, mag, src, batch_size, rng, delta)


def novelty(engine, weights, members, spec: NetSpec, states, actions):
    """Actor.get_novelty (genetic_agent.py:111-115) for n (actor, batch) pairs: states [n, B, S], actions [n, B, A]
    -> f32 [n] (device)"""
    dev = weights.device
    members = np.atleast_1d(np.asarray(members, dtype=np.int32))
    states = torch.as_tensor(states, dtype=torch.float32).to(dev).contiguous()
    actions = torch.as_tensor(actions, dtype=torch.float32).to(dev).contiguous()
    n, B = states.shape[0], states.shape[1]
    out = torch.empty(n, dtype=torch.float32, device=dev)
    m = _i32(dev, members)
    _capi.check(engine.lib.serl_ga_novelty(engine.ctx, weights.data_ptr(), weights.stride(0), *_net_args(spec), m.data_ptr(), n,
                                           states.data_ptr(), actions.data_ptr(), B, out.data_ptr(), _stream(dev)), 'serl_ga_novelty')
    return out


def sort_groups_by_distance(engine, weights, genomes, buffers, spec: NetSpec, rng=random):
    """SSNE.sort_groups_by_distance (mod_neuro_evo.py:426-445): every pair (i < j in list order) of `genomes` keyed by
    get_distance = gene1.get_novelty(batch of gene2) + gene2.get_novelty(batch of gene1), batches of
    min(256, len(buffer1), len(buffer2)) from the latest 1000 tuples of each buffer (same python `random` draws, in the
    reference's order); all 2 * pairs forward batches in ONE launch.  -> [(second, first, distance)] sorted descending."""
    from . import replay
    pairs, members, calls = [], [], []
    for i, first in enumerate(genomes):
        for second in genomes[i + 1:]:
            b1, b2 = buffers[first], buffers[second]
            bs = min(256, min(len(b1), len(b2)))
            calls += [(b1, bs), (b2, bs)]                  # sample_from_latest(bs, 1000) of gene1's, then of gene2's buffer
            pairs.append((second, first, bs))
            members += [first, second]                     # gene1 judged on gene2's batch, gene2 on gene1's
    if not pairs:
        return []
    # the draws, in the reference's order; runs of calls with the same (len(latest), batch) are replayed in bulk
    latest = {}
    for ring, _ in calls:
        if id(ring) not in latest:
            latest[id(ring)] = ring.latest_slots(1000)
    picks, k0 = [None] * len(calls), 0
    while k0 < len(calls):
        n0, bs0 = len(latest[id(calls[k0][0])]), calls[k0][1]
        k1 = k0
        while k1 < len(calls) and (len(latest[id(calls[k1][0])]), calls[k1][1]) == (n0, bs0):
            k1 += 1
        got = replay.sample_many(n0, bs0, k1 - k0, rng)
        for k in range(k0, k1):
            picks[k] = got[k - k0]
        k0 = k1
    batches = []
    for (ring, _), pick in zip(calls, picks):
        rows = ring.rows[torch.from_numpy(latest[id(ring)][pick.astype(np.int64)]).to(ring.device)]
        batches.append(ring.split(rows))
    st, ac = [], []
    for k in range(len(pairs)):
        (s1, a1, _, _, _), (s2, a2, _, _, _) = batches[2 * k], batches[2 * k + 1]
        st += [s2, s1]; ac += [a2, a1]
    sizes = {p[2] for p in pairs}
    if len(sizes) == 1:
        nov = novelty(engine, weights, members, spec, torch.stack(st), torch.stack(ac)).cpu().numpy().astype(np.float64)
    else:               # buffers of different fill: one launch per batch size
        nov = np.zeros(len(members), dtype=np.float64)
        for bs in sizes:
            idx = [k for k in range(len(members)) if pairs[k // 2][2] == bs]
            nov[idx] = novelty(engine, weights, [members[k] for k in idx], spec, torch.stack([st[k] for k in idx]),
                               torch.stack([ac[k] for k in idx])).cpu().numpy()
    groups = [(p[0], p[1], float(nov[2 * k]) + float(nov[2 * k + 1])) for k, p in enumerate(pairs)]   # two .item() floats added in f64
    return sorted(groups, key=lambda g: g[2], reverse=True)
