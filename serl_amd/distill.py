"""distilation_crossover (base/core/mod_neuro_evo.py:131-181 + GeneticAgent.update_parameters,
base/core/genetic_agent.py:22-59): the child starts from the second parent's weights and is trained by Adam to imitate,
state by state, whichever parent's action the critic rates higher (Q-filtered behaviour cloning), on a buffer built
from the latest halves of the parents' buffers.

This is gradient code around the learner's critic -- outside the rollout hot path -- and stays
PyTorch (on the GPU: PyTorch-ROCm).  What the device-resident population contributes is the data plumbing: parents and
child are rows of the packed weight tensor, the training batches are gathered from device replay rings with the
reference's own `random.sample` draws, nothing crosses to the host.  RNG parity with the reference's sequence:
`GeneticAgent(args)` draws a random initialisation from torch's generator before the hard update overwrites it -- a
throw-away Actor is built here for the same reason.
"""
import random
import torch
from torch.optim import Adam
from . import replay
from .actor import Actor, unpack_into, pack_actor


def _actor_from_row(args, spec, row, device):
    import types
    a = types.SimpleNamespace(hidden_size=spec.hidden, num_layers=spec.num_layers, activation_actor=spec.activation,
                              state_dim=spec.state_dim, action_dim=spec.action_dim, device='cpu')
    actor = Actor(a)                       # (random init consumes torch's generator like the reference's GeneticAgent(args))
    unpack_into(actor, row)
    return actor.to(device)


def update_parameters(child, optim, batch, p1, p2, critic):
    """GeneticAgent.update_parameters (genetic_agent.py:22-59)"""
    state = batch[0]
    with torch.no_grad():
        a1, a2 = p1(state), p2(state)
        q1 = torch.min(*critic(state, a1)).flatten()
        q2 = torch.min(*critic(state, a2)).flatten()
    eps = 10 ** -5
    action = torch.cat((a1[q1 - q2 > eps], a2[q2 - q1 >= eps])).detach()
    state = torch.cat((state[q1 - q2 > eps], state[q2 - q1 >= eps]))
    out = child(state)
    optim.zero_grad()
    sq = (out - action) ** 2
    loss = torch.sum(sq) + torch.mean(out ** 2)
    mse = torch.mean(sq)
    loss.backward()
    optim.step()
    return mse


def distilation_crossover(args, engine, spec, weights, first, second, buffers, critic, rng=random):
    """-> (child row f32 [P] on the device, child buffer (DeviceReplay), child critical buffer (empty DeviceReplay))"""
    if critic is None:
        raise ValueError('distilation_crossover needs the learner\'s critic: SSNE(args, engine, spec, critic=...)')
    dev = weights.device
    P = spec.param_count
    cap = int(args.individual_bs)
    buf = replay.DeviceReplay(cap, dev, engine)
    buf.add_latest_from(buffers[first], cap // 2)
    buf.add_latest_from(buffers[second], cap // 2)
    buf.shuffle(rng)
    with torch.random.fork_rng(devices=[]):       # the parents exist already in the reference: their modules draw nothing
        p1 = _actor_from_row(args, spec, weights[first, :P], dev)
        p2 = _actor_from_row(args, spec, weights[second, :P], dev)
    child = _actor_from_row(args, spec, weights[second, :P], dev)          # GeneticAgent(args) + hard_update(.., gene2.actor)
    for p in list(p1.parameters()) + list(p2.parameters()):
        p.requires_grad_(False)
    optim = Adam(child.parameters(), lr=1e-3)
    batch_size = min(128, len(buf))
    iters = len(buf) // batch_size if batch_size else 0
    losses = []
    for _ in range(12):
        for _ in range(iters):
            losses.append(update_parameters(child, optim, buf.sample(batch_size, rng), p1, p2, critic))
    row = pack_actor(child).to(dev)
    return row, buf, replay.DeviceReplay(cap, dev, engine)


# ---- all distillations of an epoch in one launch ----------------------------------------------------------------------

def sample_minibatches(n, k, calls, rng=random):
    """`calls` consecutive `rng.sample(range(n), k)` -> int32 [calls, 128] (first k columns): replay.sample_many"""
    import numpy as np
    out = np.zeros((calls, 128), dtype=np.int32)
    out[:, :k] = replay.sample_many(n, k, calls, rng)
    return out


def _batched_forward(spec, rows, states):
    """Actor.forward for K actors at once: rows f32 [K, >=P] packed parameters, states f32 [K, n, S] -> actions [K, n, A]"""
    from .actor import ACTIVATION_IDS
    S, H, L, A = spec.state_dim, spec.hidden, spec.num_layers, spec.action_dim
    act = {0: torch.tanh, 1: torch.nn.functional.elu, 2: lambda v: torch.nn.functional.leaky_relu(v, 0.01)}[ACTIVATION_IDS[spec.activation]]
    K = rows.shape[0]
    off = 0

    def take(shape):
        nonlocal off
        n = 1
        for d in shape:
            n *= d
        v = rows[:, off:off + n].reshape((K,) + shape)
        off += n
        return v
    W, b = take((H, S)), take((H,))
    h = act(torch.baddbmm(b[:, None, :], states, W.transpose(1, 2)))
    for _ in range(L):
        W, b, g, be = take((H, H)), take((H,)), take((H,)), take((H,))
        y = torch.baddbmm(b[:, None, :], h, W.transpose(1, 2))
        mean = y.mean(-1, keepdim=True)
        std = y.std(-1, keepdim=True)
        h = act(g[:, None, :] * (y - mean) / (std + 1e-6) + be[:, None, :])
    W, b = take((A, H)), take((A,))
    return torch.tanh(torch.baddbmm(b[:, None, :], h, W.transpose(1, 2)))


def distil_batch(args, engine, spec, weights, pairs, buffers, critic, rng=random):
    """All distillation crossovers of an epoch: pairs [(first, second)] in the reference's order ->
    [(child row, child buffer, empty critical buffer)].  Host draws (buffer shuffle, the throw-away initialisation, the
    minibatches) happen pair by pair in the reference's order; the parents' actions and the critic's Q-filter are evaluated
    once per pair over the child's whole buffer; the 12 x (len // 128) Adam steps of all pairs run in ONE kernel launch
    (serl_ga_distill).  Shapes the kernel is not compiled for train pair by pair in PyTorch (distilation_crossover)."""
    import ctypes
    import numpy as np
    from . import _capi
    if critic is None:
        raise ValueError('distilation_crossover needs the learner\'s critic: SSNE(args, engine, spec, critic=...)')
    dev = weights.device
    fused = dev.type == 'cuda' and spec.hidden == 32 and spec.num_layers == 3 and spec.state_dim <= 16 and spec.action_dim <= 4
    if fused and pairs:
        # probe before any host draw is consumed: n_pairs = 0 returns SERL_E_UNSUPPORTED when the training state does not fit
        # this device's LDS per workgroup (the fused kernel needs ~158 KB), and then the pairs train one by one in PyTorch
        z = ctypes.c_void_p(0)
        one = torch.zeros(4, dtype=torch.int32, device=dev)
        rc = _capi.lib().serl_ga_distill(engine.ctx, weights.data_ptr(), weights.stride(0), 0, spec.state_dim, spec.hidden, spec.num_layers,
                                         spec.action_dim, spec.activation_id, weights.data_ptr(), weights.data_ptr(), weights.data_ptr(), 1,
                                         one.data_ptr(), 0, one.data_ptr(), one.data_ptr(), ctypes.c_float(1e-3), z)
        fused = rc == 0
    if not fused or not pairs:
        return [distilation_crossover(args, engine, spec, weights, f, s, buffers, critic, rng) for f, s in pairs]
    P, S, A = spec.param_count, spec.state_dim, spec.action_dim
    cap = int(args.individual_bs)
    bufs, slots, nsteps, batch = [], [], [], []
    for first, second in pairs:
        buf = replay.DeviceReplay(cap, dev, engine)
        buf.add_latest_from(buffers[first], cap // 2)
        buf.add_latest_from(buffers[second], cap // 2)
        buf.shuffle(rng)
        _actor_from_row(args, spec, weights[second, :P], 'cpu')          # GeneticAgent(args): the draws of its random initialisation
        n = len(buf)
        B = min(128, n)
        iters = n // B if B else 0
        bufs.append(buf)
        slots.append(sample_minibatches(n, B, 12 * iters, rng))
        nsteps.append(12 * iters)
        batch.append(B)
    K, rows, steps = len(pairs), max(len(b) for b in bufs), max(nsteps)
    states = torch.zeros(K, max(rows, 1), S, dtype=torch.float32, device=dev)
    for k, b in enumerate(bufs):
        states[k, :len(b)] = b.rows[:len(b), :S]
    with torch.no_grad():
        a1 = _batched_forward(spec, weights[[f for f, _ in pairs]], states)
        a2 = _batched_forward(spec, weights[[s for _, s in pairs]], states)
        flat = states.reshape(-1, S)
        q1 = torch.min(*critic(flat, a1.reshape(-1, A))).reshape(K, -1)
        q2 = torch.min(*critic(flat, a2.reshape(-1, A))).reshape(K, -1)
        eps = 10 ** -5
        take1 = (q1 - q2) > eps                                           # genetic_agent.py:44-46
        keep = (take1 | ((q2 - q1) >= eps)).to(torch.float32).contiguous()
        targets = torch.where(take1[..., None], a1, a2).contiguous()
    child = weights[[s for _, s in pairs]].clone().contiguous()           # hard_update(new_agent.actor, gene2.actor)
    sl = np.zeros((K, max(steps, 1), 128), dtype=np.int32)
    for k, s_ in enumerate(slots):
        sl[k, :len(s_)] = s_
    sl_t = torch.from_numpy(sl).to(dev)
    ns_t = torch.tensor(nsteps, dtype=torch.int32, device=dev)
    bt_t = torch.tensor(batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _capi.check(_capi.lib().serl_ga_distill(engine.ctx, child.data_ptr(), child.stride(0), K, S, spec.hidden, spec.num_layers, A,
                                            spec.activation_id, states.data_ptr(), targets.data_ptr(), keep.data_ptr(), states.shape[1],
                                            sl_t.data_ptr(), sl.shape[1], ns_t.data_ptr(), bt_t.data_ptr(), ctypes.c_float(1e-3),
                                            ctypes.c_void_p(stream)), 'serl_ga_distill')
    return [(child[k, :P], bufs[k], replay.DeviceReplay(cap, dev, engine)) for k in range(K)]
