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
