"""Actor / GeneticAgent with the reference's class API (base/core/genetic_agent.py:10-163,
base/core/mod_utils.py:14-18,39-50,130) so shipped checkpoints load unchanged and `base/train.py`
style callers keep working; plus packing of actor parameters into the flat f32 rows the kernel reads.

The torch modules here are the *interface* (state_dict layout, extract/inject_parameters for the GA,
gradient-based learners that sit outside the hot path); the rollout itself never calls
`Actor.forward` -- it runs in the HIP kernel from the packed rows.
"""
from dataclasses import dataclass
import numpy as np
import torch
import torch.nn as nn

ACTIVATION_IDS = {'tanh': 0, 'elu': 1, 'relu': 2}   # 'relu' IS LeakyReLU in the reference (mod_utils.py:17)


class LayerNorm(nn.Module):
    """gamma * (x - mean) / (std_unbiased + eps) + beta  (mod_utils.py:39-50; not nn.LayerNorm)."""

    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(features))
        self.beta = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        mean = x.mean(-1, keepdim=True)
        std = x.std(-1, keepdim=True)
        return self.gamma * (x - mean) / (std + self.eps) + self.beta


def _activation(name):
    return {'tanh': nn.Tanh, 'elu': nn.ELU, 'relu': nn.LeakyReLU}[name.lower()]()


def is_lnorm_key(key):
    return key.startswith('lnorm')     # mod_utils.py:130 (never matches `net.N.gamma`; kept as-is)


@dataclass(frozen=True)
class NetSpec:
    state_dim: int
    action_dim: int
    hidden: int
    num_layers: int
    activation: str

    @property
    def param_count(self):
        S, H, L, A = self.state_dim, self.hidden, self.num_layers, self.action_dim
        return H * S + H + L * (H * H + 3 * H) + A * H + A

    @property
    def row_stride(self):
        """floats per member row in the packed population tensor: param_count rounded up to a multiple of 4
        (the kernel reads weight rows with 16-byte loads)"""
        return (self.param_count + 3) // 4 * 4

    @property
    def activation_id(self):
        return ACTIVATION_IDS[self.activation.lower()]

    def genome_segments(self):
        """(offset, length) of every 2-D weight inside the packed row, in named_parameters order --
        the GA genome of extract_parameters / inject_parameters (genetic_agent.py:131-155)."""
        S, H, L, A = self.state_dim, self.hidden, self.num_layers, self.action_dim
        segs, off = [(0, H * S)], H * S + H
        for _ in range(L):
            segs.append((off, H * H))
            off += H * H + 3 * H
        segs.append((off, A * H))
        return segs

    def param_layout(self):
        """(name, offset, shape) for every tensor of the state_dict, in order."""
        S, H, L, A = self.state_dim, self.hidden, self.num_layers, self.action_dim
        out, off = [], 0

        def add(name, shape):
            nonlocal off
            out.append((name, off, shape))
            off += int(np.prod(shape))
        add('net.0.weight', (H, S)); add('net.0.bias', (H,))
        for l in range(L):
            i = 2 + 3 * l
            add('net.%d.weight' % i, (H, H)); add('net.%d.bias' % i, (H,))
            add('net.%d.gamma' % (i + 1), (H,)); add('net.%d.beta' % (i + 1), (H,))
        i = 2 + 3 * L
        add('net.%d.weight' % i, (A, H)); add('net.%d.bias' % i, (A,))
        return out


class Actor(nn.Module):
    def __init__(self, args, init=False):
        super().__init__()
        self.args = args
        h, L = args.hidden_size, args.num_layers
        layers = [nn.Linear(args.state_dim, h), _activation(args.activation_actor)]
        for _ in range(L):
            layers.extend([nn.Linear(h, h), LayerNorm(h), _activation(args.activation_actor)])
        layers.extend([nn.Linear(h, args.action_dim), nn.Tanh()])
        self.net = nn.Sequential(*layers)
        self.to(getattr(args, 'device', 'cpu'))

    @property
    def spec(self):
        a = self.args
        return NetSpec(a.state_dim, a.action_dim, a.hidden_size, a.num_layers, a.activation_actor.lower())

    def forward(self, state):
        return self.net(state)

    def select_action(self, state):
        dev = next(self.parameters()).device
        state = torch.as_tensor(np.asarray(state).reshape(1, -1), dtype=torch.float32, device=dev)
        return self.forward(state).cpu().data.numpy().flatten()

    def get_novelty(self, batch):
        state_batch, action_batch, _, _, _ = batch
        novelty = torch.mean(torch.sum((action_batch - self.forward(state_batch)) ** 2, dim=-1))
        self.novelty = novelty.item()
        return self.novelty

    def _genome_params(self):
        for name, param in self.named_parameters():
            if is_lnorm_key(name) or len(param.shape) != 2:
                continue
            yield param

    def extract_grad(self):
        return torch.cat([p.grad.view(-1) for p in self._genome_params()]).detach().clone()

    def extract_parameters(self):
        return torch.cat([p.view(-1) for p in self._genome_params()]).detach().clone()

    def inject_parameters(self, pvec):
        count = 0
        for p in self._genome_params():
            sz = p.numel()
            p.data.copy_(pvec[count:count + sz].view(p.size()).data)
            count += sz

    def count_parameters(self):
        return sum(p.numel() for p in self._genome_params())


class GeneticAgent:
    """Actor + its personal buffers (genetic_agent.py:10-63).  The replay buffers and the distillation
    update are outside the hot path; any object with `.add(*transition)` can be attached."""

    def __init__(self, args, buffer=None, critical_buffer=None):
        self.args = args
        self.actor = Actor(args)
        self.buffer = buffer
        self.critical_buffer = critical_buffer


def spec_of(obj):
    """NetSpec of an Actor-like object or of a reference state_dict."""
    if isinstance(obj, NetSpec):
        return obj
    if hasattr(obj, 'spec'):
        return obj.spec
    if hasattr(obj, 'actor'):
        return spec_of(obj.actor)
    raise TypeError('cannot derive NetSpec from %r' % type(obj))


def spec_from_state_dict(sd, activation):
    w0 = sd['net.0.weight']
    H, S = w0.shape
    L = sum(1 for k in sd if k.endswith('.gamma'))
    A = sd['net.%d.weight' % (2 + 3 * L)].shape[0]
    return NetSpec(int(S), int(A), int(H), int(L), activation.lower())


def pack_actor(actor_or_sd):
    """Flat f32 row in state_dict order (W0 b0 {Wl bl gamma beta}xL Wo bo) as the kernel expects."""
    if isinstance(actor_or_sd, dict):
        vals = actor_or_sd.values()
    else:
        m = actor_or_sd.actor if hasattr(actor_or_sd, 'actor') else actor_or_sd
        # a module without buffers: parameters() walks the tensors in state_dict order (own parameters, then the children's) without building the
        # dictionary and running its hooks -- 0.02 ms per actor instead of 0.09 (a generation packs pop + 1 of them)
        vals = m.parameters() if next(m.buffers(), None) is None else m.state_dict().values()
    return torch.cat([v.detach().reshape(-1) for v in vals]).to(torch.float32).cpu()


def pad_rows(w):
    """pad the parameter axis of a packed [M, P] tensor with zeros to a multiple of 4 floats"""
    pad = (-w.shape[1]) % 4
    return torch.nn.functional.pad(w, (0, pad)).contiguous() if pad else w.contiguous()


def pack_population(actors, device=None):
    w = pad_rows(torch.stack([pack_actor(a) for a in actors]))
    return w.to(device) if device is not None else w


def unpack_into(actor, row):
    """Write a packed row back into an Actor's parameters (after device-side GA edits)."""
    sd = actor.state_dict()
    off = 0
    row = row.detach().cpu()
    for k, v in sd.items():
        n = v.numel()
        v.copy_(row[off:off + n].view(v.shape))
        off += n
