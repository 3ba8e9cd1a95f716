"""Restatement of the two pinned classes of the un-vendored `signals==0.0.1` package
(reference requirements.txt:8; call sites envs/phlabenv.py:305-344, base/evaluate.py:174-180,
base/evaluation_utils.py:51).  TEST INFRASTRUCTURE.

`SmoothedStepSequence` and `Const` are pinned by the shipped golden trajectories (all 8 001 sampled
reference values of both channels reproduce to <= 1.1e-15 rad).
`RandomizedCosineStepSequence` is NOT pinned ("parity unpinned"): it is a stand-in with the same
constructor, used only to let the reference's own `CitationEnv.reset()` run without user refs.
"""
import numpy as np


class BaseSignal:
    def __call__(self, t):
        raise NotImplementedError

    def __add__(self, other):
        return _Sum(self, other)

    __radd__ = __add__


class _Sum(BaseSignal):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def __call__(self, t):
        return self.a(t) + self.b(t)


class Const(BaseSignal):
    """v on [t0, t1], 0 outside."""

    def __init__(self, t_start, t_end, value):
        self.t0, self.t1, self.v = t_start, t_end, value

    def __call__(self, t):
        return self.v if self.t0 <= t <= self.t1 else 0.0


class SmoothedStepSequence(BaseSignal):
    """Level changes from amps[i-1] (0 before the first) to amps[i] along 0.5*(1-cos(pi*(t-t_i)/w))
    over [t_i, t_i + w]."""

    def __init__(self, times, amplitudes, smooth_width):
        self.times = [float(x) for x in times]
        self.amps = [float(x) for x in amplitudes]
        self.w = float(smooth_width)

    def __call__(self, t):
        v, prev = 0.0, 0.0
        for ti, a in zip(self.times, self.amps):
            if t >= ti:
                s = min((t - ti) / self.w, 1.0)
                v = prev + (a - prev) * (1 - np.cos(np.pi * s)) / 2
                prev = a
        return v


class RandomizedCosineStepSequence(SmoothedStepSequence):
    """UNPINNED stand-in (see module docstring)."""

    def __init__(self, t_max, ampl_max, block_width, smooth_width, n_levels, vary_timings=0.0, rng=None):
        rng = rng or np.random
        n_levels = max(int(n_levels), 2)
        block_width = max(float(block_width), 1e-6)
        levels = np.linspace(-ampl_max, ampl_max, n_levels)
        times = np.arange(0.0, t_max, block_width)
        amps = rng.choice(levels, size=len(times))
        if vary_timings:            # block starts jittered by +-vary_timings (same guess as serl_amd.refsignals.randomized_cosine_steps)
            times = np.concatenate([times[:1], times[1:] + rng.uniform(-vary_timings, vary_timings, len(times) - 1)])
        super().__init__(times, amps, max(float(smooth_width), 1e-6))


class Tabulated(BaseSignal):
    """A signal given by samples at the environment's accumulated step times (exact-key lookup).
    Lets the reference's own CitationEnv consume the same pre-tabulated references as the kernel."""

    def __init__(self, t_keys, values_deg):
        self.table = {float(t): float(v) for t, v in zip(t_keys, values_deg)}

    def __call__(self, t):
        return self.table[float(t)]


def env_times(n_steps, dt=0.01):
    """t_k as the environment accumulates it: t += dt in f64 (envs/phlabenv.py:473)."""
    t = np.empty(n_steps)
    acc = 0.0
    for k in range(n_steps):
        t[k] = acc
        acc += dt
    return t


def n_steps_for(t_max, dt=0.01):
    """Number of env steps of a full-length episode: first k with t_k >= t_max, inclusive."""
    acc, k = 0.0, 0
    while True:
        k += 1
        if acc >= t_max:
            return k
        acc += dt


def tabulate_refs(theta_sig, phi_sig, t_max, theta_trim_deg=0.22, dt=0.01):
    """ref[k] = deg2rad([theta_sig(t_k) + Const(0,t_max,trim)(t_k), phi_sig(t_k), 0]) exactly as
    envs/phlabenv.py:303-349 evaluates it with user_refs."""
    n = n_steps_for(t_max, dt)
    tk = env_times(n, dt)
    trim = Const(0.0, t_max, theta_trim_deg)
    beta = Const(0.0, t_max, 0.0)
    th = theta_sig + trim
    out = np.empty((n, 3))
    for k, t in enumerate(tk):
        out[k] = np.deg2rad(np.asarray([th(t), phi_sig(t), beta(t)]))
    return out
