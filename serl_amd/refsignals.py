"""Reference signals for the attitude-tracking task, pre-tabulated for the kernel.

The reference samples `signals` objects once per env step at the environment's *accumulated* time
(`self.t += self.dt`, envs/phlabenv.py:347-349,473) and converts degrees to radians.  The kernel
consumes that as a table `ref[k, 0:3]` (theta, phi, beta; radians; k = 0 .. n_steps-1), so signal
generation stays on the host.  `signals==0.0.1` is not vendored by the reference
(requirements.txt:8); the two classes the evaluation path uses are restated here in vectorised
form (pinned by the shipped golden trajectories).
"""
import numpy as np


def env_times(n_steps, dt=0.01):
    """t_k = fl(sum of k additions of dt), as the env accumulates it."""
    return np.concatenate(([0.0], np.cumsum(np.full(n_steps - 1, dt)))) if n_steps > 1 else np.zeros(1)


def n_steps_for(t_max, dt=0.01):
    """Steps of a full-length episode: the episode ends at the first k with t_k >= t_max
    (envs/phlabenv.py:391-399) -- 8 001 for 80 s, 2 001 for 20 s."""
    n = int(round(t_max / dt)) + 8
    t = env_times(n, dt)
    return int(np.argmax(t >= t_max)) + 1


class SmoothedStepSequence:
    """Levels `amplitudes[i]` starting at `times[i]`, blended from the previous level (0 before the
    first) along 0.5*(1-cos(pi*(t-t_i)/w)) over [t_i, t_i+w]."""

    def __init__(self, times, amplitudes, smooth_width):
        self.times = np.asarray(times, dtype=np.float64)
        self.amps = np.asarray(amplitudes, dtype=np.float64)
        self.w = float(smooth_width)

    def __call__(self, t):
        t = np.asarray(t, dtype=np.float64)
        v = np.zeros_like(t)
        prev = 0.0
        for ti, a in zip(self.times, self.amps):
            on = t >= ti
            s = np.minimum((t - ti) / self.w, 1.0)
            v = np.where(on, prev + (a - prev) * (1 - np.cos(np.pi * s)) / 2, v)
            prev = a
        return v


class Const:
    def __init__(self, t_start, t_end, value):
        self.t0, self.t1, self.v = t_start, t_end, value

    def __call__(self, t):
        t = np.asarray(t, dtype=np.float64)
        return np.where((self.t0 <= t) & (t <= self.t1), self.v, 0.0)


def tabulate(theta_sig, phi_sig, t_max, theta_trim_deg=0.22, dt=0.01):
    """-> f64 [n_steps, 3] radians: deg2rad([theta(t_k) + trim on [0,t_max], phi(t_k), 0]).
    With user-supplied references the env keeps its default theta trim 0.22 deg
    (envs/phlabenv.py:202,319-344)."""
    n = n_steps_for(t_max, dt)
    t = env_times(n, dt)
    th = np.asarray(theta_sig(t), dtype=np.float64) + Const(0.0, t_max, theta_trim_deg)(t)
    ph = np.asarray(phi_sig(t), dtype=np.float64)
    be = Const(0.0, t_max, 0.0)(t)
    return np.deg2rad(np.stack([th, ph, be], axis=1))


def base_reference(t_max=80):
    """The fixed evaluation reference of base/evaluate.py:167-180."""
    tt = np.linspace(0.0, t_max, 6)
    th = SmoothedStepSequence(tt, [0, 12, 3, -4, -8, 2], smooth_width=t_max // 10)
    ph = SmoothedStepSequence(tt, [2, -2, 2, 10, 2, -6], smooth_width=t_max // 10)
    return th, ph


def gen_refs(t_max, amp_times, ampl_max, num_trails=10, rng=np.random):
    """Random smoothed-step references, base/evaluation_utils.py:23-55 (same draw order)."""
    refs = []
    amp_times = list(amp_times)
    for _ in range(num_trails):
        choices = np.linspace(-ampl_max, ampl_max, 6)
        amplitudes = rng.choice(choices, size=6, replace=True)
        amplitudes[0] = 0.0
        amp_times = [amp_times[0]] + [t + rng.uniform(-0.05, 0.05) for t in amp_times[1:]]
        refs.append(SmoothedStepSequence(amp_times, amplitudes, smooth_width=t_max // 10))
    return refs


def synthetic_reference_tables(n_episodes, num_evals, t_max=80, seed=7):
    """Benchmark references: episode 0 of every member flies the fixed base reference,
    the others seeded smoothed-step sequences.  -> f64 [n_episodes, n_steps, 3] radians."""
    n = n_steps_for(t_max)
    out = np.empty((n_episodes, n, 3))
    th0, ph0 = base_reference(t_max)
    base = tabulate(th0, ph0, t_max)
    tt = np.linspace(0.0, t_max, 6)
    for e in range(n_episodes):
        if e % num_evals == 0:
            out[e] = base
            continue
        rng = np.random.default_rng(seed + e)
        a_th = rng.choice(np.linspace(-12, 12, 6), size=6); a_th[0] = 0.0
        a_ph = rng.choice(np.linspace(-10, 10, 6), size=6)
        out[e] = tabulate(SmoothedStepSequence(tt, a_th, t_max // 10), SmoothedStepSequence(tt, a_ph, t_max // 10), t_max)
    return out


def tabulate_batch_device(times, amps_theta, amps_phi, smooth_width, t_max, device=None, theta_trim_deg=0.22, dt=0.01):
    """Device-side tabulation of E smoothed-step references at once: the same formula as
    SmoothedStepSequence / tabulate, evaluated with torch on `device`, so that a population's reference tables never
    exist on the host.  times / amps_*: [E, K] (degrees); -> f64 [E, n_steps, 3] radians.
    The sample times t_k are the env's sequentially accumulated ones (computed once on the host: a parallel scan
    would round differently)."""
    import torch
    n = n_steps_for(t_max, dt)
    t = torch.as_tensor(env_times(n, dt), dtype=torch.float64, device=device)[None, :]            # [1, n]
    times = torch.as_tensor(np.asarray(times), dtype=torch.float64, device=device)
    out = []
    for amps in (amps_theta, amps_phi):
        amps = torch.as_tensor(np.asarray(amps), dtype=torch.float64, device=device)
        v = torch.zeros(times.shape[0], n, dtype=torch.float64, device=device)
        prev = torch.zeros(times.shape[0], 1, dtype=torch.float64, device=device)
        for k in range(times.shape[1]):
            ti, a = times[:, k:k + 1], amps[:, k:k + 1]
            s = torch.clamp((t - ti) / float(smooth_width), max=1.0)
            v = torch.where(t >= ti, prev + (a - prev) * (1 - torch.cos(np.pi * s)) / 2, v)
            prev = a
        out.append(v)
    inside = ((t >= 0.0) & (t <= t_max)).to(torch.float64)
    th = out[0] + theta_trim_deg * inside
    ref = torch.stack([th, out[1], torch.zeros_like(th)], dim=2)
    return ref * (np.pi / 180.0)
