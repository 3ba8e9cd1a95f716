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


def synthetic_reference_tables(n_episodes, num_evals, t_max=80, seed=7, first=0):
    """Benchmark references: episode 0 of every member flies the fixed base reference,
    the others seeded smoothed-step sequences.  -> f64 [n_episodes, n_steps, 3] radians.
    first: global index of the first episode (a rank's block of one sharded population: episodes first .. first + n)."""
    n = n_steps_for(t_max)
    out = np.empty((n_episodes, n, 3))
    th0, ph0 = base_reference(t_max)
    base = tabulate(th0, ph0, t_max)
    tt = np.linspace(0.0, t_max, 6)
    for e in range(n_episodes):
        if (first + e) % num_evals == 0:
            out[e] = base
            continue
        rng = np.random.default_rng(seed + first + e)
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


# ---- references as parameters: generated inside the kernel (include/serl_amd.h: serl_ref_spec) ---------------------------
REF_MAX_STEPS = 8
REF_SPEC_DTYPE = np.dtype([('n_theta', np.int32), ('n_phi', np.int32), ('w_theta', np.float64), ('w_phi', np.float64),
                           ('trim_deg', np.float64), ('t_theta', np.float64, REF_MAX_STEPS), ('a_theta', np.float64, REF_MAX_STEPS),
                           ('t_phi', np.float64, REF_MAX_STEPS), ('a_phi', np.float64, REF_MAX_STEPS)], align=True)
assert REF_SPEC_DTYPE.itemsize == 8 + 3 * 8 + 4 * 8 * REF_MAX_STEPS


def ref_specs(theta, phi, theta_trim_deg=0.22):
    """Sequences of SmoothedStepSequence objects (one pair per episode) -> structured array [E] of serl_ref_spec rows.
    The kernel evaluates them at the env's accumulated step times instead of reading a [T, 3] table per episode."""
    theta, phi = list(theta), list(phi)
    assert len(theta) == len(phi)
    out = np.zeros(len(theta), dtype=REF_SPEC_DTYPE)
    trims = np.broadcast_to(np.asarray(theta_trim_deg, dtype=np.float64), (len(theta),))
    for e, (th, ph) in enumerate(zip(theta, phi)):
        for sig, nk, wk, tk, ak in ((th, 'n_theta', 'w_theta', 't_theta', 'a_theta'), (ph, 'n_phi', 'w_phi', 't_phi', 'a_phi')):
            n = len(sig.times)
            if n > REF_MAX_STEPS:
                raise ValueError('a reference with %d steps does not fit serl_ref_spec (max %d): tabulate it instead' % (n, REF_MAX_STEPS))
            out[e][nk], out[e][wk] = n, sig.w
            out[e][tk][:n], out[e][ak][:n] = sig.times, sig.amps
        out[e]['trim_deg'] = trims[e]
    return out


def det_cospi(s):
    """cos(pi s) for 0 <= s <= 1 with the kernel's operations (vectorised; include/serl_amd.h, serl_ref_spec)"""
    s = np.asarray(s, dtype=np.float64)
    neg = s > 0.5
    r = np.where(neg, 1.0 - s, s)
    use_cos = r <= 0.25
    x = np.where(use_cos, np.pi * r, np.pi * (0.5 - r))
    x2 = x * x
    C = [1.0, -0.5, 0.041666666666666664, -0.001388888888888889, 2.48015873015873e-05, -2.755731922398589e-07,
         2.08767569878681e-09, -1.1470745597729725e-11, 4.779477332387385e-14, -1.5619206968586225e-16, 4.110317623312165e-19]
    S = [1.0, -0.16666666666666666, 0.008333333333333333, -0.0001984126984126984, 2.7557319223985893e-06, -2.505210838544172e-08,
         1.6059043836821613e-10, -7.647163731819816e-13, 2.8114572543455206e-15, -8.22063524662433e-18, 1.9572941063391263e-20]
    pc, ps = np.full_like(x2, C[10]), np.full_like(x2, S[10])
    for k in range(9, -1, -1):
        pc = C[k] + x2 * pc
        ps = S[k] + x2 * ps
    c = np.where(use_cos, pc, x * ps)
    return np.where(neg, -c, c)


def tabulate_specs(specs, t_max, dt=0.01):
    """The table the kernel's generator produces for `specs` (same operations, vectorised on the host): f64 [E, T, 3].
    Differs from `tabulate` (libm cosine, the reference's own arithmetic) by at most a few ulp."""
    n = n_steps_for(t_max, dt)
    t = env_times(n, dt)
    out = np.zeros((len(specs), n, 3))
    for e, r in enumerate(specs):
        for c, (nk, wk, tk, ak) in enumerate((('n_theta', 'w_theta', 't_theta', 'a_theta'), ('n_phi', 'w_phi', 't_phi', 'a_phi'))):
            v = np.zeros(n)
            ti, a, prev, on = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n, bool)
            for i in range(int(r[nk])):
                hit = t >= r[tk][i]
                prev = np.where(hit, np.where(on, a, 0.0), prev)
                ti = np.where(hit, r[tk][i], ti); a = np.where(hit, r[ak][i], a); on = on | hit
            s = np.minimum((t - ti) / r[wk], 1.0)
            v = np.where(on, prev + (a - prev) * (1.0 - det_cospi(np.where(on, s, 0.0))) / 2.0, 0.0)
            if c == 0:
                v = v + np.where((0.0 <= t) & (t <= t_max), r['trim_deg'], 0.0)
            out[e, :, c] = v * (np.pi / 180.0)
    return out


# ---- training references (env.reset() without user_refs, envs/phlabenv.py:316-335) ---------------------------------------
def randomized_cosine_steps(t_max, ampl_max, block_width, smooth_width, n_levels, vary_timings=0.0, rng=np.random):
    """Stand-in for `signals.stochastic_signals.RandomizedCosineStepSequence` (signals==0.0.1 is not vendored by the
    reference and not installable here: PARITY UNPINNED, SURVEY 8c -- only what the training-time state histories of
    the reference show is reproduced: one level per block drawn from linspace(-ampl_max, ampl_max, n_levels), block
    starts every `block_width` seconds jittered by +-vary_timings, cosine blend over `smooth_width`)."""
    n_levels = max(int(n_levels), 2)
    block_width = max(float(block_width), 1e-6)
    levels = np.linspace(-ampl_max, ampl_max, n_levels)
    times = np.arange(0.0, t_max, block_width)
    amps = rng.choice(levels, size=len(times))
    if vary_timings:
        times = np.concatenate([times[:1], times[1:] + rng.uniform(-vary_timings, vary_timings, len(times) - 1)])
    return SmoothedStepSequence(times, amps, max(float(smooth_width), 1e-6))


def training_references(n_episodes, t_max=20, rng=np.random, n_actions=3):
    """What CitationEnv.reset() draws per training episode (envs/phlabenv.py:316-335): theta (+-30 deg) and phi (+-20 deg)
    step sequences with block t_max//5, smooth t_max//6, t_max//2 levels, timing jitter t_max/500.  -> (thetas, phis).
    n_actions = 1 (the symmetric configuration, :304-313): one theta sequence with smooth t_max//6.7 and jitter t_max//500,
    no phi draw (-> a zero sequence)."""
    th, ph = [], []
    for _ in range(n_episodes):
        if n_actions == 1:
            th.append(randomized_cosine_steps(t_max, 30, t_max // 5, t_max // 6.7, t_max // 2, t_max // 500, rng))
            ph.append(SmoothedStepSequence(np.zeros(1), np.zeros(1), 1.0))
            continue
        th.append(randomized_cosine_steps(t_max, 30, t_max // 5, t_max // 6, t_max // 2, t_max / 500., rng))
        ph.append(randomized_cosine_steps(t_max, 20, t_max // 5, t_max // 6, t_max // 2, t_max / 500., rng))
    return th, ph
