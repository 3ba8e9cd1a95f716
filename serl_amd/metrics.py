"""Episode metrics of the reference, batched on the device with torch.fft (rocFFT):
calc_smoothness (base/core/utils.py:82-120) and calc_nMAE (base/core/utils.py:39-58)."""
import math
import numpy as np
import torch


def calc_smoothness(actions, lengths=None, dt=0.01):
    """actions: f64 [E, T, A] (rows past an episode's length ignored); lengths: int [E] (None = T).
    Per episode with N steps: Y = fft(y, N); Syy = |Y[1:N//2]|^2 * dt; S = sum_j sum_i Syy[i,j] * f_i * 2/N
    with f = linspace(dt, 1/(2dt), N//2 - 1); returns -sqrt(S) * 100 * (80 / (N*dt))  as f64 [E]."""
    actions = torch.as_tensor(actions, dtype=torch.float64)
    E, T, A = actions.shape
    if lengths is None:
        lengths = torch.full((E,), T, dtype=torch.int64)
    lengths = torch.as_tensor(lengths).to(torch.int64).cpu().abs()
    # Episodes that ran the whole table (length T: all of an evaluation's, usually) share ONE batched FFT whose plan is made
    # once per process.  The others -- a training generation of untrained actors has ~150 distinct lengths, and an FFT
    # library plans (rocFFT: compiles) per length, 0.7 s each -- go through the direct-DFT kernel (csrc/serl_metrics.hip)
    # in one launch; its O(N^2) is why the full-length episodes do not (1 000 x 8 001 steps: 60 ms against 4).
    if actions.is_cuda and A == 3 and T <= 8192:
        short = torch.nonzero(lengths != T).flatten()
        if len(short):
            out = torch.empty(E, dtype=torch.float64, device=actions.device)
            sub = actions if len(short) == E else actions.index_select(0, short.to(actions.device))
            res = _smoothness_dft(sub, lengths[short], dt)
            if res is not None:
                out[short.to(actions.device)] = res
                full = torch.nonzero(lengths == T).flatten()
                if len(full):
                    out[full.to(actions.device)] = calc_smoothness(actions.index_select(0, full.to(actions.device)), None, dt)
                return out
    out = torch.empty(E, dtype=torch.float64, device=actions.device)
    uniq = torch.unique(lengths).tolist()
    for N in uniq:      # one batched FFT per distinct episode length
        if len(uniq) == 1 and N == T:
            idx, y = None, actions                 # every episode ran the whole table (an evaluation's usual case): no gather, no copy
        else:
            idx = torch.nonzero(lengths == N).flatten().to(actions.device)
            y = actions.index_select(0, idx)[:, :N, :]
        if N < 4:
            out[idx if idx is not None else slice(None)] = 0.0
            continue
        res = _smoothness_full(y, N, dt)
        if idx is None:
            return res
        out[idx] = res
    return out


_WQ = {}      # (N, A, dt, device) -> sqrt of the frequency weight of every (bin, channel, re / im) entry of bins 1 .. N/2 - 1


def _smoothness_full(y, N, dt):
    """calc_smoothness of a batch whose episodes all have N steps: f64 [E, N, A] -> f64 [E] in three launches behind the transform:
    S = sum_i f_i sum_j |Y_ij|^2 is the squared 2-norm of the spectrum's (re, im) entries weighted by sqrt(f_i) -- one elementwise product
    and one long-row norm (a reduction over the six entries of a bin first, tried in between, is the slowest shape a reduction kernel has)."""
    # the signal is real: the half spectrum (rfft) holds every bin 1 .. N/2 - 1 the metric sums -- half the transform and half
    # the memory of fft (1 536 episodes x 8 001 steps: 0.3 GB of spectrum instead of 0.6)
    Yr = torch.view_as_real(torch.fft.rfft(y, n=N, dim=1))                      # [E, N/2 + 1, A, 2]
    E, K, A, _ = Yr.shape
    M = N // 2 - 1
    key = (N, A, float(dt), y.device)
    wq = _WQ.get(key)
    if wq is None:
        freq = torch.linspace(dt, 1 / (2 * dt), M, dtype=torch.float64, device=y.device)
        if len(_WQ) >= 8:                      # (a handful of episode lengths per process at most; never grow without bound)
            _WQ.pop(next(iter(_WQ)))
        wq = _WQ[key] = torch.sqrt(freq).repeat_interleave(2 * A)
    Yv = Yr.reshape(E, K * A * 2)[:, 2 * A:(M + 1) * 2 * A]                    # bins 1 .. N/2 - 1: a strided 2-D view, no copy
    nrm = torch.linalg.vector_norm(Yv * wq, dim=1)                              # sqrt(sum_i f_i sum_j |Y_ij|^2)
    return nrm * (-(math.sqrt(dt * 2 / N) * 100 * (80 / (N * dt))))


def calc_smoothness_speculative(actions, length_steps, dt=0.01):
    """The evaluation's usual case without a host round trip: calc_smoothness as if every episode had flown the whole table, enqueued
    on the current stream right behind the rollout kernel (nothing waits for `length_steps` on the host), plus a device flag that
    says whether that was true.  Returns (smoothness f64 [E], all_full bool scalar on the device), or None when the last batch of this
    shape was a miss (smoothness_speculation_result); when the flag reads False -- after the caller's own synchronisation -- the caller
    calls calc_smoothness(actions, length_steps) instead."""
    E, T, A = actions.shape
    if (E, T) in _SPEC_MISS:      # a batch of this shape had early endings last time (untrained / perturbed actors): the guess would be paid twice
        return None
    all_full = (length_steps.abs() == T).all()
    if T < 4:
        return torch.zeros(E, dtype=torch.float64, device=actions.device), all_full
    return _smoothness_full(actions, T, dt), all_full


_SPEC_MISS = set()


def smoothness_speculation_result(actions, all_full):
    """What the caller found when it read the flag of calc_smoothness_speculative (after its own synchronisation): remembers a miss so that the
    next batch of the same shape goes straight to the general path, forgets it after a hit.  Returns the flag as a bool."""
    ok = bool(all_full)
    E, T, _ = actions.shape
    if ok:
        _SPEC_MISS.discard((E, T))
    else:
        _SPEC_MISS.add((E, T))
    return ok


def calc_smoothness_after_miss(actions, length_steps, dt=0.01):
    """The general path for a batch whose shape calc_smoothness_speculative refused to guess (a miss is remembered per shape) -- and the place
    where the mark is taken back: when every episode of THIS batch flew the whole table again, the next one is guessed again.  One helper for
    every caller (evaluator.evaluate_pop, bench.py), so that none of them keeps paying the host round trip after a single early ending."""
    sm = calc_smoothness(actions, length_steps, dt)
    E, T, _ = actions.shape
    full = (length_steps.abs() == T).all() if torch.is_tensor(length_steps) else (np.abs(np.asarray(length_steps)) == T).all()
    if bool(full):                             # (calc_smoothness has waited for the lengths already: this read costs nothing more)
        _SPEC_MISS.discard((E, T))
    return sm


def calc_smoothness_enqueued(actions, length_steps, dt=0.01):
    """calc_smoothness of episodes of ANY lengths without a host round trip: the direct-DFT kernel (csrc/serl_metrics.hip) on the current stream with
    the lengths read ON THE DEVICE (|length_steps|, what the rollout kernel wrote) and its twiddle table sized for the whole trace.  What the
    asynchronous validation batches use (generation.validate_actor(wait=False)): nothing between the rollout launch and the results' copy waits for
    the host.  actions f64 [E, T, 3] on the GPU, T <= 8192; returns f64 [E] on the device, or None when the kernel does not take the shape."""
    E, T, A = actions.shape
    if not actions.is_cuda or A != 3 or T > 8192:
        return None
    return _smoothness_dft(actions, torch.as_tensor(length_steps).abs(), dt, max_len=max(T, 4))


def _smoothness_dft(actions, lengths, dt, max_len=None):
    import ctypes
    from . import _capi, evaluator
    with torch.cuda.device(actions.device):
        eng = evaluator.default_engine()
    if eng.device != actions.device:
        return None
    L = _capi.lib()
    a = actions.contiguous()
    E, T, _ = a.shape
    n = lengths.to(torch.int32).to(a.device)
    mx = max_len if max_len is not None else max(int(lengths.max()), 4)      # twiddle table / frequency chunks sized for the longest of them
    work = torch.empty(int(L.serl_smoothness_work_size(E, mx)), dtype=torch.float64, device=a.device)
    out = torch.empty(E, dtype=torch.float64, device=a.device)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    _capi.check(L.serl_smoothness(eng.ctx, a.data_ptr(), T * 3, n.data_ptr(), E, mx, float(dt), work.data_ptr(), out.data_ptr(),
                                  ctypes.c_void_p(stream)), 'serl_smoothness')
    return out


def calc_nMAE(errors):
    """errors: f64 [T, 3] (ref - controlled state) -> nMAE in percent."""
    errors = torch.as_tensor(errors, dtype=torch.float64)
    mae = errors.abs().mean(0)
    beta_range = max(abs(float(errors[:, -1].mean())), 3.14159 / 180)
    rng = torch.tensor([math.radians(20), math.radians(20), beta_range], dtype=torch.float64, device=errors.device)
    return float((mae / rng).mean() * 100)


def calc_nMAE_batch(errors, lengths):
    """calc_nMAE for E episodes at once on the device: errors f64 [E, T, 3] (rows past an episode's length ignored),
    lengths int [E] -> f64 [E] percent.  One masked reduction, no host round trip per episode."""
    errors = torch.as_tensor(errors, dtype=torch.float64)
    E, T, _ = errors.shape
    n = torch.as_tensor(lengths).to(errors.device).to(torch.int64).clamp(min=1)
    mask = (torch.arange(T, device=errors.device)[None, :] < n[:, None]).to(torch.float64)[:, :, None]
    nf = n.to(torch.float64)[:, None]
    mae = (errors.abs() * mask).sum(1) / nf                                   # [E, 3]
    beta_range = torch.clamp(((errors[:, :, 2:3] * mask).sum(1) / nf).abs(), min=3.14159 / 180)[:, 0]
    rng = torch.stack([torch.full_like(beta_range, math.radians(20)), torch.full_like(beta_range, math.radians(20)), beta_range], 1)
    return (mae / rng).mean(1) * 100
