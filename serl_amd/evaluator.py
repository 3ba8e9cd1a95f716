"""Population rollout on the GPU behind the reference's call shapes.

  RolloutEngine        owns the HIP context (tables of the dynamics builds) on one GPU
  evaluate_pop(...)    what the GA loop of base/core/agent.py:229-256 (and the eval_pop loop of
                       base/evaluate.py:242-256) should call: all members x num_evals episodes in one
                       fused kernel launch -> PopResult (fitness[num_evals, pop], champion index ...)
  make_evaluate(...)   adaptor with the exact signature of Agent.evaluate (agent.py:63-66), also what
                       SSNE.__init__ receives as `evaluate` (mod_neuro_evo.py:15,21)

Host code is plumbing: tensors are allocated by PyTorch-ROCm, their raw device pointers and the current
HIP stream are handed to the C ABI (include/serl_amd.h).
"""
import ctypes
from dataclasses import dataclass, field
from typing import Optional, Sequence
import numpy as np
import torch

from . import _capi, builds, refsignals, metrics
from .actor import NetSpec, pack_population, pad_rows, spec_of
from .episode import Episode


@dataclass
class PopResult:
    fitness: np.ndarray          # f64 [num_evals, pop]   episode.fitness (sum of rewards [+ smoothness])
    returns: np.ndarray          # f64 [num_evals, pop]   plain sum of rewards
    smoothness: np.ndarray       # f64 [num_evals, pop]
    length_steps: np.ndarray     # i32 [num_evals, pop]
    length_t: np.ndarray         # f64 [num_evals, pop]   info['t'] after the last increment
    cost_steps: np.ndarray       # i32 [num_evals, pop]
    pop_fitness: np.ndarray      # f64 [pop]  mean over evals (agent.py:245)
    champion: int                # argmax(pop_fitness)   (agent.py:255)
    worst: int                   # argmin(pop_fitness)   (agent.py:287)
    kernel_ms: float = 0.0
    actions: Optional[torch.Tensor] = None     # f64 [E, T, 3] device, when traces requested
    states: Optional[torch.Tensor] = None      # f64 [E, T, 12]
    rewards: Optional[torch.Tensor] = None     # f64 [E, T]
    transitions: Optional[torch.Tensor] = None  # f32 [E, T, 20]
    episode_member: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    mixed_placement: Optional[dict] = None     # a mixed-fault sweep flown as ONE launch: how its workgroups were placed (RolloutEngine.mixed_placement:
                                               # decision 1 = census of the CU pairs, 2 = tickets -- the GPU was shared or the pair mapping collided --, 0 = blockIdx ranges)


class RolloutEngine:
    """One per process / GPU.  Not thread-safe (one HIP context, calls are stream-ordered)."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError('serl_amd.RolloutEngine needs a ROCm GPU (torch.cuda.is_available() is False); '
                               'the product has no CPU path -- the CPU restatement lives in oracle/ for tests only')
        self.lib = _capi.lib()
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else int(device))
        h = ctypes.c_void_p()
        _capi.check(self.lib.serl_ctx_create(self.device.index, ctypes.byref(h)), 'serl_ctx_create')
        self.ctx = h
        self.slots = {}
        # side streams for mixed-build evaluations, made up front: the runtime binds a stream to one of its few hardware
        # queues when it is created, and streams made back to back land on different ones (measured: profiles/r01_g_mixed.md)
        self._side = [torch.cuda.Stream(self.device) for _ in range(3)]
        self.last_kernel_ms = 0.0
        self.num_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        self.multi_launches = 0          # serl_rollout_multi launches made (a mixed sweep as one launch of one code object)
        # serl_rollout_desc.kernel_hint of rollouts that do not name one (None = chosen from the episode count); tests and
        # A/B measurements set it to compare the kernel families ('team', 'team2', 'team4', 'wave', 'half')
        self.kernel_hint = None

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.serl_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def side_stream(self, i):
        """i-th side stream of this engine (mixed-build evaluations run one launch per dynamics build side by side);
        created once: a HIP stream is bound to a hardware queue when it is made"""
        while len(self._side) <= i:
            self._side.append(torch.cuda.Stream(self.device))
        return self._side[i]

    def slot_of(self, build):
        data, ent = builds.load(build)
        key = ent['data']
        if key in self.slots:
            return self.slots[key]
        slot = len(self.slots)
        bd = _capi.BuildDesc(code=builds.CODE_IDS[ent['code']], n_ro=len(data['ro']), ro_base=int(np.asarray(data['ro_base']).reshape(-1)[0]),
                             ro=data['ro'].ctypes.data, t3=data['t3'].ctypes.data, x0=data['x0'].ctypes.data,
                             dw0=data['dw0'].ctypes.data, dt=float(np.asarray(data['dt']).reshape(-1)[0]))
        _capi.check(self.lib.serl_ctx_load_build(self.ctx, slot, ctypes.byref(bd)), 'serl_ctx_load_build(%s)' % build)
        self.slots[key] = slot
        return slot

    # ------------------------------------------------------------------------------------------
    def rollout(self, weights, spec: NetSpec, member_of_episode, ref, *, build='h2000_v90', faults=None,
                err0=None, tick0=None, action_noise=None, noise_row=None, sensor_noise=None, sensor_row=None, t_max=80.0, traces=False,
                transitions=False, lanes_per_wave=0, sync=True, concurrent_episodes=0, env_config=0, incremental=False, kernel=None, launch=True):
        """Low-level: run len(member_of_episode) episodes.  weights f32 [M, >=P] (device or host),
        ref f64 [E, T, 3] or [T, 3] radians.  env_config / incremental: builds.env_config(name) (the attitude task by
        default; the per-episode tables keep their 3-column layouts, transitions have 2 S + A + 3 columns).
        Returns dict of device tensors.  launch=False: nothing is launched -- the prepared descriptor rides along as out['_desc'] for
        rollout_multi (one launch for the descriptors of several builds)."""
        S_, A_ = builds.env_dims(env_config, incremental)
        if (spec.state_dim, spec.action_dim) != (S_, A_):
            raise ValueError('actor %d -> %d does not fit the env configuration (%d observations, %d actions)'
                             % (spec.state_dim, spec.action_dim, S_, A_))
        dev = self.device
        w = torch.as_tensor(weights, dtype=torch.float32).to(dev)
        if w.shape[1] % 4 or w.stride(0) % 4 or not w.is_contiguous():
            w = pad_rows(w)
        moe = torch.as_tensor(np.asarray(member_of_episode), dtype=torch.int32).to(dev).contiguous()
        E = moe.numel()
        spec_t = None
        if isinstance(ref, np.ndarray) and ref.dtype.names:     # refsignals.ref_specs rows: generated in the kernel
            assert ref.dtype == refsignals.REF_SPEC_DTYPE and len(ref) in (1, E)
            spec_t = torch.from_numpy(np.ascontiguousarray(ref).view(np.uint8).reshape(len(ref), -1)).to(dev)
            ref_t, shared, T = None, True, refsignals.n_steps_for(t_max)
        else:
            ref_t = torch.as_tensor(ref, dtype=torch.float64).to(dev).contiguous()
            shared = ref_t.dim() == 2
            T = ref_t.shape[-2]
            assert ref_t.shape[-1] == 3 and (shared or ref_t.shape[0] == E)
        out = dict(fitness=torch.zeros(E, dtype=torch.float64, device=dev),
                   length_steps=torch.zeros(E, dtype=torch.int32, device=dev),
                   length_t=torch.zeros(E, dtype=torch.float64, device=dev),
                   cost_steps=torch.zeros(E, dtype=torch.int32, device=dev))
        d = _capi.RolloutDesc(state_dim=spec.state_dim, action_dim=spec.action_dim, hidden=spec.hidden,
                              num_layers=spec.num_layers, activation=spec.activation_id, n_members=w.shape[0],
                              weights=w.data_ptr(), weight_stride=w.stride(0) if w.shape[0] > 1 else w.shape[1], n_episodes=E,   # (a [1, P] view may carry stride 0)
                              build_slot=self.slot_of(build), member_of_episode=moe.data_ptr(),
                              ref=0 if ref_t is None else ref_t.data_ptr(), ref_stride=0 if shared else T * 3, t_max=float(t_max),
                              max_steps=T, lanes_per_wave=int(lanes_per_wave),
                              kernel_hint=_capi.KERNEL_HINTS[kernel if kernel is not None else self.kernel_hint],
                              concurrent_episodes=int(concurrent_episodes), env_config=int(env_config), incremental=int(bool(incremental)),
                              fitness=out['fitness'].data_ptr(), length_steps=out['length_steps'].data_ptr(),
                              length_t=out['length_t'].data_ptr(), cost_steps=out['cost_steps'].data_ptr())
        keep = [w, moe, ref_t, spec_t]
        if spec_t is not None:
            d.ref_spec, d.ref_spec_stride = spec_t.data_ptr(), (0 if spec_t.shape[0] == 1 else 1)
        if faults is not None:
            f = torch.as_tensor(np.asarray(faults, dtype=np.float64).reshape(E, 8)).to(dev).contiguous()
            d.faults = f.data_ptr(); keep.append(f)
        if err0 is not None:
            e0 = torch.as_tensor(np.asarray(err0, dtype=np.float64).reshape(E, 3)).to(dev).contiguous()
            d.err0 = e0.data_ptr(); keep.append(e0)
        if tick0 is not None:
            tk = torch.as_tensor(np.asarray(tick0, dtype=np.int32).reshape(E)).to(dev).contiguous()
            d.tick0 = tk.data_ptr(); keep.append(tk)
        if sensor_noise is not None:     # [rows, T + 1, 7] (builds.sensor_noise_table); sensor_row[e] = row, -1 = none
            sn = torch.as_tensor(sensor_noise, dtype=torch.float64).to(dev).contiguous()
            assert sn.dim() == 3 and sn.shape[1:] == (T + 1, 7), sn.shape
            if sensor_row is None:
                assert sn.shape[0] == E
            else:
                sr = torch.as_tensor(np.asarray(sensor_row), dtype=torch.int32).to(dev).contiguous()
                assert sr.numel() == E and int(sr.max()) < sn.shape[0]
                d.sensor_row = sr.data_ptr(); keep.append(sr)
            d.sensor_noise = sn.data_ptr(); keep.append(sn)
        if action_noise is not None:
            an = torch.as_tensor(action_noise, dtype=torch.float64).to(dev).contiguous()
            if noise_row is None:
                assert an.shape == (E, T, 3)
            else:           # an: [rows, T, 3]; noise_row[e] = row the episode adds to its actions, -1 = none
                nr = torch.as_tensor(np.asarray(noise_row), dtype=torch.int32).to(dev).contiguous()
                assert an.dim() == 3 and an.shape[1:] == (T, 3) and nr.numel() == E and int(nr.max()) < an.shape[0]
                d.noise_row = nr.data_ptr(); keep.append(nr)
            d.action_noise = an.data_ptr(); keep.append(an)
        if traces:      # True: actions + states + rewards; 'actions': only the action trace (what calc_smoothness needs)
            out['actions'] = torch.zeros(E, T, 3, dtype=torch.float64, device=dev)
            d.actions = out['actions'].data_ptr()
            if traces != 'actions':
                out['states'] = torch.zeros(E, T, 12, dtype=torch.float64, device=dev)
                out['rewards'] = torch.zeros(E, T, dtype=torch.float64, device=dev)
                d.states, d.rewards = out['states'].data_ptr(), out['rewards'].data_ptr()
        if transitions:
            out['transitions'] = torch.zeros(E, T, 2 * spec.state_dim + spec.action_dim + 3, dtype=torch.float32, device=dev)
            d.transitions = out['transitions'].data_ptr()
        out['_keep'] = keep
        if not launch:
            out['_desc'] = d
            return out
        stream = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(self.lib.serl_rollout(self.ctx, ctypes.byref(d), ctypes.c_void_p(stream)), 'serl_rollout')
        if sync:
            ms = ctypes.c_float()
            _capi.check(self.lib.serl_last_rollout_ms(self.ctx, ctypes.byref(ms)), 'serl_last_rollout_ms')
            self.last_kernel_ms = float(ms.value)
            bad = (out['length_steps'] < 0)
            if bool(bad.any()):
                raise RuntimeError('reference table too short: %d episodes were still running after %d steps'
                                   % (int(bad.sum()), T))
        return out

    def rollout_multi(self, prepared):
        """ONE launch of ONE code object for the descriptors of several dynamics builds (C ABI v7 serl_rollout_multi: a mixed-fault population,
        envs/phlabenv.py:114-165).  `prepared`: the dicts rollout(..., launch=False) returned.  Returns True when the library launched them, False
        when the combination is not eligible (SERL_E_UNSUPPORTED: nothing was launched; the caller launches them one by one)."""
        n = len(prepared)
        arr = (_capi.RolloutDesc * n)(*[p['_desc'] for p in prepared])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.serl_rollout_multi(self.ctx, n, arr, ctypes.c_void_p(stream))
        if rc == _capi.E_UNSUPPORTED:
            return False
        _capi.check(rc, 'serl_rollout_multi')
        self.multi_launches += 1
        return True

    def last_rollout_info(self):
        """WHAT the most recent rollout / rollout_multi call launched (C ABI v8 serl_last_rollout_info): dict(family = 'team' | 'teams' | 'teams2' |
        'teamx' | 'team2' | 'team2s' | 'team4' | 'team4_mixed' | 'half' | 'wave' | 'wavex' | 'lane', workgroups, episodes_per_team, work_queue,
        actor_wavefronts, actor_streamed, launches, code).  Does not wait for the device."""
        out = (ctypes.c_int32 * 8)()
        _capi.check(self.lib.serl_last_rollout_info(self.ctx, out), 'serl_last_rollout_info')
        return dict(family=_capi.FAMILIES[int(out[0])], workgroups=int(out[1]), episodes_per_team=int(out[2]), work_queue=bool(out[3]),
                    actor_wavefronts=int(out[4]), actor_streamed=bool(out[5]), launches=int(out[6]), code=int(out[7]))

    def mixed_placement(self):
        """How the most recent rollout_multi launch placed its workgroups (development aid, include/serl_amd.h serl_debug_mixed_placement):
        dict(decision = 0 blockIdx ranges | 1 census of the CU pairs | 2 tickets, registered, pairs, singles).  Waits for the device."""
        out = (ctypes.c_int32 * 4)()
        _capi.check(self.lib.serl_debug_mixed_placement(self.ctx, out), 'serl_debug_mixed_placement')
        return dict(decision=int(out[0]), registered=int(out[1]), pairs=int(out[2]), singles=int(out[3]))

    def dynamics_open_loop(self, cmds, build='h2000_v90', lanes_per_wave=0, kernel=None):
        """Dynamics only: cmds f64 [E, T, 10] -> states f64 [E, T, 12] (what the reference's raw
        initialize()/step() return for the same command sequence)."""
        c = torch.as_tensor(cmds, dtype=torch.float64).to(self.device).contiguous()
        E, T, _ = c.shape
        out = torch.zeros(E, T, 12, dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _capi.check(self.lib.serl_dyn_open_loop(self.ctx, self.slot_of(build), E, T, c.data_ptr(), out.data_ptr(),
                                                int(lanes_per_wave), _capi.KERNEL_HINTS[kernel if kernel is not None else self.kernel_hint],
                                                ctypes.c_void_p(stream)), 'serl_dyn_open_loop')
        self.kernel_ms()
        return out

    def kernel_ms(self):
        ms = ctypes.c_float()
        _capi.check(self.lib.serl_last_rollout_ms(self.ctx, ctypes.byref(ms)), 'serl_last_rollout_ms')
        self.last_kernel_ms = float(ms.value)
        return self.last_kernel_ms


_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = RolloutEngine()
    return _default_engine


def _as_weights(actors, spec):
    if isinstance(actors, torch.Tensor):
        return actors
    return pack_population(actors)


def evaluate_pop(actors, *, mode='nominal', num_evals=3, refs=None, t_max=80, smooth_fitness=False,
                 spec: Optional[NetSpec] = None, engine: Optional[RolloutEngine] = None, traces=False,
                 transitions=False, err0=None, tick0=None, lanes_per_wave=0, need_smoothness=True,
                 sensor_rng=None, concurrent=True, fused='auto') -> PopResult:
    """Evaluate a whole population: `num_evals` episodes per member (agent.py:229-256).

    actors : sequence of Actor / GeneticAgent, or a packed f32 tensor [pop, P] (then pass `spec`)
    mode   : PH-LAB mode or env name ('nominal', 'be', 'PHlab_attitude_ice', ...; envs/phlabenv.py:99-172);
             a sequence gives one mode per episode (mixed-fault sweeps: one kernel launch per dynamics build).
             Env names select the configuration too ('PHlab_symmetric_nominal', 'PHlab_full_incremental', ...:
             builds.env_config; one configuration per call, the actors must have its state_dim / action_dim)
    refs   : f64 [pop*num_evals, T, 3] / [num_evals, T, 3] / [T, 3] radians tables (refsignals.tabulate), or
             refsignals.ref_specs rows [pop*num_evals] / [num_evals] / [1] (generated inside the kernel: no table in HBM);
             None = the fixed base evaluation reference for every episode (the reference's training loop draws a new
             RandomizedCosineStepSequence per reset(): pass refsignals.ref_specs(*refsignals.training_references(E, t_max)))
    tick0  : i32 [pop*num_evals] model clock each episode starts with (None = 0).  The reference's initialize()
             does not reset the model clock, so in its sequential loop episode j of a process starts at
             tick = sum over earlier episodes of (steps + 1); only the time-switched builds (cg-shift, gust) care.
    concurrent : False = the per-build launches run one after the other (A/B switch)
    fused      : several builds as ONE launch of one code object (C ABI v7 serl_rollout_multi), its workgroups placed so that CUs which share an
                 instruction cache run the same code variant.  'auto' (default) and True: whenever the library accepts the combination (more than
                 2 x CUs episodes, attitude task, hidden 32, nominal / ice code): 768 episodes 33.8 against 30.1 M env-steps/s for a launch per build
                 side by side, 6 144 episodes 39.9 against 36.4 M (profiles/r05_experiments.md sections 9, 11); False: never
    sensor_rng : modes 'noise' / 'gust' add the reference's sensor model to what step() returns; its randn draws
             come from this legacy generator (None = np.random, like the wrappers), one block of T + 1 steps per
             noisy episode in episode order, up front (builds.sensor_noise_table)
    Episode order is member-major: e = member*num_evals + eval  (the reference's loop nest)."""
    engine = engine or default_engine()
    if spec is None:
        spec = spec_of(actors[0])
    w = _as_weights(actors, spec)
    pop = w.shape[0]
    E = pop * num_evals
    moe = np.repeat(np.arange(pop, dtype=np.int32), num_evals)
    if refs is None:
        refs = refsignals.tabulate(*refsignals.base_reference(t_max), t_max)
    generated = isinstance(refs, np.ndarray) and refs.dtype.names is not None      # refsignals.ref_specs rows
    if generated:
        if len(refs) == num_evals and num_evals != E:
            refs = np.tile(refs, pop)
        assert len(refs) in (1, E)
    else:
        refs = torch.as_tensor(refs, dtype=torch.float64)
        if refs.dim() == 3 and refs.shape[0] == num_evals and num_evals != E:
            refs = refs.repeat(pop, 1, 1)
    modes = [mode] * E if isinstance(mode, str) else list(mode)
    assert len(modes) == E
    resolved = [builds.resolve_mode(m) for m in modes]
    envs = {builds.env_config(m) for m in modes}
    if len(envs) != 1:
        raise ValueError('one env configuration per evaluate_pop call (got %s)' % sorted(envs))
    env_cfg, incremental = envs.pop()
    need_actions = traces or smooth_fitness or need_smoothness
    want = True if traces else ('actions' if need_actions else False)
    tick0 = None if tick0 is None else np.asarray(tick0, dtype=np.int32).reshape(E)
    err0 = None if err0 is None else np.asarray(err0, dtype=np.float64).reshape(E, 3)
    # one kernel launch per dynamics build (be/jr/sa/se share the nominal build as per-episode fault rows; cg, ice,
    # cg-shift ... are builds of their own); launches are stream-ordered and their results scattered back
    T_ref = refsignals.n_steps_for(t_max) if generated else refs.shape[-2]
    sensor = {e: builds.sensor_noise_table(T_ref, sensor_rng if sensor_rng is not None else np.random)
              for e in range(E) if builds.has_sensor_noise(modes[e])}
    groups = {}
    for e, (b, _) in enumerate(resolved):
        groups.setdefault(b, []).append(e)
    out, kernel_ms = None, 0.0
    # several builds: their launches run side by side on streams of their own (the library sizes each launch for the
    # episodes of all of them: `concurrent_episodes`), joined on the caller's stream afterwards
    many = len(groups) > 1 and concurrent
    cur = torch.cuda.current_stream(engine.device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if many:
        ev0.record(cur)
    parts = []

    def one_build(b, idx, side, **kw):
        rows = [resolved[e][1] for e in idx]
        faults = np.array(rows, dtype=np.float64) if any(r != builds.NOMINAL_ROW for r in rows) else None
        whole = len(idx) == E
        sn = sr = None
        noisy = [e for e in idx if e in sensor]
        if noisy:
            sn = np.stack([sensor[e] for e in noisy])
            pos = {e: j for j, e in enumerate(noisy)}
            sr = np.array([pos.get(e, -1) for e in idx], dtype=np.int32)
        with torch.cuda.stream(side):
            return engine.rollout(w, spec, moe[idx], (refs if (whole or len(refs) == 1) else refs[idx]) if generated else
                                  (refs if (whole or refs.dim() == 2) else refs[torch.as_tensor(idx)]), build=b,
                                  faults=faults, err0=None if err0 is None else err0[idx],
                                  tick0=None if tick0 is None else tick0[idx], t_max=t_max, sensor_noise=sn, sensor_row=sr,
                                  traces=want, transitions=transitions, lanes_per_wave=lanes_per_wave, env_config=env_cfg, incremental=incremental, **kw)

    # Several builds of the attitude task with the LDS-sized actor shape: ONE launch of ONE code object (C ABI v7 serl_rollout_multi,
    # rollout_team4_mixed.hip), which can place its workgroups: two code variants on CUs that share an instruction cache cost 12 %.  The library says
    # whether the combination is eligible; if not, nothing was launched and the launches below run side by side as before.
    use_multi = False
    if many and (fused is True or (fused == 'auto' and E > 2 * engine.num_cus)) and 2 <= len(groups) <= 4 and lanes_per_wave == 0 and env_cfg == 0 and not incremental and spec.hidden == 32:
        prepared = [(np.asarray(idx), one_build(b, np.asarray(idx), cur, sync=False, launch=False)) for b, idx in groups.items()]
        if engine.rollout_multi([o for _, o in prepared]):
            use_multi = True
            parts = [(idx, o, cur) for idx, o in prepared]
    for b, idx in ([] if use_multi else groups.items()):
        idx = np.asarray(idx)
        whole = len(idx) == E
        side = engine.side_stream(len(parts)) if many else cur
        side.wait_stream(cur)
        o = one_build(b, idx, side, sync=not (many or whole), concurrent_episodes=(E - len(idx)) if many else 0)
        if whole:      # (not waited for: the smoothness pass below is enqueued behind the kernel first)
            out = o
            break
        if not many:
            kernel_ms += engine.last_kernel_ms
        else:
            for v in o.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
        parts.append((idx, o, side))
    if parts:
        for _, _, side in parts:
            cur.wait_stream(side)
        ev1.record(cur)
        out = {}
        for idx, o, _ in parts:
            ti = torch.as_tensor(idx, device=o['fitness'].device)
            for k, v in o.items():
                if not torch.is_tensor(v):
                    continue
                if k not in out:
                    out[k] = torch.zeros((E,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
                out[k][ti] = v
        if many:
            ev1.synchronize()
            kernel_ms = ev0.elapsed_time(ev1)          # all launches of the evaluation, side by side
        if bool((out['length_steps'] < 0).any()):
            raise RuntimeError('reference table too short: %d episodes were still running after %d steps'
                               % (int((out['length_steps'] < 0).sum()), T_ref))
    engine.last_kernel_ms = kernel_ms
    spec_sm = None
    if need_actions:      # enqueued before the first host round trip, as if every episode flew the whole table (the usual case)
        spec_sm = metrics.calc_smoothness_speculative(out['actions'], out['length_steps'])
    ret = out['fitness'].cpu().numpy()
    ls = out['length_steps'].cpu().numpy()
    if not parts:      # one launch for the whole population: its time and its sanity check now that the host has waited
        engine.last_kernel_ms = engine.kernel_ms()
        if (ls < 0).any():
            raise RuntimeError('reference table too short: %d episodes were still running after %d steps' % (int((ls < 0).sum()), T_ref))
    if need_actions:
        hit = spec_sm is not None and metrics.smoothness_speculation_result(out['actions'], spec_sm[1])
        sm = (spec_sm[0] if hit else (metrics.calc_smoothness_after_miss if spec_sm is None else metrics.calc_smoothness)(out['actions'], ls)).cpu().numpy()
    else:
        sm = np.zeros(E)
    fit = ret + sm if smooth_fitness else ret.copy()
    sh = lambda a: np.ascontiguousarray(a.reshape(pop, num_evals).T)
    fitness = sh(fit)
    pop_fitness = np.mean(fitness, axis=0)
    placement = None
    if use_multi:
        # how the one-launch sweep placed its workgroups (the results are back: the launch is over).  Tickets instead of the census -- a GPU shared with
        # another launch, or a device whose CU-pair mapping differs from the one the census assumes -- cost ~10 %: said once, not silently
        placement = engine.mixed_placement()
        if placement['decision'] == 2 and not getattr(engine, '_warned_tickets', False):
            import warnings
            engine._warned_tickets = True
            warnings.warn('serl_amd: the mixed-fault sweep placed its workgroups by tickets, not by the census of CU pairs (%s): the GPU was shared with '
                          'another launch or the instruction-cache pairs of this device are not (1,2)(3,4)(5,6)(7,8); expect ~10 %% less throughput' % placement)
    return PopResult(mixed_placement=placement, fitness=fitness, returns=sh(ret), smoothness=sh(sm), length_steps=sh(ls),
                     length_t=sh(out['length_t'].cpu().numpy()), cost_steps=sh(out['cost_steps'].cpu().numpy()),
                     pop_fitness=pop_fitness, champion=int(np.argmax(pop_fitness)), worst=int(np.argmin(pop_fitness)),
                     kernel_ms=engine.last_kernel_ms,
                     actions=out.get('actions') if traces else None, states=out.get('states') if traces else None,
                     rewards=out.get('rewards') if traces else None, transitions=out.get('transitions'),
                     episode_member=moe)


def validate_pop(actors, refs, *, mode='nominal', t_max=80, spec: Optional[NetSpec] = None,
                 engine: Optional[RolloutEngine] = None, carry_error=False):
    """The `-eval_pop` loop of the reference (base/evaluate.py:236-256 over validate_agent :123-150 over evaluate
    :59-120) in one launch: every actor flies every reference table of `refs` (f64 [num_trails+1, T, 3] rad).

    As in the reference's evaluate(), the error history pairs ref(t_k) with the controlled state BEFORE step k
    (the state env.reset() / the previous step left), the action history is env.last_u before step k (zeros first),
    nMAE = calc_nMAE(errors) and smoothness = calc_smoothness(actions) -- both batched on the device.  Returns
    dict(nmae [R, pop], smoothness [R, pop], nmae_mean / nmae_sd / sm_mean / sm_sd [pop] (the Stats of validate_agent),
    champion = first index of the smallest mean nMAE, last_data [pop] lazily: data(i) -> [T, 19] of actor i's last episode).

    carry_error: the reference runs all episodes on ONE global env whose reset() does not clear the tracking error
    (envs/phlabenv.py:401-428), so obs0 of every episode but the first carries the final error of the episode before it
    (actor-major, reference-minor order).  True reproduces that with a second launch fed the first launch's final errors
    (the final error of an 80 s episode does not depend measurably on its own obs0: one fixed-point pass)."""
    engine = engine or default_engine()
    if spec is None:
        spec = spec_of(actors[0])
    w = _as_weights(actors, spec)
    pop = w.shape[0]
    refs = torch.as_tensor(refs, dtype=torch.float64)
    if refs.dim() == 2:
        refs = refs[None]
    R = refs.shape[0]
    E = pop * R
    moe = np.repeat(np.arange(pop, dtype=np.int32), R)
    build, row = builds.resolve_mode(mode)
    ref_e = refs.to(engine.device).repeat(pop, 1, 1)
    faults = None if row == builds.NOMINAL_ROW else [row] * E

    def fly(err0):
        return engine.rollout(w, spec, moe, ref_e, build=build, faults=faults, t_max=t_max, traces=True, err0=err0)
    out = fly(None)
    if carry_error:
        n = out['length_steps'].to(torch.int64).abs()
        last = (n - 1).clamp(min=0)
        ar = torch.arange(E, device=engine.device)
        fin = ref_e[ar, last] - out['states'][ar, last][:, [7, 6, 5]]          # env.error after each episode's last step
        err0 = torch.cat([torch.zeros(1, 3, dtype=torch.float64, device=engine.device), fin[:-1]], 0)
        out = fly(err0.cpu().numpy())
    dev = out['states'].device
    n = out['length_steps'].to(torch.int64).abs()
    data, _ = builds.load(build)
    x0 = torch.as_tensor(np.asarray(data['x0'][:12], dtype=np.float64), device=dev)
    T = refs.shape[1]
    xb = torch.cat([x0.expand(E, 1, 12), out['states'][:, :T - 1]], 1)            # env.x before step k
    ub = torch.cat([torch.zeros(E, 1, 3, dtype=torch.float64, device=dev), out['actions'][:, :T - 1]], 1)
    err = ref_e - xb[:, :, [7, 6, 5]]
    nm = metrics.calc_nMAE_batch(err, n).cpu().numpy()
    sm = metrics.calc_smoothness(ub, n).cpu().numpy()
    nm, sm = nm.reshape(pop, R).T, sm.reshape(pop, R).T
    res = dict(nmae=nm, smoothness=sm, nmae_mean=nm.mean(0), nmae_sd=nm.std(0), sm_mean=sm.mean(0), sm_sd=sm.std(0),
               length_steps=out['length_steps'].cpu().numpy().reshape(pop, R).T, kernel_ms=engine.last_kernel_ms)
    res['champion'] = int(np.argmin(res['nmae_mean']))          # `if stats.nmae < nmae_min` keeps the first minimum

    def last_data(i):
        """data[T, 19] = (ref 3, last_u 3, x 12, reward) of actor i's LAST episode, the array validate_agent returns"""
        e = i * R + R - 1
        k = int(n[e])
        return torch.cat([ref_e[e, :k], ub[e, :k], xb[e, :k], out['rewards'][e, :k, None]], 1).cpu().numpy()
    res['last_data'] = last_data
    return res


def make_evaluate(args, *, mode='nominal', t_max=20, ref_fn=None, engine=None, replay_buffer=None, counters=None):
    """-> evaluate(agent, is_action_noise, store_transition) -> Episode, the signature of
    Agent.evaluate (base/core/agent.py:63-66).  One episode per call (the batched path is evaluate_pop).

    ref_fn() -> the next episode's reference: a f64 [T,3] radians table or one refsignals.ref_specs row.  Default (None):
    what CitationEnv.reset() does without user_refs (envs/phlabenv.py:316-335) -- a fresh randomised step sequence per
    episode drawn from np.random (refsignals.training_references; the `signals` class behind it is not vendored by the
    reference: parity unpinned), theta trim = rad2deg(theta0) of the build; `ref_fn='base'` flies the fixed base
    evaluation reference (base/evaluate.py:173-180, trim 0.22 deg) every episode.
    Transitions of stored episodes are appended to `replay_buffer`, `agent.buffer` and (cost steps)
    `agent.critical_buffer` exactly as agent.py:101-112 does; `counters` (dict) receives
    num_frames / gen_frames / num_episodes increments."""
    engine = engine or default_engine()
    counters = counters if counters is not None else {}
    env_cfg, incremental = builds.env_config(mode)          # 'PHlab_symmetric_nominal', 'PHlab_full_incremental', ...
    S, A = builds.env_dims(env_cfg, incremental)
    # envs/phlabenv.py never clears self.error between episodes, and the native initialize() never resets the model clock
    state = {'err': np.zeros(3), 'tick': 0}

    def evaluate(agent, is_action_noise: bool, store_transition: bool) -> Episode:
        actor = agent.actor if hasattr(agent, 'actor') else agent
        actor.eval()
        spec = spec_of(actor)
        if ref_fn is None:
            th, ph = refsignals.training_references(1, t_max, np.random, n_actions=A)
            theta0 = float(np.asarray(builds.load(builds.resolve_mode(mode)[0])[0]['x0'])[7])
            # (one action: init_ref keeps the class default 0.22 deg, envs/phlabenv.py:202,304-313)
            ref = refsignals.tabulate(th[0], ph[0], t_max, theta_trim_deg=float(np.rad2deg(theta0)) if A == 3 else 0.22)
        elif isinstance(ref_fn, str) and ref_fn == 'base':
            ref = refsignals.tabulate(*refsignals.base_reference(t_max), t_max)
        else:
            ref = ref_fn()
        if isinstance(ref, np.ndarray) and ref.dtype.names:       # a spec row: tabulate on the host for the Episode record too
            ref_table = refsignals.tabulate_specs(ref.reshape(-1)[:1], t_max)[0]
        else:
            ref_table = np.asarray(ref, dtype=np.float64)
        T = ref_table.shape[0]
        # agent.py:90-93 / envs/noise/citation.py:71-82: this episode's np.random draws, pre-drawn in the reference's
        # interleaved per-step order; the generator is re-synchronised to the steps actually taken afterwards
        za, sn, resync = builds.draw_episode_noise(T, bool(is_action_noise), builds.has_sensor_noise(mode), np.random, n_actions=A)
        noise = None
        if za is not None:
            noise = np.zeros((1, T, 3))
            noise[0, :, :A] = np.clip(args.noise_sd * za, -args.noise_clip, args.noise_clip)
        build, row = builds.resolve_mode(mode)
        out = engine.rollout(pack_population([actor]), spec, [0], ref, build=build, sensor_noise=None if sn is None else sn[None],
                             faults=None if row == builds.NOMINAL_ROW else [row], err0=state['err'][None], tick0=[state['tick']],
                             action_noise=noise, t_max=t_max, traces=True, transitions=store_transition,
                             env_config=env_cfg, incremental=incremental)
        resync(abs(int(out['length_steps'][0])))
        n = int(out['length_steps'][0])
        state['tick'] += abs(n) + 1          # one step in reset() + n env steps
        actions = out['actions'][0, :n, :A].cpu().numpy()          # env.last_u per step (A columns)
        rewards = out['rewards'][0, :n].cpu().numpy()
        states = out['states'][0, :n].cpu().numpy()
        state['err'][:A] = (ref_table[n - 1] - states[n - 1][[7, 6, 5]])[:A]
        if store_transition:
            from . import replay
            replay.store_episodes(engine, out['transitions'], [(agent, 0, n, int(out['cost_steps'][0]))], replay_buffer, counters,
                                  state_dim=S, action_dim=A)
        smooth = float(metrics.calc_smoothness(actions[None], [n])[0])
        fitness = float(np.sum(rewards))
        if getattr(args, 'smooth_fitness', False):
            fitness += smooth
        return Episode(fitness=fitness, smoothness=smooth, length=float(out['length_t'][0]),
                       state_history=[] if store_transition else list(states), ref_signals=ref_table[n - 1][:A],
                       actions=actions, reward_lst=list(rewards))

    return evaluate
