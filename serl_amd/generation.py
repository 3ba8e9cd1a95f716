"""One ERL generation's rollouts in as few launches as their data dependencies allow.

The reference's `Agent.train` (base/core/agent.py:211-315) runs, one episode at a time on one env:

  1. pop x num_evals GA episodes, the last one of every member stored          agent.py:234-241
  2. 5 validation episodes of the champion (argmax of the mean fitness)          agent.py:255-259, 188-209
  3. SSNE.epoch                                                                  agent.py:264
  4. one exploration episode of the RL actor (action noise, stored)              agent.py:269
  5. TD3 updates, then 5 validation episodes of the updated RL actor             agent.py:271-275

(1) and (4) depend on nothing computed in this generation, so they share ONE launch here
(`evaluate_generation`: pop*num_evals + 1 episodes, the RL actor as an extra member row, its pre-drawn noise
as the only row of the noise table); (2) needs the argmax of (1) and (5) needs the weights TD3 produced, so
each is a launch of its own (`validate_actor`: all validation episodes of an actor together).  Three launches
per generation instead of pop*num_evals + 11 sequential episodes.

Replay-buffer semantics follow `Agent.evaluate` (agent.py:101-125): every stored transition goes to the shared
buffer and to the acting agent's own buffer, cost-flagged ones also to its critical buffer; `num_frames`,
`gen_frames` and `num_episodes` advance for stored episodes only.  Episodes start from a fresh env state
(error carry-over and model clock: see `evaluate_pop`); the reference's per-step `np.random.randn` draws of the
exploration noise are replaced by one up-front draw of T rows in the same order (the values coincide while the
episode runs; the generator is left T - n rows further on when it stops early).
"""
from dataclasses import dataclass
from typing import Optional, Sequence
import numpy as np
import torch

from . import builds, metrics, refsignals
from .actor import pack_population, spec_of
from .episode import Episode
from .evaluator import PopResult, RolloutEngine, default_engine


@dataclass
class GenerationResult:
    pop: PopResult                    # the GA evaluate loop (agent.py:229-256)
    rl_episode: Optional[Episode]     # the RL actor's exploration episode (agent.py:269), None without an RL actor
    kernel_ms: float
    stored: Optional[list] = None     # (_defer_store) [(agent, episode index, steps, cost steps)] of the episodes to store
    staged: Optional[torch.Tensor] = None   # (_defer_store) the staged transition rows [episodes, T, 20]


def store_transitions(rows, agent, replay_buffer=None, counters=None, engine=None):
    """rows: f32 [n, 20] = (obs7, a3, next_obs7, r, done, cost) of ONE stored episode (device tensor or array) -> the
    buffers of agent.py:101-112 and the counters of agent.py:111-125 (see replay.store_episodes)."""
    from . import replay
    rows = torch.as_tensor(rows)
    n = rows.shape[0]
    nc = int((rows[:, 19] != 0).sum())
    if rows.is_cuda:
        replay.store_episodes(engine or default_engine(), rows[None].contiguous(), [(agent, 0, n, nc)], replay_buffer, counters)
    else:                          # host rows can only feed host-side buffers (objects with add(*transition))
        replay.store_episodes(None, rows[None], [(agent, 0, n, nc)], replay_buffer, counters)


def _actor_of(agent):
    return agent.actor if hasattr(agent, 'actor') else agent


def _episode(out, e, ref_row, smooth, smooth_fitness):
    n = abs(int(out['length_steps'][e]))
    rewards = out['rewards'][e, :n].cpu().numpy()
    fitness = float(np.sum(rewards))
    if smooth_fitness:
        fitness += float(smooth)
    return Episode(fitness=fitness, smoothness=float(smooth), length=float(out['length_t'][e]),
                   state_history=list(out['states'][e, :n].cpu().numpy()), ref_signals=np.asarray(ref_row[n - 1]),
                   actions=out['actions'][e, :n].cpu().numpy(), reward_lst=list(rewards))


def evaluate_generation(pop: Sequence, rl_agent=None, *, args, mode='nominal', t_max=20, refs=None, rl_noise=None,
                        engine: Optional[RolloutEngine] = None, replay_buffer=None, counters=None,
                        store: bool = True, env_config=0, incremental=False, _defer_store: bool = False) -> GenerationResult:
    """Steps (1) and (4) of a generation in one launch.

    pop      : GeneticAgent / Actor sequence (one network shape); rl_agent: the RL learner's agent or None
    args     : needs num_evals, smooth_fitness, noise_sd, noise_clip (base/parameters.py)
    refs     : f64 [pop*num_evals (+1), T, 3] / [T, 3] radians (refsignals.tabulate), or refsignals.ref_specs rows
               [pop*num_evals (+1)] / [1] (generated in the kernel); None = base reference
    rl_noise : f64 [T, 3] clipped exploration noise; None = drawn here as agent.py:90-93 would
    store    : append the transitions of every member's last evaluation and of the RL episode to the buffers
    env_config / incremental : builds.env_config(name) for the other env configurations (the attitude task by default; the
               device replay rings hold the attitude task's rows only: use list-backed buffers with the others)
    _defer_store : (evaluate_generation_sharded) stage the transitions but leave the buffers alone; the list of stored
               episodes and the staged rows come back as result.stored / result.staged"""
    engine = engine or default_engine()
    ne = int(args.num_evals)
    actors = [_actor_of(a) for a in pop]
    for a in actors:
        if a.training:
            a.eval()                                        # agent.py:83  (Module.eval() walks every submodule: skipped when the flag is down already)
    n_pop = len(actors)
    if n_pop == 0 and rl_agent is None:
        # (evaluate_generation_sharded: a rank whose member block is empty and no RL episode to fly -- nothing to launch)
        z = np.zeros((ne, 0))
        # the staged rows must have the T every other rank stages (gather_stored_episodes sizes its all_gather from the local T): the table's
        # when references are tables, the clock's when they are specs or the base reference -- the same rule as the normal path below
        if refs is not None and not (isinstance(refs, np.ndarray) and refs.dtype.names is not None):
            rt = torch.as_tensor(refs)
            assert rt.dim() in (2, 3) and rt.shape[-1] == 3, 'reference tables are [T, 3] or [episodes, T, 3]'
            T0 = int(rt.shape[-2])
        else:
            T0 = refsignals.n_steps_for(t_max)
        res = PopResult(fitness=z, returns=z, smoothness=z, length_steps=z.astype(np.int32), length_t=z, cost_steps=z.astype(np.int32),
                        pop_fitness=np.zeros(0), champion=-1, worst=-1, kernel_ms=0.0, episode_member=np.zeros(0, np.int32))
        g = GenerationResult(pop=res, rl_episode=None, kernel_ms=0.0)
        g.stored, g.staged = [], torch.zeros(0, T0, 2 * builds.env_dims(env_config, incremental)[0] + builds.env_dims(env_config, incremental)[1] + 3)
        return g
    spec = spec_of(actors[0] if actors else _actor_of(rl_agent))
    members = list(actors)
    E = n_pop * ne
    moe = np.repeat(np.arange(n_pop, dtype=np.int32), ne)
    noise_row = None
    if rl_agent is not None:
        rl_actor = _actor_of(rl_agent)
        rl_actor.eval()
        assert spec_of(rl_actor) == spec, 'the RL actor must have the population\'s network shape'
        members.append(rl_actor)
        moe = np.append(moe, np.int32(n_pop))
    Etot = len(moe)
    if refs is None:
        refs = refsignals.tabulate(*refsignals.base_reference(t_max), t_max)
    generated = isinstance(refs, np.ndarray) and refs.dtype.names is not None      # refsignals.ref_specs rows: generated in the kernel
    if generated:
        assert len(refs) in (1, Etot), 'one reference spec per episode (pop*num_evals%s)' % (' + 1' if rl_agent is not None else '')
        T = refsignals.n_steps_for(t_max)
    else:
        refs = torch.as_tensor(refs, dtype=torch.float64)
        T = refs.shape[-2]
        if refs.dim() == 3:
            assert refs.shape[0] == Etot, 'one reference table per episode (pop*num_evals%s)' % (' + 1' if rl_agent is not None else '')
    noise = None
    if rl_agent is not None:
        if rl_noise is None:
            rl_noise = np.clip(args.noise_sd * np.random.randn(T, 3), -args.noise_clip, args.noise_clip)
        noise = np.asarray(rl_noise, dtype=np.float64).reshape(1, T, 3)
        noise_row = np.full(Etot, -1, dtype=np.int32)
        noise_row[-1] = 0
    build, row = builds.resolve_mode(mode)
    faults = None if row == builds.NOMINAL_ROW else [row] * Etot
    out = engine.rollout(pack_population(members), spec, moe, refs, build=build, faults=faults, action_noise=noise,
                         noise_row=noise_row, t_max=t_max, traces=True, transitions=store, env_config=env_config, incremental=incremental)
    ls = out['length_steps'].cpu().numpy()
    sm = metrics.calc_smoothness(out['actions'], np.abs(ls)).cpu().numpy()
    ret = out['fitness'].cpu().numpy()
    smooth_fitness = bool(getattr(args, 'smooth_fitness', False))
    fit = ret + sm if smooth_fitness else ret.copy()
    sh = lambda a: np.ascontiguousarray(np.asarray(a)[:E].reshape(n_pop, ne).T)
    fitness = sh(fit)
    pop_fitness = np.mean(fitness, axis=0)
    res = PopResult(fitness=fitness, returns=sh(ret), smoothness=sh(sm), length_steps=sh(ls),
                    length_t=sh(out['length_t'].cpu().numpy()), cost_steps=sh(out['cost_steps'].cpu().numpy()),
                    pop_fitness=pop_fitness, champion=int(np.argmax(pop_fitness)) if n_pop else -1, worst=int(np.argmin(pop_fitness)) if n_pop else -1,
                    kernel_ms=engine.last_kernel_ms, actions=out['actions'][:E], states=out['states'][:E],
                    rewards=out['rewards'][:E], transitions=out.get('transitions'), episode_member=moe[:E])
    cs = out['cost_steps'].cpu().numpy()
    stored = []
    if store:
        for m in range(n_pop):                              # the reference's order: member by member (agent.py:234-241)
            e = m * ne + ne - 1
            stored.append((pop[m], e, abs(int(ls[e])), int(cs[e])))
    rl_ep = None
    if rl_agent is not None:
        e = Etot - 1
        if generated:
            ref_row = refsignals.tabulate_specs(refs[e:e + 1] if len(refs) > 1 else refs[:1], t_max)[0]
        else:
            ref_row = (refs[e] if refs.dim() == 3 else refs).cpu().numpy()
        rl_ep = _episode(out, e, ref_row, sm[e], smooth_fitness)
        if store:
            stored.append((rl_agent, e, abs(int(ls[e])), int(cs[e])))
    if stored and not _defer_store:
        from . import replay
        replay.store_episodes(engine, out['transitions'], stored, replay_buffer, counters, state_dim=spec.state_dim, action_dim=spec.action_dim)
    g = GenerationResult(pop=res, rl_episode=rl_ep, kernel_ms=engine.last_kernel_ms)
    if _defer_store:
        g.stored, g.staged = stored, out.get('transitions')
    return g


def evaluate_generation_sharded(pop: Sequence, rl_agent=None, *, args, mode='nominal', t_max=20, refs=None, rl_noise=None,
                                engine: Optional[RolloutEngine] = None, replay_buffer=None, counters=None,
                                store: bool = True, env_config=0, incremental=False) -> GenerationResult:
    """`evaluate_generation` with the population sharded over the ranks of the default process group (one process per GPU,
    `torch.distributed` over RCCL): rank r evaluates the contiguous member block `distributed.member_block` gives
    it, ONE all_gather brings every member's result rows to every rank, ONE more the rows of every member's stored episode
    (`distributed.gather_stored_episodes`) -- after which every rank appends all `pop` episodes to its buffers in member
    order, exactly like the single-process path, so the replicated SSNE epoch finds identical replay rings everywhere.
    The RL actor's exploration episode (agent.py:269) is flown by EVERY rank (same weights, same pre-drawn noise: the host
    RNG streams are replicated), which keeps its episode and its buffer entries local.  Without a process group (or with one
    rank) this IS evaluate_generation.  The per-step traces of the population (actions / states / rewards) stay on the rank
    that flew them: `result.pop.actions` etc. are None here.

    refs: None, one [T, 3] table / one refsignals.ref_specs row, or [pop*num_evals (+1), T, 3] tables / [pop*num_evals (+1)]
    spec rows for ALL members (every rank passes the same array and uses its block's rows).  A rank whose member block is empty
    (pop < world size, or the last ranks of a population that does not divide it: distributed.member_block) flies the RL episode
    only -- or nothing -- and contributes zero rows to both gathers."""
    import torch.distributed as dist
    from . import distributed as sd, replay
    ws = dist.get_world_size() if dist.is_initialized() else 1
    if ws == 1:
        return evaluate_generation(pop, rl_agent, args=args, mode=mode, t_max=t_max, refs=refs, rl_noise=rl_noise, engine=engine,
                                   replay_buffer=replay_buffer, counters=counters, store=store, env_config=env_config, incremental=incremental)
    engine = engine or default_engine()
    rk = dist.get_rank()
    ne, n_pop = int(args.num_evals), len(pop)
    lo, hi = sd.member_block(n_pop, ws, rk)
    if rl_agent is not None and rl_noise is None:
        T_ = refsignals.n_steps_for(t_max)
        rl_noise = np.clip(args.noise_sd * np.random.randn(T_, 3), -args.noise_clip, args.noise_clip)     # the same draw on every rank
    local_refs = refs
    rows_ = list(range(lo * ne, hi * ne)) + ([n_pop * ne] if rl_agent is not None else [])
    if refs is not None and isinstance(refs, np.ndarray) and refs.dtype.names is not None:
        if len(refs) > 1:                      # refsignals.ref_specs rows, one per episode: this block's (+ the RL episode's)
            assert len(refs) == n_pop * ne + (1 if rl_agent is not None else 0), 'one reference spec per episode of the WHOLE population'
            local_refs = refs[rows_] if rows_ else refs[:1]
    elif refs is not None:
        r_ = torch.as_tensor(refs, dtype=torch.float64)
        if r_.dim() == 3:
            local_refs = r_[rows_] if rows_ else r_[:1]
    g = evaluate_generation(pop[lo:hi], rl_agent, args=args, mode=mode, t_max=t_max, refs=local_refs, rl_noise=rl_noise,
                            engine=engine, store=store, env_config=env_config, incremental=incremental, _defer_store=True)
    dev = engine.device
    p = g.pop
    rows = np.stack([p.fitness, p.returns, p.smoothness, p.length_t, p.length_steps.astype(np.float64), p.cost_steps.astype(np.float64)], -1)
    allr = sd.gather_rows(torch.as_tensor(rows).to(dev), n_pop, ws, rk, device=dev).cpu().numpy()
    fitness = allr[..., 0]
    pf = np.mean(fitness, axis=0)
    res = PopResult(fitness=fitness, returns=allr[..., 1], smoothness=allr[..., 2], length_steps=allr[..., 4].astype(np.int32),
                    length_t=allr[..., 3], cost_steps=allr[..., 5].astype(np.int32), pop_fitness=pf, champion=int(np.argmax(pf)),
                    worst=int(np.argmin(pf)), kernel_ms=p.kernel_ms, episode_member=np.repeat(np.arange(n_pop, dtype=np.int32), ne))
    if store:
        n_loc = hi - lo
        idx = [e for (_, e, _, _) in g.stored[:n_loc]]
        staged = g.staged[idx] if n_loc else g.staged[:0]          # (an empty block: zero rows of the right [T, width] shape)
        steps = [n for (_, _, n, _) in g.stored[:n_loc]]
        cost = [c for (_, _, _, c) in g.stored[:n_loc]]
        allb, asteps, acost = sd.gather_stored_episodes(torch.as_tensor(staged).to(dev), steps, cost, n_pop, ws, rk)
        items = [(pop[m], m, int(asteps[m]), int(acost[m])) for m in range(n_pop)]
        if rl_agent is not None:
            a_, e_, n_, c_ = g.stored[-1]
            allb = torch.cat([allb, torch.as_tensor(g.staged[e_:e_ + 1]).to(allb.device)])
            items.append((rl_agent, n_pop, n_, c_))
        replay.store_episodes(engine, allb, items, replay_buffer, counters)
    return GenerationResult(pop=res, rl_episode=g.rl_episode, kernel_ms=g.kernel_ms)


class ValidationHandle:
    """A validation batch in flight on a side stream (validate_actor(wait=False)).  `result()` waits for ITS launch only and returns
    validate_actor's tuple; `done()` polls.  Everything the tuple needs was copied to pinned host memory behind the kernel on the same stream."""

    def __init__(self, event, host, refs, tests, smooth_fitness, keep, t0_event):
        self._event, self._host, self._refs, self._tests, self._sf, self._keep, self._t0 = event, host, refs, tests, smooth_fitness, keep, t0_event
        self._res = None

    def done(self):
        return self._res is not None or self._event.query()

    def result(self):
        if self._res is None:
            self._event.synchronize()
            h, tests = self._host, self._tests
            self.stream_ms = float(self._t0.elapsed_time(self._event))      # rollout + smoothness + copies, as the side stream saw them
            lsr = h['length_steps'].numpy()
            if (lsr < 0).any():
                raise RuntimeError('reference table too short: %d validation episodes were still running' % int((lsr < 0).sum()))
            ls = np.abs(lsr)
            sm = h['smoothness'].numpy()
            rew = h['rewards'].numpy()
            scores = np.array([float(np.sum(rew[e, :ls[e]])) for e in range(tests)])
            lengths = h['length_t'].numpy()
            e = tests - 1
            n = int(ls[e])
            fitness = float(np.sum(rew[e, :n])) + (float(sm[e]) if self._sf else 0.0)
            ref_row = self._refs[e] if self._refs.ndim == 3 else self._refs
            last = Episode(fitness=fitness, smoothness=float(sm[e]), length=float(lengths[e]), state_history=list(h['states_last'].numpy()[:n]),
                           ref_signals=np.asarray(ref_row[n - 1]), actions=h['actions_last'].numpy()[:n].copy(), reward_lst=list(rew[e, :n]))
            self._res = (float(np.mean(scores)), float(np.std(scores)), float(np.mean(lengths)), float(np.std(lengths)), last,
                         float(np.median(sm)), float(np.std(sm)))
            self._keep = None
        return self._res


def validate_actor(agent, *, tests=5, mode='nominal', t_max=20, refs=None, smooth_fitness=False,
                   engine: Optional[RolloutEngine] = None, wait: bool = True, stream: Optional[torch.cuda.Stream] = None):
    """`Agent.validate_agent` (agent.py:188-209): `tests` episodes of one actor, nothing stored, all in one launch.
    Returns the reference's tuple (test_score, test_sd, ep_len, ep_len_sd, last_episode, sm, sm_sd):
    mean / std of sum(reward_lst), mean / std of the episode length, the last episode, median / std of smoothness.

    wait=False (SURVEY 8f-4 for real): the batch -- five one-episode teams, 2 % of the GPU -- is launched on a SIDE stream of the engine and a
    `ValidationHandle` comes back at once; its results feed statistics only (agent.py:258-259, 274-275), so the champion's validation runs beside the
    SSNE epoch and the RL actor's beside the next generation's population launch.  The side stream first waits for what the caller's stream has
    enqueued (the TD3 update that produced the weights); smoothness and the copies of the results follow the kernel on the same stream, nothing
    synchronises with the host until `handle.result()`.  Host RNG: this function draws nothing -- references are the caller's (`refs`), drawn when the
    reference draws them -- so the order of the reference's streams is the caller's call order, wait or not."""
    engine = engine or default_engine()
    actor = _actor_of(agent)
    actor.eval()
    spec = spec_of(actor)
    if refs is None:
        refs = refsignals.tabulate(*refsignals.base_reference(t_max), t_max)
    refs = torch.as_tensor(refs, dtype=torch.float64)
    if refs.dim() == 3:
        assert refs.shape[0] == tests
    build, row = builds.resolve_mode(mode)
    if not wait:
        side = stream or engine.side_stream(2)
        side.wait_stream(torch.cuda.current_stream(engine.device))
        with torch.cuda.stream(side):
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(side)
            # beside other launches: told that the rest of the GPU is taken (the one pair of timing events of the context stays the main stream's)
            out = engine.rollout(pack_population([actor]), spec, np.zeros(tests, dtype=np.int32), refs, build=build,
                                 faults=None if row == builds.NOMINAL_ROW else [row] * tests, t_max=t_max, traces=True, sync=False,
                                 concurrent_episodes=max(1, engine.num_cus - tests))
            sm = metrics.calc_smoothness_enqueued(out['actions'], out['length_steps'])
            if sm is None:
                raise RuntimeError('validate_actor(wait=False): the smoothness kernel does not take traces of %d steps' % out['actions'].shape[1])
            e = tests - 1
            host = {}
            for name, src in (('length_steps', out['length_steps']), ('length_t', out['length_t']), ('smoothness', sm), ('rewards', out['rewards']),
                              ('states_last', out['states'][e]), ('actions_last', out['actions'][e])):
                host[name] = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
                host[name].copy_(src, non_blocking=True)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(side)
        return ValidationHandle(ev, host, refs.cpu().numpy(), tests, smooth_fitness, (out, sm), t0)
    out = engine.rollout(pack_population([actor]), spec, np.zeros(tests, dtype=np.int32), refs, build=build,
                         faults=None if row == builds.NOMINAL_ROW else [row] * tests, t_max=t_max, traces=True)
    ls = np.abs(out['length_steps'].cpu().numpy())
    sm = metrics.calc_smoothness(out['actions'], ls).cpu().numpy()
    scores = np.array([float(np.sum(out['rewards'][e, :ls[e]].cpu().numpy())) for e in range(tests)])
    lengths = out['length_t'].cpu().numpy()
    e = tests - 1
    last = _episode(out, e, (refs[e] if refs.dim() == 3 else refs).cpu().numpy(), sm[e], smooth_fitness)
    return (float(np.mean(scores)), float(np.std(scores)), float(np.mean(lengths)), float(np.std(lengths)), last,
            float(np.median(sm)), float(np.std(sm)))
