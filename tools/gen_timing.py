#!/usr/bin/env python3
"""Wall time of ONE training generation's evaluation stage as a SERL user would run it through the drop-in API
(serl_amd.evaluate_generation + validate_actor: pop=50 x num_evals=3 GA episodes + the RL actor's exploration episode in one
launch, the champion's and the RL actor's validation batches), against the kernel time inside it: what the host adds.
    python tools/gen_timing.py [generations] [--profile]
Prints one JSON line; the reference's Python needs ~1.2 s per 20 s episode per core (profiles/r02_reference_cpu.json)."""
import sys, os, time, json, types, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import serl_amd
from serl_amd import refsignals, actor as A
from serl_amd.replay import DeviceReplay

G = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
args = types.SimpleNamespace(state_dim=7, action_dim=3, hidden_size=32, num_layers=3, activation_actor='tanh', num_evals=3,
                             smooth_fitness=True, noise_sd=0.2962183114680794, noise_clip=0.5)
engine = serl_amd.RolloutEngine(0)
w = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'actors.npz'))['serl50']
shared = DeviceReplay(800_000, engine.device, engine)
mk = lambda: types.SimpleNamespace(buffer=DeviceReplay(8000, engine.device, engine), critical_buffer=DeviceReplay(8000, engine.device, engine))
pop = []
for m in range(50):
    ag = mk(); ag.actor = serl_amd.Actor(args); A.unpack_into(ag.actor, torch.from_numpy(w[m])); pop.append(ag)
rl = mk(); rl.actor = serl_amd.Actor(args); A.unpack_into(rl.actor, torch.from_numpy(w[7]))
E = 50 * 3 + 1
rng = np.random.RandomState(0)
counters = {}


def generation():
    t = {}
    t0 = time.perf_counter()
    refs = refsignals.ref_specs(*refsignals.training_references(E, 20, rng), 0.2106)
    t['refs'] = time.perf_counter() - t0; t0 = time.perf_counter()
    g = serl_amd.evaluate_generation(pop, rl, args=args, t_max=20, refs=refs, engine=engine, replay_buffer=shared, counters=counters)
    torch.cuda.synchronize()
    t['evaluate_generation'] = time.perf_counter() - t0; t['kernel_generation'] = g.kernel_ms / 1e3; t0 = time.perf_counter()
    champ = pop[g.pop.champion]
    v1 = serl_amd.validate_actor(champ, tests=5, t_max=20, engine=engine)
    t['validate_champion'] = time.perf_counter() - t0; t['kernel_validate'] = engine.last_kernel_ms / 1e3; t0 = time.perf_counter()
    v2 = serl_amd.validate_actor(rl, tests=5, t_max=20, engine=engine)
    torch.cuda.synchronize()
    t['validate_rl'] = time.perf_counter() - t0
    return t


class _Critic(torch.nn.Module):
    """stand-in for TD3's twin critic (base/core/td3.py:17-85)"""

    def __init__(self):
        super().__init__()
        self.q1 = torch.nn.Sequential(torch.nn.Linear(10, 32), torch.nn.ELU(), torch.nn.Linear(32, 1))
        self.q2 = torch.nn.Sequential(torch.nn.Linear(10, 32), torch.nn.ELU(), torch.nn.Linear(32, 1))

    def forward(self, s, a):
        x = torch.cat([s, a], -1)
        return self.q1(x), self.q2(x)


def epoch(fitness, sync=True):
    """the reference's DEFAULT SSNE epoch (proximal mutation, distillation crossover, distance-sorted groups;
    base/parameters.py:110-115) on the packed population with the generation's device rings"""
    import random
    from serl_amd import ssne
    eargs = types.SimpleNamespace(pop_size=50, elite_fraction=0.2, mutation_prob=0.9, mutation_mag=0.0247682869654, mut_type='proximal',
                                  distil_crossover=True, distil_type='distance', crossover_prob=0.0, mutation_batch_size=86,
                                  individual_bs=8000)
    t0 = time.perf_counter()
    wts = serl_amd.pack_population([a.actor for a in pop], device=engine.device)
    s = ssne.SSNE(eargs, engine, serl_amd.evaluator.spec_of(pop[0].actor), critic=critic)
    s.epoch(wts, fitness, buffers=[a.buffer for a in pop], critical=[a.critical_buffer for a in pop])
    if sync:
        torch.cuda.synchronize()
    else:
        torch.cuda.current_stream().synchronize()      # (the caller's stream only: a validation batch may still be in flight on a side stream)
    return time.perf_counter() - t0


def generation_overlapped(state, fit_for_epoch=None):
    """The same generation with the validation batches in flight beside other work (validate_actor(wait=False)): the champion's batch beside the
    SSNE epoch, the RL actor's beside the NEXT generation's population launch (its handle is `state['rl']`, waited for behind that launch)."""
    t = {}
    t0 = time.perf_counter()
    refs = refsignals.ref_specs(*refsignals.training_references(E, 20, rng), 0.2106)
    t['refs'] = time.perf_counter() - t0; t0 = time.perf_counter()
    g = serl_amd.evaluate_generation(pop, rl, args=args, t_max=20, refs=refs, engine=engine, replay_buffer=shared, counters=counters)
    torch.cuda.current_stream().synchronize()
    t['evaluate_generation'] = time.perf_counter() - t0; t['kernel_generation'] = g.kernel_ms / 1e3; t0 = time.perf_counter()
    if state.get('rl') is not None:
        state['rl'].result()                       # the previous generation's RL validation: flew beside the launch above
        t['stream_ms_validate_rl'] = state['rl'].stream_ms / 1e3
    t['wait_validate_rl'] = time.perf_counter() - t0; t0 = time.perf_counter()
    h1 = serl_amd.validate_actor(pop[g.pop.champion], tests=5, t_max=20, engine=engine, wait=False)
    t['launch_validate_champion'] = time.perf_counter() - t0; t0 = time.perf_counter()
    if fit_for_epoch is not None:
        t['epoch'] = epoch(fit_for_epoch, sync=False)
        t0 = time.perf_counter()
    h1.result()
    t['wait_validate_champion'] = time.perf_counter() - t0; t['stream_ms_validate_champion'] = h1.stream_ms / 1e3; t0 = time.perf_counter()
    state['rl'] = serl_amd.validate_actor(rl, tests=5, t_max=20, engine=engine, wait=False)      # (behind the TD3 update in a training loop)
    t['launch_validate_rl'] = time.perf_counter() - t0
    return t


critic = _Critic().to(engine.device)
generation()      # warm-up (module loads, rocFFT plans)
if '--profile' in sys.argv:
    pr = cProfile.Profile(); pr.enable(); generation(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35); print(s.getvalue()[:6000])
ts = [generation() for _ in range(G)]
ep_ms = []
if '--epoch' in sys.argv:
    fit = np.random.default_rng(1).normal(-150, 50, 50)
    epoch(fit)
    if '--profile-epoch' in sys.argv:
        pr = cProfile.Profile(); pr.enable(); epoch(fit); pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
    ep_ms = [epoch(fit) * 1e3 for _ in range(3)]
mean = {k: float(np.mean([t[k] for t in ts])) * 1e3 for k in ts[0]}
total = mean['refs'] + mean['evaluate_generation'] + mean['validate_champion'] + mean['validate_rl']
steps = int(counters['num_frames'])
# the same generations with the validation batches in flight (steady state: every generation waits for the previous one's RL validation)
state = {}
fit = np.random.default_rng(1).normal(-150, 50, 50) if '--epoch' in sys.argv else None
generation_overlapped(state, fit)
t_all = time.perf_counter()
to = [generation_overlapped(state, fit) for _ in range(G)]
wall_overlapped = (time.perf_counter() - t_all) / G * 1e3
state['rl'].result()
mo = {k: float(np.mean([t[k] for t in to if k in t])) * 1e3 for k in to[-1]}
stage_overlapped = mo['refs'] + mo['evaluate_generation'] + mo['wait_validate_rl'] + mo['launch_validate_champion'] + mo['wait_validate_champion'] + mo['launch_validate_rl']
print(json.dumps(dict(what='one generation: 151 + 5 + 5 episodes of 20 s (2 001 steps), drop-in API, ms', serial=dict({k: round(v, 2) for k, v in mean.items()},
                      evaluation_stage_ms=round(total, 2), kernel_ms=round(mean['kernel_generation'] + 2 * mean['kernel_validate'], 2),
                      ssne_default_epoch_ms=[round(v, 1) for v in ep_ms]),
                      overlapped=dict({k: round(v, 2) for k, v in mo.items()}, evaluation_stage_ms=round(stage_overlapped - (0 if fit is None else 0.0), 2),
                                      evaluation_stage_minus_kernel_generation_ms=round(stage_overlapped - mo['kernel_generation'], 2),
                                      generation_wall_ms_with_epoch=round(wall_overlapped, 2) if fit is not None else None,
                                      note='validate_actor(wait=False): the champion\'s batch flies beside the SSNE epoch, the RL actor\'s beside the next generation\'s population launch; '
                                           'wait_* = what result() still waited, stream_ms_* = the batch on its side stream (rollout + smoothness + copies)'),
                      stored_frames=steps)))
