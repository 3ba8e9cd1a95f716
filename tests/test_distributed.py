"""Multi-GPU path on CPU: two processes (gloo, 127.0.0.1), members sharded in contiguous blocks, one all-gather of
the per-member result rows (serl_amd/distributed.py; RCCL on the GPU box).  The gathered result must be identical
on every rank and identical to the single-process evaluation, for a population that does not divide the world size.
The local evaluation is the CPU oracle here (test infrastructure) -- the collective logic is what is under test."""
import os, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NET32 = dict(state_dim=7, action_dim=3, hidden=32, num_layers=3, activation='tanh')
POP, NE, TMAX = 5, 2, 20


def _local_eval(lo, hi):
    from oracle import rollout as R
    from serl_amd import refsignals
    w = np.load(os.path.join(ROOT, 'tests', 'golden', 'actors.npz'))['serl50'][:POP]
    ref_all = refsignals.synthetic_reference_tables(POP * NE, NE, TMAX, seed=7)
    moe = np.repeat(np.arange(lo, hi, dtype=np.int32), NE)
    o = R.rollout(w, NET32, moe, ref_all[lo * NE:hi * NE], t_max=TMAX)
    sh = lambda a: np.ascontiguousarray(np.asarray(a).reshape(hi - lo, NE).T)
    return dict(fitness=sh(o['fitness']), returns=sh(o['fitness']), smoothness=sh(np.zeros(len(moe))),
                length_t=sh(o['length_t']), length_steps=sh(o['length_steps']), cost_steps=sh(o['cost_steps']))


def _stored(lo, hi):
    """stand-in for the rows the rollout kernel stages for members lo..hi-1: deterministic per member"""
    rs = [np.random.RandomState(100 + m) for m in range(lo, hi)]
    steps = np.array([40 + 7 * (m % 7) for m in range(lo, hi)], dtype=np.int64)
    st = np.zeros((hi - lo, 90, 20), np.float32)
    for j, r in enumerate(rs):
        st[j, :steps[j]] = r.randn(steps[j], 20).astype(np.float32)
        st[j, :steps[j], 19] = (r.rand(steps[j]) < 0.3)
    cost = np.array([int(st[j, :steps[j], 19].sum()) for j in range(hi - lo)], dtype=np.int64)
    return st, steps, cost


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from serl_amd import distributed as sd
    res = sd.evaluate_pop_sharded(_local_eval, POP, NE)
    lo, hi = sd.member_block(POP, world, rank)
    st, steps, cost = _stored(lo, hi)
    gs, gsteps, gcost = sd.gather_stored_episodes(torch.from_numpy(st), steps, cost, POP, world, rank)
    res['stored'] = (gs.numpy().copy(), gsteps.copy(), gcost.copy())
    q.put((rank, {k: (v if not isinstance(v, np.ndarray) else v.copy()) for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gather_equals_single_process():
    sys.path.insert(0, ROOT)
    from serl_amd import distributed as sd
    single = sd.evaluate_pop_sharded(_local_eval, POP, NE)       # no process group: world size 1
    assert single['block'] == (0, POP)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0]['block'] == (0, 3) and got[1]['block'] == (3, 5)
    for r in (0, 1):
        for k in ('fitness', 'returns', 'length_t', 'length_steps', 'cost_steps', 'pop_fitness'):
            np.testing.assert_array_equal(got[r][k], single[k], err_msg='rank %d %s' % (r, k))
        assert got[r]['champion'] == single['champion'] and got[r]['worst'] == single['worst']
    # the stored episodes of ALL members on every rank (what the replicated SSNE epoch reads its replay rings from)
    st, steps, cost = _stored(0, POP)
    for r in (0, 1):
        gs, gsteps, gcost = got[r]['stored']
        np.testing.assert_array_equal(gs, st); np.testing.assert_array_equal(gsteps, steps); np.testing.assert_array_equal(gcost, cost)
    from serl_amd.replay import DeviceReplay
    rings = [DeviceReplay(64, 'cpu') for _ in range(POP)]
    for m in range(POP):          # (CPU rings: plain torch indexing; on the GPU replay.store_episodes does this in one launch)
        rings[m].append_rows(torch.from_numpy(got[0]['stored'][0][m, :steps[m]]))
        assert len(rings[m]) == min(64, steps[m]) and rings[m].position == steps[m] % 64


def _fake_eval(pop, ne):
    """deterministic rows per (member, eval) without flying anything: the collective logic is what is under test"""
    def ev(lo, hi):
        m = np.arange(lo, hi, dtype=np.float64)[None, :]
        k = np.arange(ne, dtype=np.float64)[:, None]
        fit = -np.abs(np.sin(0.37 * m + 1.3 * k)) * 100.0 - 0.01 * m
        return dict(fitness=fit, returns=fit + 0.5, smoothness=np.cos(m + k), length_t=20.0 - 0.001 * m + 0 * k,
                    length_steps=(2001 - (m.astype(np.int64) % 17) + 0 * k).astype(np.int32), cost_steps=((m + k) % 5).astype(np.int32))
    return ev


def _worker_many(rank, world, port, q, pop, ne):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from serl_amd import distributed as sd
    try:
        res = sd.evaluate_pop_sharded(_fake_eval(pop, ne), pop, ne)
        lo, hi = sd.member_block(pop, world, rank)
        st, steps, cost = _stored(lo, hi)
        gs, gsteps, gcost = sd.gather_stored_episodes(torch.from_numpy(st), steps, cost, pop, world, rank)
        import hashlib
        res['stored_digest'] = (hashlib.sha256(gs.numpy().tobytes()).hexdigest(), gsteps.copy(), gcost.copy())
    except Exception:          # (a rank that dies leaves the others in a collective: tell the test at once)
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))
        os._exit(3)
    q.put((rank, {k: (v if not isinstance(v, np.ndarray) else v.copy()) for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('pop', [50, 512])
def test_eight_rank_gather_equals_single_process(pop):
    """The shape of the node the metric is quoted on: EIGHT ranks (gloo here, RCCL on the GPUs), BASELINE config 3's population
    of 50 (50 does not divide 8: seven blocks of 7 and one of 1) and config 4's 512 (blocks of 64): result rows, champion / worst
    and the stored episodes of all members come out on every rank exactly as in one process (base/core/agent.py:234-256)."""
    sys.path.insert(0, ROOT)
    from serl_amd import distributed as sd
    import hashlib
    ne, world = 3, 8
    single = sd.evaluate_pop_sharded(_fake_eval(pop, ne), pop, ne)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 30500 + (os.getpid() + pop) % 2000
    procs = [ctx.Process(target=_worker_many, args=(r, world, port, q, pop, ne)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r_, d_ = q.get(timeout=600)
        if 'error' in d_:
            for p in procs:
                p.kill()
            pytest.fail('rank %d: %s' % (r_, d_['error']))
        got[r_] = d_
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    per = (pop + world - 1) // world
    assert [got[r]['block'] for r in range(world)] == [(min(r * per, pop), min((r + 1) * per, pop)) for r in range(world)]
    st, steps, cost = _stored(0, pop)
    want = hashlib.sha256(st.tobytes()).hexdigest()
    for r in range(world):
        for k in ('fitness', 'returns', 'smoothness', 'length_t', 'length_steps', 'cost_steps', 'pop_fitness'):
            np.testing.assert_array_equal(got[r][k], single[k], err_msg='rank %d %s' % (r, k))
        assert got[r]['champion'] == single['champion'] and got[r]['worst'] == single['worst']
        dg, gsteps, gcost = got[r]['stored_digest']
        assert dg == want
        np.testing.assert_array_equal(gsteps, steps); np.testing.assert_array_equal(gcost, cost)


def test_bench_dry_partition_prints_the_member_blocks():
    """`bench.py --gpus 8 --total-pop 512 --dry-partition` (no GPU, no launcher): the blocks an 8-GPU run would evaluate"""
    import subprocess, json
    for args, want in ((['--total-pop', '512'], [[64 * r, 64 * r + 64] for r in range(8)]),
                       (['--total-pop', '50'], [[7 * r, min(7 * r + 7, 50)] for r in range(8)]),
                       (['--workload', 'mixed', '--total-pop', '2048'], [[256 * r, 256 * r + 256] for r in range(8)])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--dry-partition'] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d['covers_every_member_once'] and [b['members'] for b in d['blocks']] == want and d['scaling'] == 'strong'


def test_bench_n_gpu_command_plans_the_strong_scaling_legs():
    """The driver's N-GPU command is `bench.py --gpus N` with no --total-pop: the weak-scaling line of the metric's configuration.  It
    appends the strong-scaling legs north_star asks about -- ONE population of 512 (BASELINE config 4, base/core/agent.py:234-256 cut into N
    member blocks) and the mixed-fault sweep of 2 048 (config 5) -- and says that pop = 50 strong scaling is flat by construction.  The dry
    run shows the plan without a GPU; with one rank, --total-pop or --pop there are no legs."""
    import subprocess, json

    def dry(*args):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-partition'] + list(args), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    d = dry('--gpus', '2')
    legs = d['strong_scaling_legs']
    assert [(l['workload'], l['total_pop']) for l in legs] == [('serl50', 512), ('mixed', 2048)]
    assert legs[0]['blocks'] == [[0, 256], [256, 512]] and legs[1]['blocks'] == [[0, 1024], [1024, 2048]]
    assert 'flat by construction' in d['strong_scaling_note'] and d['scaling'] == 'weak'
    assert [l['blocks'][-1] for l in dry('--gpus', '8')['strong_scaling_legs']] == [[448, 512], [1792, 2048]]
    for args in (('--gpus', '1'), ('--gpus', '8', '--total-pop', '512'), ('--gpus', '8', '--pop', '64'), ('--gpus', '8', '--strong-legs', 'off')):
        assert dry(*args)['strong_scaling_legs'] == []


def test_bench_n_gpu_plan_fits_the_time_out_and_says_what_to_drop():
    """the N = 8 command -- weak line + the two strong-scaling legs + rank 0's partition checks, the only pieces that grow with N (the
    6 144-episode leg re-evaluated on one GPU) -- must fit a driver's time-out by the measured per-leg times (bench.PLAN_TIMES), the dry run prints the
    expected wall time piece by piece, and --no-partition-check is the documented way out when it does not."""
    import subprocess, json

    def dry(*args):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-partition'] + list(args), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])['expected']
    e8 = dry('--gpus', '8')
    names = [p['name'] for p in e8['pieces']]
    assert len(names) == 3 and names[0].startswith('line (weak') and 'serl50 pop = 512' in names[1] and 'mixed pop = 2048' in names[2]
    assert [p['episodes_per_gpu'] for p in e8['pieces']] == [150, 192, 768]
    assert e8['fits_timeout'] and e8['expected_wall_s'] < 0.5 * e8['assumed_timeout_s'], e8          # minutes, with a factor of two to spare
    assert e8['pieces'][2]['partition_check_s'] > e8['pieces'][0]['partition_check_s'] > 0          # the 6 144-episode check is the largest single piece that N adds
    assert abs(e8['expected_wall_s'] - e8['startup_and_rendezvous_s'] - sum(p['total_s'] for p in e8['pieces'])) < 0.2
    e8n = dry('--gpus', '8', '--no-partition-check')
    assert all(p['partition_check_s'] == 0 for p in e8n['pieces']) and abs(e8n['expected_wall_s'] - e8['expected_wall_s_with_no_partition_check']) < 0.2
    tight = dry('--gpus', '8', '--timeout-s', str(e8['expected_wall_s'] - 1))
    assert not tight['fits_timeout'] and 'no-partition-check' in tight['advice']
    e1 = dry('--gpus', '1')
    assert len(e1['pieces']) == 1 and e1['pieces'][0]['partition_check_s'] == 0 and e1['expected_wall_s'] < e8['expected_wall_s']


def test_member_blocks_cover_population():
    from serl_amd import distributed as sd
    for pop in (1, 5, 50, 512, 2048):
        for ws in (1, 2, 4, 8):
            blocks = [sd.member_block(pop, ws, r) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == pop
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(ws - 1))


_NCCL_ONE_RANK = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import serl_amd
from serl_amd import refsignals, distributed as sd
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=%(port)r, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)      # nccl == RCCL on ROCm
w = torch.from_numpy(np.load(os.path.join(%(root)r, 'tests', 'golden', 'actors.npz'))['serl50'][:5])
spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
refs = refsignals.synthetic_reference_tables(10, 2, 20, seed=7)
eng = serl_amd.RolloutEngine(0)
def local(lo, hi):
    r = serl_amd.evaluate_pop(w[lo:hi], spec=spec, num_evals=2, refs=refs[lo * 2:hi * 2], t_max=20, engine=eng)
    return dict(fitness=r.fitness, returns=r.returns, smoothness=r.smoothness, length_t=r.length_t,
                length_steps=r.length_steps, cost_steps=r.cost_steps)
g = sd.evaluate_pop_sharded(local, 5, 2, device=dev)          # all_gather over the RCCL communicator
one = serl_amd.evaluate_pop(w, spec=spec, num_evals=2, refs=refs, t_max=20, engine=eng)
t = torch.ones(4, device=dev, dtype=torch.float64); dist.all_reduce(t); dist.barrier()
print(json.dumps(dict(same=bool(np.array_equal(g['fitness'], one.fitness) and np.array_equal(g['length_steps'], one.length_steps)),
                      champion=[g['champion'], one.champion], backend=dist.get_backend(), allreduce=float(t.sum()))))
dist.destroy_process_group()
'''


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_sharded_evaluation_over_a_one_rank_rccl_group():
    """The RCCL code path (init with device_id, all_gather of the result rows on the device, all_reduce, barrier) on the
    one GPU a test box has: world size 1, backend nccl, in a process of its own."""
    import subprocess, json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, '-c', _NCCL_ONE_RANK % dict(root=ROOT, port=str(29600 + os.getpid() % 2000))],
                       capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])      # (RCCL prints its library path on stdout)
    assert res['same'] and res['champion'][0] == res['champion'][1] and res['backend'] == 'nccl' and res['allreduce'] == 4.0


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_starts_itself_for_n_gpus_or_says_why_not():
    """`python bench.py --gpus N` without a launcher: with fewer than N GPUs one JSON line {"skipped": ...}, exit code 0;
    under torch.distributed.run with one rank the RCCL bench path runs end to end."""
    import subprocess, json
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n + 1)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'skipped' in json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', str(29700 + os.getpid() % 2000), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1',
                        '--warmup', '0', '--pop', '4', '--no-cpu-baseline'], capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0
    assert 'strong_scaling_legs' not in line          # (--pop given: no legs)
    # ... and the strong-scaling legs an N-GPU run appends (forced here on the one rank): BASELINE configs 4 and 5 through the same process group
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', str(29750 + os.getpid() % 2000), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1',
                        '--warmup', '0', '--no-cpu-baseline', '--strong-legs', 'on'], capture_output=True, text=True, timeout=550, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith('{')][-1])
    legs = line['strong_scaling_legs']
    assert [(l['config']['total_pop'], l['scaling']) for l in legs] == [(512, 'strong'), (2048, 'strong')]
    assert all(l['value'] > 1e7 and l['rccl']['world_size'] == 1 for l in legs) and line['scaling'] == 'weak' and 'flat by construction' in line['strong_scaling_note']


# ---- a whole generation sharded over two ranks (serl_amd.generation.evaluate_generation_sharded) --------------------------
class _Buf(list):
    def add(self, *t):
        self.append(tuple(np.asarray(x).copy() if hasattr(x, 'shape') else x for x in t))


def _gen_setup(n_pop=5):
    import argparse
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import serl_amd
    from serl_amd.actor import unpack_into
    from conftest import OracleEngine
    args = argparse.Namespace(hidden_size=32, num_layers=3, activation_actor='tanh', state_dim=7, action_dim=3,
                              device=torch.device('cpu'), num_evals=2, smooth_fitness=False, noise_sd=0.2, noise_clip=0.5)
    rows = np.load(os.path.join(ROOT, 'tests', 'golden', 'actors.npz'))['serl50']

    class Agent:
        def __init__(self, row):
            self.actor = serl_amd.Actor(args)
            unpack_into(self.actor, torch.from_numpy(row[:3715].copy()))
            self.buffer, self.critical_buffer = _Buf(), _Buf()
    pop = [Agent(rows[i]) for i in ((18, 0, 7, 33, 5) if n_pop == 5 else [(7 * k + 3) % 50 for k in range(n_pop)])]
    rl = Agent(rows[9])
    from serl_amd import refsignals
    refs = refsignals.synthetic_reference_tables(n_pop * 2 + 1, 2, 20, seed=3)
    noise = np.clip(0.2 * np.random.RandomState(5).randn(refs.shape[1], 3), -0.5, 0.5)
    return serl_amd, args, pop, rl, refs, noise, OracleEngine()


def _digest(pop, rl, shared, counters, g):
    agents = pop + ([rl] if rl is not None else [])
    bufs = [shared] + [a.buffer for a in agents] + [a.critical_buffer for a in agents]
    return dict(fitness=g.pop.fitness.copy(), lengths=g.pop.length_steps.copy(), champion=g.pop.champion, counters=dict(counters),
                lens=[len(b) for b in bufs], sums=[float(sum(float(np.sum(t[0])) + float(np.sum(t[1])) + t[3] for t in b)) for b in bufs],
                rl_fitness=g.rl_episode.fitness if g.rl_episode is not None else None)


def _gen_worker(rank, world, port, q, with_rl=True, n_pop=5):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    serl_amd, args, pop, rl, refs, noise, eng = _gen_setup(n_pop)
    if not with_rl:
        rl, refs, noise = None, refs[:-1], None
    from serl_amd.generation import evaluate_generation_sharded
    shared, counters = _Buf(), {}
    try:
        g = evaluate_generation_sharded(pop, rl, args=args, t_max=20, refs=refs, rl_noise=noise, engine=eng, replay_buffer=shared, counters=counters)
    except Exception:          # (a rank that dies leaves the others in a collective: tell the test at once)
        import traceback
        q.put((rank, {'error': traceback.format_exc()}))
        os._exit(3)
    q.put((rank, _digest(pop, rl, shared, counters, g)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world,with_rl,n_pop', [(2, True, 5), (4, True, 5), (4, False, 5), (8, True, 50)])
def test_generation_sharded_over_ranks_equals_single_process(world, with_rl, n_pop):
    """evaluate_generation_sharded on gloo ranks (population of 5; two ranks: blocks 3 + 2; FOUR ranks: blocks 2 + 2 + 1 + 0 -- a
    rank with an EMPTY member block, which flies the RL actor's exploration episode only, or nothing at all without an RL actor;
    EIGHT ranks, population of 50 -- BASELINE config 3 on one node: blocks of 7 and a last one of 1):
    the gathered fitness table, the champion, the counters and EVERY buffer (shared, per agent, critical) must come out on every
    rank exactly as evaluate_generation fills them in one process.  (Local evaluation = the CPU oracle behind
    RolloutEngine.rollout's call shape: the sharding, the two all-gathers and the buffer order are what is under test.)"""
    serl_amd, args, pop, rl, refs, noise, eng = _gen_setup(n_pop)
    if not with_rl:
        rl, refs, noise = None, refs[:-1], None
    shared, counters = _Buf(), {}
    g = serl_amd.evaluate_generation(pop, rl, args=args, t_max=20, refs=refs, rl_noise=noise, engine=eng, replay_buffer=shared, counters=counters)
    single = _digest(pop, rl, shared, counters, g)
    n_agents = n_pop + (1 if with_rl else 0)
    assert single['lens'][0] == sum(single['lens'][1:1 + n_agents]) and single['counters']['num_episodes'] == n_agents
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() + 7 * world + int(with_rl)) % 2000
    procs = [ctx.Process(target=_gen_worker, args=(r, world, port, q, with_rl, n_pop)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r_, d_ = q.get(timeout=600)
        if 'error' in d_:
            for p in procs:
                p.kill()
            pytest.fail('rank %d: %s' % (r_, d_['error']))
        got[r_] = d_
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        np.testing.assert_array_equal(got[r]['fitness'], single['fitness'])
        np.testing.assert_array_equal(got[r]['lengths'], single['lengths'])
        assert got[r]['champion'] == single['champion'] and got[r]['counters'] == single['counters']
        assert got[r]['lens'] == single['lens'] and got[r]['sums'] == single['sums'] and got[r]['rl_fitness'] == single['rl_fitness']
