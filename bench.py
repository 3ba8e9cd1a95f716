#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the population-rollout fitness evaluation (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload serl50|serl10|mixed] [--pop MEMBERS_PER_GPU | --total-pop MEMBERS]
  (N > 1: one rank per GPU over RCCL -- launched by torch.distributed.run, or, when started as a plain
   `python bench.py --gpus N`, bench.py re-executes itself under torch.distributed.run; fewer than N visible GPUs ->
   one JSON line {"skipped": ...} and exit code 0)

A "step" is one population evaluation: pop members x num_evals episodes x 8 001 env steps of the
PH-LAB attitude-tracking task (t_max = 80 s), weights and reference tables already resident in HBM.
Default (weak scaling): every GPU evaluates its own block of `pop` members (shipped actors, tiled with seeded noise
beyond the shipped ones) and the per-member result rows are all-gathered (RCCL).  Prints ONE JSON line on rank 0.
--total-pop M (strong scaling, BASELINE configs 4 / 5): ONE population of M members sharded over the N ranks in contiguous
member blocks (serl_amd.distributed.member_block, the partition of base/core/agent.py:234-256's loop), "scaling": "strong";
with --gpus 1 all M x num_evals episodes run on one GPU and are checked against the CPU restatement.

Workloads (BASELINE.json configs): serl50 = config 3 (the metric's configuration, default: pop=50, actor 7-32x4-3;
--pop 64 is one GPU's share of config 4, pop=512 over 8), serl10 = config 2 (pop=10, actor 7-72x4-3), mixed = one
GPU's share of config 5 (fault mode per episode = e mod 6 over be/jr/sa/se/ice/cg; default --pop 256).
"""
import argparse, json, os, socket, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 48.0          # algorithmic HBM bytes per env-step: ref[k] 3 f64 read + u[k] 3 f64 written
F_ALG = 24059.0       # f64 arithmetic instructions the reference retires per env-step
HBM_PEAK = 8.0e12     # B/s, MI355X_MICROARCH.md chip table (6.29e12 measured)
FP64_PEAK = 78.6e12   # FLOP/s vector f64 (half the 157.3 TF f32 vector rate)
WORKLOADS = {'serl50': dict(tag='serl50', hidden=32, pop=50), 'serl10': dict(tag='serl10', hidden=72, pop=10),
             'mixed': dict(tag='serl50', hidden=32, pop=256)}
MIXED_MODES = ['be', 'jr', 'sa', 'se', 'ice', 'cg']
REFERENCE = os.environ.get('SERL_REFERENCE', '/root/reference')


def make_population(pop, rank, seed=7, tag='serl50'):
    """Members [rank*pop, (rank+1)*pop) of the benchmark population: the shipped actors, tiled, with seeded N(0, 0.01^2)
    noise on the 2-D weights of every member beyond the shipped ones.  Deterministic per (rank, pop)."""
    import torch
    from serl_amd import NetSpec
    base = np.load(os.path.join(ROOT, 'tests', 'golden', 'actors.npz'))[tag]   # [50, 3715] / [10, 16563] shipped actors
    gid = np.arange(pop) + rank * pop
    w = torch.from_numpy(base[gid % len(base)].copy())
    spec = NetSpec(7, 3, WORKLOADS[tag]['hidden'], 3, 'tanh')
    g = torch.Generator().manual_seed(seed + rank)
    noise = [0.01 * torch.randn(pop, n, generator=g) for _, n in spec.genome_segments()]
    fresh = torch.from_numpy(gid >= len(base))
    for (off, n), z in zip(spec.genome_segments(), noise):
        w[:, off:off + n] += z * fresh[:, None]
    return w


def fp64_measured(ach_f64):
    """the FP64 peak as MEASURED on an MI355X (tools/valu_latency.hip fma_peak: v_fma_f64 on every CU, 16 wavefronts per CU, 16 independent
    chains per lane; profiles/valu_latency_current.json, refreshed by tools/distill_profiles.py) beside the datasheet figure"""
    try:
        v = json.load(open(os.path.join(ROOT, 'profiles', 'valu_latency_current.json')))
        pk = float(v['fp64_fma_peak_tflops_measured'])
        return {'peak_measured': pk, 'frac_of_measured_peak': ach_f64 / (pk * 1e12), 'peak_measured_source': 'profiles/valu_latency_current.json (tools/valu_latency.hip fma_peak)'}
    except Exception:
        return {'peak_measured': None}


def cpu_port(w, hidden, ref, moe, faults_of=None, build_of=None):
    """The C restatement (oracle/rollout_ref.c) on all host cores, the same workload (the whole evaluation once)."""
    from oracle import rollout as R
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
    from time_reference import host_cores
    cores = host_cores()              # (affinity mask capped by the cgroup CPU quota: the GPU box shows 256 logical CPUs, 16 cores' worth usable)
    net = dict(state_dim=7, action_dim=3, hidden=hidden, num_layers=3, activation='tanh')
    E = len(moe)
    R.rollout(w, net, moe[:cores], ref[:cores], t_max=80.0, threads=cores, short_libm=True)       # warm-up (page-in, lib build)
    fit, ls = np.zeros(E), np.zeros(E, np.int32)
    t0 = time.perf_counter()
    groups = {None: np.arange(E)} if build_of is None else {b: np.nonzero(np.asarray(build_of) == b)[0] for b in dict.fromkeys(build_of)}
    for b, idx in groups.items():
        # (the same-libm flavour of the checker, oracle/citation_rt.h CIT_SHORT_LIBM: the kernels' own sin / cos / tan / pow compiled for the CPU, so
        # that parity_vs_cpu_port is a bit-for-bit statement; its speed is that of the glibc flavour, libm is a percent of a step)
        o = R.rollout(w, net, moe[idx], ref[idx], t_max=80.0, threads=cores, short_libm=True, **({} if b is None else dict(build=b, faults=[faults_of[e] for e in idx])))
        fit[idx], ls[idx] = o['fitness'], o['length_steps']
    dt = time.perf_counter() - t0
    steps = int(ls.sum())
    # ... and the flavour that is PINNED to the reference binary (glibc libm; tests/test_oracle_dynamics.py): the same-libm flavour shares sin / cos / tan /
    # pow / atan with the kernels, so 0.0 against it says "GPU == its own restatement"; the distance to the pinned flavour is the second claim of the line
    # (a sample of the episodes, untimed)
    pick = np.arange(0, E, max(1, E // 48))[:48]
    fit_pinned = np.zeros(len(pick))
    for b, idx in groups.items():
        sel = np.array([i for i in pick if i in set(idx.tolist())], dtype=int)
        if len(sel):
            o = R.rollout(w, net, moe[sel], ref[sel], t_max=80.0, threads=cores, short_libm=False, **({} if b is None else dict(build=b, faults=[faults_of[e] for e in sel])))
            fit_pinned[np.searchsorted(pick, sel)] = o['fitness']
    return dict(value=steps / dt, unit='env-steps/s', cores=cores, kind='port',
                sample='all %d episodes (8001 steps each) of the same workload, C restatement '
                       '(oracle/rollout_ref.c, same-libm flavour) on %d threads, %.1f s wall' % (E, cores, dt),
                pinned_sample=(pick.tolist(), fit_pinned.tolist())), fit, ls


def cpu_baseline(w, hidden, ref, moe, faults_of=None, build_of=None):
    """the reference's OWN Python path (unmodified Agent.evaluate + CitationEnv + torch Actor + its prebuilt
    dynamics library), one process per host core of THIS box, timed in THIS run (tests/tools/time_reference.py): from
    /root/reference in the build container, from the archive oracle/build.py stage_ref() packs into the git-ignored
    oracle/_ref/ on the GPU box.  The C restatement (oracle/rollout_ref.c) on all host threads rides along as `port` (it is
    also the checker of parity_vs_cpu_port)."""
    # a bounded sample for the C restatement: at most 1 536 episodes (~18 s on 256 threads)
    n = min(len(moe), 1536)
    port, fit, ls = cpu_port(w, hidden, ref[:n], moe[:n], None if faults_of is None else faults_of[:n], None if build_of is None else build_of[:n])
    from oracle import refso
    cb = None
    if refso.available():
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'tools', 'time_reference.py'), '--episodes', '2'],
                               capture_output=True, text=True, timeout=900)
            m = json.loads(r.stdout.strip().splitlines()[-1])
            cb = dict(value=m['env_steps_per_s'], unit='env-steps/s', cores=m['procs'], kind='reference-python',
                      per_core=m['env_steps_per_s_per_core'], reference=m['reference'],
                      sample='%d processes (one per usable host core) x %d full 80 s episode(s) = %d env steps of the reference\'s unmodified '
                             'Agent.evaluate (SERL50 actors, base reference, nominal build) after a 5 s warm-up episode each, %.1f s wall '
                             '(+ %.0f s process start-up, untimed)' % (m['procs'], m['episodes_per_proc'], m['env_steps'], m['seconds'], m['startup_seconds']),
                      port=port)
        except Exception as ex:          # the reference leg must never take the bench line down
            port['reference_python_error'] = repr(ex)[:300]
    if cb is None:
        cb = port
    return cb, fit, ls, n


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: become N ranks under torch.distributed.run (RCCL rendezvous on
    127.0.0.1), or report why not and exit 0."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < a.gpus:
        print(json.dumps({'skipped': 'needs %d GPUs, %d visible' % (a.gpus, have), 'n_gpus': a.gpus,
                          'metric': 'env-steps/sec (whole node), pop-rollout eval', 'value': None}))
        sys.exit(0)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    os.execvpe(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
                                '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:], env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='serl50')
    ap.add_argument('--pop', type=int, default=0, help='members per GPU (0 = the workload\'s own)')
    ap.add_argument('--total-pop', type=int, default=0, help='strong scaling: ONE population of this many members sharded over the ranks')
    ap.add_argument('--num-evals', type=int, default=3)
    ap.add_argument('--lanes', type=int, default=0, help='episodes per wavefront (0 = auto)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-partition-check', action='store_true')
    ap.add_argument('--fused', action='store_true', help='mixed workload: ONE launch of one code object (the default whenever the library takes the combination; kept for A/B scripts)')
    ap.add_argument('--no-fused', action='store_true', help='mixed workload: a launch per dynamics build side by side instead of ONE launch of one code object (A/B)')
    ap.add_argument('--strong-legs', choices=['auto', 'on', 'off'], default='auto', help='append the strong-scaling legs (pop=512, mixed pop=2048) to the line: auto = when N > 1 and no --total-pop / --pop')
    ap.add_argument('--timeout-s', type=float, default=900.0, help='--dry-partition: the time-out the plan is checked against (the driver\'s is not known to this script; "within minutes" is the contract)')
    ap.add_argument('--dry-partition', action='store_true', help='print the member / episode blocks of every rank for --gpus N [--total-pop M | --pop P] and exit: no GPU, no launcher')
    a = ap.parse_args()
    if a.dry_partition:
        from serl_amd import distributed as sd_
        wl_, ne_ = WORKLOADS[a.workload], a.num_evals
        blocks = []
        for r in range(a.gpus):
            lo_, hi_ = sd_.member_block(a.total_pop, a.gpus, r) if a.total_pop > 0 else (r * (a.pop or wl_['pop']), (r + 1) * (a.pop or wl_['pop']))
            blocks.append({'rank': r, 'members': [lo_, hi_], 'n_members': hi_ - lo_, 'episodes': [lo_ * ne_, hi_ * ne_],
                           'fault_modes_of_first_episodes': [MIXED_MODES[e % 6] for e in range(lo_ * ne_, min(lo_ * ne_ + 6, hi_ * ne_))] if a.workload == 'mixed' else None})
        total = a.total_pop if a.total_pop > 0 else a.gpus * (a.pop or wl_['pop'])
        covered = sorted(m for b in blocks for m in range(*b['members']))
        print(json.dumps({'dry_partition': True, 'n_gpus': a.gpus, 'scaling': 'strong' if a.total_pop > 0 else 'weak', 'total_pop': total,
                          'num_evals': ne_, 'blocks': blocks, 'covers_every_member_once': covered == list(range(total)),
                          'gather': 'one all_gather of [num_evals, ceil(pop / world), 6] f64 rows per evaluation (serl_amd/distributed.py gather_rows)',
                          'strong_scaling_legs': [dict(leg, blocks=[list(sd_.member_block(leg['total_pop'], a.gpus, r)) for r in range(a.gpus)]) for leg in strong_legs(a, a.gpus)],
                          'strong_scaling_note': STRONG_NOTE if strong_legs(a, a.gpus) else None,
                          'expected': expected_wall(a, a.gpus)}))
        return
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(a)

    import torch
    import torch.distributed as dist
    import serl_amd

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    grouped = 'WORLD_SIZE' in os.environ          # under a launcher, also with one rank: the RCCL path is the one that runs
    if grouped:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    ctx = dict(rank=rank, world=world, local=local, dev=dev, grouped=grouped, eng=serl_amd.RolloutEngine(local))
    res = measure(a, ctx)
    # The driver's N-GPU command carries no --total-pop: on its own it is the WEAK-scaling line of the metric's configuration (pop = 50
    # per GPU).  BASELINE config 4 -- the one numeric target of north_star: ONE population of 512 sharded over the node, base/core/agent.py:234-256 --
    # and config 5 (mixed-fault sweep, pop = 2 048) are strong-scaling questions, so the same run appends them as short legs (a few evaluations each,
    # every rank in the same process group, the partition checked against one GPU), without touching the timed region above.
    legs = strong_legs(a, world)
    if legs:
        out = []
        for leg in legs:
            a2 = argparse.Namespace(**dict(vars(a), **leg, no_cpu_baseline=True))
            r2 = measure(a2, ctx)
            if rank == 0:
                out.append({k: r2[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling', 'kernel_ms', 't_step_us', 'config',
                                               'partition_invariance', 'rccl') if k in r2})
        if rank == 0:
            res['strong_scaling_legs'] = out
            res['strong_scaling_note'] = STRONG_NOTE
    if rank == 0:
        print(json.dumps(res))
    if grouped:
        dist.destroy_process_group()


STRONG_NOTE = ('`value` of this line is WEAK scaling: every GPU evaluates its own pop = 50 (150 episodes, one per CU-resident team; 106 CUs idle).  STRONG scaling of pop = 50 is '
               'flat by construction -- 150 / N episodes per GPU take the same kernel time as 150 (each episode owns a CU and is latency-bound; ceil-blocks give 7,7,7,7,7,7,7,1 members '
               'on 8 ranks) -- so the strong-scaling figures are the legs: one population of 512 (BASELINE config 4) and the mixed-fault sweep of 2 048 (config 5) sharded by member.')


# Measured single-GPU times behind --dry-partition's expected wall time (MI355X; profiles/r05_fin_bench_*.json, refreshed by profiles/r06_*): what ONE
# population evaluation costs on one GPU by kernel class, and what the host adds around it.  Only the order of magnitude matters: the plan must fit the
# driver's time-out with a wide margin or say which check to drop.
PLAN_TIMES = {
    'serl50_one_round_ms': 109.2,          # <= CUs episodes, one per team (bench_serl50 / bench_pop64: 150 or 192 episodes x 8 001 steps)
    'serl50_two_per_team_ms': 139.0,       # <= 2 x CUs (bench_pop128)
    'serl50_four_per_team_ms': 172.0,      # <= 4 x CUs (bench_pop341)
    'serl50_queue_steps_per_s': 36.6e6,    # > 4 x CUs, four per team + work queue (bench_total512)
    'mixed_four_per_team_ms': 174.5,       # mixed sweep <= 4 x CUs in one launch (bench_mixed: 768 episodes)
    'mixed_queue_steps_per_s': 39.9e6,     # (bench_mixed_total2048: 6 144 episodes)
    'serl10_one_round_ms': 125.6,          # streamed actor, one per team (bench_serl10)
    'startup_s': 12.0,                     # python + torch + HIP context + first launch of a process (driver_run_s of the default line less its steps)
    'ref_table_s_per_episode': 2.2e-3,     # refsignals.synthetic_reference_tables on one host core (8 001 x 3 f64 per episode) + its H2D copy
    'member_s': 0.7e-3,                    # make_population per member
    'rendezvous_s': 6.0,                   # torch.distributed.run + RCCL communicator for N > 1
}


def _eval_seconds(workload, episodes, cus=256):
    P = PLAN_TIMES
    if episodes <= 0:
        return 0.0
    if workload == 'mixed':
        return P['mixed_four_per_team_ms'] / 1e3 * max(episodes / 768.0, 0.5) if episodes <= 4 * cus else episodes * 8001 / P['mixed_queue_steps_per_s']
    if workload == 'serl10':
        return P['serl10_one_round_ms'] / 1e3 * -(-episodes // cus)
    if episodes <= cus:
        return P['serl50_one_round_ms'] / 1e3
    if episodes <= 2 * cus:
        return P['serl50_two_per_team_ms'] / 1e3
    if episodes <= 4 * cus:
        return P['serl50_four_per_team_ms'] / 1e3
    return episodes * 8001 / P['serl50_queue_steps_per_s']


def expected_wall(a, world):
    """--dry-partition: what the command is expected to take on an N-GPU node, piece by piece (seconds), from PLAN_TIMES: the line itself, the
    strong-scaling legs it appends, and rank 0's partition checks (one GPU re-evaluating all members) -- the only pieces that GROW with N."""
    from serl_amd import distributed as sd_
    P = PLAN_TIMES
    pieces = []

    def piece(name, workload, total_members, per_rank_members, steps, warmup, check):
        ne = a.num_evals
        e_rank, e_all = per_rank_members * ne, total_members * ne
        t = dict(name=name, episodes_per_gpu=e_rank,
                 inputs_s=round(e_rank * P['ref_table_s_per_episode'] + per_rank_members * P['member_s'], 2),
                 timed_and_warmup_s=round((steps + warmup) * _eval_seconds(workload, e_rank), 2),
                 host_buffer_variant_s=round(2 * _eval_seconds(workload, e_rank), 2),
                 partition_check_s=round((e_all * P['ref_table_s_per_episode'] + total_members * P['member_s'] + _eval_seconds(workload, e_all)) if check else 0.0, 2))
        t['total_s'] = round(t['inputs_s'] + t['timed_and_warmup_s'] + t['host_buffer_variant_s'] + t['partition_check_s'], 2)
        pieces.append(t)

    wl = WORKLOADS[a.workload]
    check = world > 1 and not a.no_partition_check
    if a.total_pop > 0:
        per = max(sd_.member_block(a.total_pop, world, r)[1] - sd_.member_block(a.total_pop, world, r)[0] for r in range(world))
        piece('line (strong: pop = %d over %d GPUs)' % (a.total_pop, world), a.workload, a.total_pop, per, a.steps, a.warmup, check)
    else:
        per = a.pop or wl['pop']
        piece('line (weak: pop = %d per GPU)' % per, a.workload, per * world, per, a.steps, a.warmup, check)
    for leg in strong_legs(a, world):
        per = max(sd_.member_block(leg['total_pop'], world, r)[1] - sd_.member_block(leg['total_pop'], world, r)[0] for r in range(world))
        piece('leg %s pop = %d' % (leg['workload'], leg['total_pop']), leg['workload'], leg['total_pop'], per, leg['steps'], leg['warmup'], check)
    fixed = P['startup_s'] + (P['rendezvous_s'] if world > 1 else 0.0)
    total = fixed + sum(t['total_s'] for t in pieces)
    without = total - sum(t['partition_check_s'] for t in pieces)
    return dict(pieces=pieces, startup_and_rendezvous_s=fixed, expected_wall_s=round(total, 1), expected_wall_s_with_no_partition_check=round(without, 1),
                assumed_timeout_s=a.timeout_s, fits_timeout=bool(total < a.timeout_s),
                advice=None if total < a.timeout_s else ('add --no-partition-check (-%.0f s)' % (total - without) if without < a.timeout_s else 'add --strong-legs off'),
                source='bench.PLAN_TIMES: single-GPU evaluation times by kernel class and host-side rates measured on an MI355X box (profiles/r05_fin_bench_*.json); the '
                       'CPU-baseline leg runs at N = 1 only and is not part of an N-GPU plan')


def strong_legs(a, world):
    """the strong-scaling legs `bench.py --gpus N` (N > 1, no --total-pop) appends to its line; --strong-legs on / off forces or drops them"""
    if a.strong_legs == 'off' or a.total_pop > 0 or a.workload != 'serl50' or a.pop:
        return []
    if a.strong_legs != 'on' and world <= 1:
        return []
    return [dict(workload='serl50', total_pop=512, steps=3, warmup=1), dict(workload='mixed', total_pop=2048, steps=2, warmup=1)]


def measure(a, ctx):
    """W warm-up evaluations, K timed ones (barrier + synchronize on both sides, MAX over ranks), the bench line's dict on rank 0 (None elsewhere)"""
    import torch
    import torch.distributed as dist
    import serl_amd
    from serl_amd import refsignals, metrics, builds, distributed as sd
    rank, world, local, dev, grouped = ctx['rank'], ctx['world'], ctx['local'], ctx['dev'], ctx['grouped']

    wl = WORKLOADS[a.workload]
    spec = serl_amd.NetSpec(7, 3, wl['hidden'], 3, 'tanh')
    ne = a.num_evals
    strong = a.total_pop > 0
    if strong:
        # ONE population, contiguous member blocks (the reference's `for net in pop` loop, base/core/agent.py:234-256, cut into
        # `world` pieces); every rank builds only its own block of the deterministic population / references
        lo, hi = sd.member_block(a.total_pop, world, rank)
        pop = hi - lo
        w_host = make_population(a.total_pop, 0, tag=wl['tag'])[lo:hi].contiguous()
        ref_host = refsignals.synthetic_reference_tables(pop * ne, ne, 80, seed=7, first=lo * ne)
        gather_pop = a.total_pop
    else:
        lo, pop = 0, a.pop or wl['pop']
        w_host = make_population(pop, rank, tag=wl['tag'])
        ref_host = refsignals.synthetic_reference_tables(pop * ne, ne, 80, seed=7 + 100000 * rank)
        gather_pop = pop * world
    E = pop * ne
    eng = ctx['eng']
    w = w_host.to(dev)
    ref = torch.from_numpy(ref_host).to(dev)
    moe = np.repeat(np.arange(pop, dtype=np.int32), ne)
    T = ref.shape[1]
    mixed = a.workload == 'mixed'
    modes = [MIXED_MODES[(lo * ne + e) % 6] for e in range(E)] if mixed else None

    pending = []      # (all-episodes-full flag, action traces) of speculative smoothness passes not yet confirmed
    placements = []   # mixed workload: how every one-launch sweep placed its workgroups (PopResult.mixed_placement)

    def evaluate(wd, refd, moe_, n_members, modes_=None):
        """one population evaluation on this rank -> (rows f64 [ne, members, ROW] on the device, length_steps, fitness)"""
        if modes_ is None:
            out = eng.rollout(wd, spec, moe_, refd, t_max=80.0, traces='actions', lanes_per_wave=a.lanes, sync=False)
            ls, fit, lt, cs = out['length_steps'], out['fitness'], out['length_t'], out['cost_steps'].double()
            # a11, on device, enqueued behind the kernel as if every episode flew the whole table (an evaluation's usual case); the
            # flag is read in one_step, in front of the all-gather, and the general path taken if it says otherwise
            guess = metrics.calc_smoothness_speculative(out['actions'], ls)      # (None: the last evaluation of this shape had early endings)
            if guess is None:
                sm = metrics.calc_smoothness_after_miss(out['actions'], ls)      # (takes the mark back when this batch is full again)
            else:
                sm = guess[0]
                pending.append((guess[1], out['actions']))
            ls_host = None
        else:      # several dynamics builds: ONE launch of one code object, or (--no-fused) one per build side by side on streams of their own (evaluate_pop)
            r = serl_amd.evaluate_pop(wd, mode=modes_, num_evals=ne, refs=refd, t_max=80, spec=spec, engine=eng, fused=False if getattr(a, 'no_fused', False) else (True if getattr(a, 'fused', False) else 'auto'))
            tod = lambda x: torch.as_tensor(np.ascontiguousarray(x.T).reshape(-1), device=dev)
            fit, sm, lt, cs = tod(r.returns), tod(r.smoothness), tod(r.length_t), tod(r.cost_steps).double()
            ls = tod(r.length_steps)
            placements.append(r.mixed_placement)
        rows = torch.stack([fit, fit, sm, lt, ls.double(), cs], -1)
        return rows.view(n_members, ne, sd.ROW).transpose(0, 1).contiguous(), ls, fit

    def one_step():
        """one population evaluation on this rank + the fitness all-gather + index selection"""
        rows, ls, fit = evaluate(w, ref, moe, pop, modes)
        # the guess is confirmed (or replaced) BEFORE the all-gather: every rank makes exactly one collective call per step whatever its
        # own episodes did (reading the flag waits for this rank's kernel and smoothness pass -- the step's host synchronisation, moved up)
        while pending:
            full, acts = pending.pop()
            if not metrics.smoothness_speculation_result(acts, full):      # some episode ended early: the smoothness column by the general path
                pending.clear()
                sm = metrics.calc_smoothness(acts, ls)
                rows[..., 2] = sm.view(pop, ne).transpose(0, 1)
        g = sd.gather_rows(rows, gather_pop, world, rank, device=dev)
        pop_fitness = g[..., 0].mean(0)
        champion = int(torch.argmax(pop_fitness))
        return ls, fit, g, champion

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ls, fit, g, champion = one_step()
        kernel_ms.append(eng.kernel_ms() if not mixed else eng.last_kernel_ms)   # HIP events around the kernel, on its stream
    barrier()
    dt = time.perf_counter() - t0
    steps_local = int(ls.abs().sum())
    # what the collective layer saw, gathered THROUGH it: every rank's id, its device, its mean kernel time and env steps -- a line from
    # an N-GPU run proves by itself that N distinct ranks on N distinct devices took part
    rccl = None
    if grouped:
        mine = torch.tensor([float(rank), float(local), float(np.mean(kernel_ms)), float(steps_local), float(torch.cuda.current_device())],
                            dtype=torch.float64, device=dev)
        seen = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        seen = torch.stack(seen).cpu().numpy()
        rccl = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'ranks_seen': [int(v) for v in seen[:, 0]],
                'local_ranks': [int(v) for v in seen[:, 1]], 'devices': [int(v) for v in seen[:, 4]],
                'kernel_ms_per_rank': [round(float(v), 3) for v in seen[:, 2]], 'env_steps_per_rank': [int(v) for v in seen[:, 3]],
                'nccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, 'nccl') else None}
    tt = torch.tensor([dt, float(steps_local)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, steps_total = float(tmax[0]), float(tsum[1])
    else:
        steps_total = float(steps_local)
    if rank != 0:
        return None
    value = steps_total * a.steps / dt
    res_extra = {}
    if world > 1 and not a.no_partition_check:
        # the gathered result of the N-way sharded evaluation must be bit-identical to ONE GPU
        # evaluating the same pop x world members (rank 0 rebuilds every rank's block: the inputs are deterministic)
        if strong:
            ws = make_population(gather_pop, 0, tag=wl['tag']).to(dev)
            rs = torch.from_numpy(refsignals.synthetic_reference_tables(gather_pop * ne, ne, 80, seed=7)).to(dev)
            ms = [MIXED_MODES[e % 6] for e in range(gather_pop * ne)]
        else:
            ws = torch.cat([make_population(pop, r, tag=wl['tag']) for r in range(world)]).to(dev)
            rs = torch.cat([torch.from_numpy(refsignals.synthetic_reference_tables(E, ne, 80, seed=7 + 100000 * r)) for r in range(world)]).to(dev)
            ms = [MIXED_MODES[(e % E) % 6] for e in range(E * world)]
        rows1, _, _ = evaluate(ws, rs, np.repeat(np.arange(gather_pop, dtype=np.int32), ne), gather_pop, ms if mixed else None)
        res_extra['partition_invariance'] = {'members': gather_pop, 'bit_identical_to_one_gpu': bool(torch.equal(rows1, g)),
                                             'champion': champion}
    # informational (never `value`): the same evaluation when the boundary hands over HOST buffers -- weights and
    # reference tables cross PCIe inside the timed region (pinned memory, one H2D copy each per evaluation)
    w_pin, ref_pin = w_host.pin_memory(), torch.from_numpy(ref_host).pin_memory()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(2):
        w.copy_(w_pin, non_blocking=True); ref.copy_(ref_pin, non_blocking=True)
        one_step()
    torch.cuda.synchronize()
    value_host = steps_local * 2 / (time.perf_counter() - t1)
    k_ms = float(np.mean(kernel_ms))
    ach_hbm = steps_local * B_ALG / (k_ms * 1e-3)
    ach_f64 = steps_local * F_ALG / (k_ms * 1e-3)
    # HBM traffic of one launch and the issue-slot occupancy from the PMC passes committed under profiles/ (rocprofv3 --pmc,
    # collected and corrected as MI355X_MICROARCH.md prescribes); only quoted for the configuration they were measured on
    traffic, issue = None, None
    t_step_us = k_ms * 1e3 / T                   # per env step and team: measured in THIS run (HIP events around the kernel)
    try:
        # the floors are properties of the model DAG (tools/dag/critical_path.py, committed with the generated kernels); the
        # fraction printed is floor / THIS run's measured time per env step
        fl = json.load(open(os.path.join(ROOT, 'profiles', 'floors_current.json')))
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        one_per_team = not mixed and E <= cus          # the fraction is defined for one episode per team (episodes <= CUs of THIS device)
        issue = {'bound': 'valu issue slots (wave-uniform f64 glue: 63 of 64 lanes of every instruction carry the same scalar)',
                 'issue_floor_us_per_env_step': fl['issue_floor_trimmed_us_per_env_step'],
                 'issue_floor_full_dag_us_per_env_step': fl['issue_floor_us_per_env_step'],
                 'dependency_floor_us_per_env_step': fl['dependency_floor_us_per_env_step'],
                 'measured_us_per_env_step': t_step_us,
                 'issue_floor_frac': fl['issue_floor_trimmed_us_per_env_step'] / t_step_us if one_per_team else None,
                 'issue_floor_frac_full_dag': fl['issue_floor_us_per_env_step'] / t_step_us if one_per_team else None,
                 'frac_note': 'issue floor (minimal instruction count of the model DAG x 4 cycles over the 4 SIMDs of a CU, tools/dag/critical_path.py: '
                              'divisions by proved literals at 4 instructions) / measured time per env step of one team; `issue_floor_frac` is taken against the floor of '
                              'what the TRIMMED flight condition executes (gated Switch operands and the guarded exp / log10 bodies left out: %d of %d glue instructions), '
                              '`..._full_dag` against every node of the DAG; defined for one episode per team (episodes <= CUs)'
                              % (fl['glue_instructions_min_trimmed'], fl['glue_instructions_min'])}
        pm = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_current.json')))
        from serl_amd import build as hip_build
        here = hip_build.source_hash()
        if fl.get('csrc_sha256') != here:
            issue['floors_note'] = 'floors computed for another state of the kernel sources (profiles/floors_current.json csrc_sha256 != this tree): the DAG floors are properties of the model and move little, the fractions are quoted against them as they are'
        if pm.get('workload') == a.workload and pm.get('pop') == pop and ne == 3 and a.lanes == 0 and not strong and pm.get('csrc_sha256') != here:
            traffic = None             # the committed counters were collected on other kernel sources: not quoted (tools/profile_round.sh + tools/distill_profiles.py refresh them)
            issue['sq_counters'] = 'stale: profiles/pmc_current.json belongs to csrc_sha256 %s, this tree is %s' % (str(pm.get('csrc_sha256'))[:12], here[:12])
        elif pm.get('workload') == a.workload and pm.get('pop') == pop and ne == 3 and a.lanes == 0 and not strong:
            traffic = pm.get('traffic_bytes_per_launch')
            issue['sq_counters'] = dict({k: pm['issue'][k] for k in ('active_frac', 'wait_frac', 'issue_stall_frac', 'simd_issue_frac',
                                                                    'valu_per_env_step', 'salu_per_env_step', 'lds_per_env_step') if k in pm.get('issue', {})},
                                        source='committed profile (profiles/pmc_current.json: rocprofv3 --pmc passes of an earlier run of this command), not measured in this run')
    except Exception:
        pass
    name = {'serl50': 'PH-LAB nominal h2000_v90, pop=%d (SERL50 actor 7-32x4-3 tanh)',
            'serl10': 'PH-LAB nominal h2000_v90, pop=%d (SERL10 actor 7-72x4-3 tanh)',
            'mixed': 'PH-LAB mixed-fault sweep (be/jr/sa/se/ice/cg by episode), pop=%d (SERL50 actor shape)'}[a.workload] % (gather_pop if strong else pop)
    res = {
        'metric': ('env-steps/sec (whole node), pop-rollout eval, pop=%d %s sharded over the GPUs' % (a.total_pop, 'mixed-fault' if mixed else 'nominal')) if strong
                  else 'env-steps/sec (whole node), pop-rollout eval, pop=%d %s per GPU' % (pop, 'mixed-fault' if mixed else 'nominal'),
        'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s x num_evals=%d x 8001 steps (t_max=80 s) %s; shipped weights (tiled+noise beyond the '
                               'shipped ones), seeded smoothed-step references' % (name, ne, 'in all, sharded by member' if strong else 'per GPU'),
                   'pop_per_gpu': pop, 'total_pop': gather_pop, 'num_evals': ne, 'episodes_per_gpu': E, 'steps_per_episode': T,
                   'lanes_per_wave': a.lanes, 'parallelism': 'member-sharded dp%d' % world},
        'kernel_ms': k_ms,
        'roofline': {'bound': 'hbm', 'achieved': ach_hbm / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                     'frac': ach_hbm / HBM_PEAK, 'traffic': traffic,
                     'note': 'path is instruction-issue / dependent-latency bound (DESIGN.md), not HBM-bound: algorithmic traffic is '
                             '48 B per env-step; achieved = steps x 48 B / kernel time; traffic = bytes per launch from '
                             'profiles/pmc_current.json (2 x FETCH_SIZE + WRITE_SIZE)'},
        'roofline_fp64': dict({'bound': 'valu-f64', 'achieved': ach_f64 / 1e12, 'peak': FP64_PEAK / 1e12, 'unit': 'TFLOP/s',
                               'frac': ach_f64 / FP64_PEAK}, **fp64_measured(ach_f64)),
        'roofline_issue': issue,
        't_step_us': t_step_us,
        # SURVEY 8(d) words the metric "incl. H2D of weights / refs"; the bench contract of this build forbids a PCIe-inclusive rate as
        # `value`, so it rides along: this rank's evaluation with weights and reference tables crossing PCIe inside the timed region
        'value_incl_h2d_of_inputs': value_host,
        'rccl': rccl,
    }
    if mixed and placements:
        res['mixed_placement'] = placements[-1]      # (decision 1 = census of the CU pairs; 2 = tickets: shared GPU or another pair mapping, ~10 % slower; None = a launch per build)
    res.update(res_extra)
    if not a.no_cpu_baseline and world == 1:          # (the contract: the CPU baseline leg runs on rank 0 at N = 1 only)
        build_of = faults_of = None
        if mixed:
            rs_ = [builds.resolve_mode(m) for m in modes]
            build_of, faults_of = [r[0] for r in rs_], [list(r[1]) for r in rs_]
        cb, fit_cpu, ls_cpu, n_chk = cpu_baseline(w_host.numpy(), wl['hidden'], ref_host, moe, faults_of, build_of)
        res['cpu_baseline'] = cb
        fit_gpu = fit.cpu().numpy()[:n_chk]
        rel = np.abs(fit_gpu - fit_cpu) / np.abs(fit_cpu)
        res['parity_vs_cpu_port'] = {'max_rel_fitness': float(rel.max()), 'episodes_bit_identical': int((fit_gpu == fit_cpu).sum()), 'episodes_checked': int(n_chk),
                                     'lengths_equal': bool((ls.cpu().numpy()[:n_chk] == ls_cpu).all())}
        port_ = cb.get('port', cb)
        if 'pinned_sample' in port_:
            pick_, fp_ = port_.pop('pinned_sample')
            fg_ = fit.cpu().numpy()[np.asarray(pick_, dtype=int)]
            res['parity_vs_cpu_port']['max_rel_fitness_vs_pinned_oracle'] = float(np.max(np.abs(fg_ - np.asarray(fp_)) / np.abs(np.asarray(fp_))))
            res['parity_vs_cpu_port']['pinned_oracle_note'] = ('%d episodes against the oracle flavour that is bit-identical to the reference binary (glibc libm); the 0.0 above is against the '
                                                              'same-libm flavour, which shares sin / cos / tan / pow / atan with the kernels' % len(pick_))
    return res


if __name__ == '__main__':
    main()
