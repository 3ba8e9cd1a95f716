#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the population-rollout fitness evaluation (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A "step" is one population evaluation: pop members x num_evals episodes x 8 001 env steps of the
PH-LAB nominal attitude-tracking task (t_max = 80 s), SERL50 actor shape (7-32-32-32-32-3, tanh),
weights and reference tables already resident in HBM.  Weak scaling: every GPU evaluates its own
block of `pop` members (shipped SERL50 actors, tiled with seeded noise beyond the first 50) and the
per-member result rows are all-gathered.  Prints ONE JSON line on rank 0.
"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 48.0          # algorithmic HBM bytes per env-step: ref[k] 3 f64 read + u[k] 3 f64 written
F_ALG = 24059.0       # f64 arithmetic instructions the reference retires per env-step
HBM_PEAK = 8.0e12     # B/s, MI355X_MICROARCH.md chip table (6.29e12 measured)
FP64_PEAK = 78.6e12   # FLOP/s vector f64 (half the 157.3 TF f32 vector rate)


def make_population(pop, rank, seed=7):
    import torch
    base = np.load(os.path.join(ROOT, 'tests', 'golden', 'actors.npz'))['serl50']   # [50, 3715] shipped actors
    idx = (np.arange(pop) + rank * pop) % len(base)
    w = torch.from_numpy(base[idx].copy())
    if pop > len(base) or rank > 0:
        from serl_amd import NetSpec
        spec = NetSpec(7, 3, 32, 3, 'tanh')
        g = torch.Generator().manual_seed(seed + rank)
        for off, n in spec.genome_segments():
            w[:, off:off + n] += 0.01 * torch.randn(pop, n, generator=g)
        w[:min(pop, len(base)) if rank == 0 else 0] = torch.from_numpy(base[idx[:min(pop, len(base))]]) if rank == 0 else w[:0]
    return w


def cpu_baseline(w, ref, num_evals, max_episodes=None):
    """The CPU port (oracle/rollout_ref.c) on all host cores, same workload (bounded sample)."""
    from oracle import rollout as R
    cores = os.cpu_count() or 1
    pop = w.shape[0]
    E = pop * num_evals
    moe = np.repeat(np.arange(pop, dtype=np.int32), num_evals)
    n = E if max_episodes is None else min(E, max_episodes)
    net = dict(state_dim=7, action_dim=3, hidden=32, num_layers=3, activation='tanh')
    R.rollout(w, net, moe[:cores], ref[:cores], t_max=80.0, threads=cores)       # warm-up (page-in, lib build)
    t0 = time.perf_counter()
    o = R.rollout(w, net, moe[:n], ref[:n], t_max=80.0, threads=cores)
    dt = time.perf_counter() - t0
    steps = int(o['length_steps'].sum())
    return dict(value=steps / dt, unit='env-steps/s', cores=cores, kind='port',
                sample='%d of %d episodes (8001 steps each) of the same workload, C restatement '
                       '(oracle/rollout_ref.c) on %d threads, %.1f s wall' % (n, E, cores, dt)), o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--pop', type=int, default=50, help='members per GPU')
    ap.add_argument('--num-evals', type=int, default=3)
    ap.add_argument('--lanes', type=int, default=0, help='episodes per wavefront (0 = auto)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import serl_amd
    from serl_amd import refsignals, metrics, distributed as sd

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % a.gpus

    spec = serl_amd.NetSpec(7, 3, 32, 3, 'tanh')
    pop, ne = a.pop, a.num_evals
    E = pop * ne
    w_host = make_population(pop, rank)
    ref_host = refsignals.synthetic_reference_tables(E, ne, 80, seed=7 + 100000 * rank)
    eng = serl_amd.RolloutEngine(local)
    w = w_host.to(dev)
    ref = torch.from_numpy(ref_host).to(dev)
    moe = np.repeat(np.arange(pop, dtype=np.int32), ne)
    T = ref.shape[1]

    def one_step():
        """one population evaluation on this rank + the fitness all-gather"""
        out = eng.rollout(w, spec, moe, ref, t_max=80.0, traces='actions', lanes_per_wave=a.lanes, sync=False)
        ls = out['length_steps']
        sm = metrics.calc_smoothness(out['actions'], ls)                          # a11, on device
        fit = out['fitness']
        rows = torch.stack([fit, fit, sm, out['length_t'], ls.double(), out['cost_steps'].double()], -1)
        rows = rows.view(pop, ne, sd.ROW).transpose(0, 1).contiguous()
        g = sd.gather_rows(rows, pop * world, world, rank, device=dev)
        pop_fitness = g[..., 0].mean(0)
        champion = int(torch.argmax(pop_fitness))
        return out, g, champion

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out, g, champion = one_step()
        kernel_ms.append(eng.kernel_ms())      # HIP events around the kernel, on its stream (blocks on the kernel)
    barrier()
    dt = time.perf_counter() - t0
    steps_local = int(out['length_steps'].abs().sum())
    tt = torch.tensor([dt, float(steps_local)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, steps_total = float(tmax[0]), float(tsum[1])
    else:
        steps_total = float(steps_local)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    value = steps_total * a.steps / dt
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
    # HBM traffic of one launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # collected and corrected as MI355X_MICROARCH.md prescribes); only quoted for the configuration it was measured on
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01_k_pmc_traffic.json')))
        if pop == 50 and ne == 3 and a.lanes == 0:
            traffic = pm['traffic_bytes_per_launch']
    except Exception:
        pass
    res = {
        'metric': 'env-steps/sec (whole node), pop-rollout eval, pop=%d nominal per GPU' % pop,
        'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'PH-LAB nominal h2000_v90, pop=%d (SERL50 actor 7-32x4-3 tanh) x num_evals=%d x 8001 steps '
                               '(t_max=80 s) per GPU; shipped SERL50 weights (tiled+noise beyond 50), seeded '
                               'smoothed-step references' % (pop, ne),
                   'pop_per_gpu': pop, 'num_evals': ne, 'episodes_per_gpu': E, 'steps_per_episode': T,
                   'lanes_per_wave': a.lanes, 'parallelism': 'member-sharded dp%d' % world},
        'kernel_ms': k_ms,
        'roofline': {'bound': 'hbm', 'achieved': ach_hbm / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                     'frac': ach_hbm / HBM_PEAK, 'traffic': traffic,
                     'note': 'path is instruction-issue / dependent-latency bound on four wavefronts per episode, not HBM-bound (DESIGN.md): '
                             'algorithmic traffic is 48 B per env-step; achieved = steps x 48 B / kernel time; traffic = bytes '
                             'per launch from profiles/r01_k_pmc_traffic.json (2 x FETCH_SIZE + WRITE_SIZE)'},
        'roofline_fp64': {'bound': 'valu-f64', 'achieved': ach_f64 / 1e12, 'peak': FP64_PEAK / 1e12, 'unit': 'TFLOP/s',
                          'frac': ach_f64 / FP64_PEAK},
        't_step_us': k_ms * 1e3 / T,
        'value_host_buffers': value_host,      # this rank, inputs crossing PCIe per evaluation (informational)
    }
    if not a.no_cpu_baseline:
        cb, o = cpu_baseline(w_host.numpy(), ref_host, ne)
        res['cpu_baseline'] = cb
        fit_gpu = out['fitness'].cpu().numpy()
        rel = np.abs(fit_gpu - o['fitness']) / np.abs(o['fitness'])
        res['parity_vs_cpu_port'] = {'max_rel_fitness': float(rel.max()),
                                     'lengths_equal': bool((out['length_steps'].cpu().numpy() == o['length_steps']).all())}
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
