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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from serl_amd import distributed as sd
    res = sd.evaluate_pop_sharded(_local_eval, POP, NE)
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


def test_member_blocks_cover_population():
    from serl_amd import distributed as sd
    for pop in (1, 5, 50, 512, 2048):
        for ws in (1, 2, 4, 8):
            blocks = [sd.member_block(pop, ws, r) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == pop
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(ws - 1))
