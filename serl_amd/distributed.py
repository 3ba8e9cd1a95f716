"""Multi-GPU population evaluation: one process per GPU, members sharded in contiguous blocks, one
all-gather of the per-member result rows (RCCL over xGMI when the backend is `nccl`; `gloo` on CPU
for tests).  No collective touches the data path -- episodes are independent.

Results do not depend on the partition: there is no cross-episode reduction inside the kernel, the
per-member mean over evals is computed from the gathered rows identically on every rank, and index
selection (champion argmax, SSNE argsort) is replicated from the same gathered vector.
"""
import numpy as np
import torch
import torch.distributed as dist

ROW = 6   # fitness, return, smoothness, length_t, length_steps, cost_steps  (f64)


def member_block(pop, world_size, rank):
    """Contiguous member block of `rank`: members [lo, hi).  pop need not divide world_size."""
    per = (pop + world_size - 1) // world_size
    lo = min(rank * per, pop)
    return lo, min(lo + per, pop)


def gather_rows(local_rows, pop, world_size, rank, device=None):
    """local_rows: f64 [num_evals, hi-lo, ROW] of this rank's members -> f64 [num_evals, pop, ROW] on
    every rank (padded all_gather; one collective per population evaluation)."""
    per = (pop + world_size - 1) // world_size
    num_evals = local_rows.shape[0]
    buf = torch.zeros(num_evals, per, ROW, dtype=torch.float64, device=device or local_rows.device)
    buf[:, :local_rows.shape[1]] = local_rows
    if not dist.is_initialized():        # (a one-rank group still goes through the collective: the same code path as N ranks)
        return buf[:, :pop].clone()
    out = [torch.empty_like(buf) for _ in range(world_size)]
    dist.all_gather(out, buf)
    return torch.cat(out, dim=1)[:, :pop].contiguous()


def evaluate_pop_sharded(evaluate_local, pop, num_evals, device=None):
    """evaluate_local(lo, hi) -> dict(fitness, returns, smoothness, length_t, length_steps, cost_steps),
    each array [num_evals, hi-lo], for this rank's member block.  Returns the gathered dict with
    pop_fitness / champion / worst computed identically on every rank."""
    ws = dist.get_world_size() if dist.is_initialized() else 1
    rk = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = member_block(pop, ws, rk)
    if hi > lo:
        r = evaluate_local(lo, hi)
        rows = np.stack([np.asarray(r[k], dtype=np.float64) for k in
                         ('fitness', 'returns', 'smoothness', 'length_t', 'length_steps', 'cost_steps')], axis=-1)
    else:
        rows = np.zeros((num_evals, 0, ROW))
    g = gather_rows(torch.as_tensor(rows).to(device) if device is not None else torch.as_tensor(rows), pop, ws, rk,
                    device=device).cpu().numpy()
    fitness = g[..., 0]
    pop_fitness = np.mean(fitness, axis=0)
    return dict(fitness=fitness, returns=g[..., 1], smoothness=g[..., 2], length_t=g[..., 3],
                length_steps=g[..., 4].astype(np.int32), cost_steps=g[..., 5].astype(np.int32),
                pop_fitness=pop_fitness, champion=int(np.argmax(pop_fitness)), worst=int(np.argmin(pop_fitness)),
                block=(lo, hi))


def gather_stored_episodes(staged_local, steps_local, cost_local, pop, world_size, rank):
    """The stored episode of every member (agent.py:239: the last of its num_evals) on every rank.

    With the population sharded by member, the transitions of member m's stored episode exist on its owner only, but the
    SSNE epoch is replicated on all ranks (same host RNG streams, same index decisions) and proximal / safe mutation and
    the distance-sorted crossover read the members' replay rings -- so the stored rows travel: ONE all_gather of
    [ceil(pop / world), T, 20] f32 per generation (160 KB per member at 2 001 steps: 8 MB at pop = 50, 82 MB at pop = 512,
    a few ms over xGMI), after which every rank appends all `pop` episodes to its rings in member order
    (replay.store_episodes) and the rings are identical everywhere.

    staged_local f32 [n_local, T, 20] (rows the rollout kernel wrote for this rank's members' stored episodes),
    steps_local / cost_local int [n_local] -> (staged [pop, T, 20], steps [pop], cost_steps [pop])."""
    per = (pop + world_size - 1) // world_size
    n_local, T, W = staged_local.shape
    dev = staged_local.device
    buf = torch.zeros(per, T, W, dtype=staged_local.dtype, device=dev)
    buf[:n_local] = staged_local
    meta = torch.zeros(per, 2, dtype=torch.int64, device=dev)
    meta[:n_local, 0] = torch.as_tensor(steps_local, dtype=torch.int64, device=dev)
    meta[:n_local, 1] = torch.as_tensor(cost_local, dtype=torch.int64, device=dev)
    if not dist.is_initialized():
        return buf[:pop], meta[:pop, 0].cpu().numpy(), meta[:pop, 1].cpu().numpy()
    bufs = [torch.empty_like(buf) for _ in range(world_size)]
    metas = [torch.empty_like(meta) for _ in range(world_size)]
    dist.all_gather(bufs, buf)
    dist.all_gather(metas, meta)
    allb, allm = torch.cat(bufs)[:pop], torch.cat(metas)[:pop]
    return allb, allm[:, 0].cpu().numpy(), allm[:, 1].cpu().numpy()
