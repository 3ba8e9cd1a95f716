"""Device-resident replay rings behind the reference's `ReplayMemory` call shapes (base/core/replay_memory.py:12-103).

The reference keeps a Python list of `Transition` tuples per buffer (shared buffer of the RL learner, `buffer` and
`critical_buffer` of every GeneticAgent) and appends to it one tuple per env step (base/core/agent.py:101-112).  Here a
buffer is one f32 tensor `rows[capacity, 20]` in HBM -- the row layout the rollout kernel writes,
(obs[7], action[3], next_obs[7], reward, done, cost) -- and whole episodes are appended by ONE scatter kernel
(`serl_replay_scatter`, include/serl_amd.h): every stored episode of a generation to the shared ring, to its agent's
ring, and its cost-flagged rows compacted into the agent's critical ring.  Nothing crosses to the host per step.

Index semantics are the reference's, slot for slot, and every random draw is made on the host from the same python
`random` stream (`random.sample` / `random.shuffle` consume the generator as a function of the list LENGTH only, so
drawing over `range(len)` reproduces the reference's choices): `sample`, `sample_from_latest`, `get_latest`,
`add_content_of`, `add_latest_from`, `shuffle`, `reset` select the same transitions as the list-backed class.
"""
import ctypes, random
import numpy as np
import torch

ROW = 20          # obs7 | action3 | next_obs7 | reward | done | cost


def sample_many(n, k, calls, rng=random):
    """`calls` consecutive `rng.sample(range(n), k)` -> int32 [calls, k], consuming the generator exactly like the python
    calls would.  One call at a time costs ~50 - 100 us of interpreter time (a distillation draws 750 minibatches, the
    distance sort of an epoch ~1 000); here the raw 32-bit outputs of the Mersenne Twister are taken in bulk (getrandbits),
    CPython's selection algorithm is replayed by a few lines of C (serl_host_sample_slots), and the generator is rewound
    and advanced by the number of outputs the python calls would have consumed."""
    from . import _capi
    out = np.zeros((calls, k), dtype=np.int32)
    if calls == 0 or k == 0:
        return out
    if calls >= 2 and all(hasattr(rng, f) for f in ('getstate', 'setstate', 'getrandbits')):
        L = _capi.lib()
        m = int(calls * k * 2.2) + 4096
        state = rng.getstate()
        while True:
            words = np.frombuffer(rng.getrandbits(32 * m).to_bytes(4 * m, 'little'), dtype='<u4')
            used = int(L.serl_host_sample_slots(words.ctypes.data, m, int(n), int(k), int(calls), out.ctypes.data, int(k)))
            rng.setstate(state)
            if used == -1:
                m *= 2
                continue
            break
        if used >= 0:
            if used:
                rng.getrandbits(32 * used)
            return out
    for c in range(calls):                      # a single call, or a foreign generator
        out[c] = rng.sample(range(n), k)
    return out


class DeviceReplay:
    """Uniform replay ring on the GPU with the interface of replay_memory.ReplayMemory."""

    def __init__(self, capacity, device, engine=None):
        self.capacity = int(capacity)
        self.device = torch.device(device)
        self.engine = engine
        self.rows = torch.zeros(self.capacity, ROW, dtype=torch.float32, device=self.device)
        self.position = 0
        self.size = 0

    def __len__(self):
        return self.size

    def reset(self):
        self.position, self.size = 0, 0

    # ---- appends ----------------------------------------------------------------------------------------------
    def _advance(self, n):
        """book-keeping of n sequential add() calls (replay_memory.py:21-31); -> (first slot, rows to skip)"""
        first = self.position
        self.size = min(self.capacity, self.size + n)
        self.position = (self.position + n) % self.capacity
        return first, max(0, n - self.capacity)

    def add(self, obs, action, next_obs, reward, done, cost=0.0):
        """one tuple from the host (the reference's call shape; the batched path is `append_rows` / `scatter_episodes`)"""
        r = np.concatenate([np.asarray(obs, np.float32).reshape(-1), np.asarray(action, np.float32).reshape(-1),
                            np.asarray(next_obs, np.float32).reshape(-1), [np.float32(reward)], [np.float32(done)], [np.float32(cost)]])
        self.append_rows(torch.from_numpy(r.astype(np.float32))[None])

    def append_rows(self, rows):
        """rows f32 [n, 20] (device or host), appended in order like n calls of add()"""
        rows = torch.as_tensor(rows, dtype=torch.float32).to(self.device)
        n = rows.shape[0]
        if n == 0:
            return
        first, skip = self._advance(n)
        slots = (first + torch.arange(skip, n, device=self.device)) % self.capacity
        self.rows.index_copy_(0, slots, rows[skip:])

    # ---- views in the reference's order ------------------------------------------------------------------------
    def latest_slots(self, latest):
        """physical slots of ReplayMemory.get_latest(latest) (replay_memory.py:42-56), most recent last"""
        n, p, cap = self.size, self.position, self.capacity
        mem = np.arange(n, dtype=np.int64)
        latest = int(latest)
        if cap < latest:
            out = np.concatenate([mem[p:], mem[:p]])
        elif n < cap:
            out = mem[-latest:] if latest else mem
        elif p >= latest:
            out = mem[:p][-latest:] if latest else mem[:p]
        else:
            out = np.concatenate([mem[-latest + p:], mem[:p]])
        return out

    def get_latest(self, latest):
        return self.rows[torch.from_numpy(self.latest_slots(latest)).to(self.device)]

    def add_content_of(self, other):
        self.append_rows(other.get_latest(self.capacity))

    def add_latest_from(self, other, latest):
        self.append_rows(other.get_latest(latest))

    def shuffle(self, rng=random):
        """random.shuffle(self.memory): the same permutation, applied to the rows"""
        perm = list(range(self.size))
        rng.shuffle(perm)
        if self.size:
            self.rows[:self.size] = self.rows[torch.as_tensor(perm, dtype=torch.int64, device=self.device)]

    # ---- sampling ----------------------------------------------------------------------------------------------
    @staticmethod
    def split(rows):
        """rows [B, 20] -> (state [B,7], action [B,3], next_state [B,7], reward [B,1], done [B,1]) like ReplayMemory.sample"""
        return rows[:, 0:7], rows[:, 7:10], rows[:, 10:17], rows[:, 17:18], rows[:, 18:19]

    def sample_slots(self, batch_size, rng=random):
        return rng.sample(range(self.size), batch_size)

    def sample(self, batch_size, rng=random):
        idx = self.sample_slots(batch_size, rng)
        return self.split(self.rows[torch.as_tensor(idx, dtype=torch.int64, device=self.device)])

    def sample_from_latest(self, batch_size, latest, rng=random):
        slots = self.latest_slots(latest)
        pick = rng.sample(range(len(slots)), batch_size)
        return self.split(self.rows[torch.from_numpy(slots[np.asarray(pick, dtype=np.int64)]).to(self.device)])


class _Job(ctypes.Structure):
    _fields_ = [('ring', ctypes.c_void_p), ('capacity', ctypes.c_int32), ('position', ctypes.c_int32),
                ('episode', ctypes.c_int32), ('length', ctypes.c_int32), ('cost_only', ctypes.c_int32), ('skip', ctypes.c_int32)]


def scatter_episodes(engine, staged, jobs):
    """Append stored episodes to rings with one kernel launch.

    staged : f32 [E, T, 20] device tensor the rollout kernel wrote (`transitions`)
    jobs   : list of (ring: DeviceReplay, episode index, n_steps, cost_only, n_rows) in the order the reference would
             have add()-ed them (agent.py:101-112: shared buffer, agent.buffer, agent.critical_buffer per step; rings
             are independent, so per-ring order is what matters); n_rows = n_steps, or the episode's cost-step count
             for cost_only jobs (known from the rollout's `cost_steps`)."""
    from . import _capi
    if not jobs:
        return
    arr = (_Job * len(jobs))()
    # rows a later job of this launch (or the tail of the same job) overwrites are never written: a row survives iff fewer
    # than `capacity` rows follow it on its ring -- the state n sequential add() calls would leave, without write races
    total = {}
    for ring, e, n, cost_only, n_rows in jobs:
        total[id(ring)] = total.get(id(ring), 0) + int(n_rows)
    seen = {}
    for j, (ring, e, n, cost_only, n_rows) in enumerate(jobs):
        start = seen.get(id(ring), 0)
        seen[id(ring)] = start + int(n_rows)
        dead = total[id(ring)] - ring.capacity - start          # ranks of this job below `dead` are overwritten later
        first, _ = ring._advance(int(n_rows))
        arr[j] = _Job(ring.rows.data_ptr(), ring.capacity, first, int(e), int(n), int(bool(cost_only)), int(min(max(dead, 0), int(n_rows))))
    dev = staged.device
    buf = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _capi.check(engine.lib.serl_replay_scatter(engine.ctx, staged.data_ptr(), int(staged.shape[1]), buf.data_ptr(), len(jobs), stream),
                'serl_replay_scatter')


def store_episodes(engine, staged, items, replay_buffer=None, counters=None, state_dim=7, action_dim=3):
    """The buffer side of Agent.evaluate for several stored episodes at once (agent.py:101-125).

    items : [(agent, episode index in `staged`, n_steps, n_cost_steps)] in the reference's order (member by member, then
            the RL actor).  Every episode goes to `replay_buffer` (the learner's shared buffer) and to agent.buffer, its
            cost-flagged rows to agent.critical_buffer; num_frames / gen_frames advance by the steps, num_episodes by one
            per episode.  Buffers that are DeviceReplay rings are filled by ONE serl_replay_scatter launch; any other
            object with the reference's `add(*transition)` is fed tuple by tuple from a host copy (compatibility with
            the reference's list-backed ReplayMemory).  Rows are (obs S, action A, next_obs S, reward, done, cost); the
            device rings hold the attitude task's 20-float rows (S = 7, A = 3) only."""
    S, A = int(state_dim), int(action_dim)
    jobs, host = [], []
    for agent, e, n, nc in items:
        n, nc = int(n), int(nc)
        for ring, cost_only, rows in ((replay_buffer, False, n), (getattr(agent, 'buffer', None), False, n),
                                      (getattr(agent, 'critical_buffer', None), True, nc)):
            if ring is None:
                continue
            if isinstance(ring, DeviceReplay):
                if (S, A) != (7, 3):
                    raise NotImplementedError('DeviceReplay rings hold the 20-float rows of the attitude task; use list-backed '
                                              'buffers (ReplayMemory) for the other env configurations')
                jobs.append((ring, e, n, cost_only, rows))
            else:
                host.append((ring, e, n, cost_only))
        if counters is not None:
            counters['num_frames'] = counters.get('num_frames', 0) + n
            counters['gen_frames'] = counters.get('gen_frames', 0) + n
            counters['num_episodes'] = counters.get('num_episodes', 0) + 1
    if jobs:
        scatter_episodes(engine, staged, jobs)
    cache = {}
    for ring, e, n, cost_only in host:
        if e not in cache:
            cache[e] = staged[e, :n].cpu().numpy()
        for r in cache[e]:
            if cost_only and not r[2 * S + A + 2]:
                continue
            ring.add(r[0:S].astype(np.float64), r[S:S + A], r[S + A:2 * S + A].astype(np.float64), float(r[2 * S + A]), float(r[2 * S + A + 1]))
