"""Device-side replay buffers and the reference's CSV log rows.

  replay_buffer.Replay_Buffer        alg/replay_buffer.py:1-37     ring of transitions, uniform sample_batch
  replay_buffer_dual.Replay_Buffer   alg/replay_buffer_dual.py:1-63  two rings split by an "is_bad" flag
                                     (particle: scenario.collisions != 0, train_onpolicy.py:356)
  log.csv / log_century.csv columns  alg/train_onpolicy.py:200-221, :399-429
Transitions are the columns ParticleRollout / CheckersRollout export (numpy=False): each ring stores one tensor per
column; adding a whole vectorised rollout is ONE launch (cm3_rows_scatter, csrc/batch.hip), sampling is one gather launch.
"""
import torch

from . import _lib
from ._lib import Cm3Error


class RingIndex(object):
    """The bookkeeping of replay_buffer.Replay_Buffer (alg/replay_buffer.py:11-16) without the data: where the next transitions
    go and how many are stored.  Host logic (tests/test_replay.py replays the reference's recorded memory with it)."""

    def __init__(self, size):
        self.maxsize = int(size)
        self.idx = 0
        self.len = 0

    def plan_add(self, n):
        """n sequential adds -> (skip, start, kept): the first `skip` of them are overwritten again by the later ones of the same
        batch (n > maxsize), the other `kept` land at ring positions start, start + 1, ... (mod maxsize)."""
        n = int(n)
        skip = max(0, n - self.maxsize)
        start = (self.idx + skip) % self.maxsize
        kept = n - skip
        self.idx = (self.idx + n) % self.maxsize
        self.len = min(self.len + n, self.maxsize)
        return skip, start, kept


def dual_take(n1, n2, size):
    """How many transitions replay_buffer_dual.Replay_Buffer.sample_batch(size) takes from memory_1 (bad) and memory_2 (good)
    holding n1 / n2 transitions (alg/replay_buffer_dual.py:40-63) -> (k1, all1, k2, all2): k sampled without replacement, or
    everything in storage order when the flag is set."""
    half = int(size / 2.0)
    if half <= n1 and half > n2:
        return min(n1, size - n2), False, n2, True
    if half > n1 and half <= n2:
        return n1, True, min(n2, size - n1), False
    if n1 < half and n2 < half:
        return n1, True, n2, True
    return half, False, half, False


class DeviceReplayBuffer(object):
    """replay_buffer.Replay_Buffer on the device: capacity `size` transitions, overwrite oldest when full
    (replay_buffer.py:11-16), sample_batch(size) = everything if len <= size else `size` distinct uniformly
    (:28-37).  One tensor per column; `add` is ONE launch for all columns (cm3_rows_scatter onto the ring positions),
    sampling ONE launch (cm3_rows_gather).  There is no host path: the buffer lives on an AMD GPU."""

    def __init__(self, size=int(1e6), device="cuda:0"):
        self.device = _lib.require_gpu(device)
        self.ring = RingIndex(size)
        self.cols = None

    maxsize = property(lambda self: self.ring.maxsize)
    idx = property(lambda self: self.ring.idx)
    len = property(lambda self: self.ring.len)

    def __len__(self):
        return self.ring.len

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _alloc(self, cols):
        if self.cols is None:
            self.cols = {k: torch.zeros((self.maxsize,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device) for k, v in cols.items()}
        elif set(cols) != set(self.cols):
            raise Cm3Error("replay columns changed: %s vs %s" % (sorted(cols), sorted(self.cols)))

    @staticmethod
    def _src(v, device):
        return v.to(device).contiguous()

    def add(self, cols):
        """cols: dict of tensors with a common leading batch dim (B transitions, in order)."""
        B = next(iter(cols.values())).shape[0]
        if B == 0:
            return
        self._alloc(cols)
        skip, start, kept = self.ring.plan_add(B)
        pairs = [(self.cols[k], self._src(v, self.device)[skip:]) for k, v in cols.items()]
        _lib.rows_scatter(pairs, kept, self._stream(), ring_start=start, ring_size=self.maxsize)

    def add_rollout(self, rollout):
        """All transitions of a ParticleRollout's continuous float32 collection, straight from the trajectory into the ring: ONE
        launch (cm3_transitions_gather_f32 with ring positions as its output rows) instead of an export followed by an add, and no
        intermediate copy of the phase.  The reference stores the same array as v_global and v_local (train_onpolicy.py:338): so does
        the ring (one tensor under both names)."""
        env = rollout.env
        B, N, L = rollout.T * env.E, env.n, env.L
        if B > self.maxsize:                                   # (more than the ring holds: the newest survive, as sequential adds leave)
            return self.add(rollout.as_reference_batch(numpy=False))
        if self.cols is None:
            z = lambda *shape, dt=torch.float32: torch.zeros((self.maxsize,) + shape, dtype=dt, device=self.device)   # noqa: E731
            st, nst = z(N, 4), z(N, 4)
            self.cols = dict(v_global=st, obs_others=z(N, L), v_local=st, actions=z(N, dt=torch.int32), reward=z(), reward_local=z(N),
                             v_global_next=nst, obs_others_next=z(N, L), v_local_next=nst, done=z(dt=torch.bool), goals=z(N, 2))
        elif self.cols["v_global"].data_ptr() != self.cols["v_local"].data_ptr():
            return self.add(rollout.as_reference_batch(numpy=False))      # (a ring that add() allocated: separate v_local storage)
        # (validate and launch FIRST, advance the ring after: an export that raises must not leave idx / len pointing at rows that
        # were never written -- ADVICE r5)
        rollout.export_into(self.cols, self.ring.idx, self.maxsize)
        self.ring.plan_add(B)

    def add_at(self, cols, dst_row, n_added):
        """Rows b with dst_row[b] >= 0 go to ring position dst_row[b] (int64 on the device; computed by the caller, e.g. the dual
        buffer's split); the ring advances by n_added sequential adds."""
        self._alloc(cols)
        B = next(iter(cols.values())).shape[0]
        pairs = [(self.cols[k], self._src(v, self.device)) for k, v in cols.items()]
        _lib.rows_scatter(pairs, B, self._stream(), dst_row=dst_row)
        self.ring.plan_add(n_added)

    def _take(self, index):
        index = index.to(torch.int64).contiguous()
        n = index.numel()
        out = {k: torch.empty((n,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device) for k, v in self.cols.items()}
        if n:
            _lib.rows_gather([(out[k], self.cols[k]) for k in self.cols], n, index, self._stream())
        return out

    def all(self):
        """Everything stored, in storage order: VIEWS of the ring (no copy; the next add() overwrites them in place -- internal use,
        sample_batch() copies)."""
        return {k: v[:self.len] for k, v in self.cols.items()}

    def sample_batch(self, size, generator=None):
        if self.len <= size:
            # a COPY: all() hands out views of the ring, which the next add() overwrites in place -- the reference's
            # np.array(self.memory) is a copy, and a learner may still hold the batch when the next chunk is added
            return {k: v.clone() for k, v in self.all().items()}
        return self.sample_n(size, generator)

    def sample_n(self, n, generator=None):
        from .rollout import sample_distinct
        n = min(int(n), self.len)
        return self._take(sample_distinct(self.len, n, 1, generator, self.device)[0])


def _cat(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return {k: torch.cat([a[k], b[k]], dim=0) for k in a}


class DeviceDualReplayBuffer(object):
    """replay_buffer_dual.Replay_Buffer: memory_1 holds transitions of "bad" episodes, memory_2 the others; a batch
    takes half from each when both have enough, else everything from the smaller one and the remainder from the larger
    (replay_buffer_dual.py:40-63).  `add` splits a whole batch by its flag column on the device: ranks by a prefix sum, ring
    positions for both memories, one scatter launch per memory -- one host read (the number of bad transitions)."""

    def __init__(self, size=int(5e4), device="cuda:0"):
        self.mem1 = DeviceReplayBuffer(size, device)
        self.mem2 = DeviceReplayBuffer(size, device)

    def add(self, cols, is_bad):
        """is_bad: bool [B] per transition (all transitions of an episode carry the episode's flag; particle:
        scenario.collisions != 0, train_onpolicy.py:356)."""
        dev = self.mem1.device
        bad = torch.as_tensor(is_bad, device=dev).bool()
        B = bad.numel()
        if B == 0:
            return
        n1 = int(bad.sum())            # the one host read
        for mem, flag, n in ((self.mem1, bad, n1), (self.mem2, ~bad, B - n1)):
            if n == 0:
                continue
            rank = torch.cumsum(flag, 0) - 1                                   # order among the transitions of this memory
            keep = flag & (rank >= n - mem.maxsize)                             # (the earlier ones would be overwritten by this very batch)
            pos = torch.where(keep, (rank + mem.ring.idx) % mem.maxsize, torch.full_like(rank, -1))
            mem.add_at(cols, pos.contiguous(), n)

    def sample_batch(self, size, generator=None):
        n1, n2 = len(self.mem1), len(self.mem2)
        k1, all1, k2, all2 = dual_take(n1, n2, size)
        a = (self.mem1.all() if all1 else self.mem1.sample_n(k1, generator)) if n1 and k1 else None
        b = (self.mem2.all() if all2 else self.mem2.sample_n(k2, generator)) if n2 and k2 else None
        if a is None or b is None:          # (_cat would hand out the one side as is: ring views when it is "everything")
            one = a if b is None else b
            return None if one is None else {k: v.clone() for k, v in one.items()}
        return _cat(a, b)


def off_policy_batches(rollout, buffer, n_chunks, batch_size=128, generator=None, **collect_kwargs):
    """The off-policy cadence of alg/train_offpolicy.py:309-356 (the trainer the reference's README routes Checkers to): every
    transition goes into a PERSISTENT replay buffer (:337-346), and every `steps_per_train` env steps (:348) a batch is sampled
    from it (:350) for a training step.  Vectorised: `rollout` (a ParticleRollout / CheckersRollout in continuous mode with
    n_ticks = steps_per_train) collects one chunk of ticks for all its envs, all transitions of the chunk are added to `buffer`
    (DeviceReplayBuffer: ONE launch) and one batch is sampled -- yielded as device columns; `collect_kwargs` go to
    rollout.collect() (policy=..., epsilon=..., goals=... for Checkers).  The buffer outlives the chunks: old transitions are
    overwritten only when it is full (replay_buffer.py:11-16)."""
    dual = isinstance(buffer, DeviceDualReplayBuffer)
    if dual and not hasattr(rollout, "episode_is_bad"):
        raise Cm3Error("a dual replay buffer splits by the episodes' flag (scenario.collisions != 0, train_onpolicy.py:356): that is a "
                       "ParticleRollout's; use a DeviceReplayBuffer here")
    for _ in range(int(n_chunks)):
        rollout.collect(**collect_kwargs)
        if dual:
            # every transition carries the flag of the episode it belongs to: the flag is known at the tick that ENDS the episode
            # (ParticleRollout.episode_is_bad); transitions of episodes still running at the chunk's end take the flag so far
            cols = rollout.as_reference_batch(numpy=False)
            buffer.add({k: v.contiguous() for k, v in cols.items()}, _transition_flags(rollout))
        elif hasattr(rollout, "export_into") and rollout.auto_reset and rollout.state.dtype == torch.float32 and hasattr(buffer, "add_rollout"):
            buffer.add_rollout(rollout)                                   # export + add in one launch
        else:
            cols = rollout.as_reference_batch(numpy=False)
            buffer.add({k: v.contiguous() for k, v in cols.items()})
        yield buffer.sample_batch(batch_size, generator=generator)


def _transition_flags(rollout):
    """bool [T * E] in the order of as_reference_batch() of a continuous collection (time-major): transition (t, e) is "bad" when the
    episode it belongs to has collided by its end inside this chunk -- or by the chunk's last tick, for an episode that runs on."""
    coll = rollout.collisions                       # int32 [T, E]: scenario.collisions after every tick (before a same-launch reset)
    if coll is None:
        raise Cm3Error("the dual buffer's flag needs the per-tick collision counts (record_collisions=True)")
    done = rollout.done.bool()
    T = coll.shape[0]
    bad_at_end = (coll != 0)
    flag = torch.empty_like(done)
    run = bad_at_end[T - 1].clone()                 # episodes still running at the end of the chunk: their count so far
    for t in range(T - 1, -1, -1):
        run = torch.where(done[t], bad_at_end[t], run)
        flag[t] = run
    return flag.reshape(-1)


class CsvLog(object):
    """The reference's two CSV logs with identical headers and row formats (train_onpolicy.py:200-221, :399-429)."""

    def __init__(self, log_path, century_path, n_agents):
        self.log_path, self.century_path, self.n = log_path, century_path, int(n_agents)
        header = "Step,Episode,r_global" + "".join(",r_%d" % i for i in range(self.n)) + "\n"
        header_c = ("Step,Century,r_global_avg" + "".join(",r_avg_%d" % i for i in range(self.n)) + ",r_global_eval"
                    + "".join(",r_eval_%d" % i for i in range(self.n)) + ",r_eval_local,t_env (s),t_train(s)\n")
        with open(self.log_path, "w") as f:
            f.write(header)
        with open(self.century_path, "w") as f:
            f.write(header_c)

    @staticmethod
    def episode_row(step, idx_episode, reward_global, reward_local):
        return "%d,%d,%.2f," % (step, idx_episode, reward_global) + ",".join("{:.2f}".format(v) for v in reward_local) + "\n"

    @staticmethod
    def century_row(step, idx_episode, r_global_avg, r_local_avg, r_global_eval, r_local_eval, t_env, t_train):
        s = "%d,%d,%.2f," % (step, idx_episode, r_global_avg)
        s += ",".join("{:.2f}".format(v) for v in r_local_avg)
        s += ",%.2f," % r_global_eval
        s += ",".join("{:.2f}".format(v) for v in r_local_eval)
        s += ",%.2f,%d,%d\n" % (float(sum(r_local_eval)), int(t_env), int(t_train))
        return s

    def log_episode(self, *a):
        with open(self.log_path, "a") as f:
            f.write(self.episode_row(*a))

    def log_century(self, *a):
        with open(self.century_path, "a") as f:
            f.write(self.century_row(*a))
