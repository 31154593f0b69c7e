"""Device-side replay buffers and the reference's CSV log rows.

  replay_buffer.Replay_Buffer        alg/replay_buffer.py:1-37     ring of transitions, uniform sample_batch
  replay_buffer_dual.Replay_Buffer   alg/replay_buffer_dual.py:1-63  two rings split by an "is_bad" flag
                                     (particle: scenario.collisions != 0, train_onpolicy.py:356)
  log.csv / log_century.csv columns  alg/train_onpolicy.py:200-221, :399-429
Transitions are the columns ParticleRollout / CheckersRollout export (numpy=False): each ring stores one tensor per
column, so adding a whole vectorised rollout is a handful of index_copy calls and sampling is one gather.
"""
import torch


class DeviceReplayBuffer(object):
    """replay_buffer.Replay_Buffer on the device: capacity `size` transitions, overwrite oldest when full
    (replay_buffer.py:11-16), sample_batch(size) = everything if len <= size else `size` distinct uniformly
    (:28-37)."""

    def __init__(self, size=int(1e6), device="cuda:0"):
        self.maxsize = int(size)
        self.device = torch.device(device)
        self.cols = None
        self.idx = 0
        self.len = 0

    def __len__(self):
        return self.len

    def add(self, cols):
        """cols: dict of tensors with a common leading batch dim (B transitions, in order)."""
        B = next(iter(cols.values())).shape[0]
        if B == 0:
            return
        if self.cols is None:
            self.cols = {k: torch.zeros((self.maxsize,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device)
                         for k, v in cols.items()}
        if B > self.maxsize:                       # only the last maxsize survive, as sequential adds would leave
            cols = {k: v[B - self.maxsize:] for k, v in cols.items()}
            self.idx = (self.idx + B - self.maxsize) % self.maxsize
            B = self.maxsize
        pos = (self.idx + torch.arange(B, device=self.device)) % self.maxsize
        for k, v in cols.items():
            self.cols[k].index_copy_(0, pos, v.to(self.device))
        self.idx = (self.idx + B) % self.maxsize
        self.len = min(self.len + B, self.maxsize)

    def _take(self, index):
        return {k: v[index] for k, v in self.cols.items()}

    def all(self):
        return self._take(torch.arange(self.len, device=self.device))

    def sample_batch(self, size, generator=None):
        if self.len <= size:
            return self.all()
        pick = torch.randperm(self.len, generator=generator, device=self.device)[:size]
        return self._take(pick)

    def sample_n(self, n, generator=None):
        n = min(int(n), self.len)
        pick = torch.randperm(self.len, generator=generator, device=self.device)[:n]
        return self._take(pick)


def _cat(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return {k: torch.cat([a[k], b[k]], dim=0) for k in a}


class DeviceDualReplayBuffer(object):
    """replay_buffer_dual.Replay_Buffer: memory_1 holds transitions of "bad" episodes, memory_2 the others; a batch
    takes half from each when both have enough, else everything from the smaller one and the remainder from the larger
    (replay_buffer_dual.py:40-63)."""

    def __init__(self, size=int(5e4), device="cuda:0"):
        self.mem1 = DeviceReplayBuffer(size, device)
        self.mem2 = DeviceReplayBuffer(size, device)

    def add(self, cols, is_bad):
        """is_bad: bool [B] per transition (all transitions of an episode carry the episode's flag)."""
        bad = torch.as_tensor(is_bad, device=self.mem1.device).bool()
        if bool(bad.any()):
            self.mem1.add({k: v[bad] for k, v in cols.items()})
        if bool((~bad).any()):
            self.mem2.add({k: v[~bad] for k, v in cols.items()})

    def sample_batch(self, size, generator=None):
        half = int(size / 2.0)
        n1, n2 = len(self.mem1), len(self.mem2)
        e1 = self.mem1.all() if n1 else None
        e2 = self.mem2.all() if n2 else None
        if half <= n1 and half > n2:
            return _cat(self.mem1.sample_n(min(n1, size - n2), generator), e2)
        if half > n1 and half <= n2:
            return _cat(e1, self.mem2.sample_n(min(n2, size - n1), generator))
        if n1 < half and n2 < half:
            return _cat(e1, e2)
        return _cat(self.mem1.sample_n(half, generator), self.mem2.sample_n(half, generator))


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
