"""Sharding of env instances over the GPUs of a node + the one collective the path needs.

Env instances are independent (nothing in environment.py / core.py / checkers.py is shared between
envs), so stepping needs NO exchange: rank r owns the contiguous block of global env ids
[base, base+count) and every RNG draw is keyed by the GLOBAL env id, which makes results identical
for any number of ranks.  The reference itself has no collective -- its only
concurrency is n_seeds independent processes (train_multiprocess.py:31-43).

The single batch-wide statistic the build defines is advantage normalisation
    A_hat = (A - mean) / (std + eps)      over the whole [world x T x E x N] batch.
(The reference's advantage, alg_credit.py:334-357, is un-normalised: this step is optional and off for
parity runs.)  It needs three numbers per rank -- (sum, sum of squares, count) in float64 -- exchanged
with ONE all-gather (RCCL over xGMI when the backend is "nccl" on ROCm; 24 bytes per rank, latency
bound) and combined in fixed rank order so every rank computes bit-identical statistics.
"""
import torch
import torch.distributed as dist


def shard_range(n_global, rank, world):
    """Contiguous block of global env ids owned by `rank`: (base, count).  Remainders go to the low ranks."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    q, r = divmod(int(n_global), int(world))
    count = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return base, count


def local_moments(x, valid=None):
    """float64 [3] = (sum, sum of squares, count) of the selected elements of x (same device as x)."""
    x64 = x.to(torch.float64)
    if valid is not None:
        v = valid.to(torch.float64)
        while v.dim() < x64.dim():
            v = v.unsqueeze(-1)
        v = v.expand_as(x64)
        return torch.stack([(x64 * v).sum(), (x64 * x64 * v).sum(), v.sum()])
    return torch.stack([x64.sum(), (x64 * x64).sum(),
                        torch.tensor(float(x64.numel()), dtype=torch.float64, device=x.device)])


def gather_moments(m, group=None):
    """THE collective of the path: this rank's float64 (sum, sum of squares, count) -> float64 [world * 3], the triples
    of all ranks in rank order (one all_gather_into_tensor of 24 bytes per rank; RCCL over xGMI with backend "nccl").
    Returns (parts, n_parts); without a process group (or with one rank) it is the identity.  A "gloo" group whose
    tensors live on the GPU is staged through the host (gloo has no device all-gather)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return m, 1
    world = dist.get_world_size(group)
    m = m.contiguous().view(-1)          # [3], or [n_segments * 3]: the rollouts of a phase travel in ONE collective
    if m.device.type == "cuda" and dist.get_backend(group) == "gloo":
        host = torch.empty(world * m.numel(), dtype=torch.float64)
        dist.all_gather_into_tensor(host, m.cpu(), group=group)
        return host.to(m.device), world
    parts = torch.empty(world * m.numel(), dtype=torch.float64, device=m.device)
    dist.all_gather_into_tensor(parts, m, group=group)
    return parts, world


def global_moments(x, valid=None, group=None):
    """(mean, std, count) over all ranks.  One all_gather_into_tensor of 3 float64 per rank; the per-rank
    triples are summed in rank order (deterministic, identical on every rank)."""
    m = local_moments(x, valid)
    gathered, world = gather_moments(m, group)
    if world > 1:
        parts = gathered.view(world, 3)
        tot = parts[0].clone()
        for r in range(1, world):
            tot = tot + parts[r]
    else:
        tot = m
    n = tot[2].clamp(min=1.0)
    mean = tot[0] / n
    var = (tot[1] / n - mean * mean).clamp(min=0.0)
    return mean, var.sqrt(), tot[2]


def normalize_advantages(adv, valid=None, eps=1e-8, group=None):
    """A_hat = (A - mean) / (std + eps) with batch-wide (all ranks) mean / std; invalid entries -> 0."""
    mean, std, _ = global_moments(adv, valid, group)
    # one expression for the host helper and cm3_normalize_* (csrc/advantage.hip): the float64 statistics are rounded to
    # the working precision once, then one IEEE subtraction and one division per element
    out = (adv - mean.to(adv.dtype)) / (std + eps).to(adv.dtype)
    if valid is not None:
        v = valid
        while v.dim() < out.dim():
            v = v.unsqueeze(-1)
        out = torch.where(v.expand_as(out), out, torch.zeros_like(out))
    return out


def returns_to_go(reward, done, gamma=0.99, bootstrap=None):
    """Discounted return G_t = r_t + gamma * (1 - done_t) * G_{t+1} over a time-major trajectory
    (reward [T, ...], done [T, E] broadcast over trailing dims).  Stays on the trajectory's device."""
    T = reward.shape[0]
    out = torch.empty_like(reward)
    nxt = torch.zeros_like(reward[0]) if bootstrap is None else bootstrap
    nd = (~done.bool()).to(reward.dtype)
    while nd.dim() < reward.dim():
        nd = nd.unsqueeze(-1)
    for t in range(T - 1, -1, -1):
        nxt = reward[t] + gamma * nd[t] * nxt
        out[t] = nxt
    return out


def normalized_returns(reward, done, valid=None, gamma=0.99, eps=1e-8, group=None, normalize=True):
    """GPU path of the advantage-normalisation step over a device trajectory.

    reward [T,E] or [T,E,C] (float32/float64, contiguous, on the GPU), done / valid uint8-or-bool [T,E].
    Two launches and ONE collective per rollout: cm3_returns_moments_* (discounted returns + this rank's
    float64 moments, deterministic), all_gather_into_tensor of the 3 moments (RCCL), cm3_normalize_* (rank-ordered
    global sums, mean / std, normalisation).  Returns (returns_or_normalised [same shape], (mean, std, count))."""
    from . import _lib
    if reward.device.type != "cuda":
        raise _lib.Cm3Error("normalized_returns runs on the GPU; use returns_to_go / normalize_advantages for host tensors")
    lib = _lib.lib()
    x = reward.contiguous()
    T, E = x.shape[0], x.shape[1]
    C = 1 if x.dim() == 2 else int(x.shape[2])
    suffix = {torch.float32: "f32", torch.float64: "f64"}[x.dtype]
    d8 = done.to(torch.uint8).contiguous()
    v8 = None if valid is None else valid.to(torch.uint8).contiguous()
    out = torch.empty_like(x)
    stream = _lib.current_stream_handle(x.device)
    scratch = _scratch(lib, x.device, stream)
    buf = torch.empty(6, dtype=torch.float64, device=x.device)      # [0:3] this rank's moments, [3:6] (mean, std, count)
    moments, stats = buf[0:3], buf[3:6]
    _lib.check(getattr(lib, "cm3_returns_moments_" + suffix)(
        x.data_ptr(), d8.data_ptr(), _lib.ptr(v8), out.data_ptr(), scratch.data_ptr(), moments.data_ptr(),
        T, E, C, float(gamma), stream))
    parts, n_parts = gather_moments(moments, group)
    # rank-ordered sum of the triples, mean / std and the normalisation itself: one launch
    _lib.check(getattr(lib, "cm3_normalize_" + suffix)(
        out.data_ptr(), _lib.ptr(v8), parts.data_ptr(), n_parts, stats.data_ptr(), out.numel(), C, float(eps),
        1 if normalize else 0, stream))
    return out, (stats[0], stats[1], stats[2])


class ReturnsNormalizer(object):
    """normalized_returns() with PERSISTENT buffers and separately enqueueable halves, so that a collector can capture
    `collect -> returns + moments -> (all-gather) -> normalise` into hipGraphs (ParticleRollout.collect_normalized):
        enqueue_moments(stream)            returns -> self.out, this rank's triple(s) -> self.moments
        enqueue_normalize(stream, parts)   cm3_normalize_* over `parts` (self.moments itself when there is one rank)
        enqueue_fused(stream, shift)       both for ONE rank, two launches (+ the rollout's slot bookkeeping)
    reward [T,E] or [T,E,C] float32/float64 and done uint8 [T,E] are the trajectory's own tensors (read in place).
    segments = K > 1: the trajectory holds K consecutive rollouts of T / K ticks (one collection phase); every rollout gets its
    own returns (zero beyond its last tick), moments [K,3] and statistics [K,3] -- the K rollouts' advantage steps are the SAME
    two launches (cm3_returns_normalize_segments_*), and with several ranks ONE all-gather of K triples."""

    def __init__(self, reward, done, gamma=0.99, eps=1e-8, normalize=True, segments=1):
        from . import _lib
        if reward.device.type != "cuda" or not reward.is_contiguous() or done.dtype != torch.uint8 or not done.is_contiguous():
            raise _lib.Cm3Error("ReturnsNormalizer needs contiguous device tensors (reward float, done uint8)")
        self._lib_mod, self.lib = _lib, _lib.lib()
        self.reward, self.done = reward, done
        self.K = int(segments)
        if self.K < 1 or int(reward.shape[0]) % self.K:
            raise _lib.Cm3Error("segments must divide the %d ticks of the trajectory" % int(reward.shape[0]))
        self.T, self.E = int(reward.shape[0]) // self.K, int(reward.shape[1])      # T = ticks per segment
        self.C = 1 if reward.dim() == 2 else int(reward.shape[2])
        self.suffix = {torch.float32: "f32", torch.float64: "f64"}[reward.dtype]
        self.gamma, self.eps, self.apply = float(gamma), float(eps), 1 if normalize else 0
        self.out = torch.empty_like(reward)
        self.buf = torch.zeros(2, self.K, 3, dtype=torch.float64, device=reward.device)
        self.moments, self.stats = self.buf[0], self.buf[1]                      # [K, 3] each
        if self.K == 1:
            self.moments, self.stats = self.buf[0, 0], self.buf[1, 0]            # [3]: the single-rollout interface
        self.scratch = torch.zeros(self.lib.cm3_returns_segments_scratch_bytes(self.K) // 8, dtype=torch.float64,
                                   device=reward.device)

    def enqueue_moments(self, stream):
        if self.K == 1:
            self._lib_mod.check(getattr(self.lib, "cm3_returns_moments_" + self.suffix)(
                self.reward.data_ptr(), self.done.data_ptr(), 0, self.out.data_ptr(), self.scratch.data_ptr(),
                self.moments.data_ptr(), self.T, self.E, self.C, self.gamma, stream))
        else:       # returns + per-segment moments, nothing applied (the all-gather and cm3_normalize_segments_* follow)
            self._segments_call(stream, None, 0)

    def _segments_call(self, stream, shift, apply):
        cs = None
        if shift:
            cs = self._lib_mod.CopyShift()
            cs.n = len(shift)
            for r, (a, m, c) in enumerate(shift):
                cs.first_dst[r], cs.mid[r], cs.last_src[r] = a.data_ptr(), m.data_ptr(), c.data_ptr()
                cs.bytes[r] = m.numel() * m.element_size()
        import ctypes
        self._lib_mod.check(getattr(self.lib, "cm3_returns_normalize_segments_" + self.suffix)(
            self.reward.data_ptr(), self.done.data_ptr(), 0, self.out.data_ptr(), self.scratch.data_ptr(),
            self.moments.data_ptr(), self.stats.data_ptr(), self.T, self.K, self.E, self.C, self.gamma, self.eps, int(apply),
            ctypes.byref(cs) if cs is not None else None, stream))

    def enqueue_fused(self, stream, shift=None):
        """enqueue_moments + enqueue_normalize on this rank's own moments (two launches, the first of them in one memory round
        trip for T <= 40), bit for bit; shift: optional list of up to 4 (first_dst, mid, last_src) tensor triples -- first_dst <-
        mid, then mid <- last_src, done by the first launch (include/cm3_amd.h cm3_copy_shift)."""
        self._segments_call(stream, shift, self.apply)

    def enqueue_normalize(self, stream, parts=None, n_parts=1):
        parts = self.moments if parts is None else parts
        self._lib_mod.check(getattr(self.lib, "cm3_normalize_segments_" + self.suffix)(
            self.out.data_ptr(), 0, parts.data_ptr(), int(n_parts), self.K, self.stats.data_ptr(),
            self.out.numel() // self.K, self.C, self.eps, self.apply, stream))


_SCRATCH = {}


def _scratch(lib, device, stream):
    """Scratch of cm3_returns_moments_* per (device, stream): launches on one stream are ordered, so they can share it
    (every call zeroes its arrival counter itself)."""
    key = (device.type, device.index, int(stream or 0))
    if key not in _SCRATCH:
        _SCRATCH[key] = torch.zeros(lib.cm3_returns_scratch_bytes() // 8, dtype=torch.float64, device=device)
    return _SCRATCH[key]
