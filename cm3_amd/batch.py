"""Batch reformatting on the device.

The reference reshapes every sampled batch on the host with NumPy before each of its 6-8 sess.run calls:
  alg_credit.Alg.process_actions       alg/alg_credit.py:406-443
  alg_credit.Alg.process_batch         alg/alg_credit.py:445-499
  alg_credit.Alg.process_goals         alg/alg_credit.py:501-526
  alg_credit.Alg.process_global_state  alg/alg_credit.py:528-557
  the n x n "credit" repeats of train_step   alg/alg_credit.py:614-658
Here the same arrays are produced from the device trajectory columns (ParticleRollout.as_reference_batch(...,
numpy=False)) with gathers / views, so a learner living on the GPU never pulls the batch through the host.
Pure data movement: outputs are bit-identical to the reference's (tests/test_batch.py pins them to arrays produced
by the REAL reference functions, tests/golden/batch_particle.npz).
"""
import torch


def rep_rows(x, n):
    """every row of x repeated n times consecutively -- x.repeat_interleave(n, dim=0) as ONE copy launch (expand + reshape; torch's
    repeat_interleave builds a repeats tensor, a cumulative sum and an index_select: four launches, and train_step has a dozen)."""
    return x.unsqueeze(1).expand(x.shape[0], int(n), *x.shape[1:]).reshape(x.shape[0] * int(n), *x.shape[1:])


class _Tiler(object):
    """Collects "destination row r <- source row f(r)" outputs and builds them 16 per launch (cm3_rows_tile, csrc/batch.hip)."""

    def __init__(self, device):
        self.device, self.specs, self.keep = device, [], []

    def add(self, src, n_rows, epr, dtype, kind, terms=((1, 0, 1), (1, 0, 0)), others=None, shape=None, out=None):
        from . import _lib
        want = tuple((n_rows, epr) if shape is None else shape)
        if out is None:
            out = torch.empty(want, dtype=dtype, device=self.device)
        elif tuple(out.shape) != want or out.dtype != dtype or not out.is_contiguous():
            raise ValueError("persistent feed tensor has shape %s / %s, expected %s / %s" % (tuple(out.shape), out.dtype, want, dtype))
        c = _lib.TileCol()
        c.dst, c.src, c.n_rows, c.elems_per_row, c.kind = out.data_ptr(), (0 if src is None else src.data_ptr()), n_rows, epr, kind
        c.elem_bytes = src.element_size() if (src is not None and kind == _lib.TILE_COPY) else 0
        if others is not None:
            c.others_n, c.others_divq, c.others_divn = others
        else:
            for i, (d, m, mul) in enumerate(terms):
                c.div[i], c.mod[i], c.mul[i] = d, m, mul
        self.specs.append(c)
        self.keep.append(src)
        return out

    def run(self):
        from . import _lib
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for k in range(0, len(self.specs), 16):
            chunk = self.specs[k:k + 16]
            arr = (_lib.TileCol * len(chunk))(*chunk)
            _lib.check(_lib.lib().cm3_rows_tile(arr, len(chunk), stream))
        self.specs, self.keep = [], []


def particle_static_feeds_device(cols, l_action=5, out=None):
    """Every feed of the reference's train_step that does not depend on a network output (particle env, N > 1, Q-credit variant), from
    the float32 columns of ParticleRollout.as_reference_batch(numpy=False), in TWO launches of cm3_rows_tile: process_actions /
    process_global_state (alg_credit.py:406-443, :528-557), the n x n credit repeats (:614-658), the n x n x l_action counterfactual
    tiling (:730-751).  Values and dtypes are those of the torch composition in train_step_feeds (tests/test_batch.py compares the two
    and both with the arrays the REAL train_step fed).  out: the dict an earlier call returned for the same batch size -- the feeds are
    written into those tensors again (persistent addresses for a captured consumer)."""
    from . import _lib
    vg, vgn = cols["v_global"].contiguous(), cols["v_global_next"].contiguous()
    B, N, l = vg.shape
    A, R, dev = int(l_action), B * N, vg.device
    actions = cols["actions"].to(torch.int32).contiguous()
    done = cols["done"].contiguous()
    done_u8 = done.view(torch.uint8) if done.dtype == torch.bool else done.to(torch.uint8)
    goals = cols["goals"].contiguous()
    lg = goals.shape[2]
    reward, reward_local = cols["reward"].contiguous(), cols["reward_local"].contiguous()
    f64, i64 = torch.float64, torch.int64
    by_n = lambda outer=1: ((outer * N * N, 0, N), (outer, N, 1))          # noqa: E731  rows (b, m, n[, a]) <- row b N + n
    by_m = lambda outer=1: ((outer * N, 0, 1), (1, 0, 0))                   # noqa: E731  rows (r, m[, a])    <- row r
    t = _Tiler(dev)
    S, O = {}, (out or {})
    S["a1"] = t.add(actions, R, A, i64, _lib.TILE_ONEHOT_I64, out=O.get("a1"))
    S["ao"] = t.add(actions, R * (N - 1), A, f64, _lib.TILE_ONEHOT_F64, others=(N, N * (N - 1), N - 1), shape=(R, N - 1, A), out=O.get("ao"))
    S["reward_rep"] = t.add(reward, R, 1, reward.dtype, _lib.TILE_COPY, terms=by_m(), shape=(R,), out=O.get("reward_rep"))
    S["done_rep"] = t.add(done_u8, R, 1, torch.bool, _lib.TILE_COPY, terms=by_m(), shape=(R,), out=O.get("done_rep"))
    S["not_done"] = t.add(done_u8, R, 1, i64, _lib.TILE_NOT_I64, terms=by_m(), shape=(R,), out=O.get("not_done"))
    S["others"] = t.add(vg, R * (N - 1), l, f64, _lib.TILE_F32_TO_F64, others=(N, N * (N - 1), N - 1), shape=(R, (N - 1) * l), out=O.get("others"))
    S["others_next"] = t.add(vgn, R * (N - 1), l, f64, _lib.TILE_F32_TO_F64, others=(N, N * (N - 1), N - 1), shape=(R, (N - 1) * l), out=O.get("others_next"))
    S["goals_self_rep"] = t.add(goals, R * N, lg, goals.dtype, _lib.TILE_COPY, terms=by_n(), out=O.get("goals_self_rep"))
    S["one_next_rep_n"] = t.add(vgn, R * N, l, vgn.dtype, _lib.TILE_COPY, terms=by_n(), out=O.get("one_next_rep_n"))
    S["one_next_rep_m"] = t.add(vgn, R * N, l, vgn.dtype, _lib.TILE_COPY, terms=by_m(), out=O.get("one_next_rep_m"))
    S["others_next_rep_n"] = t.add(vgn, R * N * (N - 1), l, f64, _lib.TILE_F32_TO_F64, others=(N, N * N * (N - 1), N - 1),
                                   shape=(R * N, (N - 1) * l), out=O.get("others_next_rep_n"))
    S["r_rep"] = t.add(reward_local, R * N, 1, reward_local.dtype, _lib.TILE_COPY, terms=by_n(), shape=(R * N,), out=O.get("r_rep"))
    # (done per time step -> per agent row -> repeated by n: row (b, m, n) <- done[b])
    S["nd_rep"] = t.add(done_u8, R * N, 1, i64, _lib.TILE_NOT_I64, terms=((N * N, 0, 1), (1, 0, 0)), shape=(R * N,), out=O.get("nd_rep"))
    S["s_n_rep"] = t.add(vg, R * N, l, vg.dtype, _lib.TILE_COPY, terms=by_n(), out=O.get("s_n_rep"))
    S["s_m_rep"] = t.add(vg, R * N, l, vg.dtype, _lib.TILE_COPY, terms=by_m(), out=O.get("s_m_rep"))
    S["s_others_rep"] = t.add(vg, R * N * (N - 1), l, f64, _lib.TILE_F32_TO_F64, others=(N, N * N * (N - 1), N - 1),
                              shape=(R * N, (N - 1) * l), out=O.get("s_others_rep"))
    t.run()
    S["a1_rep_m"] = t.add(actions, R * N, A, i64, _lib.TILE_ONEHOT_I64, terms=by_m(), out=O.get("a1_rep_m"))
    S["cf_s_n"] = t.add(vg, R * N * A, l, vg.dtype, _lib.TILE_COPY, terms=by_n(A), out=O.get("cf_s_n"))
    S["cf_goals"] = t.add(goals, R * N * A, lg, goals.dtype, _lib.TILE_COPY, terms=by_n(A), out=O.get("cf_goals"))
    S["cf_eye"] = t.add(None, R * N * A, A, f64, _lib.TILE_EYE_F64, out=O.get("cf_eye"))
    S["cf_s_m"] = t.add(vg, R * N * A, l, vg.dtype, _lib.TILE_COPY, terms=by_m(A), out=O.get("cf_s_m"))
    S["cf_s_others"] = t.add(vg, R * N * A * (N - 1), l, f64, _lib.TILE_F32_TO_F64, others=(N, N * N * A * (N - 1), A * (N - 1)),
                             shape=(R * N * A, (N - 1) * l), out=O.get("cf_s_others"))
    t.run()
    return S


def phase_static_feeds(cols_all, n_minibatches, l_action=5):
    """The static feeds of EVERY minibatch of an on-policy phase (train_onpolicy.py:359-377: 24 minibatches of 128) from the phase's
    one export (ParticleRollout.on_policy_phase: columns [n_minibatches * batch_size, ...]) -- ONE pair of cm3_rows_tile launches;
    returns a list of per-minibatch dicts of VIEWS (every feed is transition-major, so a minibatch is a contiguous block of rows).
    Round 5 built them per minibatch: 24 x 0.09 ms of launch latency for 0.13 ms of work."""
    S = particle_static_feeds_device(cols_all, l_action)
    M = int(n_minibatches)
    out = []
    for k in range(M):
        d = {}
        for name, t in S.items():
            r = t.shape[0] // M
            d[name] = t[k * r:(k + 1) * r]
        out.append(d)
    return out


class OnPolicyPhasePlan(object):
    """One on-policy phase of the reference's loop (train_onpolicy.py:359-377: after a collection phase, 24 x (sample 128 transitions,
    train_step)) with PERSISTENT device tensors, so that the data movement of every train_step can be a hipGraph replay:

        plan = OnPolicyPhasePlan(rollout, run, gamma, epsilon)        # once
        rollout.collect(); plan.refresh(generator)                     # per phase: one export launch + one pair of tiling launches
        for k in range(plan.epochs): calls = plan.step(k)              # per minibatch: ONE graph replay

    refresh() draws the phase's minibatches and writes their columns and the static feeds of all of them INTO THE SAME tensors every
    phase; step(k) replays the graph captured from cm3_amd.batch.train_step_feeds(columns_k, run, gamma, epsilon, static=static_k) --
    every launch that function and `run` enqueue (the session's networks included, if `run` launches them on the current stream
    into tensors of its own that it keeps) -- and returns the same list of (ops, feed) with this phase's values in the same tensors.
    `run` is CALLED only while a graph is captured (the first step(k) of each k): it must not synchronise or allocate per call what
    it does not keep.  Without graphs (use_graph=False) step(k) simply calls train_step_feeds: the specification the replay is
    tested against (tests/test_batch.py::test_phase_plan_replays_equal_eager_feeds)."""

    def __init__(self, rollout, run, gamma, epsilon, epochs=24, batch_size=128, l_action=5, use_graph=True):
        self.ro, self.run, self.gamma, self.epsilon = rollout, run, float(gamma), float(epsilon)
        self.epochs, self.batch_size, self.l_action, self.use_graph = int(epochs), int(batch_size), int(l_action), bool(use_graph)
        self.cols = self.static = None
        self.k = 0
        self._graphs, self._calls = {}, {}

    def refresh(self, generator=None):
        first = self.cols is None
        self.cols, self.k = self.ro._phase_export(self.epochs, self.batch_size, generator, out=self.cols)
        vg = self.cols["v_global"]
        if not (vg.is_cuda and vg.dtype == torch.float32 and vg.shape[1] > 1):
            raise ValueError("OnPolicyPhasePlan covers float32 device rollouts with more than one agent (the tiling kernels' case)")
        self.static = particle_static_feeds_device(self.cols, self.l_action, out=self.static)
        if first:
            M, k = self.epochs, self.k
            self._mb = [({n: v[m * k:(m + 1) * k] for n, v in self.cols.items()},
                         {n: t[m * (t.shape[0] // M):(m + 1) * (t.shape[0] // M)] for n, t in self.static.items()}) for m in range(M)]
        return self

    def minibatch(self, k):
        """(columns, static feeds) of minibatch k: views of the persistent phase tensors"""
        return self._mb[k]

    def step(self, k):
        from . import _lib
        cols, static = self._mb[k]
        if not self.use_graph:
            return train_step_feeds(cols, self.run, self.gamma, self.epsilon, l_action=self.l_action, static=static)
        dev = cols["v_global"].device
        if k not in self._graphs:
            # once eagerly (the allocator then holds blocks of every size the capture will ask for), then captured; the tensors the
            # captured pass created are KEPT (self._calls): a replay writes to their addresses
            train_step_feeds(cols, self.run, self.gamma, self.epsilon, l_action=self.l_action, static=static)
            torch.cuda.synchronize(dev)
            box = {}

            def enqueue(stream):
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                    box["calls"] = train_step_feeds(cols, self.run, self.gamma, self.epsilon, l_action=self.l_action, static=static)
            self._graphs[k] = _lib.capture_graph(dev, enqueue)
            self._calls[k] = box["calls"]
        _lib.check(_lib.lib().cm3_graph_launch(self._graphs[k], torch.cuda.current_stream(dev).cuda_stream))
        return self._calls[k]

    def close(self):
        from . import _lib
        if self._graphs:
            torch.cuda.synchronize()
            for g in self._graphs.values():
                _lib.lib().cm3_graph_destroy(g)
        self._graphs, self._calls = {}, {}


def td_target(reward, q, multiplier, gamma):
    """reward + gamma * q * multiplier in NumPy's order (alg_credit.py:594, :640, :684), float64: one launch of cm3_td_target_f64 for
    device tensors (reward float32 / float64, multiplier int64), the torch composition otherwise -- the same bits."""
    q = q.reshape(-1)
    if (q.is_cuda and q.dtype == torch.float64 and q.is_contiguous() and multiplier.dtype == torch.int64 and multiplier.is_contiguous()
            and reward.dtype in (torch.float32, torch.float64) and reward.is_contiguous() and reward.numel() == q.numel()):
        from . import _lib
        out = torch.empty_like(q)
        _lib.check(_lib.lib().cm3_td_target_f64(reward.data_ptr(), 1 if reward.dtype == torch.float64 else 0, q.data_ptr(),
                                                multiplier.data_ptr(), float(gamma), out.data_ptr(), q.numel(),
                                                torch.cuda.current_stream(q.device).cuda_stream))
        return out
    return reward.to(torch.float64) + gamma * q * multiplier


def process_actions_device(acts, n_steps, n_agents, l_action=5, rep_m=False):
    """process_actions(acts.reshape(n_steps, N)) -- and optionally the one-hot rows repeated N times (`rep_m`, alg_credit.py:628) -- in
    ONE cm3_rows_tile launch for a device tensor of sampled actions (any integer dtype)."""
    from . import _lib
    N, A, R = n_agents, int(l_action), n_steps * n_agents
    a32 = acts.reshape(-1).to(torch.int32).contiguous()
    t = _Tiler(acts.device)
    one = t.add(a32, R, A, torch.int64, _lib.TILE_ONEHOT_I64)
    others = t.add(a32, R * (N - 1), A, torch.float64, _lib.TILE_ONEHOT_F64, others=(N, N * (N - 1), N - 1), shape=(R, N - 1, A))
    rep = t.add(a32, R * N, A, torch.int64, _lib.TILE_ONEHOT_I64, terms=((N, 0, 1), (1, 0, 0))) if rep_m else None
    t.run()
    return one, others, rep


def others_index(n_agents, device=None):
    """[N, N-1] long: row n lists the other agents in ascending order (np.arange(N) != n)."""
    idx = [[j for j in range(n_agents) if j != n] for n in range(n_agents)]
    return torch.tensor(idx, dtype=torch.long, device=device).reshape(n_agents, max(n_agents - 1, 0))


def gather_others(x):
    """x [B, N, d] -> [B*N, N-1, d]: row b*N+n holds x[b, m] for the agents m != n in ascending order
    (the reference's `out[n::N] = x[:, np.arange(N) != n]` interleave)."""
    B, N = x.shape[0], x.shape[1]
    idx = others_index(N, x.device)
    return x[:, idx].reshape(B * N, N - 1, *x.shape[2:])


def process_actions(actions, l_action=5):
    """actions int [B, N] -> (actions_1hot int64 [B*N, A], actions_others_1hot float64 [B*N, N-1, A])
    (alg_credit.py:406-443; dtypes as the reference: np.zeros(dtype=int) and np.zeros default float64)."""
    B, N = actions.shape
    one = torch.nn.functional.one_hot(actions.long(), l_action)            # [B, N, A] int64
    others = gather_others(one).to(torch.float64)
    return one.reshape(B * N, l_action), others


def process_goals(goals):
    """goals [B, N, l_goal] -> (goals_self [B*N, l_goal], goals_others [B*N, (N-1)*l_goal]) (alg_credit.py:501-526)."""
    B, N, lg = goals.shape
    return goals.reshape(B * N, lg), gather_others(goals).reshape(B * N, (N - 1) * lg).to(torch.float64)


def process_global_state(v_global):
    """v_global [B, N, l] -> (one_agent [B*N, l], others [B*N, (N-1)*l], state [B*N, N*l]) (alg_credit.py:528-557)."""
    B, N, l = v_global.shape
    one = v_global.reshape(B * N, l)
    others = gather_others(v_global).reshape(B * N, (N - 1) * l).to(torch.float64)
    state = rep_rows(v_global.reshape(B, N * l), N)
    return one, others, state


def process_batch(cols, l_action=5):
    """Columns of ParticleRollout.as_reference_batch(numpy=False) -> the 13-tuple of alg_credit.process_batch
    (alg_credit.py:445-499): per-agent rows, global quantities repeated N times."""
    v_global = cols["v_global"]
    B, N = v_global.shape[0], v_global.shape[1]
    a1, ao = process_actions(cols["actions"], l_action)
    return (B, v_global, cols["obs_others"].reshape(B * N, -1), cols["v_local"].reshape(B * N, -1), a1, ao,
            rep_rows(cols["reward"], N), cols["reward_local"].reshape(B * N),
            cols["v_global_next"], cols["obs_others_next"].reshape(B * N, -1), cols["v_local_next"].reshape(B * N, -1),
            rep_rows(cols["done"], N), cols["goals"])


def repeat_indexed_by_n(x, n_agents):
    """[B*N, d] -> [B*N*N, d]: each time step's block of N rows repeated N times (alg_credit.py:621-623, :648-649)."""
    d = x.shape[1:]
    return rep_rows(x.reshape(-1, n_agents, *d), n_agents).reshape(-1, *d)


def repeat_indexed_by_m(x, n_agents):
    """[B*N, d] -> [B*N*N, d]: every row repeated N times consecutively (alg_credit.py:628-629, :651-652)."""
    return rep_rows(x, n_agents)


# ---- Checkers (alg_credit_checkers.py:375-535) ---------------------------------------------------------------------------

def process_batch_checkers(cols, l_action=5):
    """Columns of CheckersRollout.as_reference_batch(numpy=False) -> the 18-tuple of
    alg_credit_checkers.Alg.process_batch (alg_credit_checkers.py:414-482): per-agent rows; grid and done repeated N
    times, `reward` left per time step (unlike the particle variant), both action columns one-hot."""
    vec = cols["vec"]
    B, N = vec.shape[0], vec.shape[1]
    rep = lambda x: rep_rows(x, N)                             # noqa: E731
    rows = lambda x: x.reshape(B * N, *x.shape[2:])            # noqa: E731
    a1, ao = process_actions(cols["actions"], l_action)
    prev1 = torch.nn.functional.one_hot(cols["actions_prev"].long(), l_action).reshape(B * N, l_action)
    return (B, rep(cols["grid"]), vec, rows(cols["obs_others"]), rows(cols["obs_self_t"]), rows(cols["obs_self_v"]),
            prev1, a1, ao, cols["reward"], cols["local_rewards"].reshape(B * N), rep(cols["next_grid"]),
            cols["next_vec"], rows(cols["next_obs_others"]), rows(cols["next_obs_self_t"]),
            rows(cols["next_obs_self_v"]), rep(cols["done"]), cols["goals"])


CHECKERS_BATCH_NAMES = ("n_steps", "state_env", "state_agents", "obs_others", "obs_self_t", "obs_self_v",
                        "actions_prev_1hot", "actions_1hot", "actions_others_1hot", "reward", "reward_local",
                        "state_env_next", "state_agents_next", "obs_others_next", "obs_self_t_next", "obs_self_v_next",
                        "done", "goals")
# process_goals / process_global_state of alg_credit_checkers.py:484-535 are the particle functions above, verbatim
# in behaviour (l_goal = 2, l_state_one_agent = 4): use process_goals(goals) and process_global_state(state_agents).


# ---- every feed_dict of train_step (alg_credit.py:558-800, alg_credit_checkers.py:536-780) -------------------------------

def _rep_n(x, n):
    """things indexed by n: each time step's block of N rows repeated N times (alg_credit.py:621-623)"""
    return repeat_indexed_by_n(x, n)


def train_step_feeds(cols, run, gamma, epsilon, env="particle", use_Q_credit=True, use_V=True, l_action=5, device_tiling=True,
                     static=None):
    """The data movement of the reference's train_step on the device: builds, in the reference's order, the feed_dict of
    every sess.run -- TD targets, the n x n credit repeats (alg_credit.py:614-658) and the n x n x l_action counterfactual
    tiling (:730-751; Checkers twin alg_credit_checkers.py:590-760) -- from the columns of
    {Particle,Checkers}Rollout.as_reference_batch(numpy=False).

    ``run(ops, feed)`` plays sess.run: `ops` is a list of the reference's op attribute names (e.g. ["Q_global_op",
    "Q_global"]), `feed` a dict keyed by the reference's placeholder attribute names; it returns one tensor per op (None
    for optimiser ops).  Returns the list of (ops, feed) in call order.  Pure gathers / repeats plus the float64 TD
    arithmetic: bit-identical to the arrays the REAL train_step feeds (tests/golden/trainstep_*.npz, recorded by
    oracle/gen_golden_trainstep.py from the reference code itself under a recording session).
    static: this minibatch's entry of phase_static_feeds() -- the static feeds then cost no launch here."""
    checkers = env == "checkers"
    calls = []

    def call(ops, feed):
        calls.append((ops, feed))
        return run(ops, feed)

    if checkers:
        (n_steps, state_env, state_agents, obs_others, obs_self_t, obs_self_v, actions_prev_1hot, actions_1hot,
         actions_others_1hot, reward, reward_local, state_env_next, state_agents_next, obs_others_next, obs_self_t_next,
         obs_self_v_next, done, goals) = process_batch_checkers(cols, l_action)
        v_global, v_global_next = state_agents, state_agents_next
    else:
        vg0 = cols["v_global"]
        # float32 columns on the GPU: everything that does not depend on a network output comes from two launches of
        # cm3_rows_tile (particle_static_feeds_device); other inputs (the float64 parity path, host tensors of the CPU tests, N = 1,
        # the variants without Q-credit) go through the torch composition below -- the specification both are tested against
        S = static if static is not None else (
            particle_static_feeds_device(cols, l_action)
            if (vg0.is_cuda and vg0.dtype == torch.float32 and vg0.shape[1] > 1 and use_Q_credit and device_tiling) else None)
        if S is not None:
            B_, N_ = vg0.shape[0], vg0.shape[1]
            (n_steps, v_global, obs_others, v_local, actions_1hot, actions_others_1hot, reward, reward_local, v_global_next,
             obs_others_next, v_local_next, done, goals) = (
                B_, vg0, cols["obs_others"].reshape(B_ * N_, -1), cols["v_local"].reshape(B_ * N_, -1), S["a1"], S["ao"], S["reward_rep"],
                cols["reward_local"].reshape(B_ * N_), cols["v_global_next"], cols["obs_others_next"].reshape(B_ * N_, -1),
                cols["v_local_next"].reshape(B_ * N_, -1), S["done_rep"], cols["goals"])
        else:
            (n_steps, v_global, obs_others, v_local, actions_1hot, actions_others_1hot, reward, reward_local, v_global_next,
             obs_others_next, v_local_next, done, goals) = process_batch(cols, l_action)
    if checkers:
        S = None
    N = v_global.shape[1]
    goals_self, _ = process_goals(goals) if S is None else (goals.reshape(-1, goals.shape[2]), None)
    if S is None:
        one, others, _ = process_global_state(v_global)
        one_next, others_next, _ = process_global_state(v_global_next)
    else:
        one, others = v_global.reshape(-1, v_global.shape[2]), S["others"]
        one_next, others_next = v_global_next.reshape(-1, v_global_next.shape[2]), S["others_next"]
    f64 = torch.float64
    not_done = (-(done.to(torch.int64) - 1)) if S is None else S["not_done"]      # if true, then 0, else 1 (:590)

    def actor_feed(oo, *obs):
        if checkers:
            return {"obs_others": oo, "obs_self_t": obs[0], "obs_self_v": obs[1], "actions_prev": obs[2],
                    "v_goal": goals_self, "epsilon": epsilon}
        return {"obs_others": oo, "v_obs": obs[0], "v_goal": goals_self, "epsilon": epsilon}

    def with_env(feed, s_env=None, ot=None, ov=None):
        if checkers:
            feed["state_env"] = s_env
            if ot is not None:
                feed["obs_self_t"], feed["obs_self_v"] = ot, ov
        return feed

    # ---- Q_n(s, a): target actions a', TD target, optimiser step (alg_credit.py:574-612) ----
    if checkers:      # run_actor_target(actions_1hot, obs_others_next, ...) (alg_credit_checkers.py:551)
        acts = call(["action_samples_target"], actor_feed(obs_others_next, obs_self_t_next, obs_self_v_next, actions_1hot))[0]
    else:
        acts = call(["action_samples_target"], actor_feed(obs_others_next, v_local_next))[0]
    a_next_rep = None
    if S is not None and acts.is_cuda:    # one launch: both one-hot forms and the rows repeated by m (Q-credit target, :628)
        a_next, a_others_next, a_next_rep = process_actions_device(acts, n_steps, N, l_action, rep_m=N > 1 and use_Q_credit)
    else:
        a_next, a_others_next = process_actions(acts.reshape(n_steps, N), l_action)
    q_t = call(["Q_global_target"], with_env({"v_state_one_agent": one_next, "v_goal": goals_self, "action_one": a_next,
                                              "v_state_other_agents": others_next, "action_others": a_others_next},
                                             state_env_next if checkers else None,
                                             obs_self_t_next if checkers else None, obs_self_v_next if checkers else None))[0]
    td = td_target(reward_local, q_t, not_done, gamma) if S is not None else reward_local.to(f64) + gamma * q_t.reshape(-1) * not_done
    q_res = call(["Q_global_op", "Q_global"],
                 with_env({"Q_global_td_target": td, "v_state_one_agent": one, "v_goal": goals_self, "action_one": actions_1hot,
                           "v_state_other_agents": others, "action_others": actions_others_1hot},
                          state_env if checkers else None, obs_self_t if checkers else None,
                          obs_self_v if checkers else None))[1]
    q_res_rep = rep_rows(q_res.reshape(n_steps, N), N)                              # :612

    rep_m = lambda x: rep_rows(x, N)                                                # noqa: E731  things indexed by m
    s_n_rep = s_m_rep = s_others_rep = goals_self_rep = s_env_rep = ot_rep = ov_rep = None
    if N > 1 and use_Q_credit:      # ---- Q_n(s, a^m) (:616-674) ----
        goals_self_rep = _rep_n(goals_self, N) if S is None else S["goals_self_rep"]
        feed = {"v_state_one_agent": _rep_n(one_next, N) if S is None else S["one_next_rep_n"], "v_goal": goals_self_rep,
                "action_one": rep_m(a_next) if a_next_rep is None else a_next_rep,
                "v_state_m": rep_m(one_next) if S is None else S["one_next_rep_m"],
                "v_state_other_agents": _rep_n(others_next, N) if S is None else S["others_next_rep_n"]}
        if checkers:
            feed = with_env(feed, rep_m(state_env_next), rep_m(obs_self_t_next), rep_m(obs_self_v_next))
        qc_t = call(["Q_credit_target"], feed)[0]
        if S is None:
            r_rep = _rep_n(reward_local.reshape(-1, 1), N).reshape(-1)
            nd_rep = -(_rep_n(done.reshape(-1, 1).to(torch.int64), N).reshape(-1) - 1)
            s_n_rep, s_others_rep, s_m_rep = _rep_n(one, N), _rep_n(others, N), rep_m(one)
        else:
            r_rep, nd_rep = S["r_rep"], S["nd_rep"]
            s_n_rep, s_others_rep, s_m_rep = S["s_n_rep"], S["s_others_rep"], S["s_m_rep"]
        td_c = td_target(r_rep, qc_t, nd_rep, gamma) if S is not None else r_rep.to(f64) + gamma * qc_t.reshape(-1) * nd_rep
        feed = {"Q_credit_td_target": td_c, "v_state_one_agent": s_n_rep, "v_goal": goals_self_rep,
                "action_one": rep_m(actions_1hot) if S is None else S["a1_rep_m"], "v_state_m": s_m_rep,
                "v_state_other_agents": s_others_rep}
        if checkers:
            s_env_rep, ot_rep, ov_rep = rep_m(state_env), rep_m(obs_self_t), rep_m(obs_self_v)
            feed = with_env(feed, s_env_rep, ot_rep, ov_rep)
        call(["Q_credit_op"], feed)
    v_res_rep = None
    if N > 1 and use_V:             # ---- V_n(s) (:676-699) ----
        v_t = call(["V_target"], with_env({"v_state_one_agent": one_next, "v_goal": goals_self,
                                           "v_state_other_agents": others_next}, state_env_next if checkers else None))[0]
        td_v = td_target(reward_local, v_t, not_done, gamma) if S is not None else reward_local.to(f64) + gamma * v_t.reshape(-1) * not_done
        v_res = call(["V_op", "V"], with_env({"V_td_target": td_v, "v_state_one_agent": one, "v_goal": goals_self,
                                              "v_state_other_agents": others}, state_env if checkers else None))[1]
        v_res_rep = rep_rows(v_res.reshape(n_steps, N), N)

    # ---- policy: probabilities + counterfactual Q for every action (:704-757) ----
    pol_obs = (obs_self_t, obs_self_v, actions_prev_1hot) if checkers else (v_local,)
    rep_a = lambda x: rep_rows(x, l_action)                                         # noqa: E731
    eye = torch.eye(l_action, dtype=f64, device=one.device)                         # self.actions = np.eye(l_action)
    if N == 1:                      # stage 1: Q(s, a = every action, g)
        probs = call(["probs"], actor_feed(obs_others, *pol_obs))[0]
        feed = {"v_state_one_agent": rep_a(one), "v_goal": rep_a(goals_self), "action_one": eye.repeat(n_steps, 1),
                "v_state_other_agents": others, "action_others": actions_others_1hot}
        if checkers:
            feed = with_env(feed, rep_a(state_env), rep_a(obs_self_t), rep_a(obs_self_v))
        q_cf = call(["Q_global"], feed)[0].reshape(n_steps, l_action)
    elif use_Q_credit:
        probs = rep_rows(call(["probs"], actor_feed(obs_others, *pol_obs))[0], N)
        if S is None:
            feed = {"v_state_one_agent": rep_a(s_n_rep), "v_goal": rep_a(goals_self_rep),
                    "action_one": eye.repeat(N * N * n_steps, 1), "v_state_m": rep_a(s_m_rep),
                    "v_state_other_agents": rep_a(s_others_rep)}
        else:
            feed = {"v_state_one_agent": S["cf_s_n"], "v_goal": S["cf_goals"], "action_one": S["cf_eye"], "v_state_m": S["cf_s_m"],
                    "v_state_other_agents": S["cf_s_others"]}
        if checkers:
            feed = with_env(feed, rep_a(s_env_rep), rep_a(ot_rep), rep_a(ov_rep))
        q_cf = call(["Q_credit"], feed)[0].reshape(n_steps * N * N, l_action)
    else:
        probs = torch.zeros(1, l_action, dtype=f64, device=one.device)
        q_cf = torch.zeros(1, l_action, dtype=f64, device=one.device)
    feed = actor_feed(obs_others, *pol_obs)
    feed.update({"action_taken": actions_1hot, "Q_actual": q_res_rep, "probs_evaluated": probs, "Q_cf": q_cf})
    if N > 1 and use_V:
        feed["V_evaluated"] = v_res_rep
    call(["policy_op"], feed)
    call(["list_update_target_ops"], {})
    return calls
