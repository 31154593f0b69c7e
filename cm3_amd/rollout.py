"""Trajectory collection -- the `while not done` loop of alg/train_onpolicy.py:281-350 for E envs.

The reference appends one object-array row per tick to a Python list (replay_buffer.py:11-16):
  particle (11 columns, train_onpolicy.py:338)
    [global_state, obs_others, obs_self, actions, reward, local_rewards,
     next_global_state, next_obs_others, next_obs_self, done, goals]
  Checkers (16 columns, train_onpolicy.py:336)
    [grid, vec, obs_others, obs_self_t, obs_self_v, actions_prev, actions, reward, local_rewards,
     next_grid, next_vec, next_obs_others, next_obs_self_t, next_obs_self_v, done, goals]
Here the trajectory lives on the device, time-major and de-duplicated: slot t of the state /
observation arrays is "before tick t", slot t+1 is "after tick t", so `next_*` of transition t is
slot t+1 (or the captured terminal state when the env was re-initialised in the same launch).  The
step kernel writes slot t+1 directly (zero-copy, cm3_particle_rollout_*).  ``as_reference_batch``
gathers any set of (tick, env) pairs into exactly what ``np.stack(batch[:, k])`` yields in
alg_credit.process_batch (alg_credit.py:458-470) / alg_credit_checkers.process_batch
(alg_credit_checkers.py:427-444), and ``as_reference_rows`` rebuilds the reference's object rows so
the real ``process_batch`` consumes them unchanged.
"""
import collections
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import FLAG_AUTO_RESET, FLAG_GEN_ACTIONS, Cm3Error


def rows_from_columns(cols, order):
    """Object ndarray [B, len(order)]: row b = the reference's per-tick np.array([...fields...]) of
    train_onpolicy.py:336/:338 (dtype=object, which NumPy >= 1.24 requires for ragged rows), so that
    ``np.stack(rows[:, k])`` in the reference's process_batch returns cols[order[k]]."""
    B = len(cols[order[0]])
    rows = np.empty((B, len(order)), dtype=object)
    for k, name in enumerate(order):
        col = cols[name]
        for b in range(B):
            rows[b, k] = col[b]
    return rows


PARTICLE_ORDER = ("v_global", "obs_others", "v_local", "actions", "reward", "reward_local", "v_global_next",
                  "obs_others_next", "v_local_next", "done", "goals")
CHECKERS_ORDER = ("grid", "vec", "obs_others", "obs_self_t", "obs_self_v", "actions_prev", "actions", "reward",
                  "local_rewards", "next_grid", "next_vec", "next_obs_others", "next_obs_self_t",
                  "next_obs_self_v", "done", "goals")


def _valid_from_done(done_u8, finished0=None):
    """valid[t, e] = env e had not finished before tick t (episode-synchronous collection: an env that
    finishes early stops producing transitions, like the reference's `while not done`).  finished0 (bool [E]):
    envs whose episode had already ended when this collection started (the collector carries that across
    collect(reset=False) calls on an env without auto-reset) produce no valid transition at all."""
    d = done_u8.to(torch.int32)
    finished_before = torch.cumsum(d, dim=0) - d
    valid = finished_before == 0
    if finished0 is not None:
        valid = valid & ~finished0.unsqueeze(0)
    return valid


def _pairs_aligned(pairs):
    return all((d.numel() * d.element_size()) % 16 == 0 and d.data_ptr() % 16 == 0 and s_.data_ptr() % 16 == 0
               and d.is_contiguous() and s_.is_contiguous() and d.numel() * d.element_size() == s_.numel() * s_.element_size()
               for d, s_ in pairs)


def _copy_pairs(pairs, stream):
    """(dst, src) tensor pairs ON `stream`: ONE cm3_copy_list launch per 8 regions when every region is 16-byte sized /
    aligned, torch copies otherwise -- issued on `stream` too (an ExternalStream context), so that they are ordered with the
    launches around them and, while `stream` is being captured into a hipGraph, become nodes of that graph instead of
    running once, unordered, at capture time."""
    ok = _pairs_aligned(pairs)
    if ok:
        for k in range(0, len(pairs), 8):
            _lib.copy_list(pairs[k:k + 8], stream)
    else:
        dev = pairs[0][0].device
        # (handle 0 = the legacy default stream: torch.cuda.ExternalStream(0) would hand out a fresh POOL stream instead)
        ts = torch.cuda.ExternalStream(int(stream), device=dev) if stream else torch.cuda.default_stream(dev)
        with torch.cuda.stream(ts):
            for d, s_ in pairs:
                d.copy_(s_)


_Mode = collections.namedtuple("_Mode", "kind live sparse")


class _ActorGraphCache(object):
    """One captured (actor launch, step launch) x T graph per rollout object.  The key is the policy OBJECT (a strong
    reference: `id()` of a collected actor can be reused); epsilon is not part of it -- the actor launches read it from
    a device float that collect() refreshes, so the annealing of train_onpolicy.py:369 never triggers a re-capture."""

    def __init__(self, device):
        self.device = device
        self.graph = None
        self.policy = None
        self.key = None
        self.eps = torch.zeros(1, dtype=torch.float32, device=device)

    def launch(self, lib, policy, epsilon, capture, stream, key=None):
        """key: anything else the captured launches bake in (the per-tick fused policy kernel takes epsilon by value)."""
        self.eps.fill_(float(epsilon))          # ordered before the replay on the same stream
        if self.graph is None or self.policy is not policy or self.key != key:
            self.destroy(lib)
            self.graph = _lib.capture_graph(self.device, capture)
            self.policy, self.key = policy, key
        _lib.check(lib.cm3_graph_launch(self.graph, stream))

    def destroy(self, lib):
        if self.graph is not None:
            torch.cuda.synchronize(self.device)     # a previous replay may still be in flight
            lib.cm3_graph_destroy(self.graph)
            self.graph = self.policy = None


class ParticleRollout(object):
    """T-tick trajectory over a VecParticleEnv.

    env.auto_reset False: episode-synchronous -- reset all envs, run T = max_steps ticks, transitions of
        an env after its `done` are flagged invalid (mirrors one reference episode per env).
    env.auto_reset True: continuous -- finished envs restart inside the launch; the true terminal
        next-state/obs are captured per tick, goals are recorded per slot; every transition is valid.
    """

    def __init__(self, env, n_ticks=None, use_graph=True, fused=False, record_collisions=True,
                 fused_policy_tick=False, live_state=None, policy_mode="auto", sparse_goals=None):
        self.env = env
        self.T = int(n_ticks or env.max_steps)
        self.use_graph = bool(use_graph)
        # fused: the random-action branch runs all T ticks in ONE launch (CM3_FLAG_FUSED_TICKS; state in
        # registers, identical results).  Policy-driven collection always launches once per tick.
        self.fused = bool(fused)
        # fused_policy_tick: policy-driven collection with ONE launch per tick -- actor forward pass, sampling and env step of
        # a tick in one cm3_policy_rollout_f32 launch (n_ticks = 1) instead of an actor launch followed by a step launch;
        # bit-identical to both other policy modes (n_agents in {1, 2, 4, 8})
        self.fused_policy_tick = bool(fused_policy_tick)
        # policy_mode: how collect(policy=<device actor>) launches (all three are bit-identical, tests/test_gpu_actor.py):
        #   "episode"  the whole policy-driven episode in ONE launch (csrc/policy.hip) -- the fastest (C2: 5.6 vs 10.6 us per tick);
        #   "tick"     an actor launch and a step launch per tick inside one hipGraph;
        #   "auto"     (default, round 3) "episode" whenever the fused kernel applies -- n_agents in {1, 2, 4, 8}, float32 env,
        #              actor.seed == env.seed (one Philox key) -- else "tick".  fused=True / fused_policy_tick=True
        #              still force their modes.
        if policy_mode not in ("auto", "episode", "tick"):
            raise Cm3Error("policy_mode must be 'auto', 'episode' or 'tick'")
        self.policy_mode = policy_mode
        # live_state: None = by size (see collect); True / False force stepping in place on the env's buffers with slot copies /
        # chaining the ticks through the slots.  Identical trajectories either way (tests/test_gpu_rollout.py).
        self.live_state = live_state
        # sparse_goals: random-action collection at streaming sizes writes a goals slot only where an env restarts (landmarks move
        # only at an episode start; cm3_particle_traj.goals_live alone) instead of all of them every tick -- 7 % of the bytes of a
        # tick that runs at the chip's copy rate (profiles/r04_trajectory_stream.txt).  None = by size (see collect), True / False
        # force it.  `goals` (and as_reference_batch) look the same either way: the dense array is filled in on first access.
        self.sparse_goals = sparse_goals
        self._goals_sparse = False
        self._graph_mode = None          # the _Mode the captured graphs were built with (+ the env's buffer parity when live)
        self.n_captures = 0              # hipGraph captures of the random-action collection so far (tests)
        self._goal_src = None
        self._goal_src32 = self._goal_src32_of = None
        E, N, L, T, dev, dt = env.E, env.n, env.L, self.T, env.device, env.dtype
        z = lambda *s, d=dt: torch.zeros(*s, dtype=d, device=dev)  # noqa: E731
        self.state = z(T + 1, N, E, 4)
        self.obs_others = z(T + 1, E, N, L)
        self.actions = z(T, E, N, d=torch.int32)
        self.reward = z(T, E)
        self.reward_n = z(T, E, N)
        self.done = z(T, E, d=torch.uint8)
        self.auto_reset = env.auto_reset
        if self.auto_reset:
            self._goals_buf = z(T + 1, N, E, 2)
            self.term_state = z(T, N, E, 4)
            self.term_obs_others = z(T, E, N, L)
        else:
            self._goals_buf = None
            self.term_state = self.term_obs_others = None
        # scenario.collisions after every tick, before any same-launch reset: the per-episode value the reference reads
        # at train_onpolicy.py:356 sits at the tick that ends the episode
        # (record_collisions=False drops the slot: episode_is_bad() is then unavailable)
        self.collisions = z(T, E, d=torch.int32) if record_collisions else None
        self._graph = None
        self._actor_graph = _ActorGraphCache(dev)
        self._live, self._live_cur = False, env._cur
        # episode-synchronous mode: which envs' episodes have ended since their last reset (carried across collects)
        self._finished = torch.zeros(E, dtype=torch.bool, device=dev)
        self._finished0 = None
        self._lib = _lib.lib()
        self.collected = False
        if hasattr(env, "_rollouts"):
            env._rollouts.add(self)          # env.reset() / env.set_state() outside the collector clear `_finished` (mark_reset)

    def mark_reset(self, mask=None):
        """Envs (bool / uint8 [E] mask; None = all) were re-seeded outside this collector -- env.reset(mask), env.set_state():
        their next episode is fresh, so transitions collected with reset=False are valid again.  VecParticleEnv calls this
        itself; call it by hand only for an env object that does not."""
        if mask is None:
            self._finished.zero_()
        else:
            m = torch.as_tensor(mask, device=self.env.device).bool().reshape(-1)
            self._finished &= ~m

    # ---- plumbing ------------------------------------------------------------------------------------
    def _traj(self, t0=0, live=False, sparse_goals=False):
        """live: per-tick launches step IN PLACE on the env's own state / goals buffers and write every tick's state / goals
        to its slot as a copy (cm3_particle_traj.state_live): a tick then loads lines its predecessor read and overwrote
        instead of a fresh slot that was only ever written -- 0.19 us of 2.83 per tick at C2 (tools/trajectory_gap.py)."""
        env, es = self.env, self.state.element_size()
        E, N, L = env.E, env.n, env.L
        t = _lib.ParticleTraj()
        if live:
            t.state_live = env._state[env._cur].data_ptr()
            t.goals_live = env._goals.data_ptr()
        t.state = self.state[t0].data_ptr()
        t.state_stride = N * E * 4 * es
        if self._goals_buf is not None:
            t.goals = self._goals_buf[t0].data_ptr()
            t.goals_stride = N * E * 2 * es
            if sparse_goals:
                t.goals_live = env._goals.data_ptr()
        else:
            t.goals = env._goals.data_ptr()
            t.goals_stride = 0
        t.obs_others = self.obs_others[t0].data_ptr()
        t.obs_others_stride = E * N * L * es
        t.actions = self.actions[t0].data_ptr()
        t.actions_stride = E * N * 4
        t.reward_n = self.reward_n[t0].data_ptr()
        t.reward_n_stride = E * N * es
        t.reward = self.reward[t0].data_ptr()
        t.reward_stride = E * es
        t.done = self.done[t0].data_ptr()
        t.done_stride = E
        t.meta = env._meta.data_ptr()
        t.episode = env._episode.data_ptr()
        if self.term_state is not None:
            t.term_state = self.term_state[t0].data_ptr()
            t.term_state_stride = N * E * 4 * es
            t.term_obs_others = self.term_obs_others[t0].data_ptr()
            t.term_obs_others_stride = E * N * L * es
        if self.collisions is not None:
            t.collisions = self.collisions[t0].data_ptr()
            t.collisions_stride = E * 4
        return t

    def _enqueue(self, t0, n, flags, stream=None, live=False, sparse_goals=False):
        env = self.env
        env._desc.flags = flags
        traj = self._traj(t0, live, sparse_goals)
        stream = env._stream() if stream is None else stream
        fn = getattr(self._lib, "cm3_particle_rollout_" + env._suffix)
        _lib.check(fn(ctypes.byref(env._desc), ctypes.byref(traj), int(n), stream))

    def _enqueue_tick0_from_env(self, flags, stream):
        """Tick 0 as a plain step launch that READS the env's own state / goals buffers (not slot 0) and writes slot 1 and the
        per-tick slots 0: the copy that records slot 0 then no longer stands between the caller and the first step launch
        (collect_normalized runs it on a parallel branch of its graph).  Values are those of `_enqueue(0, 1, ...)`."""
        env = self.env
        b = _lib.ParticleBufs()
        b.state_in = env._state[env._cur].data_ptr()
        b.state_out = self.state[1].data_ptr()
        b.goals_in = env._goals.data_ptr()
        b.goals_out = self._goals_buf[1].data_ptr() if self._goals_buf is not None else env._goals.data_ptr()
        b.meta_in = b.meta_out = env._meta.data_ptr()
        b.episode = env._episode.data_ptr()
        b.actions = self.actions[0].data_ptr()
        b.obs_others = self.obs_others[1].data_ptr()
        b.reward_n = self.reward_n[0].data_ptr()
        b.reward = self.reward[0].data_ptr()
        b.done = self.done[0].data_ptr()
        if self.term_state is not None:
            b.term_state = self.term_state[0].data_ptr()
            b.term_obs_others = self.term_obs_others[0].data_ptr()
        if self.collisions is not None:
            b.collisions_tick = self.collisions[0].data_ptr()
        env._desc.flags = flags
        fn = getattr(self._lib, "cm3_particle_step_" + env._suffix)
        _lib.check(fn(ctypes.byref(env._desc), ctypes.byref(b), stream))

    def _copy(self, pairs):
        """(dst, src) tensor pairs: one cm3_copy_list launch when every region is 16-byte sized / aligned."""
        _copy_pairs(pairs, self.env._stream())

    def _load_slot0(self):
        env = self.env
        pairs = [(self.state[0], env._state[env._cur]), (self.obs_others[0], env._obs_others[env._cur])]
        if self._goals_buf is not None:
            pairs.append((self._goals_buf[0], env._goals))
        self._copy(pairs)

    def _store_back(self, live=False):
        env = self.env
        pairs = [(env._obs_others[env._cur], self.obs_others[self.T])]
        if not live:    # (live: the env's state / goals buffers ARE the final state)
            pairs.append((env._state[env._cur], self.state[self.T]))
            if self._goals_buf is not None and not self._goals_sparse:     # (sparse goal slots: the live goals array is current)
                pairs.append((env._goals, self._goals_buf[self.T]))
        self._copy(pairs)

    # ---- collection ------------------------------------------------------------------------------------
    def _enqueue_actor_rollout(self, actor, epsilon, base_flags, stream):
        """T x (actor launch, step launch) on `stream`: the policy reads slot t, writes actions[t]; the step kernel
        consumes them and writes slot t+1 (train_onpolicy.py:311-323 without leaving the device)."""
        env = self.env
        live = self._live
        for t in range(self.T):
            goals = self._goals_buf[t] if (self._goals_buf is not None and not live) else env._goals   # (live: goals in place)
            actor.enqueue(env.E, self.obs_others[t], self.state[t], goals, env._meta, env._episode, self.actions[t],
                          epsilon, stream=stream, env_id_base=env.env_id_base)
            self._enqueue(t, 1, base_flags, stream, live=live)

    def _enqueue_fused_policy_ticks(self, actor, epsilon, base_flags, stream):
        """T launches of the fused policy kernel, one tick each (slot t -> slot t+1)."""
        env = self.env
        env._desc.flags = base_flags & FLAG_AUTO_RESET
        ad = actor._desc(env.E, epsilon, env.env_id_base)
        for t in range(self.T):
            traj = self._traj(t)
            _lib.check(self._lib.cm3_policy_rollout_f32(ctypes.byref(env._desc), ctypes.byref(traj), ctypes.byref(ad),
                                                        ctypes.byref(actor._wt), None, 0, 1, stream))

    def _mode(self, policy):
        """How this collect() launches -- the one place that decides it.  -> _Mode(kind, live, sparse):
          kind    random_fused | random            the reference's random-action branch, all ticks in one launch / a launch per tick
                  policy_episode                   on-device actor, the whole episode in ONE launch (csrc/policy.hip)
                  policy_fused_tick | policy_tick  on-device actor, one fused launch / an actor + a step launch per tick
                  host_policy                      a Python callable per tick
          live    per-tick step launches step IN PLACE on the env's buffers and copy every tick's state to its slot -- only while a
                  tick's state is small (<= 1 MiB: the slot copy is extra write traffic, the gain is load latency: C2 2.87 -> 2.74 us)
                  and the trajectory is streaming-size (>= 128 MB of observation slots: a small trajectory collected over and over stays
                  cache-resident and the copies only cost); measured in profiles/r02_trajectory_gap_live_state.txt,
                  r03_live_state_crossover.txt.  live_state = True / False forces it.
          sparse  goal slots are written only where an env restarts: always with live state (the goals live in place), and for the
                  random-action branch at streaming sizes (sparse_goals = True / False forces that)."""
        env = self.env
        dev_policy = policy is not None and hasattr(policy, "enqueue") and hasattr(policy, "act")
        if dev_policy and env.dtype != torch.float32:
            raise Cm3Error("the device actor reads float32 env buffers")
        same_key = dev_policy and getattr(policy, "seed", None) == env.seed
        if policy is None:
            kind = "random_fused" if self.fused else "random"
        elif not dev_policy:
            kind = "host_policy"
        else:
            episode_ok = env.n in (1, 2, 4, 8) and same_key and not self.fused_policy_tick
            if self.fused or (self.policy_mode in ("auto", "episode") and episode_ok):
                kind = "policy_episode"
            elif self.policy_mode == "episode":
                raise Cm3Error("policy_mode='episode' needs n_agents in {1, 2, 4, 8} and actor.seed == env.seed")
            else:
                kind = "policy_fused_tick" if self.fused_policy_tick else "policy_tick"
            if kind in ("policy_episode", "policy_fused_tick") and not same_key:
                raise Cm3Error("fused policy launches need actor.seed == env.seed (one Philox key)")
        es = self.state.element_size()
        stream_size = env.E * env.n * env.L * es * self.T >= (128 << 20)
        small = env.n * env.E * 4 * es <= (1 << 20) and stream_size
        if self.live_state is not None:
            small = bool(self.live_state)
        per_tick_steps = kind in ("random", "policy_tick", "host_policy") and not self.fused
        live = bool(self._goals_buf is not None and small and per_tick_steps)
        sparse = bool(self._goals_buf is not None and kind != "random_fused" and (
            live or (kind == "random" and bool(stream_size if self.sparse_goals is None else self.sparse_goals))))
        return _Mode(kind, live, sparse)

    def collect(self, policy=None, reset=None, epsilon=0.0):
        """Runs T ticks.  policy None = the reference's random-action branch (train_onpolicy.py:305-307,
        drawn in-kernel; the whole rollout is one hipGraph replay); otherwise ``policy(obs_others [E,N,L],
        obs_self [E,N,4], goals [E,N,2]) -> int actions [E,N]`` is called every tick (:311-313).
        reset: None -> reset at the start iff the env does not auto-reset."""
        env = self.env
        if reset is None:
            reset = not self.auto_reset
        if reset:
            env.reset()
            self._finished.zero_()
        # envs whose episode already ended in an earlier collect (no auto-reset, reset=False) stay invalid
        self._finished0 = None if self.auto_reset else self._finished.clone()
        self._load_slot0()
        base = (FLAG_AUTO_RESET if self.auto_reset else 0) | env.kernel_flags
        mode = self._mode(policy)
        live, sparse = mode.live, mode.sparse
        self._live = live
        # ONE key for every captured graph of this object: the mode, plus the env's buffer parity where the launches hold the
        # addresses of the env's current buffers (live state; env.step() flips them)
        key = (mode, env._cur if live else None)
        if self._graph_mode is not None and self._graph_mode != key:
            self._drop_graphs()
        self._graph_mode = key
        if self._live_cur != env._cur:
            self._drop_norm_graph()       # (collect_normalized's graph reads the env's current buffers in every mode)
            self._live_cur = env._cur
        # `_goals_sparse` only says whether the goal slots of the LAST collection still need completing (the `goals` getter clears it)
        self._goals_sparse, self._goal_src = sparse, None
        stream = env._stream()
        if mode.kind == "random_fused":
            self._enqueue(0, self.T, base | FLAG_GEN_ACTIONS | _lib.FLAG_FUSED_TICKS)
        elif mode.kind == "random":
            flags = base | FLAG_GEN_ACTIONS
            if self.use_graph:
                if self._graph is None:
                    self.n_captures += 1
                    self._graph = _lib.capture_graph(env.device, lambda s: self._enqueue(0, self.T, flags, s, live=live, sparse_goals=sparse))
                _lib.check(self._lib.cm3_graph_launch(self._graph, stream))
            else:
                self._enqueue(0, self.T, flags, live=live, sparse_goals=sparse)
        elif mode.kind == "policy_episode":
            # the whole policy-driven episode in ONE launch (csrc/policy.hip): weights, observation tile and env state stay in
            # LDS / registers for all T ticks; bit-identical to alternating actor / step launches
            env._desc.flags = base & FLAG_AUTO_RESET
            traj = self._traj(0)
            ad = policy._desc(env.E, epsilon, env.env_id_base)
            _lib.check(self._lib.cm3_policy_rollout_f32(ctypes.byref(env._desc), ctypes.byref(traj), ctypes.byref(ad),
                                                        ctypes.byref(policy._wt), None, 0, self.T, stream))
        elif mode.kind == "policy_fused_tick":
            if self.use_graph:      # epsilon is a by-value argument of these launches: part of the graph's key
                self._actor_graph.launch(self._lib, policy, epsilon, lambda s: self._enqueue_fused_policy_ticks(policy, epsilon, base, s),
                                         stream, key=("fused_tick", float(epsilon)))
            else:
                self._enqueue_fused_policy_ticks(policy, epsilon, base, stream)
        elif mode.kind == "policy_tick":
            if self.use_graph:
                cache = self._actor_graph
                cache.launch(self._lib, policy, epsilon, lambda s: self._enqueue_actor_rollout(policy, cache.eps, base, s), stream)
            else:
                self._enqueue_actor_rollout(policy, epsilon, base, stream)
        else:                       # a host policy: one call and one step launch per tick
            for t in range(self.T):
                goals = (self._goals_buf[t] if (self._goals_buf is not None and not live) else env._goals).permute(1, 0, 2)
                a = policy(self.obs_others[t], self.state[t].permute(1, 0, 2), goals)
                self.actions[t].copy_(torch.as_tensor(a, device=env.device).reshape(env.E, env.n))
                self._enqueue(t, 1, base, live=live)
        self._store_back(live)
        if not self.auto_reset:
            self._finished |= self.done.bool().any(0)
        self.collected = True
        return self

    def collect_normalized(self, gamma=0.99, eps=1e-8, normalize=True, group=None, time_collective=False, segments=1):
        """Random-action collection (continuous mode) FOLLOWED by the advantage-normalisation step over reward_n -- BASELINE
        configs[3]: discounted returns + this rank's float64 moments, ONE 24-byte all-gather over the ranks, normalisation
        with the global statistics (cm3_amd.shard).  With one rank the collective is the identity and the whole step --
        slot copy in, T step launches, slot copy out, returns + moments, normalise -- is ONE hipGraph replay; with several
        ranks the graph ends at the moments, the all-gather and the normalise launch follow eagerly (RCCL / gloo).
        Returns (normalised returns [T,E,N], (mean, std, count) device scalars); with time_collective=True also the seconds
        the host spent in the all-gather call (0.0 for one rank).
        segments = K > 1 (round 4): the T ticks are K consecutive rollouts of T / K ticks -- one collection phase
        (train_onpolicy.py:359: episodes_per_train rollouts between two training steps) -- each with its own returns and its own
        normalisation statistics ((mean, std, count) become [K] tensors); the K advantage steps are the same two launches at the
        end of ONE chain of T step launches, the whole phase one hipGraph replay, and with several ranks the K moment triples
        travel in ONE all-gather.  Values per rollout are bit-identical to K calls with segments = 1 on a T / K-tick collector."""
        import time
        from .shard import ReturnsNormalizer, gather_moments
        import torch.distributed as dist
        env = self.env
        if not self.auto_reset or self.fused:
            raise Cm3Error("collect_normalized runs the continuous, one-launch-per-tick random-action collection")
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        key = (float(gamma), float(eps), bool(normalize), world > 1, int(segments))
        if getattr(self, "_norm_key", None) != key:
            self._drop_norm_graph()
            self._norm = ReturnsNormalizer(self.reward_n, self.done, gamma, eps, normalize, segments=segments)
            self._norm_key = key
        self._finished0 = None
        es = self.state.element_size()
        small = env.n * env.E * 4 * es <= (1 << 20) and env.E * env.n * env.L * es * self.T >= (128 << 20)
        live = self._live = bool(small if self.live_state is None else self.live_state)
        if self._graph_mode is not None and self._graph_mode != (live, live):
            self._drop_graphs()
        self._graph_mode = (live, live)
        self._goals_sparse, self._goal_src = live, None      # (the live-state launches write a goals slot only where an env restarts)
        # The captured graph bakes in the addresses of env._state[env._cur] / env._obs_others[env._cur] (tick 0 reads them, the
        # slot bookkeeping writes them) -- live or not -- and VecParticleEnv.step() flips env._cur: re-capture after a flip.
        if self._live_cur != env._cur:
            self._drop_graphs()
            self._live_cur = env._cur
        flags = FLAG_AUTO_RESET | env.kernel_flags | FLAG_GEN_ACTIONS

        def enqueue(s):
            pairs = [(self.state[0], env._state[env._cur]), (self.obs_others[0], env._obs_others[env._cur]),
                     (self._goals_buf[0], env._goals)]
            back = [(env._obs_others[env._cur], self.obs_others[self.T])]
            if not live:
                back += [(env._state[env._cur], self.state[self.T]), (env._goals, self._goals_buf[self.T])]
            if not live and world == 1 and _pairs_aligned(pairs + back):
                # one rank, one graph: tick 0 reads the env's buffers directly (not slot 0), and the launch that computes the
                # returns and their partial moments also records slot 0 from the env's buffers and then leaves slot T in them --
                # the chain of dependent launches is T step launches + 2, not T + 4
                self._enqueue_tick0_from_env(flags, s)
                if self.T > 1:
                    self._enqueue(1, self.T - 1, flags, s, live=False)
                self._norm.enqueue_fused(s, [(self.state[0], env._state[env._cur], self.state[self.T]),
                                             (self.obs_others[0], env._obs_others[env._cur], self.obs_others[self.T]),
                                             (self._goals_buf[0], env._goals, self._goals_buf[self.T])])
                return
            _copy_pairs(pairs, s)
            self._enqueue(0, self.T, flags, s, live=live)
            _copy_pairs(back, s)
            if world == 1:
                self._norm.enqueue_fused(s)        # (= enqueue_moments + enqueue_normalize on the own moments, bit for bit)
            else:
                self._norm.enqueue_moments(s)

        stream = env._stream()
        if self.use_graph:
            if getattr(self, "_norm_graph", None) is None:
                self._norm_graph = _lib.capture_graph(env.device, enqueue)
            _lib.check(self._lib.cm3_graph_launch(self._norm_graph, stream))
        else:
            enqueue(stream)
        t_coll = 0.0
        if world > 1:
            t0 = time.perf_counter()
            parts, n_parts = gather_moments(self._norm.moments, group)
            t_coll = time.perf_counter() - t0
            self._norm.enqueue_normalize(stream, parts, n_parts)
        self.collected = True
        st = self._norm.stats
        res = (self._norm.out, (st[0], st[1], st[2]) if st.dim() == 1 else (st[:, 0], st[:, 1], st[:, 2]))
        return res + (t_coll,) if time_collective else res

    def _drop_norm_graph(self):
        if getattr(self, "_norm_graph", None) is not None:
            torch.cuda.synchronize(self.env.device)
            self._lib.cm3_graph_destroy(self._norm_graph)
        self._norm_graph = None

    def _drop_graphs(self):
        self._drop_norm_graph()
        if self._graph is not None:
            torch.cuda.synchronize(self.env.device)
            self._lib.cm3_graph_destroy(self._graph)
            self._graph = None
        self._actor_graph.destroy(self._lib)

    def close(self):
        self._drop_graphs()

    # ---- views --------------------------------------------------------------------------------------------
    @property
    def goals(self):
        """[T + 1, N, E, 2] landmark positions per slot (None for an episode-synchronous collector: they do not change).  After a
        collection with sparse goal slots the array is completed here, once, on first access."""
        if self._goals_sparse:
            idx = self._goal_source_slots().view(self.T + 1, 1, self.env.E, 1).expand_as(self._goals_buf)
            self._goals_buf.copy_(self._goals_buf.gather(0, idx))
            self._goals_sparse = False
        return self._goals_buf

    @goals.setter
    def goals(self, value):
        self._goals_buf, self._goals_sparse = value, False

    def _goal_source_slots(self):
        """int64 [T + 1, E]: the slot that holds the goals in effect at slot t of env e -- the last slot <= t that was written
        (slot 0, or slot k + 1 of a tick k that ended an episode)."""
        if self._goal_src is None:
            T, E = self.T, self.env.E
            src = torch.zeros(T + 1, E, dtype=torch.int64, device=self.env.device)
            ticks = torch.arange(1, T + 1, device=self.env.device, dtype=torch.int64).view(T, 1)
            src[1:] = torch.where(self.done.bool(), ticks, torch.zeros_like(ticks))
            self._goal_src = torch.cummax(src, dim=0).values
        return self._goal_src

    @property
    def valid(self):
        """bool [T, E]"""
        if self.auto_reset:
            return torch.ones(self.T, self.env.E, dtype=torch.bool, device=self.env.device)
        return _valid_from_done(self.done, self._finished0)

    def episode_is_bad(self):
        """The reference's dual-buffer flag `scenario.collisions != 0` per finished episode (train_onpolicy.py:356), read
        from the per-tick collision counts of the trajectory at the tick that ENDS the episode -- collisions of an env
        that is stepped on after its `done` (episode-synchronous mode) do not count, as in the reference, whose loop
        stops there (:302).
        Episode-synchronous mode: bool [E]; an env still running at the end of this trajectory reports its count so far.
        Continuous mode: bool [T, E], set only where ``done`` is set (the count of the episode that ended at that tick,
        captured before the same-launch reset zeroes the live counter)."""
        if self.collisions is None:
            raise Cm3Error("episode_is_bad() needs ParticleRollout(record_collisions=True)")
        d = self.done.bool()
        if self.auto_reset:
            return (self.collisions != 0) & d
        T = self.T
        first = torch.where(d.any(0), d.to(torch.uint8).argmax(0), torch.full_like(d[0], T - 1, dtype=torch.long))
        return self.collisions.gather(0, first.unsqueeze(0)).squeeze(0) != 0

    def episode_returns(self):
        """(reward_global [E], reward_local [E,N]) accumulated over each env's episode
        (train_onpolicy.py:349-350); episode-synchronous mode."""
        v = self.valid.to(self.reward.dtype)
        return (self.reward * v).sum(0), (self.reward_n * v.unsqueeze(2)).sum(0)

    def _next(self, name, tt, ee):
        """next_* of transitions (tt, ee): slot t+1, or the captured terminal values where the env restarted."""
        if name == "state":
            nxt = self.state[tt + 1, :, ee]                    # [B, N, 4]
            if self.term_state is not None:
                term = self.term_state[tt, :, ee]
                d = self.done[tt, ee].bool().view(-1, 1, 1)
                nxt = torch.where(d, term, nxt)
            return nxt
        nxt = self.obs_others[tt + 1, ee]
        if self.term_obs_others is not None:
            term = self.term_obs_others[tt, ee]
            d = self.done[tt, ee].bool().view(-1, 1, 1)
            nxt = torch.where(d, term, nxt)
        return nxt

    def valid_indices(self):
        """(tt, ee) int64 tensors of all valid transitions, time-major."""
        idx = self.valid.nonzero(as_tuple=False)
        return idx[:, 0], idx[:, 1]

    def as_reference_batch(self, tt=None, ee=None, numpy=True, out=None):
        """Columns of the reference's transition batch for the (tick, env) pairs (tt, ee) (default: all valid
        ones), each equal to np.stack(batch[:, k]) in alg_credit.process_batch (alg_credit.py:458-470).
        out: a dict this call returned earlier for the same number of transitions (float32 path, numpy=False): the columns are
        written into those tensors again (persistent addresses: what a captured hipGraph of the consumer needs).
        float32 trajectories: ONE launch of cm3_transitions_gather_f32 (csrc/batch.hip) fills all columns; the float64 parity
        instantiation goes through the torch composition below (as_reference_batch_torch: same values, ~25 launches)."""
        everything = tt is None and self.auto_reset and self.state.dtype == torch.float32     # (all T x E transitions are valid)
        if tt is None and not everything:
            tt, ee = self.valid_indices()
        if not everything:
            tt = torch.as_tensor(tt, device=self.env.device, dtype=torch.long).contiguous()
            ee = torch.as_tensor(ee, device=self.env.device, dtype=torch.long).contiguous()
        if self.state.dtype != torch.float32:
            return self.as_reference_batch_torch(tt, ee, numpy)
        env = self.env
        B, N, L, dev = (self.T * env.E if everything else tt.numel()), env.n, env.L, env.device
        if out is not None:
            if numpy or out["v_global"].shape[0] != B or out["v_global"].data_ptr() != out["v_local"].data_ptr():
                raise Cm3Error("as_reference_batch(out=...): pass the dict an earlier call returned for %d transitions (numpy=False)" % B)
            state, obs, nstate, nobs = out["v_global"], out["obs_others"], out["v_global_next"], out["obs_others_next"]
            reward, reward_n, goals, actions, done = out["reward"], out["reward_local"], out["goals"], out["actions"], out["done"]
        else:
            f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)      # noqa: E731
            state, obs, nstate, nobs = f(B, N, 4), f(B, N, L), f(B, N, 4), f(B, N, L)
            reward, reward_n, goals = f(B), f(B, N), f(B, N, 2)
            actions = torch.empty(B, N, dtype=torch.int32, device=dev)
            done = torch.empty(B, dtype=torch.bool, device=dev)
        traj = self._traj(0)
        goal_slot, gs_stride = None, 0
        if self._goals_buf is not None and self._goals_sparse:
            if self._goal_src32 is None or self._goal_src32_of is not self._goal_source_slots():
                self._goal_src32_of = self._goal_source_slots()
                self._goal_src32 = self._goal_src32_of.to(torch.int32)
            goal_slot, gs_stride = self._goal_src32, env.E * 4
        out = _lib.TransitionCols()
        out.state, out.obs_others, out.actions, out.reward, out.reward_n = (state.data_ptr(), obs.data_ptr(), actions.data_ptr(),
                                                                            reward.data_ptr(), reward_n.data_ptr())
        out.next_state, out.next_obs_others, out.done, out.goals = nstate.data_ptr(), nobs.data_ptr(), done.data_ptr(), goals.data_ptr()
        _lib.check(self._lib.cm3_transitions_gather_f32(ctypes.byref(env._desc), ctypes.byref(traj), _lib.ptr(goal_slot), gs_stride,
                                                        None if everything else tt.data_ptr(), None if everything else ee.data_ptr(), B,
                                                        ctypes.byref(out), env._stream()))
        cols = dict(v_global=state, obs_others=obs, v_local=state, actions=actions, reward=reward, reward_local=reward_n,
                    v_global_next=nstate, obs_others_next=nobs, v_local_next=nstate, done=done, goals=goals)
        if numpy:
            cols = {k: v.detach().cpu().numpy() for k, v in cols.items()}
        return cols

    def export_into(self, columns, ring_start, ring_size):
        """Every transition of the trajectory (continuous collection, float32) written straight into the rows (ring_start + b) mod
        ring_size of `columns` -- a dict with the nine distinct tensors of as_reference_batch (v_global, obs_others, actions, reward,
        reward_local, v_global_next, obs_others_next, done, goals), each with ring_size rows: export and replay add in ONE launch
        (DeviceReplayBuffer.add_rollout).  Returns the number of transitions."""
        env = self.env
        if not (self.auto_reset and self.state.dtype == torch.float32):
            raise Cm3Error("export_into needs a continuous float32 collection (every transition valid)")
        B = self.T * env.E
        traj = self._traj(0)
        goal_slot, gs_stride = None, 0
        if self._goals_buf is not None and self._goals_sparse:
            if self._goal_src32 is None or self._goal_src32_of is not self._goal_source_slots():
                self._goal_src32_of = self._goal_source_slots()
                self._goal_src32 = self._goal_src32_of.to(torch.int32)
            goal_slot, gs_stride = self._goal_src32, env.E * 4
        out = _lib.TransitionCols()
        for field, name in (("state", "v_global"), ("obs_others", "obs_others"), ("actions", "actions"), ("reward", "reward"),
                            ("reward_n", "reward_local"), ("next_state", "v_global_next"), ("next_obs_others", "obs_others_next"),
                            ("done", "done"), ("goals", "goals")):
            t = columns[name]
            want_dt = {"actions": torch.int32, "done": (torch.bool, torch.uint8)}.get(name, torch.float32)
            want_tail = {"v_global": (env.n, 4), "v_global_next": (env.n, 4), "obs_others": (env.n, env.L), "obs_others_next": (env.n, env.L),
                         "actions": (env.n,), "reward": (), "reward_local": (env.n,), "done": (), "goals": (env.n, 2)}[name]
            if (not t.is_contiguous() or t.shape[0] != ring_size or tuple(t.shape[1:]) != want_tail
                    or (t.dtype not in want_dt if isinstance(want_dt, tuple) else t.dtype != want_dt) or not t.is_cuda):
                raise Cm3Error("export_into: column %s must be a contiguous device tensor [%d%s] of %s (a wrong layout would be an "
                               "out-of-bounds device write)" % (name, ring_size, "".join(", %d" % d for d in want_tail), want_dt))
            setattr(out, field, t.data_ptr())
        out.ring_start, out.ring_size = int(ring_start), int(ring_size)
        _lib.check(self._lib.cm3_transitions_gather_f32(ctypes.byref(env._desc), ctypes.byref(traj), _lib.ptr(goal_slot), gs_stride,
                                                        None, None, B, ctypes.byref(out), env._stream()))
        return B

    def as_reference_batch_torch(self, tt, ee, numpy=True):
        """The same columns as a composition of torch indexing operations (any dtype; what the kernel path is tested against)."""
        tt = torch.as_tensor(tt, device=self.env.device, dtype=torch.long)
        ee = torch.as_tensor(ee, device=self.env.device, dtype=torch.long)
        state = self.state[tt, :, ee]                                   # [B, N, 4]
        nxt_state = self._next("state", tt, ee)
        if self._goals_buf is None:
            goals = self.env._goals[:, ee].permute(1, 0, 2)             # [B, N, 2]
        elif self._goals_sparse:                                        # (gathered from the last written slot: nothing is filled in)
            goals = self._goals_buf[self._goal_source_slots()[tt, ee], :, ee]
        else:
            goals = self._goals_buf[tt, :, ee]
        cols = dict(
            v_global=state, obs_others=self.obs_others[tt, ee], v_local=state,
            actions=self.actions[tt, ee], reward=self.reward[tt, ee], reward_local=self.reward_n[tt, ee],
            v_global_next=nxt_state, obs_others_next=self._next("obs_others", tt, ee), v_local_next=nxt_state,
            done=self.done[tt, ee].bool(), goals=goals)
        if numpy:
            cols = {k: v.detach().cpu().numpy() for k, v in cols.items()}
        return cols

    ORDER = ("v_global", "obs_others", "v_local", "actions", "reward", "reward_local", "v_global_next",
             "obs_others_next", "v_local_next", "done", "goals")

    def as_reference_rows(self, tt=None, ee=None):
        """Object ndarray [B, 11]: row b is the reference's np.array([...11 fields...]) of train_onpolicy.py:338
        (dtype=object, as NumPy >= 1.24 requires for ragged rows)."""
        return rows_from_columns(self.as_reference_batch(tt, ee), self.ORDER)

    def _sample_positions(self, size, rows, generator):
        """int64 [rows, k] positions into the valid transitions (time-major), each row `size` distinct ones uniformly at random
        (random.sample, replay_buffer.py:34), or all of them when there are <= size -- and the (tt, ee) of all valid transitions,
        None when every transition is valid (continuous collection: position b = t E + e, nothing to enumerate)."""
        if self.auto_reset:
            n, tt, ee = self.T * self.env.E, None, None
        else:
            tt, ee = self.valid_indices()
            n = tt.numel()
        if n <= size:
            return torch.arange(n, device=self.env.device).unsqueeze(0).expand(rows, n), tt, ee
        return sample_distinct(n, size, rows, generator, self.env.device), tt, ee

    def sample_batch(self, size, generator=None, numpy=True):
        """replay_buffer.sample_batch (replay_buffer.py:28-37): all transitions if there are <= size of them,
        else `size` distinct ones uniformly at random."""
        pos, tt, ee = self._sample_positions(size, 1, generator)
        pos = pos[0]
        if tt is None:
            E = self.env.E
            if pos.numel() == self.T * E:
                return self.as_reference_batch(numpy=numpy)
            return self.as_reference_batch(torch.div(pos, E, rounding_mode="floor"), pos % E, numpy=numpy)
        return self.as_reference_batch(tt[pos], ee[pos], numpy=numpy)

    def on_policy_minibatches(self, epochs=24, batch_size=128, generator=None, numpy=False):
        """The on-policy cadence of train_onpolicy.py:359-377: after a collection phase, `epochs` minibatches of
        `batch_size` transitions are sampled from the freshly collected buffer, which is then discarded (the next
        collect() overwrites the trajectory).  The reference collects episodes_per_train = 10 episodes (<= 330
        transitions) per phase; a vectorised phase holds n_envs episodes.
        All minibatches of the phase are drawn together and exported by ONE launch (round 5: 24 separate samples cost 24 permutations
        of the phase's 1.35 M transitions, 7 ms against the 0.8 ms of collecting them); each yielded dict holds views of that export."""
        cols, k = self._phase_export(epochs, batch_size, generator)
        for m in range(int(epochs)):
            mb = {name: v[m * k:(m + 1) * k] for name, v in cols.items()}
            yield ({name: v.detach().cpu().numpy() for name, v in mb.items()} if numpy else mb)

    def _phase_export(self, epochs, batch_size, generator, out=None):
        """-> (columns of all `epochs` minibatches back to back [epochs * k, ...], k = transitions per minibatch): one draw, one launch"""
        epochs = int(epochs)
        pos, tt, ee = self._sample_positions(batch_size, epochs, generator)
        self.last_sample_positions = pos          # (tests: which transitions the minibatches hold)
        k = pos.shape[1]
        flat = pos.reshape(-1)
        if tt is None:
            E = self.env.E
            cols = self.as_reference_batch(torch.div(flat, E, rounding_mode="floor"), flat % E, numpy=False, out=out)
        else:
            cols = self.as_reference_batch(tt[flat], ee[flat], numpy=False, out=out)
        return cols, k

    def on_policy_phase(self, epochs=24, batch_size=128, generator=None, l_action=5):
        """The minibatches of on_policy_minibatches() TOGETHER WITH the static feeds of the reference's train_step for each of them
        (cm3_amd.batch.phase_static_feeds: process_actions / process_global_state, the n x n credit repeats, the n x n x l_action
        counterfactual tiling of alg_credit.py:406-557, :614-658, :730-751) -- for the whole phase one export launch and one pair of
        tiling launches; a list of (columns, static) pairs of views, static None where the tiling kernel does not apply (N = 1, float64).
        Feed them to cm3_amd.batch.train_step_feeds(columns, run, gamma, epsilon, static=static)."""
        from . import batch as B
        cols, k = self._phase_export(epochs, batch_size, generator)
        vg = cols["v_global"]
        if vg.is_cuda and vg.dtype == torch.float32 and vg.shape[1] > 1:
            statics = B.phase_static_feeds(cols, int(epochs), l_action)
        else:
            statics = [None] * int(epochs)
        return [({name: v[m * k:(m + 1) * k] for name, v in cols.items()}, statics[m]) for m in range(int(epochs))]


def sample_distinct(n, size, rows, generator, device):
    """int64 [rows, size]: every row `size` DISTINCT integers of range(n), uniformly at random over the size-subsets and their orders
    (the reference's random.sample).  n small: a permutation per row.  n >> size (a phase holds 10^6 transitions, a minibatch 128):
    2 size draws with replacement per row and the first `size` distinct values in draw order -- sequential drawing with rejection of
    repeats is exactly sampling without replacement -- found with a stable sort, no host synchronisation: O(size log size) work instead
    of a permutation of all n.  (Fewer than `size` distinct values among 2 size draws from n >= 64 size needs more than `size`
    collisions: probability below 1e-100; the positions such a row would leave unset hold 0 .. size - 1.)"""
    if n < 64 * size:
        return torch.stack([torch.randperm(n, generator=generator, device=device)[:size] for _ in range(rows)])
    m = 2 * size
    x = torch.randint(0, n, (rows, m), generator=generator, device=device)
    sx, order = torch.sort(x, dim=1, stable=True)
    first_sorted = torch.ones_like(sx, dtype=torch.bool)
    first_sorted[:, 1:] = sx[:, 1:] != sx[:, :-1]                    # first of each run of equal values = its earliest draw (stable)
    first = torch.zeros_like(first_sorted).scatter_(1, order, first_sorted)      # ... in draw order
    rank = torch.cumsum(first, dim=1) - 1
    dst = torch.where(first & (rank < size), rank, torch.full_like(rank, size))
    out = torch.arange(size + 1, device=device).repeat(rows, 1)
    out.scatter_(1, dst, x)
    return out[:, :size].contiguous()


class CheckersRollout(object):
    """T-tick trajectory over a VecCheckersEnv (16-column transitions, train_onpolicy.py:336).

    env.auto_reset False: episode-synchronous -- the env is reset at the start of every collect(); transitions of an
        env after its `done` are flagged invalid (one reference episode per env).
    env.auto_reset True: continuous -- finished envs restart inside the launch (CM3_FLAG_AUTO_RESET); slot t+1 then
        holds the FRESH episode's observation, the true terminal next_* of the transition are captured per tick
        (term_*; the reference stores them before it resets, train_onpolicy.py:336-347), the goals in effect are
        recorded per slot (they change when a single-agent env restarts, :288-291), and `actions_prev` of the first
        transition of an episode is zeros (:295).  Every transition is valid; collect() continues where the previous
        one stopped.
    """

    def __init__(self, env, n_ticks=None, use_graph=True, fused=False, policy_mode="auto", record_probs=False):
        """policy_mode (collect(policy=<CheckersActor>)): "auto" = the whole rollout in ONE launch (cm3_policy_rollout_checkers)
        wherever that kernel applies (CheckersActor.fused_rollout_ok), else -- and with "tick" -- an actor launch and a step
        launch per tick inside one hipGraph; the two produce the same bits.  record_probs: keep the mixed probabilities the
        actions were drawn from, float32 [T, E, N, 5] (self.probs)."""
        self.env = env
        self.T = int(n_ticks or env.max_steps)
        self.use_graph = bool(use_graph)
        self.fused = bool(fused)      # random-action branch in ONE launch (CM3_FLAG_FUSED_TICKS; fast kernel only)
        if policy_mode not in ("auto", "tick"):
            raise Cm3Error("policy_mode must be 'auto' or 'tick'")
        self.policy_mode = policy_mode
        self.auto_reset = bool(env.auto_reset)
        self._graph = None
        E, N, T, dev = env.E, env.n, self.T, env.device
        z = lambda *s, d: torch.zeros(*s, dtype=d, device=dev)  # noqa: E731
        self._grid_raw = z(T + 1, E, env.grid_stride, d=torch.int8)
        self._obst_raw = z(T + 1, E, env.obst_stride, d=torch.int8)
        self.grid = env.grid_view(self._grid_raw)              # [T+1, E, R, C+1, 2] view (padding hidden)
        self.vec = z(T + 1, E, N, 4, d=torch.int32)
        self.obs_others = z(T + 1, E, N, env.Lo, d=torch.float64)
        self.obs_self_t = env.obst_view(self._obst_raw)        # [T+1, E, N, K, K, 3] view
        self.obs_self_v = z(T + 1, E, N, 4, d=torch.float64)
        self.actions = z(T, E, N, d=torch.int32)
        self.local_rewards = z(T, E, N, d=torch.float64)
        self.reward = z(T, E, d=torch.float64)
        self.done = z(T, E, d=torch.uint8)
        self._prev_bufs = (z(E, N, d=torch.int32), z(E, N, d=torch.int32), z(E, N, d=torch.int32))
        self.prev0 = self._prev_bufs[1]                        # actions_prev of slot 0 (zeros at an episode start)
        self._rolled = None
        self.probs = z(T, E, N, 5, d=torch.float32) if record_probs else None
        if self.auto_reset:
            self._term_grid_raw = z(T, E, env.grid_stride, d=torch.int8)
            self._term_obst_raw = z(T, E, env.obst_stride, d=torch.int8)
            self.term_grid = env.grid_view(self._term_grid_raw)
            self.term_obs_self_t = env.obst_view(self._term_obst_raw)
            self.term_vec = z(T, E, N, 4, d=torch.int32)
            self.term_obs_others = z(T, E, N, env.Lo, d=torch.float64)
            self.term_obs_self_v = z(T, E, N, 4, d=torch.float64)
            self.goal_slots = z(T + 1, E, N, d=torch.uint8)
        else:
            self.goal_slots = None
        self._started = False
        self._goals_onehot = None
        self._traj_cache = None               # (the ctypes structs of the one-launch rollout: the buffers never move)
        self._actor_graph = _ActorGraphCache(dev)
        self._lib = _lib.lib()

    def _bufs(self, t):
        env = self.env
        b = _lib.CheckersBufs()
        b.mask, b.agents, b.steps = env._mask.data_ptr(), env._agents.data_ptr(), env._steps.data_ptr()
        b.episode, b.goals = env._episode.data_ptr(), env._goals.data_ptr()
        b.actions = self.actions[t].data_ptr()
        b.grid, b.vec = self._grid_raw[t + 1].data_ptr(), self.vec[t + 1].data_ptr()
        b.obs_others = self.obs_others[t + 1].data_ptr()
        b.obs_self_t, b.obs_self_v = self._obst_raw[t + 1].data_ptr(), self.obs_self_v[t + 1].data_ptr()
        b.local_rewards, b.reward, b.done = (self.local_rewards[t].data_ptr(), self.reward[t].data_ptr(),
                                             self.done[t].data_ptr())
        if self.auto_reset:
            b.term_grid, b.term_vec = self._term_grid_raw[t].data_ptr(), self.term_vec[t].data_ptr()
            b.term_obs_others = self.term_obs_others[t].data_ptr()
            b.term_obs_self_t, b.term_obs_self_v = self._term_obst_raw[t].data_ptr(), self.term_obs_self_v[t].data_ptr()
            b.goals_next = self.goal_slots[t + 1].data_ptr()
        return b

    def _traj(self):
        env = self.env
        t = _lib.CheckersTraj()
        t.mask, t.agents, t.steps = env._mask.data_ptr(), env._agents.data_ptr(), env._steps.data_ptr()
        t.episode, t.goals = env._episode.data_ptr(), env._goals.data_ptr()
        t.action_block = env._action_block.data_ptr()
        def slot(x):
            return x.data_ptr(), x[0].numel() * x.element_size()
        t.actions, t.actions_stride = slot(self.actions)
        t.grid, t.grid_slot_stride = slot(self._grid_raw)
        t.vec, t.vec_stride = slot(self.vec)
        t.obs_others, t.obs_others_stride = slot(self.obs_others)
        t.obs_self_t, t.obs_self_t_slot_stride = slot(self._obst_raw)
        t.obs_self_v, t.obs_self_v_stride = slot(self.obs_self_v)
        t.local_rewards, t.local_rewards_stride = slot(self.local_rewards)
        t.reward, t.reward_stride = slot(self.reward)
        t.done, t.done_stride = slot(self.done)
        if self.auto_reset:
            t.term_grid, t.term_grid_slot_stride = slot(self._term_grid_raw)
            t.term_vec, t.term_vec_stride = slot(self.term_vec)
            t.term_obs_others, t.term_obs_others_stride = slot(self.term_obs_others)
            t.term_obs_self_t, t.term_obs_self_t_slot_stride = slot(self._term_obst_raw)
            t.term_obs_self_v, t.term_obs_self_v_stride = slot(self.term_obs_self_v)
            t.goals_slots, t.goals_slots_stride = slot(self.goal_slots)
        return t

    def _base_flags(self):
        return FLAG_AUTO_RESET if self.auto_reset else 0

    def _enqueue_random(self, stream, fused):
        env = self.env
        env._desc.flags = self._base_flags() | FLAG_GEN_ACTIONS | (_lib.FLAG_FUSED_TICKS if fused else 0)
        traj = self._traj()
        _lib.check(self._lib.cm3_checkers_rollout(ctypes.byref(env._desc), ctypes.byref(traj), self.T, stream))
        env._desc.flags = 0

    def _enqueue_actor_rollout(self, actor, epsilon, stream):
        """T x (actor launch, step launch) on `stream`: the policy reads trajectory slot t (+ actions[t-1] as
        actions_prev -- prev0 at t = 0, zeros where the previous tick ended an episode: train_onpolicy.py:295,345) and
        writes actions[t]; the step kernel consumes them and writes slot t+1 (train_onpolicy.py:309-321 without leaving
        the device)."""
        env = self.env
        for t in range(self.T):
            goals = self.goal_slots[t] if self.auto_reset else env._goals
            actor.enqueue(env.E, self._obst_raw[t], env.obst_stride, self.obs_self_v[t], self.obs_others[t], goals,
                          self.actions[t - 1] if t > 0 else self.prev0, env._steps, env._episode, self.actions[t], epsilon,
                          probs=None if self.probs is None else self.probs[t], stream=stream, env_id_base=env._desc.env_id_base,
                          prev_done=self.done[t - 1] if (t > 0 and self.auto_reset) else None)
            env._desc.flags = self._base_flags()
            b = self._bufs(t)
            _lib.check(self._lib.cm3_checkers_step(ctypes.byref(env._desc), ctypes.byref(b), stream))
        env._desc.flags = 0

    def _enqueue_policy_rollout(self, actor, epsilon, stream, prev0_next=None):
        """The same T ticks as _enqueue_actor_rollout in ONE launch (csrc/policy_checkers.hip), slot 0 and the env's
        current-observation buffers (what _load_slot0 / _store_back copy for the other modes) included."""
        env = self.env
        env._desc.flags = self._base_flags()
        if self._traj_cache is None or self._traj_cache[0] != env._cur:
            self._traj_cache = (env._cur, self._traj(), env._bufs(env._cur))
        _, traj, final = self._traj_cache
        actor.enqueue_rollout(env._desc, traj, env.E, env.obst_stride, self.T, epsilon, prev0=self.prev0, probs=self.probs,
                              stream=stream, final_obs=final, prev0_next=prev0_next)
        env._desc.flags = 0

    def _load_slot0(self):
        """slot 0 <- the env's current observation (after a reset, or where the previous collect() stopped)."""
        env = self.env
        s = env._slots[env._cur]
        pairs = [(self._grid_raw[0], s["grid_raw"]), (self._obst_raw[0], s["obs_self_t_raw"]), (self.vec[0], s["vec"]),
                 (self.obs_others[0], s["obs_others"]), (self.obs_self_v[0], s["obs_self_v"])]
        if self.goal_slots is not None:
            pairs.append((self.goal_slots[0], env._goals))
        _copy_pairs(pairs, env._stream())

    def _store_back(self):
        """the env's current-observation buffers <- slot T, so that get_obs() / CheckersActor.act(env) / the next
        collect() see the state the rollout left (the compact live state is advanced in place by the launches)."""
        env, T = self.env, self.T
        s = env._slots[env._cur]
        _copy_pairs([(s["grid_raw"], self._grid_raw[T]), (s["obs_self_t_raw"], self._obst_raw[T]), (s["vec"], self.vec[T]),
                     (s["obs_others"], self.obs_others[T]), (s["obs_self_v"], self.obs_self_v[T]),
                     (s["actions"], self.actions[T - 1])], env._stream())

    def collect(self, goals=None, policy=None, epsilon=0.0, reset=None):
        """goals: one-hot [N,2] / [E,N,2] (train_onpolicy.py:287-293); needed whenever the env is reset.
        policy None = uniform random actions drawn in-kernel; a cm3_amd.actor.CheckersActor = the on-device policy (the whole
        rollout in one launch where that kernel applies, else actor and step launches alternating inside one hipGraph: policy_mode);
        else policy(actions_prev, obs_others, obs_self_t, obs_self_v, goals) -> [E,N] on the host.  reset: None -> always for an
        episode-synchronous env, only the first time for a continuous (auto-reset) one."""
        env = self.env
        one_launch = (policy is not None and hasattr(policy, "enqueue_rollout") and self.policy_mode == "auto"
                      and policy.fused_rollout_ok(env))
        if reset is None:
            reset = (not self.auto_reset) or (not self._started)
        # actions_prev of tick 0 (self.prev0; train_onpolicy.py:295,345): zeros after a reset, else the previous rollout's last actions
        # (zeros where its last tick ended an episode).  Three buffers: one that holds zeros and is never written, two that take
        # turns -- the one-launch kernel writes the NEXT rollout's prev0 into the spare one, so nothing is launched for it here.
        pz, pa, pb = self._prev_bufs
        if reset:
            if goals is None:
                raise Cm3Error("collect() resets the env here and needs goals")
            env.reset(goals, quiet=True)
            if one_launch:
                self.prev0 = pz
            else:
                self.prev0 = pa
                self.prev0.zero_()
        elif self._started:
            if self._rolled is not None and one_launch:
                self.prev0 = self._rolled
            else:
                # (the launch-per-tick modes read prev0 from a CAPTURED graph: always the same buffer, pa)
                if self._rolled is not None:
                    nxt = self._rolled
                else:
                    keep = (self.done[self.T - 1] == 0).unsqueeze(1)
                    nxt = torch.where(keep, self.actions[self.T - 1], torch.zeros_like(self.actions[0]))
                self.prev0 = pa
                if nxt is not pa:
                    pa.copy_(nxt)
        self._rolled = None
        self._started = True
        self._goals_onehot = None             # (lazy: see goals_onehot)
        stream = env._stream()
        if one_launch:
            # cm3_policy_rollout_checkers writes slot 0 from the live state and hands the state it leaves to the env's own
            # current-observation buffers: no copies around the launch
            spare = pa if self.prev0 is not pa else pb
            self._enqueue_policy_rollout(policy, epsilon, stream, spare)
            self._rolled = spare
            return self
        self._load_slot0()
        if policy is None:
            # random-action branch (train_onpolicy.py:305-307): cm3_checkers_rollout -- one fused launch, or T step
            # launches bound to their trajectory slots and replayed as one hipGraph
            if self.fused:
                self._enqueue_random(stream, True)
            elif self.use_graph:
                if self._graph is None:
                    self._graph = _lib.capture_graph(env.device, lambda st: self._enqueue_random(st, False))
                _lib.check(self._lib.cm3_graph_launch(self._graph, stream))
            else:
                self._enqueue_random(stream, False)
        elif hasattr(policy, "enqueue") and hasattr(policy, "act"):            # on-device actor, a launch pair per tick
            if self.use_graph:
                cache = self._actor_graph
                cache.launch(self._lib, policy, epsilon,
                             lambda st: self._enqueue_actor_rollout(policy, cache.eps, st), stream)
            else:
                self._enqueue_actor_rollout(policy, epsilon, stream)
        else:
            for t in range(self.T):                                              # host policy: one call per tick
                a = policy(self.actions_prev_at(t), self.obs_others[t], self.obs_self_t[t], self.obs_self_v[t],
                           self.goals_at(t))
                self.actions[t].copy_(torch.as_tensor(a, device=env.device).reshape(env.E, env.n))
                env._desc.flags = self._base_flags()
                b = self._bufs(t)
                _lib.check(self._lib.cm3_checkers_step(ctypes.byref(env._desc), ctypes.byref(b), stream))
            env._desc.flags = 0
        self._store_back()
        return self

    @property
    def goals_onehot(self):
        """[E,N,2] one-hot goals of the rollout's episodes when no goal slots are recorded (episode-synchronous envs: the goals the
        env was reset with; they cannot change inside a collect()).  Built on first use: a collect() does not pay for it."""
        if self._goals_onehot is None:
            self._goals_onehot = self.env.goals.clone()
        return self._goals_onehot

    def actions_prev_at(self, t):
        """int32 [E,N]: the actions_prev fed at tick t (train_onpolicy.py:295,345)."""
        if t == 0:
            return self.prev0
        if not self.auto_reset:
            return self.actions[t - 1]
        return torch.where((self.done[t - 1] == 0).unsqueeze(1), self.actions[t - 1], torch.zeros_like(self.actions[0]))

    def goals_at(self, t):
        """one-hot [E,N,2] goals in effect at tick t."""
        if self.goal_slots is None:
            return self.goals_onehot
        return torch.nn.functional.one_hot(self.goal_slots[t].long(), 2)

    def close(self):
        if self._graph is not None:
            torch.cuda.synchronize(self.env.device)
            self._lib.cm3_graph_destroy(self._graph)
            self._graph = None
        self._actor_graph.destroy(self._lib)

    @property
    def valid(self):
        if self.auto_reset:
            return torch.ones(self.T, self.env.E, dtype=torch.bool, device=self.env.device)
        return _valid_from_done(self.done)

    def valid_indices(self):
        idx = self.valid.nonzero(as_tuple=False)
        return idx[:, 0], idx[:, 1]

    ORDER = ("grid", "vec", "obs_others", "obs_self_t", "obs_self_v", "actions_prev", "actions", "reward",
             "local_rewards", "next_grid", "next_vec", "next_obs_others", "next_obs_self_t", "next_obs_self_v",
             "done", "goals")

    def _next(self, name, tt, ee):
        """next_<name> of transitions (tt, ee): slot t+1, or the captured terminal observation where the env restarted
        in the same launch (continuous mode)."""
        nxt = getattr(self, name)[tt + 1, ee]
        if self.auto_reset:
            term = getattr(self, "term_" + name)[tt, ee]
            d = self.done[tt, ee].bool().view(-1, *([1] * (nxt.dim() - 1)))
            nxt = torch.where(d, term, nxt)
        return nxt

    def as_reference_batch(self, tt=None, ee=None, numpy=True):
        """16 columns equal to np.stack(batch[:, k]) of alg_credit_checkers.process_batch
        (alg_credit_checkers.py:427-444); integer-valued columns are cast to the reference's float64."""
        if tt is None:
            tt, ee = self.valid_indices()
        dev = self.env.device
        tt = torch.as_tensor(tt, device=dev, dtype=torch.long)
        ee = torch.as_tensor(ee, device=dev, dtype=torch.long)
        f = lambda x: x.to(torch.float64)  # noqa: E731
        prev = self.actions[(tt - 1).clamp(min=0), ee]
        if self.auto_reset:                                            # a fresh episode starts from zeros (:295)
            fresh = self.done[(tt - 1).clamp(min=0), ee].bool().view(-1, 1)
            prev = torch.where(fresh, torch.zeros_like(prev), prev)
        prev = torch.where((tt > 0).view(-1, 1), prev, self.prev0[ee])
        goals = (self.goals_onehot[ee] if self.goal_slots is None
                 else torch.nn.functional.one_hot(self.goal_slots[tt, ee].long(), 2))
        cols = dict(
            grid=f(self.grid[tt, ee]), vec=f(self.vec[tt, ee]), obs_others=self.obs_others[tt, ee],
            obs_self_t=f(self.obs_self_t[tt, ee]), obs_self_v=self.obs_self_v[tt, ee],
            actions_prev=prev, actions=self.actions[tt, ee], reward=self.reward[tt, ee],
            local_rewards=self.local_rewards[tt, ee],
            next_grid=f(self._next("grid", tt, ee)), next_vec=f(self._next("vec", tt, ee)),
            next_obs_others=self._next("obs_others", tt, ee), next_obs_self_t=f(self._next("obs_self_t", tt, ee)),
            next_obs_self_v=self._next("obs_self_v", tt, ee), done=self.done[tt, ee].bool(),
            goals=goals)
        if numpy:
            cols = {k: v.detach().cpu().numpy() for k, v in cols.items()}
        return cols

    def as_reference_rows(self, tt=None, ee=None):
        return rows_from_columns(self.as_reference_batch(tt, ee), self.ORDER)

    def episode_returns(self):
        v = self.valid.to(torch.float64)
        return (self.reward * v).sum(0), (self.local_rewards * v.unsqueeze(2)).sum(0)
