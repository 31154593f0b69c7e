"""VecParticleEnv -- E copies of the reference's particle env stepped by ONE HIP launch per tick.

Host-side mirror of the reference interface for this path:

    reference (per env)                                           here (E envs)
    --------------------------------------------------------      ---------------------------------
    scenario.make_world(n_agents, config, prob_random)            VecParticleEnv(config, n_agents,
      (multi-goal_spread.py:19-63)                                   prob_random, max_steps, n_envs, ...)
    MultiAgentEnv(world, ..., max_steps) (environment.py:14-16)
    env.reset() -> (global_state, obs_others_n, obs_n, done)      same tuple; leading E dim, agent lists
      (environment.py:125-149)                                     become an N dim
    env.step(action_n) -> (global_state, obs_others_n, obs_n,     same tuple, batched
      reward, reward_n, done)  (environment.py:81-123)
    env.world.landmarks[i].state.p_pos (train_onpolicy.py:285)    env.goals        [E, N, 2]
    scenario.collisions (train_onpolicy.py:356)                   env.collisions   [E]

All arithmetic happens in libcm3_hip.so (cm3_amd/csrc/particle.hip) on tensors owned by PyTorch;
this module only allocates, binds pointers and shapes views.  No CPU fallback exists.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import FLAG_AUTO_RESET, FLAG_GEN_ACTIONS, Cm3Error


def _fill_desc(desc, config, n_agents, prob_random, max_steps, n_envs, seed, env_id_base, flags):
    desc.n_envs = int(n_envs)
    desc.n_agents = int(n_agents)
    desc.max_steps = int(max_steps)
    desc.flags = int(flags)
    desc.env_id_base = int(env_id_base)
    desc.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    desc.prob_random = float(prob_random)
    desc.initial_std = float(config["initial_std"])
    for key in ("agents_x", "agents_y", "landmarks_x", "landmarks_y"):
        vals = config[key]
        if len(vals) < n_agents:
            raise Cm3Error("config[%r] has %d entries, need n_agents=%d" % (key, len(vals), n_agents))
        arr = getattr(desc, key)
        for i in range(_lib.MAX_AGENTS):
            arr[i] = float(vals[i]) if i < n_agents else 0.0


class VecParticleEnv(object):
    """E independent cooperative-navigation envs (multi-goal_spread) on one GPU.

    dtype       torch.float32 (production) or torch.float64 (parity mode; same kernel template)
    auto_reset  finished episodes are re-initialised inside the step launch (CM3_FLAG_AUTO_RESET)
    env_id_base global id of local env 0 -- the RNG is keyed by global env id, so a sharded run
                reproduces the single-GPU run
    Tensors returned by reset()/step() are views of double-buffered device storage: they stay valid
    until the step AFTER the next one (enough to assemble (state, next_state) transitions); clone to
    keep them longer.
    """

    def __init__(self, config_particle, n_agents, prob_random, max_steps, n_envs, device="cuda:0",
                 seed=12341, dtype=torch.float32, auto_reset=False, env_id_base=0, kernel="auto"):
        self.device = _lib.require_gpu(device)
        if dtype not in (torch.float32, torch.float64):
            raise Cm3Error("dtype must be float32 or float64")
        if not (1 <= n_agents <= _lib.MAX_AGENTS):
            raise Cm3Error("n_agents must be in 1..%d" % _lib.MAX_AGENTS)
        self.config = config_particle
        self.n = self.n_agents = int(n_agents)
        self.E = self.n_envs = int(n_envs)
        self.max_steps = int(max_steps)
        self.prob_random = float(prob_random)
        self.dtype = dtype
        self.seed = int(seed)
        self.auto_reset = bool(auto_reset)
        self.env_id_base = int(env_id_base)
        if kernel not in _lib.KERNEL_FLAGS:
            raise Cm3Error("kernel must be one of %s" % sorted(_lib.KERNEL_FLAGS))
        # step-kernel mapping: "env" = one lane per env, "pair" = one lane per ordered agent pair,
        # "auto" = chosen from n_envs by the library.  Results are identical.
        self.kernel_flags = _lib.KERNEL_FLAGS[kernel] if self.n >= 2 else 0
        self.L = 4 * max(self.n - 1, 1)
        self._suffix = "f32" if dtype == torch.float32 else "f64"
        E, N, L, dev = self.E, self.n, self.L, self.device
        # Separate allocations on purpose: carving all streams out of one arena at 256-byte offsets cost 5-15 %
        # of the HBM bandwidth at >= 1M envs (5.48 vs 5.80 TB/s at 2^20, 4.84 vs 5.66 TB/s at 2^22 -- channel
        # aliasing between the nine concurrent streams) and bought nothing at small batches.
        z = lambda *shape, dt=dtype: torch.zeros(*shape, dtype=dt, device=dev)  # noqa: E731
        # live, in place
        self._goals = z(N, E, 2)
        self._meta = z(E, 2, dt=torch.int32)
        self._episode = z(E, dt=torch.int32)
        # double-buffered outputs (slot = tick parity)
        self._state = [z(N, E, 4), z(N, E, 4)]
        self._actions = [z(E, N, dt=torch.int32), z(E, N, dt=torch.int32)]
        self._reward_n = [z(E, N), z(E, N)]
        self._reward = [z(E), z(E)]
        self._done = [z(E, dt=torch.uint8), z(E, dt=torch.uint8)]
        self._obs_others = [z(E, N, L), z(E, N, L)]
        self._term_state = None
        self._term_obs_others = None
        self._collisions_tick = None
        self._cur = 0
        self._rollouts = weakref.WeakSet()       # collectors to tell about resets / state injections (_notify_reset)
        self._desc = _lib.ParticleDesc()
        _fill_desc(self._desc, config_particle, N, prob_random, max_steps, E, seed, env_id_base, 0)
        self._lib = _lib.lib()
        # hot-path caches for step(): bound C functions, argument structs and result tuples per buffer parity
        self._step_fn = self._fn("step")
        self._desc_ref = ctypes.byref(self._desc)
        self._step_bufs = {}
        self._step_out = {}
        self._pin, self._pin_np, self._pin_k = None, None, 0      # pinned staging of host actions (step())

    # ---- plumbing --------------------------------------------------------------------------------
    def _fn(self, op):
        return getattr(self._lib, "cm3_particle_%s_%s" % (op, self._suffix))

    def _stream(self):
        return _lib.current_stream_handle(self.device)

    def _bufs(self, src, dst):
        b = _lib.ParticleBufs()
        b.state_in = self._state[src].data_ptr()
        b.state_out = self._state[dst].data_ptr()
        b.goals_in = b.goals_out = self._goals.data_ptr()
        b.meta_in = b.meta_out = self._meta.data_ptr()
        b.episode = self._episode.data_ptr()
        b.actions = self._actions[dst].data_ptr()
        b.obs_others = self._obs_others[dst].data_ptr()
        b.reward_n = self._reward_n[dst].data_ptr()
        b.reward = self._reward[dst].data_ptr()
        b.done = self._done[dst].data_ptr()
        b.term_state = _lib.ptr(self._term_state)
        b.term_obs_others = _lib.ptr(self._term_obs_others)
        b.collisions_tick = _lib.ptr(self._collisions_tick)
        return b

    def enable_terminal_capture(self):
        """Allocate term_state / term_obs_others / the per-tick collision count so AUTO_RESET keeps the true terminal
        next-state and the finished episode's scenario.collisions (both are overwritten by the re-initialisation)."""
        if self._term_state is None:
            self._term_state = torch.zeros_like(self._state[0])
            self._term_obs_others = torch.zeros_like(self._obs_others[0])
            self._collisions_tick = torch.zeros(self.E, dtype=torch.int32, device=self.device)
            self._step_bufs.clear()

    # ---- reference surface -----------------------------------------------------------------------
    def reset(self, mask=None):
        """environment.py:125-149 for every env (or those selected by the uint8/bool ``mask`` [E])."""
        cur = self._cur
        b = self._bufs(cur, cur)
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if m.numel() != self.E:
                raise Cm3Error("mask must have n_envs elements")
        self._desc.flags = 0
        _lib.check(self._fn("reset")(ctypes.byref(self._desc), ctypes.byref(b), _lib.ptr(m), self._stream()))
        done = torch.zeros(self.E, dtype=torch.bool, device=self.device)   # np.any(done_n) is False after reset
        gs = self.global_state
        self._notify_reset(m)
        return gs, self._obs_others[cur], gs, done

    def _notify_reset(self, mask=None):
        """A ParticleRollout carries which envs' episodes have already ended across collect(reset=False) calls; an env that
        is re-seeded OUTSIDE the collector (reset(), set_state()) starts a fresh episode and must not stay flagged."""
        for ro in list(self._rollouts):
            ro.mark_reset(mask)

    def step(self, actions=None):
        """environment.py:81-123.  ``actions`` int [E, N]; None draws uniform actions in-kernel
        (the reference's random-action branch, train_onpolicy.py:305-307)."""
        src, dst = self._cur, self._cur ^ 1
        flags = (FLAG_AUTO_RESET if self.auto_reset else 0) | self.kernel_flags
        direct = None
        self._last_actions = self._actions[dst]
        if actions is None:
            flags |= FLAG_GEN_ACTIONS
        elif isinstance(actions, torch.Tensor) and actions.dtype == torch.int32 and actions.is_cuda and actions.is_contiguous():
            # device int32 actions (e.g. from ParticleActor.act): the launch reads them where they are
            if actions.shape != (self.E, self.n):
                raise Cm3Error("actions must have shape [n_envs, n_agents] = [%d, %d], got %s" % (self.E, self.n, tuple(actions.shape)))
            direct = actions
            self._last_actions = actions
        elif isinstance(actions, np.ndarray) and actions.shape == (self.E, self.n) and actions.dtype.kind in "iu":
            # host integer actions (np.random.randint, a NumPy policy: train_onpolicy.py:307): converted to int32 into a PINNED
            # staging buffer and copied asynchronously -- torch.as_tensor(ndarray, device=...) is a pageable, synchronous copy
            # (33 us per tick at C2, round 6).  A ring of four buffers, each with the event of its last copy.
            if self._pin is None:
                self._pin = [(torch.empty(self.E, self.n, dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(4)]
                self._pin_np = [b.numpy() for b, _ in self._pin]
            k = self._pin_k = (self._pin_k + 1) & 3
            buf, ev = self._pin[k]
            ev.synchronize()
            np.copyto(self._pin_np[k], actions, casting="unsafe")
            self._actions[dst].copy_(buf, non_blocking=True)
            ev.record()
        else:
            a = torch.as_tensor(actions, device=self.device)
            if a.shape != (self.E, self.n):
                raise Cm3Error("actions must have shape [n_envs, n_agents] = [%d, %d], got %s"
                               % (self.E, self.n, tuple(a.shape)))
            self._actions[dst].copy_(a)
        self._desc.flags = flags
        b = self._step_bufs.get(dst)
        if b is None:
            b = self._step_bufs[dst] = self._bufs(src, dst)
            gs = self._state[dst].permute(1, 0, 2)
            self._step_out[dst] = (gs, self._obs_others[dst], gs, self._reward[dst], self._reward_n[dst],
                                   self._done[dst].view(torch.bool))
        b.actions = direct.data_ptr() if direct is not None else self._actions[dst].data_ptr()
        rc = self._step_fn(self._desc_ref, ctypes.byref(b), torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _lib.check(rc)
        self._cur = dst
        return self._step_out[dst]

    def get_obs(self):
        """(obs_self [E,N,4], obs_others [E,N,L]) of the current state (multi-goal_spread.py:145-154)."""
        return self.global_state, self._obs_others[self._cur]

    # ---- extra surface read by the reference's callers ------------------------------------------------
    @property
    def global_state(self):
        """[E, N, 4] rows (vx, vy, px, py) (environment.py:113-116): a permuted view of the SoA state."""
        return self._state[self._cur].permute(1, 0, 2)

    @property
    def goals(self):
        """[E, N, 2] landmark positions (train_onpolicy.py:283-285)."""
        return self._goals.permute(1, 0, 2)

    @property
    def collisions(self):
        """[E] scenario.collisions (multi-goal_spread.py:93,137; read at train_onpolicy.py:356).  Exactly the reference's
        counter: it keeps counting if an env is stepped after its episode ended (MultiAgentEnv.step does not look at a
        previous `done`); the collectors read the per-episode value from the trajectory (ParticleRollout.episode_is_bad)."""
        return self._meta[:, 1]

    @property
    def steps(self):
        """[E] MultiAgentEnv.steps (environment.py:93)."""
        return self._meta[:, 0]

    @property
    def collisions_after_last_step(self):
        """int32 [E]: scenario.collisions right after the most recent step(), BEFORE auto_reset zeroed it for the envs
        whose episode ended there (needs enable_terminal_capture())."""
        return self._collisions_tick

    @property
    def episode(self):
        return self._episode

    @property
    def last_actions(self):
        """int32 [E, N] actions consumed (or drawn in-kernel) by the most recent step()."""
        return getattr(self, "_last_actions", self._actions[self._cur])

    @property
    def terminal_state(self):
        return None if self._term_state is None else self._term_state.permute(1, 0, 2)

    @property
    def terminal_obs_others(self):
        return self._term_obs_others

    # ---- state injection / checkpoint ---------------------------------------------------------------------
    def get_state(self):
        gs = self.global_state
        return dict(pos=gs[..., 2:4].clone(), vel=gs[..., 0:2].clone(), landmarks=self.goals.clone(),
                    steps=self.steps.clone(), collisions=self.collisions.clone(), episode=self._episode.clone())

    def set_state(self, pos, vel, landmarks, steps=None, collisions=None, episode=None):
        """Inject [E,N,2] positions / velocities / landmarks (parity tests, resume) and refresh obs."""
        cur = self._cur
        st = self._state[cur]
        t = lambda x: torch.as_tensor(x, device=self.device).to(self.dtype)  # noqa: E731
        st[:, :, 0:2] = t(vel).permute(1, 0, 2)
        st[:, :, 2:4] = t(pos).permute(1, 0, 2)
        self._goals.copy_(t(landmarks).permute(1, 0, 2))
        self._meta[:, 0] = 0 if steps is None else torch.as_tensor(steps, device=self.device).to(torch.int32)
        self._meta[:, 1] = 0 if collisions is None else torch.as_tensor(collisions, device=self.device).to(torch.int32)
        if episode is not None:
            self._episode.copy_(torch.as_tensor(episode, device=self.device).to(torch.int32))
        self._notify_reset(None)
        b = self._bufs(cur, cur)
        _lib.check(self._fn("observe")(ctypes.byref(self._desc), ctypes.byref(b), self._stream()))
        return self.global_state, self._obs_others[cur]
