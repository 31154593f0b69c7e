"""VecCheckersEnv -- E copies of the reference's Checkers grid world, one HIP launch per tick.

    reference (per env, env/checkers.py)                          here (E envs)
    ---------------------------------------------------------     --------------------------------------
    Checkers(n_rows, n_columns, n_obs, agents_r, agents_c,        VecCheckersEnv(init, n_agents, max_steps,
             n_agents, max_steps)                 (:5-35)            n_envs, ...)   init = config "init" block
    reset(goals) -> ((grid, vec), obs_others, obs_self_t,         same tuple, leading E dim
             obs_self_v, False)                   (:265-291)
    step(actions) -> ((grid, vec), obs_others, obs_self_t,        same tuple, batched
             obs_self_v, total, local_rewards, done) (:228-262)

Value parity is exact.  Integer-valued arrays come back as integer tensors (grid / obs_self_t int8,
vec int32); ``.double()`` gives the reference's float64 arrays bit for bit.  The live state on the
device is compact (64-bit collected mask + one packed word per agent), see csrc/checkers.hip.
"""
import ctypes

import torch

from . import _lib
from ._lib import FLAG_AUTO_RESET, FLAG_GEN_ACTIONS, Cm3Error


class VecCheckersEnv(object):
    def __init__(self, init, n_agents, max_steps, n_envs, device="cuda:0", seed=12341, auto_reset=False,
                 env_id_base=0, padded_records=None):
        self.device = _lib.require_gpu(device)
        self.n = self.n_agents = int(n_agents)
        self.E = self.n_envs = int(n_envs)
        self.max_steps = int(max_steps)
        self.R, self.C, self.O = int(init["n_rows"]), int(init["n_columns"]), int(init["n_obs"])
        if self.R % 2 != 1 or self.C % 2 != 0:          # checkers.py:16-17
            raise Cm3Error("n_rows must be odd and n_columns even")
        if len(init["agents_r"]) < self.n or len(init["agents_c"]) < self.n:
            raise Cm3Error("init lists shorter than n_agents")
        self.K = 2 * self.O + 1
        self.Lo = 2 * max(self.n - 1, 1)
        self.auto_reset = bool(auto_reset)
        E, N, dev = self.E, self.n, self.device
        z = lambda *shape, dt: torch.zeros(*shape, dtype=dt, device=dev)  # noqa: E731
        self._mask = z(E, dt=torch.int64)             # uint64 bit pattern
        self._agents = z(N, E, dt=torch.int32)        # packed r | c<<8 | green<<16 | orange<<24
        self._steps = z(E, dt=torch.int32)
        self._episode = z(E, dt=torch.int32)
        self._goals = z(E, N, dt=torch.uint8)
        self._goals_arg = None      # the [N,2] one-hot array the goal bytes were last set from (reset(): skip re-deriving them)
        # env records of the two byte grids; the reference geometry gets 4-byte padded records (54 -> 56,
        # 75 N -> multiple of 4), which selects the multi-lane fast kernel.  Views hide the padding.
        self.grid_rec = self.R * (self.C + 1) * 2
        self.obst_rec = N * self.K * self.K * 3
        fast = (self.R, self.C, self.O) == (3, 8, 2) if padded_records is None else bool(padded_records)
        pad4 = lambda n: (n + 3) // 4 * 4 if fast else n  # noqa: E731
        self.grid_stride, self.obst_stride = pad4(self.grid_rec), pad4(self.obst_rec)
        self._slots = []
        for _ in range(2):                             # double-buffered outputs
            grid_raw = z(E, self.grid_stride, dt=torch.int8)
            obst_raw = z(E, self.obst_stride, dt=torch.int8)
            self._slots.append(dict(
                actions=z(E, N, dt=torch.int32),
                grid_raw=grid_raw, obs_self_t_raw=obst_raw,
                grid=self.grid_view(grid_raw),
                vec=z(E, N, 4, dt=torch.int32),
                obs_others=z(E, N, self.Lo, dt=torch.float64),
                obs_self_t=self.obst_view(obst_raw),
                obs_self_v=z(E, N, 4, dt=torch.float64),
                local_rewards=z(E, N, dt=torch.float64),
                reward=z(E, dt=torch.float64),
                done=z(E, dt=torch.uint8)))
        self._cur = 0
        self._term = None
        d = self._desc = _lib.CheckersDesc()
        d.n_envs, d.n_agents, d.n_rows, d.n_columns, d.n_obs = E, N, self.R, self.C, self.O
        d.max_steps = self.max_steps
        d.grid_stride, d.obs_self_t_stride = self.grid_stride, self.obst_stride
        d.flags = 0
        d.env_id_base = int(env_id_base)
        d.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        for i in range(_lib.MAX_AGENTS):
            d.agents_r[i] = int(init["agents_r"][i]) if i < N else 0
            d.agents_c[i] = int(init["agents_c"][i]) if i < N else 0
        self._lib = _lib.lib()
        # stage 1 of the in-kernel action stream: a Philox block per env, a constant of (seed, global env id) -- computed once here,
        # loaded by every step launch with its state (cm3_checkers_bufs.action_block)
        import ctypes
        self._action_block = torch.zeros(E, 4, dtype=torch.int32, device=dev)
        _lib.check(self._lib.cm3_checkers_action_blocks(ctypes.byref(d), self._action_block.data_ptr(),
                                                        torch.cuda.current_stream(dev).cuda_stream))

    def grid_view(self, raw):
        """[..., E, grid_stride] int8 storage -> [..., E, R, C+1, 2] view without the padding bytes."""
        return raw[..., :self.grid_rec].unflatten(-1, (self.R, self.C + 1, 2))

    def obst_view(self, raw):
        return raw[..., :self.obst_rec].unflatten(-1, (self.n, self.K, self.K, 3))

    def _bufs(self, slot):
        s = self._slots[slot]
        b = _lib.CheckersBufs()
        b.mask = self._mask.data_ptr()
        b.agents = self._agents.data_ptr()
        b.steps = self._steps.data_ptr()
        b.episode = self._episode.data_ptr()
        b.goals = self._goals.data_ptr()
        b.action_block = self._action_block.data_ptr()
        for k in ("actions", "vec", "obs_others", "obs_self_v", "local_rewards", "reward", "done"):
            setattr(b, k, s[k].data_ptr())
        b.grid = s["grid_raw"].data_ptr()
        b.obs_self_t = s["obs_self_t_raw"].data_ptr()
        if self._term is not None:
            t = self._term
            b.term_grid, b.term_obs_self_t = t["grid_raw"].data_ptr(), t["obs_self_t_raw"].data_ptr()
            b.term_vec, b.term_obs_others = t["vec"].data_ptr(), t["obs_others"].data_ptr()
            b.term_obs_self_v = t["obs_self_v"].data_ptr()
        return b

    def enable_terminal_capture(self):
        """Allocate the term_* arrays so that auto_reset keeps the TRUE post-step observation of an env whose episode
        ended (the next_* columns of the terminal transition, train_onpolicy.py:336-347); read them with
        ``terminal_obs()`` where ``done`` is set."""
        if self._term is None:
            E, N = self.E, self.n
            z = lambda *shape, dt: torch.zeros(*shape, dtype=dt, device=self.device)  # noqa: E731
            grid_raw, obst_raw = z(E, self.grid_stride, dt=torch.int8), z(E, self.obst_stride, dt=torch.int8)
            self._term = dict(grid_raw=grid_raw, obs_self_t_raw=obst_raw, grid=self.grid_view(grid_raw),
                              obs_self_t=self.obst_view(obst_raw), vec=z(E, N, 4, dt=torch.int32),
                              obs_others=z(E, N, self.Lo, dt=torch.float64), obs_self_v=z(E, N, 4, dt=torch.float64))

    def terminal_obs(self):
        """((grid, vec), obs_others, obs_self_t, obs_self_v) captured by the most recent step() for the envs it
        re-initialised (rows of other envs are stale)."""
        t = self._term
        return None if t is None else ((t["grid"], t["vec"]), t["obs_others"], t["obs_self_t"], t["obs_self_v"])

    def _stream(self):
        return _lib.current_stream_handle(self.device)

    def _obs_tuple(self, slot):
        s = self._slots[slot]
        return (s["grid"], s["vec"]), s["obs_others"], s["obs_self_t"], s["obs_self_v"]

    # ---- reference surface --------------------------------------------------------------------------
    def reset(self, goals=None, mask=None, goal_index=None, quiet=False):
        """checkers.py:265-291.  ``goals``: one-hot [N,2] (np.eye(n_agents), train_onpolicy.py:293) or
        [E,N,2]; alternatively ``goal_index`` int [E,N] (0 green, 1 orange).  quiet: return nothing (collectors)."""
        import numpy as np
        # the per-episode call of the training loop passes the SAME small [N,2] array every time (np.eye(n), train_onpolicy.py:293):
        # five tiny launches to turn it into the goal bytes the env already holds (42 us per reset, mostly those)
        same = (goal_index is None and mask is None and self.n > 1 and isinstance(goals, np.ndarray) and goals.shape == (self.n, 2)
                and self._goals_arg is not None and np.array_equal(goals, self._goals_arg))
        if same:
            pass
        elif goal_index is not None:
            g = torch.as_tensor(goal_index, device=self.device)
        else:
            if goals is None:
                raise Cm3Error("reset needs goals (one-hot) or goal_index")
            g = torch.as_tensor(goals, device=self.device)
            if g.dim() == 2:
                if tuple(g.shape) != (self.n, 2):
                    raise Cm3Error("one-hot goals must be [N,2] or [E,N,2]")
                g = g.unsqueeze(0).expand(self.E, self.n, 2)
            g = g.argmax(dim=2)
        m = None
        if not same:
            if tuple(g.shape) != (self.E, self.n):
                raise Cm3Error("goals must be [N,2] / [E,N,2] one-hot, or goal_index [E,N]")
            if mask is None:
                self._goals.copy_(g.to(torch.uint8))
            else:
                m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
                self._goals.copy_(torch.where(m.bool().unsqueeze(1), g.to(torch.uint8), self._goals))
            whole = goal_index is None and mask is None and self.n > 1 and isinstance(goals, np.ndarray) and goals.shape == (self.n, 2)
            self._goals_arg = goals.copy() if whole else None
        self._desc.flags = 0
        b = self._bufs(self._cur)
        _lib.check(self._lib.cm3_checkers_reset(ctypes.byref(self._desc), ctypes.byref(b), _lib.ptr(m),
                                                self._stream()))
        if quiet:
            return None
        gs, oo, ot, ov = self._obs_tuple(self._cur)
        return gs, oo, ot, ov, torch.zeros(self.E, dtype=torch.bool, device=self.device)

    def step(self, actions=None):
        """checkers.py:228-262.  ``actions`` int [E,N]; None draws uniform actions in-kernel."""
        dst = self._cur ^ 1
        flags = FLAG_AUTO_RESET if self.auto_reset else 0
        if actions is None:
            flags |= FLAG_GEN_ACTIONS
        else:
            a = torch.as_tensor(actions, device=self.device)
            if tuple(a.shape) != (self.E, self.n):
                raise Cm3Error("actions must have shape [n_envs, n_agents]")
            self._slots[dst]["actions"].copy_(a)
        self._desc.flags = flags
        b = self._bufs(dst)
        _lib.check(self._lib.cm3_checkers_step(ctypes.byref(self._desc), ctypes.byref(b), self._stream()))
        self._cur = dst
        s = self._slots[dst]
        gs, oo, ot, ov = self._obs_tuple(dst)
        return gs, oo, ot, ov, s["reward"], s["local_rewards"], s["done"].view(torch.bool)

    def get_obs(self):
        return self._obs_tuple(self._cur)

    # ---- state access ---------------------------------------------------------------------------------
    @property
    def goals(self):
        """[E,N,2] one-hot goals (train_onpolicy.py:293)."""
        return torch.nn.functional.one_hot(self._goals.long(), 2)

    @property
    def steps(self):
        return self._steps

    @property
    def last_actions(self):
        return self._slots[self._cur]["actions"]

    def get_state(self):
        a = self._agents
        return dict(mask=self._mask.clone(), r=(a & 0xff).t().clone(), c=((a >> 8) & 0xff).t().clone(),
                    n_green=((a >> 16) & 0xff).t().clone(), n_orange=((a >> 24) & 0xff).t().clone(),
                    steps=self._steps.clone(), goals=self._goals.clone())

    def set_state(self, mask, r, c, n_green, n_orange, steps, goals=None):
        """Inject a compact state ([E] mask as int64 bit pattern; [E,N] integer arrays) and refresh obs."""
        dev = self.device
        self._mask.copy_(torch.as_tensor(mask, device=dev).to(torch.int64))
        tt = lambda x: torch.as_tensor(x, device=dev).to(torch.int32).t()  # noqa: E731
        self._agents.copy_(tt(r) | (tt(c) << 8) | (tt(n_green) << 16) | (tt(n_orange) << 24))
        self._steps.copy_(torch.as_tensor(steps, device=dev).to(torch.int32))
        if goals is not None:
            self._goals.copy_(torch.as_tensor(goals, device=dev).to(torch.uint8))
            self._goals_arg = None
        none = torch.zeros(self.E, dtype=torch.uint8, device=dev)
        self._desc.flags = 0
        b = self._bufs(self._cur)
        _lib.check(self._lib.cm3_checkers_reset(ctypes.byref(self._desc), ctypes.byref(b), none.data_ptr(),
                                                self._stream()))
        return self._obs_tuple(self._cur)
