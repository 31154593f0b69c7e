"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's Checkers env.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this; the product (``cm3_amd``) never does.

Parity status: PINNED by ``tests/golden/checkers_*.npz`` (recorded from the reference's own
``env/checkers.py`` by ``oracle/gen_golden.py``; the reference has no tests of its own).

* ``CheckersEnvOracle``  one environment on the reference's dense ``[rows, cols, 3]`` world
                         (float64 holding {-1,0,1}), scalar call structure like the reference.
* ``VecCheckersOracle``  E environments stepped with the compact state the HIP kernel uses
                         (collected bit-mask + agent cells), outputs rebuilt with NumPy; it is
                         checked against the dense one so the compact encoding itself is pinned.

Reference: /root/reference/env/checkers.py -- __init__ :5-35, populate_world :38-63,
get_valid_grid :66-76, get_global_state :79-94, get_obs :97-109, normalize :112-125,
get_local_observation :128-154, agent_act :157-187, get_reward :190-225, step :228-262,
reset :265-291.
"""
import numpy as np


class CheckersEnvOracle(object):
    def __init__(self, n_rows=3, n_columns=16, n_obs=2, agents_r=(0, 2), agents_c=(16, 16),
                 n_agents=1, max_steps=50):
        assert n_rows % 2 == 1 and n_columns % 2 == 0           # checkers.py:16-17
        self.n_rows, self.n_columns, self.n_obs = n_rows, n_columns, n_obs
        self.total_rows = n_rows + 2 * n_obs                      # :24
        self.total_columns = n_columns + 2 * n_obs + 1            # :25
        self.max_collectible = n_rows * n_columns                 # :28
        self.n_agents = n_agents
        self.max_steps = max_steps
        self.start_r = np.array(agents_r) + n_obs                 # :34
        self.start_c = np.array(agents_c) + n_obs                 # :35

    # ---- world construction (populate_world :38-63) ------------------------------------
    def _populate(self):
        w, o, R, C = self.world, self.n_obs, self.n_rows, self.n_columns
        w[:, 0:o, 2] = 1
        w[0:o, :, 2] = 1
        w[o + R:, :, 2] = 1
        w[o:o + R, o + C + 1:, 2] = 1
        for i in range(self.n_agents):
            r, c = self.loc[i]
            w[r, c, 2] = -1
        green_first = True
        for row in range(o, o + R):
            first, second = (0, 1) if green_first else (1, 0)
            w[row, o:o + C:2, first] = -1
            w[row, o + 1:o + C:2, second] = -1
            green_first = not green_first

    def reset(self, goals):
        """checkers.py:265-291."""
        self.world = np.zeros((self.total_rows, self.total_columns, 3))
        self.steps = 0
        self.goals = np.asarray(goals)
        if self.n_agents == 1:                                    # :271-276
            goal = int(np.where(self.goals[0] == 1)[0][0])
            self.start_r = np.array([0 if goal == 0 else 2]) + self.n_obs
        self.loc = np.zeros((self.n_agents, 2), dtype=int)
        self.loc[:, 0] = self.start_r
        self.loc[:, 1] = self.start_c
        self._populate()
        self.collected = np.zeros((self.n_agents, 2))
        gs = self.global_state()
        oo, ot, ov = self.local_observation()
        return gs, oo, ot, ov, False

    # ---- observations --------------------------------------------------------------------
    def global_state(self):
        """:66-94.  grid is returned BY VALUE here (the reference hands out a live view,
        SURVEY.md §7.3 item 8)."""
        o = self.n_obs
        grid = self.world[o:o + self.n_rows, o:o + self.n_columns + 1, 0:2].copy()
        vec = [np.concatenate([self.loc[i].astype(float), self.collected[i]])
               for i in range(self.n_agents)]
        return grid, vec

    def _normalize(self, loc):
        """:112-125."""
        loc = np.array(loc, dtype=float)
        if loc.ndim == 1:
            loc[0] = (loc[0] - self.total_rows / 2.0) / self.total_rows
            loc[1] = (loc[1] - self.total_columns / 2.0) / self.total_columns
        else:
            loc[:, 0] = (loc[:, 0] - self.total_rows / 2.0) / self.total_rows
            loc[:, 1] = (loc[:, 1] - self.total_columns / 2.0) / self.total_columns
        return loc

    def local_observation(self):
        """:97-109, :128-154."""
        o = self.n_obs
        obs_t, obs_v, obs_o = [], [], []
        for i in range(self.n_agents):
            r, c = self.loc[i]
            win = np.array(self.world[r - o:r + o + 1, c - o:c + o + 1, :])
            win[o, o, 2] = 0
            v = np.concatenate([self._normalize(self.loc[i]),
                                self.collected[i] / (self.max_collectible / 2.0)])
            obs_t.append(win)
            obs_v.append(v)
            if self.n_agents == 1:
                others = np.reshape(self._normalize(self.loc[i]), 2)
            else:
                mask = np.arange(self.n_agents) != i
                others = np.reshape(self._normalize(self.loc[mask, :]), (self.n_agents - 1) * 2)
            obs_o.append(others)
        return obs_o, obs_t, obs_v

    # ---- dynamics ------------------------------------------------------------------------
    _MOVES = {1: (-1, 0), 2: (+1, 0), 3: (0, -1), 4: (0, +1)}

    def _act(self, i, action):
        """agent_act :157-187."""
        r, c = self.loc[i]
        if action == 0:
            return 0
        if action in self._MOVES:
            dr, dc = self._MOVES[action]
            if self.world[r + dr, c + dc, 2] == 0:
                self.world[r + dr, c + dc, 2] = -1
                self.world[r, c, 2] = 0
                self.loc[i] = (r + dr, c + dc)
                return 0
        return -0.1

    def _collect(self, i, goal):
        """get_reward :190-225."""
        r, c = self.loc[i]
        if goal not in (0, 1):
            raise ValueError("goal index must be 0 or 1")
        for ch in (0, 1):          # green is tested before orange
            if self.world[r, c, ch] == -1:
                self.world[r, c, ch] = 1
                self.collected[i, ch] += 1
                return 1.0 if ch == goal else -0.5
        return 0

    def step(self, actions):
        """:228-262."""
        local = []
        for i in range(self.n_agents):
            penalty = self._act(i, actions[i])
            goal = int(np.where(self.goals[i] == 1)[0][0])
            local.append(penalty + self._collect(i, goal))
        gs = self.global_state()
        oo, ot, ov = self.local_observation()
        total = np.sum(local)
        self.steps += 1
        if self.steps == self.max_steps:
            done = True
        elif self.n_agents == 1:
            goal = int(np.where(self.goals[0] == 1)[0][0])
            done = bool(np.sum(self.world[:, :, goal]) == self.max_collectible / 2.0)
        else:
            done = bool(np.sum(self.world[:, :, 0:2]) == self.max_collectible)
        return gs, oo, ot, ov, total, local, done


class VecCheckersOracle(object):
    """E environments on the compact state (what the HIP kernel keeps in HBM):

      mask[e]      uint64, bit (k*n_columns + j) set <=> reward cell (row k, col j) collected
      loc[e,i,:]   (r, c) of agent i in expanded-grid coordinates
      count[e,i,:] (#green, #orange) collected by agent i
      steps[e]

    A reward cell (k, j) is green iff (k + j) is even (populate_world :54-63).  Channel 2 of the
    dense world is derived: 1 on walls, -1 where an agent stands, 0 elsewhere -- valid because
    agents can never share a cell (agent_act refuses occupied targets) provided they start on
    distinct cells, which the constructor enforces.
    """

    def __init__(self, n_rows, n_columns, n_obs, agents_r, agents_c, n_agents, max_steps, n_envs):
        assert n_rows % 2 == 1 and n_columns % 2 == 0
        assert n_rows * n_columns <= 64
        self.R, self.C, self.O = n_rows, n_columns, n_obs
        self.TR = n_rows + 2 * n_obs
        self.TC = n_columns + 2 * n_obs + 1
        self.N = n_agents
        self.E = n_envs
        self.max_steps = max_steps
        self.max_collectible = n_rows * n_columns
        self.start_r = np.array(agents_r[:n_agents]) + n_obs
        self.start_c = np.array(agents_c[:n_agents]) + n_obs
        if n_agents > 1:
            cells = set(zip(self.start_r.tolist(), self.start_c.tolist()))
            assert len(cells) == n_agents, "agents must start on distinct cells"
        # static wall map (channel 2 == 1)
        wall = np.zeros((self.TR, self.TC), bool)
        o, R, C = n_obs, n_rows, n_columns
        wall[:, 0:o] = True
        wall[0:o, :] = True
        wall[o + R:, :] = True
        wall[o:o + R, o + C + 1:] = True
        self.wall = wall

    def reset(self, goals):
        """goals int [E,N] (index of the wanted colour) or one-hot [E,N,2] / [N,2]."""
        g = np.asarray(goals)
        if g.ndim == 2 and g.shape == (self.N, 2):
            g = np.broadcast_to(g, (self.E, self.N, 2))
        if g.ndim == 3:
            g = np.argmax(g, axis=2)
        self.goal = np.ascontiguousarray(g).astype(np.int64).reshape(self.E, self.N)
        self.mask = np.zeros(self.E, np.uint64)
        self.loc = np.zeros((self.E, self.N, 2), np.int64)
        self.loc[:, :, 0] = self.start_r
        self.loc[:, :, 1] = self.start_c
        if self.N == 1:
            self.loc[:, 0, 0] = np.where(self.goal[:, 0] == 0, 0, 2) + self.O
        self.count = np.zeros((self.E, self.N, 2), np.int64)
        self.steps = np.zeros(self.E, np.int64)
        return self.outputs()

    def reset_envs(self, sel, goal_index=None):
        """Restart the envs selected by the bool mask `sel` (the outer loop of train_onpolicy.py:281-294 calling
        Checkers.reset for a fresh episode); goal_index int [E,N] replaces the goals of those envs (N == 1 draws a new
        one-hot goal per episode, :288-291)."""
        sel = np.asarray(sel, bool)
        if goal_index is not None:
            self.goal = np.where(sel[:, None], np.asarray(goal_index).reshape(self.E, self.N), self.goal)
        self.mask = np.where(sel, np.uint64(0), self.mask)
        self.loc[sel, :, 0] = self.start_r
        self.loc[sel, :, 1] = self.start_c
        if self.N == 1:
            self.loc[sel, 0, 0] = (np.where(self.goal[:, 0] == 0, 0, 2) + self.O)[sel]
        self.count[sel] = 0
        self.steps = np.where(sel, 0, self.steps)
        return self.outputs()

    # -- dense reconstruction ---------------------------------------------------------------
    def dense_world(self):
        E, TR, TC, O, R, C = self.E, self.TR, self.TC, self.O, self.R, self.C
        w = np.zeros((E, TR, TC, 3))
        k = np.arange(R)[:, None]
        j = np.arange(C)[None, :]
        bit = (k * C + j).astype(np.uint64)
        coll = ((self.mask[:, None, None] >> bit[None]) & np.uint64(1)).astype(bool)   # [E,R,C]
        green = ((k + j) % 2 == 0)[None]
        val = np.where(coll, 1.0, -1.0)
        w[:, O:O + R, O:O + C, 0] = np.where(green, val, 0.0)
        w[:, O:O + R, O:O + C, 1] = np.where(~green, val, 0.0)
        w[:, :, :, 2] = self.wall[None].astype(float)
        e = np.arange(E)
        for i in range(self.N):
            w[e, self.loc[:, i, 0], self.loc[:, i, 1], 2] = -1.0
        return w

    def outputs(self):
        """-> grid[E,R,C+1,2], vec[E,N,4], obs_others[E,N,2*max(N-1,1)], obs_self_t[E,N,2O+1,2O+1,3],
        obs_self_v[E,N,4]   (all float64, the reference's values)."""
        E, N, O = self.E, self.N, self.O
        w = self.dense_world()
        grid = w[:, O:O + self.R, O:O + self.C + 1, 0:2].copy()
        vec = np.concatenate([self.loc.astype(float), self.count.astype(float)], axis=2)
        nr = (self.loc[:, :, 0] - self.TR / 2.0) / self.TR
        nc = (self.loc[:, :, 1] - self.TC / 2.0) / self.TC
        norm = np.stack([nr, nc], axis=2)                                  # [E,N,2]
        obs_v = np.concatenate([norm, self.count / (self.max_collectible / 2.0)], axis=2)
        K = 2 * O + 1
        obs_t = np.zeros((E, N, K, K, 3))
        e = np.arange(E)
        for i in range(N):
            for dr in range(K):
                for dc in range(K):
                    obs_t[:, i, dr, dc, :] = w[e, self.loc[:, i, 0] - O + dr, self.loc[:, i, 1] - O + dc, :]
            obs_t[:, i, O, O, 2] = 0.0
        if N == 1:
            obs_o = norm.reshape(E, 1, 2).copy()
        else:
            obs_o = np.zeros((E, N, (N - 1) * 2))
            for i in range(N):
                others = [j for j in range(N) if j != i]
                obs_o[:, i, :] = norm[:, others, :].reshape(E, (N - 1) * 2)
        return grid, vec, obs_o, obs_t, obs_v

    # -- dynamics ---------------------------------------------------------------------------
    def step(self, actions):
        """actions int [E,N] -> (grid, vec, obs_others, obs_self_t, obs_self_v, total[E], local[E,N], done[E])."""
        E, N, O, R, C = self.E, self.N, self.O, self.R, self.C
        actions = np.asarray(actions).reshape(E, N)
        e = np.arange(E)
        local = np.zeros((E, N))
        dr_tab = np.array([0, -1, +1, 0, 0])
        dc_tab = np.array([0, 0, 0, -1, +1])
        for i in range(N):                                  # sequential: agent i sees agents < i moved
            a = actions[:, i]
            inrange = (a >= 1) & (a <= 4)
            ai = np.where(inrange, a, 0)
            tr = self.loc[:, i, 0] + dr_tab[ai]
            tc = self.loc[:, i, 1] + dc_tab[ai]
            blocked = self.wall[tr, tc].copy()
            for j in range(N):
                if j != i:
                    blocked |= (self.loc[:, j, 0] == tr) & (self.loc[:, j, 1] == tc)
            move = inrange & ~blocked
            penalty = np.where((a != 0) & ~move, -0.1, 0.0)
            self.loc[:, i, 0] = np.where(move, tr, self.loc[:, i, 0])
            self.loc[:, i, 1] = np.where(move, tc, self.loc[:, i, 1])
            k = self.loc[:, i, 0] - O
            j_ = self.loc[:, i, 1] - O
            incell = (k >= 0) & (k < R) & (j_ >= 0) & (j_ < C)
            bit = np.where(incell, k * C + j_, 0).astype(np.uint64)
            fresh = incell & (((self.mask >> bit) & np.uint64(1)) == 0)
            colour = np.where(((k + j_) % 2) == 0, 0, 1)
            rew = np.where(fresh, np.where(colour == self.goal[:, i], 1.0, -0.5), 0.0)
            self.mask = np.where(fresh, self.mask | (np.uint64(1) << bit), self.mask)
            self.count[e, i, colour] += fresh.astype(np.int64)
            local[:, i] = penalty + rew
        total = local[:, 0].copy()
        for i in range(1, N):
            total = total + local[:, i]
        self.steps = self.steps + 1
        popcnt = np.array([bin(int(m)).count("1") for m in self.mask]) if E <= 4096 else _popcount64(self.mask)
        if N == 1:
            k = np.arange(R)[:, None]
            j = np.arange(C)[None, :]
            bits = (k * C + j)
            gmask = np.uint64(sum(1 << int(b) for b in bits[((k + j) % 2) == 0]))
            omask = np.uint64(sum(1 << int(b) for b in bits[((k + j) % 2) == 1]))
            want = np.where(self.goal[:, 0] == 0, gmask, omask)
            all_got = (self.mask & want) == want
        else:
            all_got = popcnt == self.max_collectible
        done = (self.steps == self.max_steps) | all_got
        grid, vec, oo, ot, ov = self.outputs()
        return grid, vec, oo, ot, ov, total, local, done


def _popcount64(x):
    x = x.astype(np.uint64)
    c = np.zeros(x.shape, np.int64)
    for s in range(64):
        c += ((x >> np.uint64(s)) & np.uint64(1)).astype(np.int64)
    return c
