"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's particle env.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this; the product (``cm3_amd``) never does.

Parity status: PINNED by ``tests/golden/particle_*.npz`` -- vectors recorded in the build
container from the reference's own ``MultiAgentEnv`` + ``multi-goal_spread`` scenario by
``oracle/gen_golden.py`` (the reference ships no tests or golden vectors of its own,
SURVEY.md §4).  ``tests/test_oracle_particle.py`` checks both classes below bit-for-bit
(float64) against those vectors.

Two restatements of the same algorithm:

* ``ParticleEnvOracle``   one environment, scalar call structure shaped like the reference
                          (per-agent / per-pair Python loops over 2-vectors).  This is the
                          "reference-shaped" CPU baseline of SURVEY.md §8(d).
* ``VecParticleOracle``   ``[E, N, ...]`` vectorised NumPy, dtype-generic (float64 is
                          bit-identical to the scalar one; float32 mirrors the kernel).

Reference files (relative to /root/reference/env/multiagent-particle-envs/multiagent):
  core.py          World.step :117-131, apply_action_force :134-140,
                   apply_environment_force :143-155, integrate_state :158-169,
                   get_collision_force :180-196, constants :82-99
  environment.py   step :81-123, reset :125-149, _set_action :177-225
  scenarios/multi-goal_spread.py  make_world :19-63, reset_world :65-93, is_collision :114-118,
                   reward :121-138, done :140-143, observation :145-154
"""
import numpy as np

# core.py:94-99 and multi-goal_spread.py:41-57
DT = 0.1
DAMPING = 0.25
CONTACT_FORCE = 1e+2
CONTACT_MARGIN = 1e-3
AGENT_SIZE = 0.15
SENSITIVITY = 5.0          # environment.py:211 (agent.accel is None)
REACH_THRESHOLD = -0.05    # multi-goal_spread.py:126
MASS = 1.0                 # core.py:45-49


def np_list_sum(values):
    """Value of ``np.sum(list_of_float64)`` as the reference computes it (environment.py:107).

    NumPy's add.reduce on a contiguous 1-D float64 array: plain left-to-right for n < 8,
    8 interleaved accumulators combined as a fixed tree for 8 <= n <= 128 (verified against
    NumPy 2.2.6 in tests/test_oracle_particle.py::test_np_list_sum_order).
    """
    vals = [np.float64(v) for v in values]
    n = len(vals)
    if n == 0:
        return np.float64(0.0)
    if n < 8:
        acc = vals[0]
        for v in vals[1:]:
            acc = acc + v
        return acc
    assert n <= 128
    r = vals[:8]
    i = 8
    while i < n - (n % 8):
        for k in range(8):
            r[k] = r[k] + vals[i + k]
        i += 8
    acc = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        acc = acc + vals[i]
        i += 1
    return acc


def action_to_force(action, dtype=np.float64):
    """environment.py:193-214: discrete action -> u in {0, +-1 e_axis}, then u *= 5.0.

    Actions outside 1..4 (including 0) leave u at zero (no branch matches, :197-200).
    """
    u = np.zeros(2, dtype=dtype)
    if action == 1:
        u[0] = -1.0
    if action == 2:
        u[0] = +1.0
    if action == 3:
        u[1] = -1.0
    if action == 4:
        u[1] = +1.0
    u *= dtype(SENSITIVITY)
    return u


class ParticleEnvOracle(object):
    """One environment; float64 unless ``dtype`` says otherwise."""

    def __init__(self, n_agents, config, prob_random, max_steps, dtype=np.float64):
        self.n = int(n_agents)
        self.cfg = config
        self.prob_random = prob_random
        self.max_steps = int(max_steps)
        self.dtype = dtype
        self.pos = np.zeros((self.n, 2), dtype)
        self.vel = np.zeros((self.n, 2), dtype)
        self.landmarks = np.zeros((self.n, 2), dtype)
        self.reached = [False] * self.n
        self.collisions = 0
        self.steps = 0

    # ---- reset -------------------------------------------------------------------------
    def reset(self, py_rng, np_rng):
        """multi-goal_spread.py:65-93 + environment.py:125-149.

        ``py_rng`` is a ``random.Random`` (or the ``random`` module) and ``np_rng`` an
        ``np.random.RandomState`` (or the ``np.random`` module); draws are made in the
        reference's order so identically-seeded generators give identical states:
        one ``random()``; per agent ``uniform(-1,1,2)`` or two ``normal(0,std)``; per
        landmark ``uniform(-1,1,2)`` only in the random branch.
        """
        c = self.cfg
        rand_num = py_rng.random()
        for i in range(self.n):
            if rand_num < self.prob_random:
                self.pos[i] = np_rng.uniform(-1, +1, 2)
            else:
                x = c['agents_x'][i] + np_rng.normal(0, c['initial_std'])
                y = c['agents_y'][i] + np_rng.normal(0, c['initial_std'])
                self.pos[i] = (x, y)
            self.vel[i] = 0.0
            self.reached[i] = False
        for i in range(self.n):
            if rand_num < self.prob_random:
                self.landmarks[i] = np_rng.uniform(-1, +1, 2)
            else:
                self.landmarks[i] = (c['landmarks_x'][i], c['landmarks_y'][i])
        self.collisions = 0
        self.steps = 0
        obs_self, obs_others = self._observe()
        done_n = [self.reached[i] for i in range(self.n)]
        return self.global_state(), obs_others, obs_self, bool(np.any(done_n))

    def set_state(self, pos, vel, landmarks, steps=0, collisions=0):
        """State injection (parity tests, checkpoint)."""
        self.pos[...] = pos
        self.vel[...] = vel
        self.landmarks[...] = landmarks
        self.steps = int(steps)
        self.collisions = int(collisions)
        self.reached = [False] * self.n

    # ---- pieces of step ----------------------------------------------------------------
    def _contact_force(self, a, b):
        """core.py:180-196 for two colliding, movable agents a < b."""
        dt = self.dtype
        delta = self.pos[a] - self.pos[b]
        dist = np.sqrt(np.sum(np.square(delta)))
        dist_min = dt(AGENT_SIZE) + dt(AGENT_SIZE)
        k = dt(CONTACT_MARGIN)
        penetration = np.logaddexp(dt(0), -(dist - dist_min) / k) * k
        return dt(CONTACT_FORCE) * delta / dist * penetration

    def _world_step(self, actions):
        """core.py:117-131 with the scenario's constants."""
        dt = self.dtype
        # environment.py:89-90 then core.py:134-140 (u_noise is None -> + 0.0)
        force = [action_to_force(actions[i], dt) + dt(0.0) for i in range(self.n)]
        # core.py:143-155: entity pairs a < b; landmarks have collide=False (:181)
        for a in range(self.n):
            for b in range(a + 1, self.n):
                f = self._contact_force(a, b)
                force[a] = f + force[a]
                force[b] = (-f) + force[b]
        # core.py:158-169 (max_speed None)
        for i in range(self.n):
            self.vel[i] = self.vel[i] * dt(1 - DAMPING)
            self.vel[i] += (force[i] / dt(MASS)) * dt(DT)
            self.pos[i] += self.vel[i] * dt(DT)

    def _is_collision(self, i, j):
        """multi-goal_spread.py:114-118 (agent1 = i, agent2 = j)."""
        delta = self.pos[i] - self.pos[j]
        dist = np.sqrt(np.sum(np.square(delta)))
        return bool(dist < self.dtype(AGENT_SIZE) + self.dtype(AGENT_SIZE))

    def _reward(self, i):
        """multi-goal_spread.py:121-138."""
        rew = 0
        rew -= np.sqrt(np.sum(np.square(self.pos[i] - self.landmarks[i])))
        self.reached[i] = bool(rew >= REACH_THRESHOLD)
        for j in range(self.n):
            if j == i:
                continue
            if self._is_collision(j, i):
                rew -= 1
                self.collisions += 1      # double counts by design (:135-137)
        return rew

    def _observe_one(self, i):
        """multi-goal_spread.py:145-154."""
        others = []
        for j in range(self.n):
            if j == i and self.n > 1:
                continue
            others.append(self.vel[j] - self.vel[i])
            others.append(self.pos[j] - self.pos[i])
        return np.concatenate([self.vel[i], self.pos[i]]), np.concatenate(others)

    def _observe(self):
        obs_self, obs_others = [], []
        for i in range(self.n):
            s, o = self._observe_one(i)
            obs_self.append(s)
            obs_others.append(o)
        return obs_self, obs_others

    def global_state(self):
        """environment.py:113-116: rows (vx, vy, px, py)."""
        return np.concatenate([self.vel, self.pos], axis=1)

    # ---- step --------------------------------------------------------------------------
    def step(self, actions):
        """environment.py:81-123."""
        self._world_step(actions)
        self.steps += 1
        obs_self, obs_others, reward_n, done_n = [], [], [], []
        for i in range(self.n):
            s, o = self._observe_one(i)
            obs_self.append(s)
            obs_others.append(o)
            reward_n.append(self._reward(i))
            done_n.append(self.reached[i])
        reward = np_list_sum(reward_n)
        done = bool(self.steps == self.max_steps or all(done_n))
        return self.global_state(), obs_others, obs_self, reward, reward_n, done


class VecParticleOracle(object):
    """E independent environments, vectorised over the leading dim.  Same algorithm,
    same operation order per env (so float64 results equal ``ParticleEnvOracle``)."""

    def __init__(self, n_agents, config, prob_random, max_steps, n_envs, dtype=np.float64):
        self.n = int(n_agents)
        self.E = int(n_envs)
        self.cfg = config
        self.prob_random = prob_random
        self.max_steps = int(max_steps)
        self.dtype = np.dtype(dtype).type
        E, N = self.E, self.n
        self.pos = np.zeros((E, N, 2), dtype)
        self.vel = np.zeros((E, N, 2), dtype)
        self.landmarks = np.zeros((E, N, 2), dtype)
        self.steps = np.zeros(E, np.int64)
        self.collisions = np.zeros(E, np.int64)
        self.n_others = max(N - 1, 1)

    def set_state(self, pos, vel, landmarks, steps=None, collisions=None):
        self.pos[...] = pos
        self.vel[...] = vel
        self.landmarks[...] = landmarks
        self.steps[...] = 0 if steps is None else steps
        self.collisions[...] = 0 if collisions is None else collisions

    def set_from_global_state(self, gs, landmarks, steps=None, collisions=None):
        gs = np.asarray(gs)
        self.set_state(gs[..., 2:4], gs[..., 0:2], landmarks, steps, collisions)

    def global_state(self):
        return np.concatenate([self.vel, self.pos], axis=2)            # [E,N,4]

    def observe(self):
        """-> obs_self [E,N,4], obs_others [E,N,max(N-1,1)*4]  (multi-goal_spread.py:145-154)."""
        E, N = self.E, self.n
        gs = self.global_state()
        oo = np.zeros((E, N, self.n_others, 4), self.pos.dtype)
        for i in range(N):
            k = 0
            for j in range(N):
                if j == i and N > 1:
                    continue
                oo[:, i, k, 0:2] = self.vel[:, j] - self.vel[:, i]
                oo[:, i, k, 2:4] = self.pos[:, j] - self.pos[:, i]
                k += 1
        return gs.copy(), oo.reshape(E, N, self.n_others * 4)

    @staticmethod
    def _sum_agents(x):
        """np.sum order of environment.py:107 applied along axis 1."""
        n = x.shape[1]
        if n < 8:
            acc = x[:, 0].copy()
            for i in range(1, n):
                acc = acc + x[:, i]
            return acc
        r = [x[:, k].copy() for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = r[k] + x[:, i + k]
            i += 8
        acc = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            acc = acc + x[:, i]
            i += 1
        return acc

    def step(self, actions):
        """actions int [E,N] -> (global_state[E,N,4], obs_others[E,N,L], obs_self[E,N,4],
        reward[E], reward_n[E,N], done[E] bool); also updates ``collisions`` and ``steps``
        and leaves ``self.reached`` [E,N]."""
        dt = self.dtype
        E, N = self.E, self.n
        actions = np.asarray(actions).reshape(E, N)
        # _set_action (environment.py:193-214)
        u = np.zeros((E, N, 2), self.pos.dtype)
        u[..., 0] = np.where(actions == 1, dt(-1.0), u[..., 0])
        u[..., 0] = np.where(actions == 2, dt(+1.0), u[..., 0])
        u[..., 1] = np.where(actions == 3, dt(-1.0), u[..., 1])
        u[..., 1] = np.where(actions == 4, dt(+1.0), u[..., 1])
        u *= dt(SENSITIVITY)
        force = u + dt(0.0)
        # pair forces (core.py:143-155,180-196)
        k = dt(CONTACT_MARGIN)
        dist_min = dt(AGENT_SIZE) + dt(AGENT_SIZE)
        with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
            for a in range(N):
                for b in range(a + 1, N):
                    delta = self.pos[:, a] - self.pos[:, b]
                    dist = np.sqrt(delta[:, 0] * delta[:, 0] + delta[:, 1] * delta[:, 1])
                    pen = np.logaddexp(dt(0), -(dist - dist_min) / k) * k
                    f = dt(CONTACT_FORCE) * delta / dist[:, None] * pen[:, None]
                    force[:, a] = f + force[:, a]
                    force[:, b] = (-f) + force[:, b]
        # integrate (core.py:158-169)
        self.vel = self.vel * dt(1 - DAMPING)
        self.vel = self.vel + (force / dt(MASS)) * dt(DT)
        self.pos = self.pos + self.vel * dt(DT)
        self.steps = self.steps + 1
        # observation / reward / done (environment.py:95-121)
        obs_self, obs_others = self.observe()
        reward_n = np.zeros((E, N), self.pos.dtype)
        reached = np.zeros((E, N), bool)
        for i in range(N):
            d = self.pos[:, i] - self.landmarks[:, i]
            rew = dt(0) - np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
            reached[:, i] = rew >= dt(REACH_THRESHOLD)
            for j in range(N):
                if j == i:
                    continue
                dd = self.pos[:, j] - self.pos[:, i]
                hit = np.sqrt(dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) < dist_min
                rew = np.where(hit, rew - dt(1), rew)
                self.collisions = self.collisions + hit
            reward_n[:, i] = rew
        self.reached = reached
        reward = self._sum_agents(reward_n)
        done = (self.steps == self.max_steps) | reached.all(axis=1)
        return self.global_state(), obs_others, obs_self, reward, reward_n, done

    def pair_margins(self):
        """Distances to the two discontinuities: min over pairs of
        | |p_i-p_j| - 0.3 | and min over agents of | |p_i-L_i| - 0.05 |, per env.  Parity tests
        mask envs whose margin is below the float32 noise floor."""
        E, N = self.E, self.n
        m_col = np.full(E, np.inf)
        for i in range(N):
            for j in range(i + 1, N):
                dd = self.pos[:, j] - self.pos[:, i]
                m_col = np.minimum(m_col, np.abs(np.sqrt((dd * dd).sum(-1)) - 0.3))
        d = self.pos - self.landmarks
        m_reach = np.abs(np.sqrt((d * d).sum(-1)) - 0.05).min(axis=1)
        return m_col, m_reach
