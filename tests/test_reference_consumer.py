"""CPU, build container only: the REAL alg_credit.process_batch / alg_credit_checkers.process_batch
(alg_credit.py:445-499, alg_credit_checkers.py:414-470) consume the exporter's rows unchanged.

Runs only where /root/reference exists (it cannot travel to the GPU box); the transition columns are
taken from the golden vectors, packed by cm3_amd.rollout.rows_from_columns exactly as
ParticleRollout.as_reference_rows / CheckersRollout.as_reference_rows pack device trajectories."""
import os
import sys
import types

import numpy as np
import pytest

from cm3_amd.rollout import CHECKERS_ORDER, PARTICLE_ORDER, rows_from_columns
from tests.helpers import load_golden

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "alg")), reason="reference tree not present")


def _import_alg(name):
    """Imports a reference alg module with a permissive stub `tensorflow` (TF1 is not installed); only the
    pure-NumPy batch reshapers are exercised."""
    if "tensorflow" not in sys.modules:
        class _Any(types.ModuleType):
            def __getattr__(self, k):
                return _Any(k)

            def __call__(self, *a, **k):
                return _Any("call")
        sys.modules["tensorflow"] = _Any("tensorflow")
    sys.dont_write_bytecode = True
    p = os.path.join(REF, "alg")
    if p not in sys.path:
        sys.path.insert(0, p)
    if not hasattr(np, "int"):
        np.int = int
    return __import__(name)


def _particle_columns(g, ep):
    T = int(g["ep_len"][ep])
    N = g["meta"]["n_agents"]
    gs = np.concatenate([g["init_gs"][ep][None], g["gs"][ep, :T]])
    oo = np.concatenate([g["init_obs_others"][ep][None], g["obs_others"][ep, :T]])
    goals = np.repeat(g["landmarks"][ep][None], T, axis=0)
    return dict(v_global=gs[:-1], obs_others=oo[:-1], v_local=gs[:-1], actions=g["actions"][ep, :T],
                reward=g["reward"][ep, :T], reward_local=g["reward_n"][ep, :T], v_global_next=gs[1:],
                obs_others_next=oo[1:], v_local_next=gs[1:], done=g["done"][ep, :T], goals=goals), T, N


def test_real_particle_process_batch_consumes_rows():
    alg_credit = _import_alg("alg_credit")
    g = load_golden("particle_antipodal_greedy")
    cols, T, N = _particle_columns(g, 0)
    rows = rows_from_columns(cols, PARTICLE_ORDER)
    alg = alg_credit.Alg.__new__(alg_credit.Alg)
    alg.n_agents, alg.l_action, alg.experiment = N, 5, "particle"
    alg.l_obs_others, alg.l_obs, alg.l_goal = 4 * (N - 1), 4, 2
    alg.l_state_one_agent, alg.l_state = 4, 4 * N
    out = alg.process_batch(rows)
    (n_steps, v_global, obs_others, v_local, a_1hot, a_others_1hot, reward, reward_local, v_global_next,
     obs_others_next, v_local_next, done, goals) = out
    assert n_steps == T
    assert np.array_equal(v_global, cols["v_global"]) and np.array_equal(v_global_next, cols["v_global_next"])
    assert np.array_equal(obs_others, cols["obs_others"].reshape(T * N, -1))
    assert np.array_equal(v_local, cols["v_local"].reshape(T * N, 4))
    assert np.array_equal(reward, np.repeat(cols["reward"], N))
    assert np.array_equal(reward_local, cols["reward_local"].reshape(-1))
    assert np.array_equal(done, np.repeat(cols["done"], N))
    assert np.array_equal(goals, cols["goals"])
    assert a_1hot.shape == (T * N, 5) and a_others_1hot.shape == (T * N, N - 1, 5)
    assert np.array_equal(a_1hot.argmax(1), cols["actions"].reshape(-1))
    gself, gothers = alg.process_goals(goals, n_steps)
    assert gself.shape == (T * N, 2) and gothers.shape == (T * N, (N - 1) * 2)
    one, others, state = alg.process_global_state(v_global, n_steps)
    assert state.shape == (T * N, 4 * N) and others.shape == (T * N, (N - 1) * 4)


def test_real_checkers_process_batch_consumes_rows():
    mod = _import_alg("alg_credit_checkers")
    g = load_golden("checkers_stage2_uniform")
    ep, T, N = 0, int(g["ep_len"][0]), 2

    def seq(init, per_tick):
        return np.concatenate([g[init][ep][None], g[per_tick][ep, :T]])
    grid, vec = seq("init_grid", "grid"), seq("init_vec", "vec")
    oo, ot, ov = seq("init_obs_others", "obs_others"), seq("init_obs_self_t", "obs_self_t"), seq("init_obs_self_v", "obs_self_v")
    acts = g["actions"][ep, :T]
    prev = np.concatenate([np.zeros((1, N), acts.dtype), acts[:-1]])
    cols = dict(grid=grid[:-1], vec=vec[:-1], obs_others=oo[:-1], obs_self_t=ot[:-1], obs_self_v=ov[:-1],
                actions_prev=prev, actions=acts, reward=g["reward"][ep, :T], local_rewards=g["local_rewards"][ep, :T],
                next_grid=grid[1:], next_vec=vec[1:], next_obs_others=oo[1:], next_obs_self_t=ot[1:],
                next_obs_self_v=ov[1:], done=g["done"][ep, :T],
                goals=np.repeat(g["goals"][ep][None], T, axis=0).astype(float))
    rows = rows_from_columns(cols, CHECKERS_ORDER)
    alg = mod.Alg.__new__(mod.Alg)
    d = g["meta"]["config"]["dimensions"]
    alg.n_agents, alg.l_action, alg.experiment = N, d["l_action"], "checkers"
    for k, v in d.items():
        setattr(alg, k, v)
    alg.l_obs_others, alg.l_obs_self, alg.l_goal = d["l_obs_others"], d["l_obs_self"], d["l_goal"]
    alg.rows_obs, alg.columns_obs, alg.channels_obs = d["rows_obs"], d["columns_obs"], d["channels_obs"]
    alg.l_state_one_agent = d["l_state_one"]
    out = alg.process_batch(rows)
    assert out[0] == T
    flat = [o for o in out[1:] if isinstance(o, np.ndarray)]
    assert any(o.shape == (T * N, 3, 9, 2) and np.array_equal(o, np.repeat(cols["grid"], N, axis=0)) for o in flat)
    assert any(o.shape == (T, N, 4) and np.array_equal(o, cols["vec"]) for o in flat)
    assert any(o.shape == (T * N, 5, 5, 3) and np.array_equal(o, cols["obs_self_t"].reshape(T * N, 5, 5, 3)) for o in flat)
