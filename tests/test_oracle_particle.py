"""CPU: the particle oracle (oracle/particle_oracle.py) against the golden vectors recorded
from the reference's own MultiAgentEnv (tests/golden/particle_*.npz).  float64, bit-exact."""
import random

import numpy as np
import pytest

from oracle.particle_oracle import ParticleEnvOracle, VecParticleOracle, np_list_sum
from tests.helpers import golden_names, load_golden

NAMES = golden_names("particle_")


def test_fixtures_present():
    assert len(NAMES) >= 12


def test_np_list_sum_order():
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 10, 16):
        for _ in range(300):
            a = list(rng.normal(size=n) * rng.choice([1e-3, 1.0, 1e3], size=n))
            assert np_list_sum(a) == np.sum(a)


@pytest.mark.parametrize("name", NAMES)
def test_scalar_oracle_matches_reference(name):
    g = load_golden(name)
    m = g["meta"]
    N = m["n_agents"]
    env = ParticleEnvOracle(N, m["config"], m["prob_random"], m["max_steps"])
    for ep in range(len(g["ep_len"])):
        gs0 = g["init_gs"][ep]
        env.set_state(gs0[:, 2:4], gs0[:, 0:2], g["landmarks"][ep])
        s, o = env._observe()
        assert np.array_equal(np.array(o), g["init_obs_others"][ep])
        assert np.array_equal(np.array(s), g["init_obs_self"][ep])
        for t in range(g["ep_len"][ep]):
            gs, oo, os_, rew, rew_n, done = env.step(g["actions"][ep, t])
            assert np.array_equal(gs, g["gs"][ep, t]), (name, ep, t)
            assert np.array_equal(np.array(oo), g["obs_others"][ep, t])
            assert np.array_equal(np.array(os_), g["obs_self"][ep, t])
            assert rew == g["reward"][ep, t]
            assert np.array_equal(np.array(rew_n), g["reward_n"][ep, t])
            assert done == g["done"][ep, t]
            assert env.collisions == g["collisions"][ep, t]
            assert np.array_equal(np.array(env.reached), g["reached"][ep, t])


@pytest.mark.parametrize("name", NAMES)
def test_vector_oracle_matches_reference(name):
    """All episodes of a fixture stepped side by side as E = n_episodes environments."""
    g = load_golden(name)
    m = g["meta"]
    N = m["n_agents"]
    Ep = len(g["ep_len"])
    env = VecParticleOracle(N, m["config"], m["prob_random"], m["max_steps"], Ep)
    env.set_from_global_state(g["init_gs"], g["landmarks"])
    s, o = env.observe()
    assert np.array_equal(o, g["init_obs_others"])
    for t in range(int(g["ep_len"].max())):
        live = g["ep_len"] > t
        acts = np.where(live[:, None], g["actions"][:, t], 0)
        gs, oo, os_, rew, rew_n, done = env.step(acts)
        assert np.array_equal(gs[live], g["gs"][live, t]), (name, t)
        assert np.array_equal(oo[live], g["obs_others"][live, t])
        assert np.array_equal(os_[live], g["obs_self"][live, t])
        assert np.array_equal(rew[live], g["reward"][live, t])
        assert np.array_equal(rew_n[live], g["reward_n"][live, t])
        assert np.array_equal(done[live], g["done"][live, t])
        assert np.array_equal(env.collisions[live], g["collisions"][live, t])
        assert np.array_equal(env.reached[live], g["reached"][live, t])


@pytest.mark.parametrize("name", ["particle_stage1_uniform", "particle_antipodal_uniform",
                                  "particle_cross_uniform", "particle_merge8_uniform"])
def test_reset_replays_reference_rng_streams(name):
    """reset() consumes random.random / np.random in the reference's order
    (multi-goal_spread.py:75-91) and actions are drawn as train_onpolicy.py:307 does, so a run
    seeded like train_onpolicy.py:38-39 reproduces the recorded episodes end to end."""
    g = load_golden(name)
    m = g["meta"]
    N = m["n_agents"]
    py_rng = random.Random(m["seed"])
    np_rng = np.random.RandomState(m["seed"])
    env = ParticleEnvOracle(N, m["config"], m["prob_random"], m["max_steps"])
    env.reset(py_rng, np_rng)               # make_world() performs one reset_world (mgs.py:62)
    for ep in range(len(g["ep_len"])):
        gs, oo, os_, done = env.reset(py_rng, np_rng)
        assert np.array_equal(gs, g["init_gs"][ep])
        assert np.array_equal(env.landmarks, g["landmarks"][ep])
        assert done == g["init_done"][ep]
        for t in range(g["ep_len"][ep]):
            acts = np_rng.randint(0, 5, N)
            assert np.array_equal(acts, g["actions"][ep, t])
            gs, oo, os_, rew, rew_n, done = env.step(acts)
            assert np.array_equal(gs, g["gs"][ep, t])
        assert done


def test_kat_p1_values():
    """Known-answer values quoted in SURVEY.md §8(c) KAT-P1."""
    g = load_golden("particle_kat_headon")
    assert g["gs"][0, 0].tolist() == [[0.5, 0, -0.15000000000000002, 0], [-0.5, 0, 0.15000000000000002, 0]]
    assert g["reward"][0, 0] == -2.1
    assert g["collisions"][0, :4].tolist() == [0, 2, 4, 4]
    assert g["gs"][0, 1, 0, 0] == 0.8680685281944008
    assert g["reward"][0, 3] == -2.2443594226267303


def test_float32_oracle_close_to_float64():
    """The f32 instantiation (what the kernel computes in) stays within 1e-5 of f64 per tick when
    the f64 state is re-injected every tick."""
    g = load_golden("particle_antipodal_greedy")
    m = g["meta"]
    Ep = len(g["ep_len"])
    e32 = VecParticleOracle(4, m["config"], 0.2, 33, Ep, dtype=np.float32)
    worst = 0.0
    prev = g["init_gs"]
    for t in range(33):
        e32.set_from_global_state(prev.astype(np.float32), g["landmarks"].astype(np.float32))
        gs, *_ = e32.step(g["actions"][:, t])
        worst = max(worst, np.abs(gs.astype(np.float64) - g["gs"][:, t]).max())
        prev = g["gs"][:, t]
    assert worst < 1e-5
