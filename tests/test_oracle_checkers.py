"""CPU: the Checkers oracle (oracle/checkers_oracle.py) against golden vectors recorded from the
reference's own env/checkers.py (tests/golden/checkers_*.npz).  Bit-exact (value equality in
float64)."""
import numpy as np
import pytest

from oracle.checkers_oracle import CheckersEnvOracle, VecCheckersOracle
from tests.helpers import golden_names, load_golden

NAMES = golden_names("checkers_")


def _mk(m, cls=CheckersEnvOracle, **kw):
    i = m["config"]["init"]
    return cls(i["n_rows"], i["n_columns"], i["n_obs"], i["agents_r"], i["agents_c"],
               m["n_agents"], m["max_steps"], **kw)


def test_fixtures_present():
    assert len(NAMES) >= 5


@pytest.mark.parametrize("name", NAMES)
def test_dense_oracle_matches_reference(name):
    g = load_golden(name)
    m = g["meta"]
    env = _mk(m)
    for ep in range(len(g["ep_len"])):
        gs, oo, ot, ov, done = env.reset(g["goals"][ep])
        assert np.array_equal(gs[0], g["init_grid"][ep])
        assert np.array_equal(np.array(gs[1]), g["init_vec"][ep])
        assert np.array_equal(np.array(oo), g["init_obs_others"][ep])
        assert np.array_equal(np.array(ot), g["init_obs_self_t"][ep])
        assert np.array_equal(np.array(ov), g["init_obs_self_v"][ep])
        assert done is False
        for t in range(g["ep_len"][ep]):
            gs, oo, ot, ov, total, local, done = env.step(g["actions"][ep, t])
            assert np.array_equal(gs[0], g["grid"][ep, t]), (name, ep, t)
            assert np.array_equal(np.array(gs[1]), g["vec"][ep, t])
            assert np.array_equal(np.array(oo), g["obs_others"][ep, t])
            assert np.array_equal(np.array(ot), g["obs_self_t"][ep, t])
            assert np.array_equal(np.array(ov), g["obs_self_v"][ep, t])
            assert total == g["reward"][ep, t]
            assert np.array_equal(np.array(local, dtype=float), g["local_rewards"][ep, t])
            assert done == g["done"][ep, t]


@pytest.mark.parametrize("name", NAMES)
def test_compact_vector_oracle_matches_reference(name):
    g = load_golden(name)
    m = g["meta"]
    Ep = len(g["ep_len"])
    env = _mk(m, VecCheckersOracle, n_envs=Ep)
    grid, vec, oo, ot, ov = env.reset(g["goals"])
    assert np.array_equal(grid, g["init_grid"])
    assert np.array_equal(vec, g["init_vec"])
    assert np.array_equal(oo, g["init_obs_others"])
    assert np.array_equal(ot, g["init_obs_self_t"])
    assert np.array_equal(ov, g["init_obs_self_v"])
    for t in range(int(g["ep_len"].max())):
        live = g["ep_len"] > t
        acts = np.where(live[:, None], g["actions"][:, t], 0)
        grid, vec, oo, ot, ov, total, local, done = env.step(acts)
        assert np.array_equal(grid[live], g["grid"][live, t]), (name, t)
        assert np.array_equal(vec[live], g["vec"][live, t])
        assert np.array_equal(oo[live], g["obs_others"][live, t])
        assert np.array_equal(ot[live], g["obs_self_t"][live, t])
        assert np.array_equal(ov[live], g["obs_self_v"][live, t])
        assert np.array_equal(total[live], g["reward"][live, t])
        assert np.array_equal(local[live], g["local_rewards"][live, t])
        assert np.array_equal(done[live], g["done"][live, t])


def test_kat_c1_values():
    """Known-answer values quoted in SURVEY.md §8(c) KAT-C1 (stage 2, goals = eye(2))."""
    env = CheckersEnvOracle(3, 8, 2, [0, 2], [8, 8], 2, 33)
    gs, oo, ot, ov, done = env.reset(np.eye(2))
    assert np.array(gs[1]).tolist() == [[2, 10, 0, 0], [4, 10, 0, 0]]
    assert np.allclose(np.array(oo), [[0.07142857, 0.26923077], [-0.21428571, 0.26923077]], atol=1e-8)
    for acts, total, local, vec in [
            ([4, 1], -0.1, [-0.1, 0], [[2, 10, 0, 0], [3, 10, 0, 0]]),
            ([3, 4], -0.6, [-0.5, -0.1], [[2, 9, 0, 1], [3, 10, 0, 0]]),
            ([3, 3], 0.5, [1.0, -0.5], [[2, 8, 1, 1], [3, 9, 1, 0]])]:
        gs, oo, ot, ov, tot, loc, done = env.step(acts)
        assert tot == total and list(loc) == local and np.array(gs[1]).tolist() == vec


def test_sweep_collects_everything_and_ends_early():
    g = load_golden("checkers_stage2_script")
    # episode 3: agent 0 sweeps all 24 cells -> done on tick 24 (index 23)
    assert g["ep_len"][3] == 24 and g["done"][3, 23]
    assert g["vec"][3, 23, 0, 2:].tolist() == [12.0, 12.0]


def test_masked_restart_equals_reset():
    """VecCheckersOracle.reset_envs (the outer loop restarting an episode, train_onpolicy.py:281-294): restarting every env
    is reset(); restarting some leaves the others untouched."""
    from oracle.checkers_oracle import VecCheckersOracle
    rng = np.random.default_rng(3)
    for N, goals in ((2, np.eye(2)), (1, None)):
        E = 40
        a = VecCheckersOracle(3, 8, 2, [0, 2], [8, 8], N, 33, E)
        b = VecCheckersOracle(3, 8, 2, [0, 2], [8, 8], N, 33, E)
        g0 = goals if goals is not None else rng.integers(0, 2, (E, 1))
        a.reset(g0)
        b.reset(g0)
        for _ in range(7):
            acts = rng.integers(0, 5, (E, N))
            a.step(acts)
            b.step(acts)
        sel = rng.random(E) < 0.5
        new_goal = rng.integers(0, 2, (E, N)) if N == 1 else None
        out = a.reset_envs(sel, new_goal)
        assert np.array_equal(a.steps[sel], np.zeros(sel.sum())) and np.array_equal(a.steps[~sel], b.steps[~sel])
        assert np.array_equal(a.mask[~sel], b.mask[~sel]) and not a.mask[sel].any()
        fresh = VecCheckersOracle(3, 8, 2, [0, 2], [8, 8], N, 33, E)
        want = fresh.reset(a.goal)
        for x, y, z in zip(out, want, b.outputs()):
            assert np.array_equal(x[sel], y[sel]) and np.array_equal(x[~sel], z[~sel])
