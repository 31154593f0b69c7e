"""GPU parity: the HIP Checkers kernels against the reference's golden vectors and the NumPy oracle.
Integer grid world => BIT-EXACT: every output of every tick must equal the reference's float64 value."""
import numpy as np
import pytest
import torch

from oracle import philox
from oracle.checkers_oracle import VecCheckersOracle
from tests.helpers import golden_names, load_cfg, load_golden

pytestmark = pytest.mark.gpu
NAMES = golden_names("checkers_")


def _env(cfg, E, max_steps=33, **kw):
    from cm3_amd.checkers import VecCheckersEnv
    return VecCheckersEnv(cfg["init"], cfg["n_agents"], max_steps, E, device="cuda:0", **kw)


def _f64(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _check_obs(got, want, sel=slice(None)):
    (grid, vec), oo, ot, ov = got
    w_grid, w_vec, w_oo, w_ot, w_ov = want
    assert np.array_equal(_f64(grid)[sel], w_grid[sel])
    assert np.array_equal(_f64(vec)[sel], w_vec[sel])
    assert np.array_equal(_f64(oo)[sel], w_oo[sel])
    assert np.array_equal(_f64(ot)[sel], w_ot[sel])
    assert np.array_equal(_f64(ov)[sel], w_ov[sel])


@pytest.mark.parametrize("padded", [True, False])     # fast multi-lane kernel / generic lane-per-env kernel
@pytest.mark.parametrize("name", NAMES)
def test_bit_exact_vs_reference_golden(name, padded):
    g = load_golden(name)
    m = g["meta"]
    Ep, T = len(g["ep_len"]), int(g["ep_len"].max())
    env = _env(m["config"], Ep, padded_records=padded)
    out = env.reset(torch.as_tensor(g["goals"]))
    _check_obs(out[:4], (g["init_grid"], g["init_vec"], g["init_obs_others"], g["init_obs_self_t"],
                         g["init_obs_self_v"]))
    assert not bool(out[4].any())
    for t in range(T):
        live = g["ep_len"] > t
        acts = np.where(live[:, None], g["actions"][:, t], 0)
        gs, oo, ot, ov, total, local, done = env.step(torch.as_tensor(acts))
        _check_obs((gs, oo, ot, ov), (g["grid"][:, t], g["vec"][:, t], g["obs_others"][:, t],
                                      g["obs_self_t"][:, t], g["obs_self_v"][:, t]), live)
        assert np.array_equal(_f64(total)[live], g["reward"][live, t]), (name, t)
        assert np.array_equal(_f64(local)[live], g["local_rewards"][live, t])
        assert np.array_equal(done.cpu().numpy()[live], g["done"][live, t])


@pytest.mark.parametrize("cfg_name,E,lo,hi", [("checkers_stage2.json", 8192, 0, 5), ("checkers_stage2.json", 1000, -1, 7),
                                              ("checkers_stage1.json", 4096 + 13, 0, 5), ("checkers_stage1.json", 1, 0, 5),
                                              ("checkers_stage2.json", 10007, 0, 5)])   # 313 workgroups: XCD 256-tiles, ragged
@pytest.mark.parametrize("padded", [True, False])
def test_bit_exact_vs_oracle_full_episodes(cfg_name, E, lo, hi, padded):
    """BASELINE C3 (N=2, E=8192) and ragged sizes (32 envs per workgroup: 1000 -> plain block order, 4109 / 8192 -> eighths, 10007 ->
    tiles of 256 with idle logical blocks; csrc/common.h): 33 free-running ticks, 100 % of ticks compared."""
    cfg = load_cfg(cfg_name)
    N = cfg["n_agents"]
    i = cfg["init"]
    rng = np.random.default_rng(E)
    goal_idx = rng.integers(0, 2, (E, N)) if N == 1 else np.tile(np.arange(N) % 2, (E, 1))
    orc = VecCheckersOracle(i["n_rows"], i["n_columns"], i["n_obs"], i["agents_r"], i["agents_c"], N, 33, E)
    want = orc.reset(goal_idx)
    env = _env(cfg, E, padded_records=padded)
    out = env.reset(goal_index=torch.as_tensor(goal_idx))
    _check_obs(out[:4], want)
    for t in range(33):
        acts = rng.integers(lo, hi, (E, N))
        # bias toward "left" so that episodes collect many cells
        acts = np.where(rng.random((E, N)) < 0.3, 3, acts)
        w = orc.step(acts)
        gs, oo, ot, ov, total, local, done = env.step(torch.as_tensor(acts))
        _check_obs((gs, oo, ot, ov), w[:5])
        assert np.array_equal(_f64(total), w[5])
        assert np.array_equal(_f64(local), w[6])
        assert np.array_equal(done.cpu().numpy(), w[7])
    assert bool(done.all())


def test_generic_geometry_three_agents():
    """Not a reference config: 5x6 band, n_obs=1, 3 agents -- exercises the runtime geometry."""
    init = dict(n_rows=5, n_columns=6, n_obs=1, agents_r=[0, 2, 4], agents_c=[6, 6, 6])
    cfg = dict(n_agents=3, init=init)
    E, N = 257, 3
    rng = np.random.default_rng(0)
    goal_idx = rng.integers(0, 2, (E, N))
    orc = VecCheckersOracle(5, 6, 1, init["agents_r"], init["agents_c"], N, 40, E)
    want = orc.reset(goal_idx)
    env = _env(cfg, E, max_steps=40)
    out = env.reset(goal_index=torch.as_tensor(goal_idx))
    _check_obs(out[:4], want)
    for t in range(40):
        acts = rng.integers(0, 5, (E, N))
        w = orc.step(acts)
        gs, oo, ot, ov, total, local, done = env.step(torch.as_tensor(acts))
        _check_obs((gs, oo, ot, ov), w[:5])
        assert np.array_equal(_f64(total), w[5]) and np.array_equal(_f64(local), w[6])
        assert np.array_equal(done.cpu().numpy(), w[7])


@pytest.mark.parametrize("case", range(12))
def test_generic_geometry_fuzz(case):
    """Random geometries the reference's constructor accepts (odd rows, even columns, rows x columns <= 64, n_obs 1..3,
    1..4 agents on distinct start cells; 4-byte padded records on or off) through the generic kernel: 25 ticks of random
    (also out-of-range) actions, every output bit-exact against the oracle."""
    rng = np.random.default_rng(1000 + case)
    R = int(rng.choice([1, 3, 5, 7]))
    C = int(rng.choice([c for c in (2, 4, 6, 8, 10, 12) if R * c <= 64]))
    O = int(rng.integers(1, 4))          # n_obs 0 leaves the world without its wall border: the reference indexes out of range
    N = int(rng.integers(1, min(4, R) + 1))
    if N == 1 and R < 3:
        N, R = 1, 3                      # the single-agent start rule puts the agent on row 0 or 2 (checkers.py:271-276)
        C = min(C, 12)
    rows = rng.permutation(R)[:N]
    init = dict(n_rows=R, n_columns=C, n_obs=O, agents_r=[int(r) for r in rows], agents_c=[C] * N)   # start column, distinct rows
    cfg = dict(n_agents=N, init=init)
    E = int(rng.integers(1, 300))
    goal_idx = rng.integers(0, 2, (E, N))
    orc = VecCheckersOracle(R, C, O, init["agents_r"], init["agents_c"], N, 30, E)
    want = orc.reset(goal_idx)
    env = _env(cfg, E, max_steps=30, padded_records=bool(case % 2) and (R, C, O) == (3, 8, 2))
    out = env.reset(goal_index=torch.as_tensor(goal_idx))
    _check_obs(out[:4], want)
    for t in range(25):
        acts = rng.integers(-1, 7, (E, N))
        w = orc.step(acts)
        gs, oo, ot, ov, total, local, done = env.step(torch.as_tensor(acts))
        _check_obs((gs, oo, ot, ov), w[:5])
        assert np.array_equal(_f64(total), w[5]) and np.array_equal(_f64(local), w[6])
        assert np.array_equal(done.cpu().numpy(), w[7])


def test_generated_actions_and_auto_reset():
    cfg = load_cfg("checkers_stage2.json")
    E, N, seed = 640, 2, 17
    env = _env(cfg, E, max_steps=6, seed=seed, auto_reset=True)
    ref = _env(cfg, E, max_steps=6, seed=seed)
    env.reset(np.eye(2))
    fresh = [x.clone() if torch.is_tensor(x) else tuple(y.clone() for y in x) for x in env.get_obs()]
    ref.reset(np.eye(2))
    for t in range(6):
        a = env.step()
        want = philox.expected_actions(seed, np.arange(E), 1, t, N, checkers=True)      # reset() made this episode 1
        assert np.array_equal(env.last_actions.cpu().numpy(), want)
        b = ref.step(env.last_actions)
        assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5]) and torch.equal(a[6], b[6])
    assert bool(a[6].all())
    # after the terminal tick the auto-reset env shows the fresh episode again
    (grid, vec), oo, ot, ov = env.get_obs()
    assert torch.equal(grid, fresh[0][0]) and torch.equal(vec, fresh[0][1]) and torch.equal(ot, fresh[2])
    assert int(env.steps.max()) == 0


def test_state_roundtrip_and_bad_configs():
    from cm3_amd import Cm3Error
    cfg = load_cfg("checkers_stage2.json")
    env = _env(cfg, 64)
    env.reset(np.eye(2))
    for _ in range(7):
        env.step()
    st = env.get_state()
    obs = env.get_obs()
    keep = [obs[0][0].clone(), obs[0][1].clone(), obs[1].clone(), obs[2].clone(), obs[3].clone()]
    env.reset(np.eye(2))
    got = env.set_state(st["mask"], st["r"], st["c"], st["n_green"], st["n_orange"], st["steps"], st["goals"])
    assert torch.equal(got[0][0], keep[0]) and torch.equal(got[0][1], keep[1]) and torch.equal(got[2], keep[3])
    with pytest.raises(Cm3Error):
        bad = dict(n_agents=2, init=dict(n_rows=3, n_columns=8, n_obs=2, agents_r=[0, 0], agents_c=[8, 8]))
        _env(bad, 4).reset(np.eye(2))
    with pytest.raises(Cm3Error):
        bad = dict(n_agents=2, init=dict(n_rows=4, n_columns=8, n_obs=2, agents_r=[0, 2], agents_c=[8, 8]))
        _env(bad, 4)


def test_single_agent_auto_reset_draws_goal_and_start_row():
    """N = 1: every fresh episode draws a goal (train_onpolicy.py:288-291) and starts on row 0 (green) or 2 (orange)
    accordingly (checkers.py:271-276)."""
    cfg = load_cfg("checkers_stage1.json")
    E = 4096
    env = _env(cfg, E, max_steps=4, seed=3, auto_reset=True)
    env.reset(goal_index=torch.zeros(E, 1, dtype=torch.int64))
    seen = []
    for t in range(12):
        out = env.step()
        done = out[6]
        if bool(done.all()):                                   # all envs restart together every 4 ticks
            st_ = env.get_state()
            g = st_["goals"][:, 0].long()
            assert torch.equal(st_["r"][:, 0].long(), torch.where(g == 0, 2, 4))     # expanded rows 0+2 / 2+2
            assert int(st_["steps"].max()) == 0 and int(st_["mask"].abs().max()) == 0
            seen.append(g.float().mean().item())
            want = philox.reset_words(3, np.arange(E), int(env._episode[0]), 0)[0] & 1
            assert np.array_equal(g.cpu().numpy(), want.astype(np.int64))
    assert len(seen) == 3 and all(0.4 < m < 0.6 for m in seen)
