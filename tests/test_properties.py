"""Size-independent invariants: hypothesis-driven on the CPU oracles, and at the full
BASELINE sizes on the GPU kernels."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle.checkers_oracle import VecCheckersOracle
from oracle.particle_oracle import VecParticleOracle
from tests.helpers import load_cfg


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(2, 6))
def test_oracle_contact_forces_conserve_momentum(seed, n):
    """Pair forces are +f / -f (core.py:194-195): with zero actions the total momentum only decays by damping."""
    rng = np.random.default_rng(seed)
    cfg = dict(agents_x=[0] * n, agents_y=[0] * n, landmarks_x=[0] * n, landmarks_y=[0] * n, initial_std=0)
    env = VecParticleOracle(n, cfg, 0.0, 33, 16)
    pos = rng.uniform(-0.4, 0.4, (16, n, 2))
    vel = rng.normal(0, 0.5, (16, n, 2))
    env.set_state(pos, vel, rng.uniform(-1, 1, (16, n, 2)))
    before = env.vel.sum(axis=1)
    env.step(np.zeros((16, n), int))
    after = env.vel.sum(axis=1)
    assert np.allclose(after, 0.75 * before, rtol=0, atol=1e-9 * (1 + np.abs(env.vel).max()))


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2 ** 31 - 1))
def test_oracle_checkers_invariants(seed):
    rng = np.random.default_rng(seed)
    env = VecCheckersOracle(3, 8, 2, [0, 2], [8, 8], 2, 33, 32)
    env.reset(np.tile([0, 1], (32, 1)))
    prev = env.mask.copy()
    for t in range(33):
        _, vec, _, obs_t, _, total, local, done = env.step(rng.integers(-1, 7, (32, 2)))
        assert np.all((env.mask & prev) == prev)                         # collected cells never revert
        pop = np.array([bin(int(m)).count("1") for m in env.mask])
        assert np.array_equal(pop, env.count.sum(axis=(1, 2)))           # every pick-up is counted once
        assert np.all(~env.wall[env.loc[..., 0], env.loc[..., 1]])       # nobody stands in a wall
        assert np.all((env.loc[:, 0] != env.loc[:, 1]).any(axis=1))      # the two agents never share a cell
        assert set(np.unique(local)) <= {0.0, 1.0, -0.5, -0.1}
        assert np.array_equal(done, (env.steps == 33) | (pop == 24))
        prev = env.mask.copy()


@pytest.mark.gpu
def test_gpu_checkers_invariants_full_size():
    from cm3_amd.checkers import VecCheckersEnv
    cfg = load_cfg("checkers_stage2.json")
    E = 8192
    env = VecCheckersEnv(cfg["init"], 2, 33, E, device="cuda:0", seed=1)
    env.reset(np.eye(2))
    prev = env.get_state()["mask"]
    wall = torch.zeros(7, 13, dtype=torch.bool, device="cuda")
    wall[:, :2] = True
    wall[:2] = True
    wall[5:] = True
    wall[2:5, 11:] = True
    for t in range(33):
        (grid, vec), _, ot, _, total, local, done = env.step()
        st_ = env.get_state()
        assert bool(((st_["mask"] & prev) == prev).all())
        pop = torch.tensor([bin(int(m) & (2 ** 64 - 1)).count("1") for m in st_["mask"].cpu()], device="cuda")
        assert torch.equal(pop, (st_["n_green"] + st_["n_orange"]).sum(1).long())
        assert not bool(wall[st_["r"].long(), st_["c"].long()].any())
        assert bool(((st_["r"][:, 0] != st_["r"][:, 1]) | (st_["c"][:, 0] != st_["c"][:, 1])).all())
        assert torch.equal((grid == 1).sum(dim=(1, 2, 3)), pop)           # +1 cells of the grid = collected cells
        assert torch.equal(total, local.sum(1))
        prev = st_["mask"]
    assert bool(done.all())


@pytest.mark.gpu
def test_gpu_particle_momentum_and_idempotent_observe():
    from cm3_amd.particle import VecParticleEnv
    cfg = load_cfg("particle_stage2_antipodal.json")
    E, N = 4096, 4
    for kernel in ("env", "pair", "agent"):
        env = VecParticleEnv(cfg, N, 0.2, 33, E, device="cuda:0", dtype=torch.float64, kernel=kernel)
        rng = np.random.default_rng(0)
        pos = rng.uniform(-0.4, 0.4, (E, N, 2))
        vel = rng.normal(0, 0.5, (E, N, 2))
        gs0, oo0 = env.set_state(pos, vel, rng.uniform(-1, 1, (E, N, 2)))
        oo0 = oo0.clone()
        _, oo1 = env.set_state(pos, vel, env.goals)                      # observe is idempotent
        assert torch.equal(oo0, oo1)
        before = gs0[..., 0:2].sum(1).clone()
        gs, *_ = env.step(torch.zeros(E, N, dtype=torch.int32))
        after = gs[..., 0:2].sum(1)
        assert torch.allclose(after, 0.75 * before, rtol=0, atol=1e-8)


def test_sample_distinct_is_sampling_without_replacement():
    """cm3_amd.rollout.sample_distinct (the minibatch sampler of the on-policy cadence, replay_buffer.py:28-37 random.sample): every
    row distinct and in range on both of its paths (permutation for small n, first distinct draws for n >> size), marginals uniform."""
    import torch
    from cm3_amd.rollout import sample_distinct
    g = torch.Generator().manual_seed(3)
    for n, size, rows in ((1_000_000, 128, 24), (64 * 128, 128, 300), (500, 128, 7), (129, 128, 5)):
        o = sample_distinct(n, size, rows, g, "cpu")
        assert o.shape == (rows, size) and o.dtype == torch.int64
        assert int(o.min()) >= 0 and int(o.max()) < n
        assert all(len(set(r.tolist())) == size for r in o)
    n, size = 64 * 8, 8                                 # the rejection path at its smallest n / size ratio
    cnt = torch.zeros(n)
    draws = 3000
    for _ in range(draws):
        cnt += torch.bincount(sample_distinct(n, size, 8, g, "cpu").reshape(-1), minlength=n)
    expected = draws * 8 * size / n
    chi2 = float(((cnt - expected) ** 2 / expected).sum())
    assert abs(chi2 - (n - 1)) < 6 * (2 * (n - 1)) ** 0.5, chi2          # chi-square with n - 1 degrees of freedom, 6 sigma
    # first position uniform as well (the ORDER of a sample is random too)
    firsts = torch.cat([sample_distinct(n, size, 64, g, "cpu")[:, 0] for _ in range(200)])
    c0 = torch.bincount(firsts, minlength=n).float()
    e0 = firsts.numel() / n
    assert abs(float(((c0 - e0) ** 2 / e0).sum()) - (n - 1)) < 6 * (2 * (n - 1)) ** 0.5
