"""GPU parity: the HIP particle kernels (through the C ABI / VecParticleEnv) against
  (a) the golden vectors recorded from the reference (tests/golden/particle_*.npz), and
  (b) the NumPy oracle on fresh random states at BASELINE sizes.

Tolerances (BASELINE.json north_star): float32 dynamics within 1e-5 PER TICK with the oracle's
float64 state injected before every tick; samples within 2e-6 of the two
discontinuities (collision radius 0.3, reach radius 0.05) are excluded from reward / done / collision
comparisons and counted (item 3).  The float64 instantiation free-runs whole episodes and must stay
within 1e-9 of the reference (only exp/log1p ulps differ); free-running float32 episodes are characterised (measured
drift table) and held to regression bounds set from it, DRIFT32_T10 / DRIFT32_RANDOM.
"""
import numpy as np
import pytest
import torch

from oracle import philox
from oracle.particle_oracle import VecParticleOracle
from tests.helpers import golden_names, load_cfg, load_golden

pytestmark = pytest.mark.gpu

TOL32 = 1e-5
EDGE = 2e-6
NAMES = golden_names("particle_")


KERNELS = ["env", "pair", "agent"]      # every step-kernel mapping must satisfy every parity test


def _env(cfg, N, E, dtype=torch.float32, prob_random=0.2, max_steps=33, **kw):
    from cm3_amd.particle import VecParticleEnv
    return VecParticleEnv(cfg, N, prob_random, max_steps, E, device="cuda:0", dtype=dtype, **kw)


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _maxabs(x):
    return float(np.abs(x).max()) if x.size else 0.0


def _margins(pos, landmarks):
    """distance of each env's post-step configuration from the two thresholds"""
    E, N, _ = pos.shape
    m_col = np.full(E, np.inf)
    for i in range(N):
        for j in range(i + 1, N):
            d = np.sqrt(((pos[:, i] - pos[:, j]) ** 2).sum(-1))
            m_col = np.minimum(m_col, np.abs(d - 0.3))
    m_reach = np.abs(np.sqrt(((pos - landmarks) ** 2).sum(-1)) - 0.05).min(axis=1)
    return m_col, m_reach


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", NAMES)
def test_f32_teacher_forced_vs_reference_golden(name, kernel):
    g = load_golden(name)
    m = g["meta"]
    if kernel == "pair" and m["n_agents"] > 8:
        pytest.skip("the lane-per-pair mapping holds N (N - 1) <= 64 lanes per env: N <= 8")
    N, Ep, T = m["n_agents"], len(g["ep_len"]), int(g["ep_len"].max())
    env = _env(m["config"], N, Ep, prob_random=m["prob_random"], kernel=kernel)
    prev_gs, prev_col = g["init_gs"], np.zeros(Ep, np.int64)
    skipped = 0
    for t in range(T):
        live = g["ep_len"] > t
        gs_in = np.where(live[:, None, None], prev_gs, 0.0)
        gs0, oo0 = env.set_state(gs_in[..., 2:4], gs_in[..., 0:2], g["landmarks"], steps=np.full(Ep, t),
                                 collisions=prev_col)
        if t == 0:
            assert _maxabs(_np(oo0) - g["init_obs_others"]) < TOL32
        acts = np.where(live[:, None], g["actions"][:, t], 0)
        gs, oo, os_, rew, rew_n, done = env.step(torch.as_tensor(acts))
        gs, oo, os_, rew, rew_n = map(_np, (gs, oo, os_, rew, rew_n))
        done = done.cpu().numpy()
        col = env.collisions.cpu().numpy()
        want = g["gs"][:, t]
        assert _maxabs(gs[live] - want[live]) < TOL32, (name, t)
        assert _maxabs(os_[live] - g["obs_self"][live, t]) < TOL32
        assert _maxabs(oo[live] - g["obs_others"][live, t]) < TOL32
        m_col, m_reach = _margins(want[..., 2:4], g["landmarks"])
        safe = live & (m_col > EDGE) & (m_reach > EDGE)
        skipped += int((live & ~safe).sum())
        assert _maxabs(rew_n[safe] - g["reward_n"][safe, t]) < TOL32
        assert _maxabs(rew[safe] - g["reward"][safe, t]) < TOL32
        assert np.array_equal(done[safe], g["done"][safe, t]), (name, t)
        assert np.array_equal(col[safe], g["collisions"][safe, t])
        prev_gs = np.where(live[:, None, None], want, prev_gs)
        prev_col = np.where(live, g["collisions"][:, t], prev_col)
    assert skipped <= 3        # near-threshold samples are rare (SURVEY: ~1 in 100k pair-steps)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", NAMES)
def test_f64_free_running_vs_reference_golden(name, kernel):
    """Whole episodes in the float64 instantiation, no re-injection: state, rewards, done, collisions."""
    g = load_golden(name)
    m = g["meta"]
    if kernel == "pair" and m["n_agents"] > 8:
        pytest.skip("the lane-per-pair mapping holds N (N - 1) <= 64 lanes per env: N <= 8")
    N, Ep, T = m["n_agents"], len(g["ep_len"]), int(g["ep_len"].max())
    env = _env(m["config"], N, Ep, dtype=torch.float64, prob_random=m["prob_random"], kernel=kernel)
    gs0 = g["init_gs"]
    env.set_state(gs0[..., 2:4], gs0[..., 0:2], g["landmarks"])
    for t in range(T):
        live = g["ep_len"] > t
        acts = np.where(live[:, None], g["actions"][:, t], 0)
        gs, oo, os_, rew, rew_n, done = env.step(torch.as_tensor(acts))
        gs, oo, rew, rew_n = map(_np, (gs, oo, rew, rew_n))
        assert _maxabs(gs[live] - g["gs"][live, t]) < 1e-9, (name, t)
        assert _maxabs(oo[live] - g["obs_others"][live, t]) < 1e-9
        assert _maxabs(rew_n[live] - g["reward_n"][live, t]) < 1e-9
        assert _maxabs(rew[live] - g["reward"][live, t]) < 1e-8
        assert np.array_equal(done.cpu().numpy()[live], g["done"][live, t])
        assert np.array_equal(env.collisions.cpu().numpy()[live], g["collisions"][live, t])
        assert np.array_equal(env.steps.cpu().numpy()[live], np.full(live.sum(), t + 1))


# Free-running float32 drift against the float64 reference goldens, MEASURED on MI355X (profiles/r02_f32_free_running_drift.txt;
# identical for the three mappings, which are bit-identical, and identical run to run):
#   first 10 ticks      every fixture <= 7.3e-6 (inside the 1e-5 parity tolerance, the 8-agent greedy fixture included)
#   whole episode (33)  random-action fixtures <= 1.8e-5 (SURVEY.md section 7.3-2 measured 5.2e-5 on its own probe);
#                       greedy fixtures (agents pressed against each other for most of the episode; contact stiffness
#                       100 / 1e-3): merge 8.0e-6, cross 1.6e-4, antipodal 7.6e-4, merge8 8.7e-2 -- the 8-agent one DIVERGES:
#                       multi-body contact is chaotic and a float32 rounding difference grows until the trajectories are
#                       no longer comparable.  Per-tick parity (teacher-forced, 1e-5) and float64 free-running parity
#                       (1e-9) hold on that same fixture, so this is the dynamics, not a kernel defect.
# History of this test's bar (all three kept visible on purpose): (1) 5e-4 for every fixture, chosen before measuring as
# "10x the survey" -- failed on antipodal_greedy and merge8_greedy; (2) two tiers 1e-4 / 5e-3 set from a first table --
# failed on merge8_greedy once its state drift was recorded; (3) below, set from the complete table: a bound at a horizon
# where the comparison is well-posed for every fixture, a whole-episode bound for the random-action fixtures only, and
# greedy whole-episode drift recorded but not bounded.  These are regression bounds on a characterised quantity, NOT a
# parity claim: parity is the per-tick teacher-forced 1e-5 test above and the float64 free-running test.
DRIFT32_T10 = 2e-5          # first 10 free-running ticks, every fixture (measured max 7.3e-6)
# Round 4: the 9- / 10-agent ring fixtures (every path crosses the centre; greedy policy: all agents pressed against each other from
# tick ~5 on) measured 3.4e-5 after 10 ticks -- the same chaotic multi-body contact as merge8_greedy, reached earlier.  Their bar
# at that horizon is set from the measurement, like the others; per-tick parity (1e-5) and float64 free-running parity hold on them.
DRIFT32_T10_RING_GREEDY = 1e-4
DRIFT32_RANDOM = 1e-4       # whole episode, random-action fixtures (measured max 1.8e-5)


def _record_drift(name, kernel, worst, worst_rew, worst10, n_safe, n_live):
    """every fixture's measured worst state / reward_n drift, appended to gpurun_out/f32_free_running_drift.txt (the table
    DESIGN.md quotes; copied to profiles/)"""
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:                                                       # scratch output only: never let it fail a test
        os.makedirs(out, exist_ok=True)
        f = open(os.path.join(out, "f32_free_running_drift.txt"), "a")
    except OSError:
        return
    with f:
        f.write("%-32s %-6s state_end %.3e  state_t10 %.3e  reward_n(safe) %.3e  compared %d of %d env-ticks\n"
                % (name, kernel, worst, worst10, worst_rew, n_safe, n_live))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", NAMES)
def test_f32_free_running_drift_vs_reference_golden(name, kernel):
    """Whole episodes in float32 with NO re-injection against the float64 reference trajectory: the contact stiffness
    (100 / 1e-3) amplifies rounding, so this is a drift characterisation with regression bounds (see DRIFT32_* above), not
    the 1e-5 parity gate.  rewards / done / collisions are compared only where the reference sits further than the bound
    from a threshold -- the recorded table says how many env-ticks that leaves (few, for the greedy fixtures).  Every
    fixture's numbers are written to gpurun_out/f32_free_running_drift.txt before anything asserts."""
    g = load_golden(name)
    m = g["meta"]
    if kernel == "pair" and m["n_agents"] > 8:
        pytest.skip("the lane-per-pair mapping holds N (N - 1) <= 64 lanes per env: N <= 8")
    N, Ep, T = m["n_agents"], len(g["ep_len"]), int(g["ep_len"].max())
    env = _env(m["config"], N, Ep, prob_random=m["prob_random"], kernel=kernel)
    gs0 = g["init_gs"]
    env.set_state(gs0[..., 2:4], gs0[..., 0:2], g["landmarks"])
    greedy = "greedy" in name            # the fixture's action policy (oracle/gen_golden.py), known before any measurement
    bound = DRIFT32_RANDOM
    worst, worst_rew, ok_env, failures = 0.0, 0.0, np.ones(Ep, bool), []
    worst10, n_live, n_safe = 0.0, 0, 0
    for t in range(T):
        live = g["ep_len"] > t
        acts = np.where(live[:, None], g["actions"][:, t], 0)
        gs, oo, os_, rew, rew_n, done = env.step(torch.as_tensor(acts))
        gs, rew_n = _np(gs), _np(rew_n)
        want = g["gs"][:, t]
        if not np.isfinite(want[live]).all():         # dist == 0 NaN episodes of a KAT: parity is covered per tick
            break
        worst = max(worst, _maxabs(gs[live] - want[live]))
        if t < 10:
            worst10 = worst
        m_col, m_reach = _margins(want[..., 2:4], g["landmarks"])
        ok_env &= ~live | ((m_col > bound) & (m_reach > bound))     # once a flip is possible the env's counters may differ
        safe = live & ok_env
        n_live += int(live.sum())
        n_safe += int(safe.sum())
        worst_rew = max(worst_rew, _maxabs(rew_n[safe] - g["reward_n"][safe, t]))
        if not np.array_equal(done.cpu().numpy()[safe], g["done"][safe, t]):
            failures.append(("done", t))
        if not np.array_equal(env.collisions.cpu().numpy()[safe], g["collisions"][safe, t]):
            failures.append(("collisions", t))
    _record_drift(name, kernel, worst, worst_rew, worst10, n_safe, n_live)
    t10 = DRIFT32_T10_RING_GREEDY if (greedy and name.startswith("particle_ring")) else DRIFT32_T10
    assert np.isfinite(worst) and worst10 < t10, (name, worst10)
    if not greedy:
        assert worst < DRIFT32_RANDOM, (name, worst)
        assert worst_rew < 2 * DRIFT32_RANDOM, (name, worst_rew)     # |d dist| <= sqrt(2) |d pos|
        assert not failures, (name, failures[:5])


@pytest.mark.parametrize("kernel", KERNELS)
def test_f32_contact_force_is_exactly_zero_from_the_skip_distance(kernel):
    """The float32 kernels skip the contact chain from dist >= 0.32 (thresholds.h, Contact<float>::kSkip) on the argument that
    the hardware soft-plus is exactly 0 there.  Checked on the device just BELOW the cut, where the chain still runs: from
    0.3170 up the computed force must already be exactly +-0 (velocities stay bit-zero), so it is for every larger distance;
    at 0.31 the force is non-zero (the test does see forces)."""
    cfg = load_cfg("particle_stage2_merge.json")
    rng = np.random.default_rng(7)
    E, N = 4096, 2
    dist = np.concatenate([np.linspace(0.3170, 0.31999, E // 2), np.linspace(0.32, 0.45, E // 2)])
    ang = rng.uniform(0, 2 * np.pi, E)
    centre = rng.uniform(-0.5, 0.5, (E, 2))
    half = 0.5 * dist[:, None] * np.stack([np.cos(ang), np.sin(ang)], -1)
    pos = np.stack([centre + half, centre - half], 1).astype(np.float32).astype(np.float64)
    d32 = np.sqrt(((pos[:, 0].astype(np.float32) - pos[:, 1].astype(np.float32)) ** 2).sum(-1, dtype=np.float32))
    keep = d32 >= np.float32(0.3168)                       # float32 rounding of the placed positions
    env = _env(cfg, N, E, kernel=kernel)
    env.set_state(pos, np.zeros((E, N, 2)), np.full((E, N, 2), 5.0))
    gs = env.step(torch.zeros(E, N, dtype=torch.int64))[0].cpu().numpy()
    assert keep.sum() > E * 0.9
    assert np.all(gs[keep][..., 0:2] == 0.0)               # velocities: exactly zero
    assert np.array_equal(gs[keep][..., 2:4], pos[keep].astype(np.float32))
    pos[:, 0] = centre + half * (0.31 / dist[:, None])
    pos[:, 1] = centre - half * (0.31 / dist[:, None])
    env.set_state(pos, np.zeros((E, N, 2)), np.full((E, N, 2), 5.0))
    gs = env.step(torch.zeros(E, N, dtype=torch.int64))[0].cpu().numpy()
    assert np.all(np.abs(gs[..., 0:2]).max(axis=(1, 2)) > 1e-8)


@pytest.mark.parametrize("cfg_name,N,bound", [("particle_stage2_merge.json", 2, 3e-6), ("particle_stage2_antipodal.json", 4, 4e-6),
                                               ("particle_merge8.json", 8, 7e-6)])
def test_f32_one_tick_error_keeps_its_margin(cfg_name, N, bound):
    """The float32 step (soft-plus on the hardware exp2 / log2 units, everything else IEEE) against the float64 oracle over
    10 x 4096 crowded random states: the worst one-tick errors measured when that soft-plus went in were state 7.6e-7 / 1.6e-6 /
    3.0e-6 and obs_others 1.6e-6 / 2.3e-6 / 4.3e-6 for N = 2 / 4 / 8 (identical to the libm build's,
    profiles/r02_f32_softplus_hw.txt).  The contract is 1e-5; this test holds the MARGIN (bounds ~1.6x the measured values), so a
    change that eats into it is seen before it reaches the contract."""
    cfg = load_cfg(cfg_name)
    rng = np.random.default_rng(99 + N)
    E = 4096
    env = _env(cfg, N, E)
    orc = VecParticleOracle(N, cfg, 0.2, 33, E)
    worst_state = worst_obs = 0.0
    for it in range(10):
        pos, vel, lm = _random_states(rng, E, N)
        pos, vel, lm = (x.astype(np.float32).astype(np.float64) for x in (pos, vel, lm))
        acts = rng.integers(0, 5, (E, N))
        orc.set_state(pos, vel, lm)
        w_gs, w_oo = orc.step(acts)[:2]
        env.set_state(pos, vel, lm)
        gs, oo = env.step(torch.as_tensor(acts))[:2]
        worst_state = max(worst_state, _maxabs(_np(gs) - w_gs))
        worst_obs = max(worst_obs, _maxabs(_np(oo) - w_oo))
    assert worst_state < bound and worst_obs < bound, (worst_state, worst_obs)


def _random_states(rng, E, N, crowd=0.5):
    """positions in [-1,1]^2 with a fraction of envs squeezed so that contacts are common"""
    pos = rng.uniform(-1, 1, (E, N, 2))
    squeeze = rng.random(E) < crowd
    pos[squeeze] *= 0.25
    vel = rng.normal(0, 0.7, (E, N, 2))
    lm = rng.uniform(-1, 1, (E, N, 2))
    near = rng.random((E, N)) < 0.1           # some agents sit near their landmark
    lm[near] = pos[near] + rng.normal(0, 0.03, (int(near.sum()), 2))
    return pos, vel, lm


@pytest.mark.parametrize("cfg_name,N,E", [("particle_stage1.json", 1, 1), ("particle_stage2_antipodal.json", 4, 4096),
                                           ("particle_stage2_merge.json", 2, 1000),
                                           ("particle_stage2_cross.json", 4, 4096 + 37),
                                           ("particle_merge8.json", 8, 8192), ("particle_merge8.json", 3, 777),
                                           ("particle_merge8.json", 5, 300), ("particle_merge8.json", 6, 129),
                                           ("particle_merge8.json", 7, 64), ("particle_ring10.json", 9, 2000),
                                           ("particle_ring10.json", 10, 4099)])
@pytest.mark.parametrize("kernel", KERNELS)
def test_f32_random_states_vs_oracle(cfg_name, N, E, kernel):
    """BASELINE configs C1/C2/C4(per-GPU)/C5(per-GPU) + ragged sizes and every agent count."""
    if kernel == "pair" and N > 8:
        pytest.skip("the lane-per-pair mapping needs N <= 8")
    cfg = load_cfg(cfg_name)
    rng = np.random.default_rng(1234 + N * 1000 + E)
    env = _env(cfg, N, E, kernel=kernel)
    orc = VecParticleOracle(N, cfg, 0.2, 33, E)
    bad = 0
    for it in range(3):
        pos, vel, lm = _random_states(rng, E, N)
        pos32, vel32, lm32 = (x.astype(np.float32).astype(np.float64) for x in (pos, vel, lm))
        steps = rng.integers(0, 33, E)
        col = rng.integers(0, 50, E)
        acts = rng.integers(-1, 7, (E, N))
        orc.set_state(pos32, vel32, lm32, steps, col)
        w_gs, w_oo, w_os, w_rew, w_rn, w_done = orc.step(acts)
        env.set_state(pos32, vel32, lm32, steps, col)
        gs, oo, os_, rew, rew_n, done = env.step(torch.as_tensor(acts))
        gs, oo, rew, rew_n = map(_np, (gs, oo, rew, rew_n))
        assert _maxabs(gs - w_gs) < TOL32
        assert _maxabs(oo - w_oo) < TOL32
        m_col, m_reach = orc.pair_margins()
        safe = (m_col > EDGE) & (m_reach > EDGE)
        bad += int((~safe).sum())
        assert _maxabs(rew_n[safe] - w_rn[safe]) < TOL32
        assert _maxabs(rew[safe] - w_rew[safe]) < TOL32
        assert np.array_equal(done.cpu().numpy()[safe], w_done[safe])
        assert np.array_equal(env.collisions.cpu().numpy()[safe], orc.collisions[safe])
        assert np.array_equal(env.steps.cpu().numpy(), orc.steps)
    assert bad <= max(2, E * 3 // 5000)


@pytest.mark.parametrize("kernel", KERNELS)
def test_f64_random_states_vs_oracle(kernel):
    cfg = load_cfg("particle_stage2_antipodal.json")
    rng = np.random.default_rng(5)
    E, N = 2048, 4
    env = _env(cfg, N, E, dtype=torch.float64, kernel=kernel)
    orc = VecParticleOracle(N, cfg, 0.2, 33, E)
    pos, vel, lm = _random_states(rng, E, N)
    acts = rng.integers(0, 5, (E, N))
    orc.set_state(pos, vel, lm)
    w_gs, w_oo, _, w_rew, w_rn, w_done = orc.step(acts)
    env.set_state(pos, vel, lm)
    gs, oo, _, rew, rew_n, done = env.step(torch.as_tensor(acts))
    assert _maxabs(_np(gs) - w_gs) < 1e-11
    assert _maxabs(_np(oo) - w_oo) < 1e-11
    assert _maxabs(_np(rew_n) - w_rn) < 1e-11
    assert np.array_equal(done.cpu().numpy(), w_done)
    assert np.array_equal(env.collisions.cpu().numpy(), orc.collisions)


@pytest.mark.parametrize("kernel", KERNELS)
def test_structural_properties_full_size(kernel):
    """Size-independent properties at the BASELINE C2 size after a 33-tick in-kernel random rollout:
    obs rows are exact differences of state rows, reward == ordered sum of reward_n, done <=> rule."""
    cfg = load_cfg("particle_stage2_antipodal.json")
    E, N = 4096, 4
    env = _env(cfg, N, E, kernel=kernel)
    env.reset()
    for t in range(33):
        gs, oo, os_, rew, rew_n, done = env.step()
    gs, oo, rew, rew_n = gs.cpu(), oo.cpu(), rew.cpu(), rew_n.cpu()
    for i in range(N):
        for k in range(N - 1):
            j = k if k < i else k + 1
            assert torch.equal(oo[:, i, 4 * k:4 * k + 4], gs[:, j] - gs[:, i])
    acc = rew_n[:, 0].clone()
    for i in range(1, N):
        acc = acc + rew_n[:, i]
    assert torch.equal(acc, rew)
    assert bool(done.all())                      # steps == max_steps
    assert torch.equal(env.steps.cpu(), torch.full((E,), 33, dtype=torch.int32))
    assert torch.isfinite(gs).all()


def test_generated_actions_and_reset_match_philox_spec_and_are_shard_invariant():
    cfg = load_cfg("particle_stage2_merge.json")          # initial_std = 0.05: exercises Box-Muller
    E, N, seed = 1024, 2, 99
    full = _env(cfg, N, E, dtype=torch.float64, seed=seed)
    full.reset()
    pos, lm, rnd = philox.expected_reset(seed, np.arange(E), 1, cfg, N, 0.2)
    gs = _np(full.global_state)
    assert _maxabs(gs[..., 2:4] - pos) < 1e-12 and np.all(gs[..., 0:2] == 0)
    assert _maxabs(_np(full.goals) - lm) < 1e-15
    assert 0.1 < rnd.mean() < 0.3
    full.step()
    want = philox.expected_actions(seed, np.arange(E), 1, 0, N)
    assert np.array_equal(full.last_actions.cpu().numpy(), want)
    # second half of the envs as its own shard: identical results
    half = _env(cfg, N, E // 2, dtype=torch.float64, seed=seed, env_id_base=E // 2)
    half.reset()
    half.step()
    assert torch.equal(half.global_state, full.global_state[E // 2:])
    assert torch.equal(half.last_actions, full.last_actions[E // 2:])
    # float32 reset is the rounded float64 reset
    f32 = _env(cfg, N, E, dtype=torch.float32, seed=seed)
    f32.reset()
    assert _maxabs(_np(f32.global_state)[..., 2:4] - pos) < 5e-7      # float Box-Muller in the float kernel


@pytest.mark.parametrize("kernel", KERNELS)
def test_auto_reset_semantics(kernel):
    """Under AUTO_RESET a finished env returns terminal reward/done, the fresh episode's state/obs, and the
    true terminal next-state in term_state; the step counter restarts (train_onpolicy.py:282 folded in)."""
    cfg = load_cfg("particle_stage2_antipodal.json")
    E, N, seed = 512, 4, 3
    env = _env(cfg, N, E, dtype=torch.float64, seed=seed, auto_reset=True, max_steps=5, kernel=kernel)
    ref = _env(cfg, N, E, dtype=torch.float64, seed=seed, auto_reset=False, max_steps=5, kernel="env")
    env.enable_terminal_capture()
    env.reset()
    ref.reset()
    for t in range(5):
        a = env.step()
        b = ref.step(env.last_actions)
        assert torch.equal(a[3], b[3]) and torch.equal(a[5], b[5])
        if t < 4:
            assert torch.equal(a[0], b[0])
    assert bool(a[5].all())
    assert torch.equal(env.terminal_state, b[0])                 # true terminal next-state
    assert torch.equal(env.terminal_obs_others, b[1])
    pos, lm, _ = philox.expected_reset(seed, np.arange(E), 2, cfg, N, 0.2)
    assert _maxabs(_np(env.global_state)[..., 2:4] - pos) < 1e-12
    assert _maxabs(_np(env.goals) - lm) < 1e-15
    assert int(env.steps.max()) == 0 and int(env.collisions.max()) == 0
    assert torch.equal(env.episode.cpu(), torch.full((E,), 2, dtype=torch.int32))


def test_partial_reset_mask_and_out_of_range_actions():
    cfg = load_cfg("particle_stage2_cross.json")
    E, N = 300, 4
    env = _env(cfg, N, E, dtype=torch.float64)
    env.reset()
    for _ in range(3):
        env.step()
    before = env.global_state.clone()
    mask = torch.zeros(E, dtype=torch.bool)
    mask[::3] = True
    env.reset(mask=mask)
    after = env.global_state
    assert torch.equal(after[~mask.cuda()], before[~mask.cuda()])
    assert (after[mask.cuda()][..., 0:2] == 0).all()
    # actions outside 1..4 apply no force (environment.py:197-200)
    st = env.get_state()
    a5 = torch.full((E, N), 5, dtype=torch.int32)
    a0 = torch.zeros((E, N), dtype=torch.int32)
    g5 = env.step(a5)[0].clone()
    env.set_state(st["pos"], st["vel"], st["landmarks"], st["steps"], st["collisions"])
    g0 = env.step(a0)[0]
    assert torch.equal(g5, g0)


def test_bad_arguments_raise():
    from cm3_amd import Cm3Error
    cfg = load_cfg("particle_stage2_antipodal.json")
    env = _env(cfg, 4, 8)
    with pytest.raises(Cm3Error):
        env.step(torch.zeros(8, 3, dtype=torch.int32))
    with pytest.raises(Cm3Error):
        _env(cfg, 11, 8)


@pytest.mark.parametrize("N,cfg_name", [(2, "particle_stage2_merge.json"), (4, "particle_stage2_cross.json"),
                                        (8, "particle_merge8.json"), (5, "particle_merge8.json"),
                                        (9, "particle_ring10.json"), (10, "particle_ring10.json")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("other", ["pair", "agent"])
def test_both_kernel_mappings_agree_bitwise(N, cfg_name, dtype, other):
    """Same inputs, same bits: 40 free-running ticks with in-kernel actions and auto-reset (max_steps 7) through
    the lane-per-env kernel and the lane-per-pair / lane-per-agent kernels, including terminal capture."""
    if other == "pair" and N > 8:
        pytest.skip("the lane-per-pair mapping needs N <= 8")
    cfg = load_cfg(cfg_name)
    E = 1000
    a = _env(cfg, N, E, dtype=dtype, seed=21, auto_reset=True, max_steps=7, kernel="env")
    b = _env(cfg, N, E, dtype=dtype, seed=21, auto_reset=True, max_steps=7, kernel=other)
    a.enable_terminal_capture()
    b.enable_terminal_capture()
    a.reset()
    b.reset()
    for t in range(40):
        ra, rb = a.step(), b.step()
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), t
        assert torch.equal(a.last_actions, b.last_actions)
        assert torch.equal(a.collisions, b.collisions) and torch.equal(a.steps, b.steps)
        assert torch.equal(a.goals, b.goals) and torch.equal(a.episode, b.episode)
        # terminal capture of the envs that finished in this tick
        d = ra[-1]
        if bool(d.any()):
            assert torch.equal(a.terminal_state[d], b.terminal_state[d]), t
            assert torch.equal(a.terminal_obs_others[d], b.terminal_obs_others[d]), t
    assert torch.equal(a.terminal_state, b.terminal_state)
    assert torch.equal(a.terminal_obs_others, b.terminal_obs_others)


@pytest.mark.parametrize("N,cfg_name,E,other", [
    (8, "particle_merge8.json", 5003, "agent"),    # two lanes per agent: 313 workgroups -> 256-tiles, 199 idle logical blocks
    (8, "particle_merge8.json", 2051, "agent"),    # 129 workgroups -> eighths (G = 17), ragged last block
    (8, "particle_merge8.json", 3001, "pair"),     # one env per wave: 751 workgroups -> tiles, ragged
    (4, "particle_stage2_cross.json", 5121, "pair"),    # 257 workgroups: one block into the second tile
    (4, "particle_stage2_cross.json", 1270, "pair"),    # 64 workgroups: the smallest launch that takes the XCD order
    (5, "particle_merge8.json", 20011, "agent"),   # one lane per agent, 8 lanes per env: tiles, ragged
])
def test_xcd_block_order_covers_ragged_batches(N, cfg_name, E, other):
    """The XCD-aware block order (csrc/common.h: plain below 64 workgroups, eighths up to 256, tiles of 256 above; grids rounded up,
    logical blocks beyond the batch idle) must cover every env exactly once whatever the batch size: batches chosen at the edges
    of the three modes, stepped 12 ticks with in-kernel actions and auto-reset, against the lane-per-env kernel (plain order)."""
    cfg = load_cfg(cfg_name)
    a = _env(cfg, N, E, seed=5, auto_reset=True, max_steps=5, kernel="env")
    b = _env(cfg, N, E, seed=5, auto_reset=True, max_steps=5, kernel=other)
    a.enable_terminal_capture()
    b.enable_terminal_capture()
    a.reset()
    b.reset()
    for t in range(12):
        ra, rb = a.step(), b.step()
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), t
        assert torch.equal(a.last_actions, b.last_actions)
        assert torch.equal(a.collisions, b.collisions) and torch.equal(a.steps, b.steps) and torch.equal(a.episode, b.episode)
    assert torch.equal(a.terminal_state, b.terminal_state) and torch.equal(a.terminal_obs_others, b.terminal_obs_others)
    assert int(a.episode.min()) >= 2           # every env went through resets: none was skipped


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
@pytest.mark.parametrize("kernel", KERNELS)
def test_f64_free_running_random_configs_all_agent_counts(N, kernel):
    """33 free-running ticks in float64 from random crowded states, random (also out-of-range) actions, for every
    agent count and both kernel mappings: state / obs / rewards / done / collisions against the NumPy oracle."""
    if (N == 1 and kernel in ("pair", "agent")) or (N > 8 and kernel == "pair"):
        pytest.skip("the pair mapping needs 2 .. 8 agents")
    rng = np.random.default_rng(100 + N)
    cfg = dict(n_agents=N, agents_x=rng.uniform(-1, 1, N).tolist(), agents_y=rng.uniform(-1, 1, N).tolist(),
               landmarks_x=rng.uniform(-1, 1, N).tolist(), landmarks_y=rng.uniform(-1, 1, N).tolist(), initial_std=0.1)
    E = 500
    env = _env(cfg, N, E, dtype=torch.float64, kernel=kernel, max_steps=20)
    orc = VecParticleOracle(N, cfg, 0.2, 20, E)
    pos, vel, lm = _random_states(rng, E, N, crowd=0.7)
    env.set_state(pos, vel, lm)
    orc.set_state(pos, vel, lm)
    for t in range(33):
        acts = rng.integers(-1, 7, (E, N))
        w_gs, w_oo, _, w_rew, w_rn, w_done = orc.step(acts)
        gs, oo, _, rew, rew_n, done = env.step(torch.as_tensor(acts))
        scale = 1.0 + np.abs(w_gs).max()
        assert _maxabs(_np(gs) - w_gs) < 1e-9 * scale, (N, t)
        assert _maxabs(_np(oo) - w_oo) < 2e-9 * scale
        m_col, m_reach = orc.pair_margins()
        safe = (m_col > 1e-9) & (m_reach > 1e-9)
        assert safe.mean() > 0.99
        assert _maxabs(_np(rew_n)[safe] - w_rn[safe]) < 1e-9 * scale
        assert _maxabs(_np(rew)[safe] - w_rew[safe]) < 1e-8 * scale
        assert np.array_equal(done.cpu().numpy()[safe], w_done[safe])
        assert np.array_equal(env.steps.cpu().numpy(), orc.steps)


def test_step_takes_host_and_device_actions_alike():
    """VecParticleEnv.step(actions): device int32 tensors are read in place, host integer arrays go through a ring of pinned staging
    buffers (round 6: the reference's loop hands np.random.randint output to env.step every tick, train_onpolicy.py:307, :321) --
    ten ticks driven three ways leave the same state, observations and rewards; bad shapes are still refused."""
    from cm3_amd import Cm3Error
    from cm3_amd.particle import VecParticleEnv
    from tests.helpers import load_cfg
    cfg = load_cfg("particle_stage2_antipodal.json")
    rng = np.random.default_rng(3)
    acts = rng.integers(0, 5, (10, 300, 4))
    outs = []
    for how in ("device_int32", "host_int64", "host_list"):
        env = VecParticleEnv(cfg, 4, 0.2, 33, 300, device="cuda:0", seed=5, auto_reset=True)
        env.reset()
        for t in range(10):
            a = acts[t]
            if how == "device_int32":
                a = torch.as_tensor(a, dtype=torch.int32, device="cuda:0")
            elif how == "host_list":
                a = a.tolist()
            gs, oo, _, rew, rew_n, done = env.step(a)
        torch.cuda.synchronize()
        outs.append((gs.clone(), oo.clone(), rew.clone(), rew_n.clone(), done.clone(), env.last_actions.clone() if hasattr(env, "last_actions") else None))
    for o in outs[1:]:
        for x, y in zip(outs[0][:5], o[:5]):
            assert torch.equal(x, y)
    env = VecParticleEnv(cfg, 4, 0.2, 33, 300, device="cuda:0", seed=5)
    env.reset()
    with pytest.raises(Cm3Error):
        env.step(np.zeros((300, 3), np.int64))
    with pytest.raises(Cm3Error):
        env.step(torch.zeros(299, 4, dtype=torch.int32, device="cuda:0"))
