"""GPU: the SIZE-GATED instantiations of the particle step kernels.

`kernel="auto"` (cm3::launch_n, csrc/particle.hip) chooses a build of one template from (N, E): the mapping (lane per pair /
per agent / two lanes per agent / per env), waves per workgroup, the observation store policy (plain / non-temporal /
write-through), the max-ILP translation unit, the early-store variant.  Every crossover of that table is exercised here at one
size just below and one just above it, and the test ASSERTS WHICH BUILD RAN (cm3_last_kernel_variant, ABI 5) against an
independent restatement of the table -- a retune that routes a size to another build fails here until the restatement, and
with it the list of sizes, follows.

(a) bitwise: 12 free-running ticks, float32, in-kernel actions, auto-reset with terminal capture, against the ONE-WAVE
    lane-per-env build `k_particle_step<f32,N,waves=1>` -- run over shards of <= 131 072 envs with `env_id_base` offsets where
    the batch is larger (every random stream is keyed by the global env id, so the shards ARE the big batch).
(b) directly against the float64 oracle (oracle/particle_oracle.py, pinned to the reference's own outputs): one teacher-forced
    tick of environment.py:81-123 at 1e-5 on the builds behind the large-batch roofline figures, and reset() / observe() of the
    four-wave builds against the Philox specification.
(c) the same gates seen from the collector: ParticleRollout at sizes that select the streaming stores of the large builds.
"""
import numpy as np
import pytest
import torch

from oracle import philox
from oracle.particle_oracle import VecParticleOracle
from tests.helpers import load_cfg

gpu = pytest.mark.gpu       # (the table-scan test below needs no GPU and also runs in the CPU suite)

CFG = {2: "particle_stage2_merge.json", 3: "particle_merge8.json", 4: "particle_stage2_cross.json", 5: "particle_merge8.json",
       6: "particle_merge8.json", 7: "particle_merge8.json", 8: "particle_merge8.json", 9: "particle_ring10.json",
       10: "particle_ring10.json"}
ONE_WAVE_MAX = 128 * 1024          # launch_n: lane-per-env runs one wave per workgroup up to here, four above
WT_MIN = 3 << 20                   # kWtMinObsBytes
ILP_MAX_WAVES = 16384              # kIlpMaxWaves


def _pow2ceil(v):
    r = 1
    while r < v:
        r <<= 1
    return r


def expected_variant(N, E, forced=None):
    """Restatement of cm3::launch_n / launch_pairs / launch_agents / launch_one for a float32 in-place step launch of E envs
    (n_ticks = 1, no streaming flag): the fields of cm3_last_kernel_variant() that the choice determines."""
    NO = max(N - 1, 1)
    obs_bytes = E * N * NO * 16
    pair_max = {2: 32768, 3: 24576, 4: 12288, 9: 0, 10: 0}.get(N, 16384)      # (N > 8: no pair mapping -- N (N - 1) lanes exceed a wave)
    agent_lo = {4: 12289, 5: 10240, 6: 8192, 7: 6144, 8: 4096, 9: 1024, 10: 1024}.get(N, 1 << 62)
    agent_hi = {4: 40960, 5: 40960, 6: 1572864, 7: 786432, 8: 786432, 9: 786432, 10: 786432}.get(N, 0)
    pairs = N >= 2 and E <= pair_max
    agents = N >= 4 and agent_lo <= E <= agent_hi
    if forced == "env":
        pairs = agents = False
    if agents:
        epw = 64 // _pow2ceil(N)
        waves = (E + epw - 1) // epw
        tu = "ilp" if waves <= ILP_MAX_WAVES else "default"
        wg = 1 if waves < 256 else 4
        wt = obs_bytes >= WT_MIN
        if N == 8 and E <= 32768:
            return dict(kernel="k_particle_step_agents2", waves=wg, sp="wt" if wt else "plain", early=int(wt and E <= 16384), tu=tu)
        return dict(kernel="k_particle_step_agents", waves=wg, sp="wt" if wt else "plain", early=0, tu=tu)
    if pairs:
        la = {3: 2, 4: 4, 5: 4}.get(N, N - 1)
        epw = 64 // _pow2ceil(N * la)
        waves = (E + epw - 1) // epw
        return dict(kernel="k_particle_step_pairs", waves=1 if waves < 256 else 4, sp="plain", early=0,
                    tu="ilp" if waves <= ILP_MAX_WAVES else "default")
    return dict(kernel="k_particle_step", waves=1 if E <= ONE_WAVE_MAX else 4, sp="wt" if obs_bytes >= WT_MIN else "plain",
                early=0, tu="default")


def _variant():
    from cm3_amd import _lib
    s = _lib.last_kernel_variant()
    name, _, rest = s.partition("<")
    f = dict(kv.split("=") for kv in rest.rstrip(">").split(",")[1:])
    return dict(kernel=name, real=rest.split(",")[0], n=int(f["N"]), waves=int(f["waves"]), fused=int(f["fused"]), sp=f["sp"],
                live=int(f["live"]), early=int(f["early"]), tu=f["tu"], raw=s)


def _check_variant(N, E, forced=None, **also):
    got, want = _variant(), dict(expected_variant(N, E, forced), n=N, real="f32", **also)
    for k, v in want.items():
        assert got[k] == v, "E=%d N=%d: ran %s, the dispatch table says %s" % (E, N, got["raw"], want)


def _env(N, E, **kw):
    from cm3_amd.particle import VecParticleEnv
    kw.setdefault("max_steps", 5)
    kw.setdefault("auto_reset", True)
    kw.setdefault("seed", 5)
    return VecParticleEnv(load_cfg(CFG[N]), N, 0.2, kw.pop("max_steps"), E, device="cuda:0", dtype=kw.pop("dtype", torch.float32), **kw)


class _Shards(object):
    """E envs as shards of <= ONE_WAVE_MAX envs each, every shard stepped by the one-wave lane-per-env build."""

    def __init__(self, N, E, **kw):
        self.N = N
        self.bases = list(range(0, E, ONE_WAVE_MAX))
        self.envs = [_env(N, min(ONE_WAVE_MAX, E - b), env_id_base=b, kernel="env", **kw) for b in self.bases]
        for e in self.envs:
            e.enable_terminal_capture()
            e.reset()

    def step(self):
        outs = [e.step() for e in self.envs]
        v = _variant()             # the reference build really is the one-wave lane-per-env kernel
        assert v["kernel"] == "k_particle_step" and v["waves"] == 1 and v["n"] == self.N, v["raw"]
        return [torch.cat([o[k] for o in outs], dim=0) for k in range(6)]

    def cat(self, attr, dim=0):
        return torch.cat([getattr(e, attr) for e in self.envs], dim=dim)


# (N, crossover): sizes E = crossover and crossover + 1 are both run (the table's bounds are inclusive on the low side)
CROSSOVERS = [
    (2, 8160), (2, 32768), (2, 98303), (2, ONE_WAVE_MAX),
    (3, 2040), (3, 24576), (3, 32767), (3, ONE_WAVE_MAX),
    (4, 1020), (4, 12288), (4, 16383), (4, 40960), (4, ONE_WAVE_MAX),
    (5, 510), (5, 10239), (5, 40960), (5, ONE_WAVE_MAX),
    (6, 510), (6, 8191), (6, ONE_WAVE_MAX), (6, 1572864),
    (7, 255), (7, 6143), (7, ONE_WAVE_MAX), (7, 786432),
    (8, 255), (8, 4095), (8, 16384), (8, 32768), (8, ONE_WAVE_MAX), (8, 786432),
    # N = 9, 10: lane per env -> lane per agent (16 lanes per env); plain -> write-through rows at 3 MiB; max-ILP -> default unit at
    # 16 384 waves of 4 envs; -> lane per env
    (9, 1023), (9, 2730), (9, 65536), (9, 786432),
    (10, 1023), (10, 2184), (10, 65536), (10, 786432),
]


def test_the_crossover_list_covers_every_change_of_the_table():
    """The sizes below are where expected_variant() changes -- checked by scanning the table itself, so an edit of the
    restatement without an edit of CROSSOVERS fails here (no GPU work)."""
    for N in range(2, 11):
        marks = sorted(c for n, c in CROSSOVERS if n == N)
        probe = sorted(set([1, 2, 3] + [m + d for m in marks for d in (-1, 0, 1, 2)] +
                           [1 << k for k in range(3, 22)] + [3 << k for k in range(3, 20)] + [2 ** 21 + 1]))
        changes = [a for a, b in zip(probe, probe[1:]) if expected_variant(N, a) != expected_variant(N, b) and b == a + 1]
        coarse = [(a, b) for a, b in zip(probe, probe[1:]) if expected_variant(N, a) != expected_variant(N, b) and b != a + 1]
        assert not coarse, (N, coarse)                 # every change lies between two ADJACENT probed sizes ...
        assert changes == marks, (N, changes, marks)   # ... and is a listed crossover


@gpu
@pytest.mark.parametrize("N,cross", CROSSOVERS)
@pytest.mark.parametrize("side", [0, 1])
def test_auto_dispatch_is_bit_identical_to_the_one_wave_build_at_every_crossover(N, cross, side):
    E = cross + side
    assert expected_variant(N, cross) != expected_variant(N, cross + 1)
    ticks = 12 if E <= 300000 else 7
    ref = _Shards(N, E)
    env = _env(N, E, kernel="auto")
    env.enable_terminal_capture()
    env.reset()
    assert torch.equal(env.global_state, ref.cat("global_state"))
    for t in range(ticks):
        rb = env.step()
        _check_variant(N, E, fused=0, live=0)
        ra = ref.step()
        for k, (x, y) in enumerate(zip(ra, rb)):
            assert torch.equal(x, y), (t, k)
        assert torch.equal(ref.cat("last_actions"), env.last_actions), t
        assert torch.equal(ref.cat("collisions"), env.collisions) and torch.equal(ref.cat("steps"), env.steps), t
        assert torch.equal(ref.cat("episode"), env.episode), t
        assert torch.equal(ref.cat("collisions_after_last_step"), env.collisions_after_last_step), t
    assert torch.equal(ref.cat("terminal_state"), env.terminal_state)
    assert torch.equal(ref.cat("terminal_obs_others"), env.terminal_obs_others)
    assert torch.equal(ref.cat("goals"), env.goals)
    assert int(env.episode.min()) >= (3 if ticks == 12 else 2)          # every env went through resets: none was skipped


def _random_states(rng, E, N, crowd=0.5):
    pos = rng.uniform(-1, 1, (E, N, 2))
    pos[rng.random(E) < crowd] *= 0.25
    vel = rng.normal(0, 0.7, (E, N, 2))
    lm = rng.uniform(-1, 1, (E, N, 2))
    near = rng.random((E, N)) < 0.1
    lm[near] = pos[near] + rng.normal(0, 0.03, (int(near.sum()), 2))
    return pos, vel, lm


@gpu
@pytest.mark.parametrize("N,E,forced", [
    (4, 200003, None),       # k_particle_step<f32,4,waves=4,wt>: every N = 4 sweep point >= 2^18 of bench.py
    (4, 1048576, None),
    (4, 20011, "env"),       # the one-wave lane-per-env build with write-through stores (kernel="env" from 16 384 envs)
    (8, 40009, None),        # k_particle_step_agents<f32,8,waves=4,wt>, max-ILP unit
    (8, 150001, None),       # ... default unit
    (8, 1000003, None),      # lane-per-env above kAgentHi
    (6, 300007, None), (7, 200003, None),
    (2, 1 << 20, None), (3, 500009, None), (5, 262147, None),
    (10, 100003, None), (9, 20011, None), (10, 900001, None),
])
def test_large_batch_builds_vs_f64_oracle_teacher_forced(N, E, forced):
    """One tick of environment.py:81-123 from injected random (crowded) float32-representable states with actions in -1..6,
    `kernel` as a user gets it, against the float64 oracle at the north-star's 1e-5; rewards / done / collision counts exact
    outside 2e-6 of the two thresholds (those samples are counted and bounded)."""
    TOL, EDGE = 1e-5, 2e-6
    cfg = load_cfg(CFG[N])
    rng = np.random.default_rng(4000 + 100 * N + E % 97)
    env = _env(N, E, kernel=forced or "auto", auto_reset=False, max_steps=33)
    orc = VecParticleOracle(N, cfg, 0.2, 33, E)
    pos, vel, lm = (x.astype(np.float32).astype(np.float64) for x in _random_states(rng, E, N))
    steps, col = rng.integers(0, 33, E), rng.integers(0, 50, E)
    acts = rng.integers(-1, 7, (E, N))
    orc.set_state(pos, vel, lm, steps, col)
    w_gs, w_oo, _, w_rew, w_rn, w_done = orc.step(acts)
    gs0, oo0 = env.set_state(pos, vel, lm, steps, col)
    v = _variant()
    assert v["kernel"] == "k_particle_observe" and v["waves"] == (1 if E <= ONE_WAVE_MAX else 4), v["raw"]
    # observe(): rows are exact float32 differences of the injected state rows (multi-goal_spread.py:145-154)
    for i in range(N):
        for k in range(N - 1):
            j = k if k < i else k + 1
            assert torch.equal(oo0[:, i, 4 * k:4 * k + 4], gs0[:, j] - gs0[:, i]), (i, k)
    gs, oo, _, rew, rew_n, done = env.step(torch.as_tensor(acts))
    _check_variant(N, E, forced, fused=0, live=0)
    f64 = lambda x: x.detach().cpu().numpy().astype(np.float64)      # noqa: E731
    assert np.abs(f64(gs) - w_gs).max() < TOL
    assert np.abs(f64(oo) - w_oo).max() < TOL
    m_col, m_reach = orc.pair_margins()
    safe = (m_col > EDGE) & (m_reach > EDGE)
    assert int((~safe).sum()) <= max(3, E * N // 2000)
    assert np.abs(f64(rew_n)[safe] - w_rn[safe]).max() < TOL
    assert np.abs(f64(rew)[safe] - w_rew[safe]).max() < 4 * TOL
    assert np.array_equal(done.cpu().numpy()[safe], w_done[safe])
    assert np.array_equal(env.collisions.cpu().numpy()[safe], orc.collisions[safe])
    assert np.array_equal(env.steps.cpu().numpy(), orc.steps)


@gpu
@pytest.mark.parametrize("N,E", [(2, 200003), (4, 1048576 + 5), (8, 150001)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_reset_of_the_four_wave_build_matches_the_philox_spec(N, E, dtype):
    """reset() above 131 072 envs = k_particle_reset<R,N,waves=4> (multi-goal_spread.py:65-93 from the Philox reset stream):
    positions, zero velocities, landmarks, counters, the observation of the fresh state; then a masked reset of every third env."""
    cfg = load_cfg(CFG[N])
    seed = 99
    env = _env(N, E, dtype=dtype, seed=seed, auto_reset=False)
    gs, oo, _, done = env.reset()
    v = _variant()
    assert v["kernel"] == "k_particle_reset" and v["waves"] == 4 and v["n"] == N, v["raw"]
    # float32 kernel: Box-Muller in float (|rad| <= ~5.5 at 1.2 M draws, cosf / logf at ~1 ulp) + the rounding of the position
    # itself; 5e-7 holds at the 1024 envs of test_gpu_particle.py, the largest of 1.2 M x 2 draws here measured 6.4e-7
    tol = 1e-12 if dtype == torch.float64 else 1.5e-6
    ids = np.arange(E)
    pos, lm, rnd = philox.expected_reset(seed, ids, 1, cfg, N, 0.2)
    g = gs.cpu().numpy().astype(np.float64)
    assert np.abs(g[..., 2:4] - pos).max() < tol and np.all(g[..., 0:2] == 0)
    assert np.abs(env.goals.cpu().numpy().astype(np.float64) - lm).max() < (1e-15 if dtype == torch.float64 else 1e-7)
    assert 0.19 < rnd.mean() < 0.21
    assert int(env.steps.abs().max()) == 0 and int(env.collisions.abs().max()) == 0 and not bool(done.any())
    assert torch.equal(env.episode.cpu(), torch.ones(E, dtype=torch.int32))
    for i in range(N):
        for k in range(N - 1):
            j = k if k < i else k + 1
            assert torch.equal(oo[:, i, 4 * k:4 * k + 4], gs[:, j] - gs[:, i])
    mask = torch.zeros(E, dtype=torch.bool)
    mask[::3] = True
    before = env.global_state.clone()
    env.reset(mask=mask)
    m = mask.cuda()
    assert torch.equal(env.global_state[~m], before[~m])
    pos2, _, _ = philox.expected_reset(seed, ids[::3], 2, cfg, N, 0.2)
    assert np.abs(env.global_state[m].cpu().numpy().astype(np.float64)[..., 2:4] - pos2).max() < tol
    assert torch.equal(env.episode.cpu()[::3], torch.full((len(ids[::3]),), 2, dtype=torch.int32))


@gpu
@pytest.mark.parametrize("N,E,T,want", [
    (8, 40009, 5, dict(kernel="k_particle_step_agents", waves=4, sp="wt", live=0, tu="ilp")),       # 180 MB of observation slots
    (8, 150001, 3, dict(kernel="k_particle_step_agents", waves=4, sp="wt", live=0, tu="default")),
    (4, 200003, 5, dict(kernel="k_particle_step", waves=4, sp="wt", live=0)),
    (2, 300007, 16, dict(kernel="k_particle_step", waves=4, sp="wt", live=0)),
    (2, 70001, 60, dict(kernel="k_particle_step", waves=1, sp="nt", live=0)),                       # non-temporal, below the wt gate
    (4, 16384, 44, dict(kernel="k_particle_step_agents", waves=4, sp="wt", live=1, tu="ilp")),       # exactly 1 MiB of state: live
    (8, 8192, 20, dict(kernel="k_particle_step_agents2", waves=4, sp="wt", live=1, early=1)),       # C5's trajectory build
    (6, 9000, 33, dict(kernel="k_particle_step_agents", waves=4, sp="wt", live=1, tu="ilp")),
    (7, 9000, 25, dict(kernel="k_particle_step_agents", waves=4, sp="wt", live=1, tu="ilp")),
])
def test_collector_at_streaming_sizes_equals_stepwise(N, E, T, want):
    """ParticleRollout (one launch per tick, hipGraph) at sizes whose observation slots are a stream (>= 128 MB): every slot
    against the same envs advanced by env.step() through the ONE-WAVE lane-per-env build, and the build the collector's
    launches took is the one the gates (rollout.py `small`, particle_rollout's obs_store_nt, the launchers' wt) prescribe."""
    from cm3_amd.rollout import ParticleRollout
    L = 4 * (N - 1)
    assert E * N * L * 4 * T >= (128 << 20)
    env = _env(N, E, kernel="auto", max_steps=3)
    env.reset()
    ref = _Shards(N, E, max_steps=3)
    ro = ParticleRollout(env, n_ticks=T, use_graph=True)
    ro.collect(reset=False)
    got = _variant()                    # the launches were enqueued (captured) by this thread: the last one is a step launch
    for k, val in dict(want, n=N, real="f32", fused=0).items():
        assert got[k] == val, (got["raw"], want)
    assert ro._live == bool(want["live"])
    assert torch.equal(ro.state[0].permute(1, 0, 2), ref.cat("global_state"))
    n_done = 0
    for t in range(T):
        gs, oo, _, rew, rew_n, done = ref.step()
        assert torch.equal(ref.cat("last_actions"), ro.actions[t]), t
        assert torch.equal(rew, ro.reward[t]) and torch.equal(rew_n, ro.reward_n[t]) and torch.equal(done, ro.done[t].bool()), t
        assert torch.equal(gs, ro.state[t + 1].permute(1, 0, 2)), t
        assert torch.equal(oo, ro.obs_others[t + 1]), t
        assert torch.equal(ref.cat("goals"), ro.goals[t + 1].permute(1, 0, 2)), t
        assert torch.equal(ref.cat("collisions_after_last_step"), ro.collisions[t]), t
        if bool(done.any()):
            n_done += int(done.sum())
            assert torch.equal(ref.cat("terminal_obs_others")[done], ro.term_obs_others[t][done])
            assert torch.equal(ref.cat("terminal_state")[done], ro.term_state[t].permute(1, 0, 2)[done])
    assert n_done >= E
    assert torch.equal(env.global_state, ref.cat("global_state"))        # the env holds the final state
    ro.close()
