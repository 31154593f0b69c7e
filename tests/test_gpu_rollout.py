"""GPU: trajectory collection (train_onpolicy.py:281-350) -- the device trajectory equals stepping the env
tick by tick, and the exported columns follow the reference's 11-/16-column transition layout."""
import numpy as np
import pytest
import torch

from oracle.particle_oracle import VecParticleOracle
from oracle.checkers_oracle import VecCheckersOracle
from tests.helpers import load_cfg

pytestmark = pytest.mark.gpu


def _penv(E, N=4, dtype=torch.float64, cfg="particle_stage2_antipodal.json", **kw):
    from cm3_amd.particle import VecParticleEnv
    return VecParticleEnv(load_cfg(cfg), N, 0.2, kw.pop("max_steps", 33), E, device="cuda:0", dtype=dtype, **kw)


@pytest.mark.parametrize("use_graph", [False, True])
def test_particle_rollout_equals_stepwise_env_and_oracle(use_graph):
    from cm3_amd.rollout import ParticleRollout
    E, N, T = 300, 4, 33
    env = _penv(E, seed=5)
    ro = ParticleRollout(env, use_graph=use_graph).collect()
    ro.collect()                                   # second rollout (graph replay) must be a fresh episode
    # replay the same actions through a second env stepped tick by tick, and through the NumPy oracle
    ref = _penv(E, seed=5)
    ref.set_state(ro.state[0, :, :, 2:4].permute(1, 0, 2), ro.state[0, :, :, 0:2].permute(1, 0, 2), env.goals)
    orc = VecParticleOracle(N, load_cfg("particle_stage2_antipodal.json"), 0.2, 33, E)
    orc.set_from_global_state(ro.state[0].permute(1, 0, 2).cpu().numpy(), env.goals.cpu().numpy())
    for t in range(T):
        gs, oo, os_, rew, rew_n, done = ref.step(ro.actions[t])
        assert torch.equal(gs, ro.state[t + 1].permute(1, 0, 2))
        assert torch.equal(oo, ro.obs_others[t + 1])
        assert torch.equal(rew, ro.reward[t]) and torch.equal(rew_n, ro.reward_n[t])
        assert torch.equal(done, ro.done[t].bool())
        w = orc.step(ro.actions[t].cpu().numpy())
        assert np.abs(gs.cpu().numpy() - w[0]).max() < 1e-9
        assert np.array_equal(done.cpu().numpy(), w[5])
    assert torch.equal(env.global_state, ro.state[T].permute(1, 0, 2))     # live env holds the last slot
    assert bool(ro.done[T - 1].all())
    ro.close()


def test_particle_reference_batch_layout_and_valid_mask():
    from cm3_amd.rollout import ParticleRollout, PARTICLE_ORDER
    E, N, T = 64, 4, 33
    env = _penv(E, seed=11)
    ro = ParticleRollout(env, use_graph=False)
    # a policy that walks every agent toward its landmark: some envs finish early (all reached)
    def policy(obs_others, obs_self, goals):
        d = goals - obs_self[..., 2:4]
        horiz = d[..., 0].abs() > d[..., 1].abs()
        ax = torch.where(d[..., 0] > 0, 2, 1)
        ay = torch.where(d[..., 1] > 0, 4, 3)
        a = torch.where(horiz, ax, ay)
        slow = obs_self[..., 0:2].norm(dim=-1) > 0.6           # coast when fast
        return torch.where(slow, torch.zeros_like(a), a).to(torch.int32)
    ro.collect(policy=policy)
    valid = ro.valid
    d = ro.done.bool()
    first_done = torch.where(d.any(0), d.float().argmax(0), torch.full((E,), T - 1, device=d.device))
    assert torch.equal(valid.sum(0), first_done + 1)
    cols = ro.as_reference_batch()
    B = int(valid.sum())
    shapes = dict(v_global=(B, N, 4), obs_others=(B, N, 12), v_local=(B, N, 4), actions=(B, N), reward=(B,),
                  reward_local=(B, N), v_global_next=(B, N, 4), obs_others_next=(B, N, 12), v_local_next=(B, N, 4),
                  done=(B,), goals=(B, N, 2))
    assert list(cols) == list(PARTICLE_ORDER)
    for k, shp in shapes.items():
        assert cols[k].shape == shp, k
    tt, ee = ro.valid_indices()
    b = B // 2
    t, e = int(tt[b]), int(ee[b])
    assert np.array_equal(cols["v_global"][b], ro.state[t, :, e].cpu().numpy())
    assert np.array_equal(cols["v_global_next"][b], ro.state[t + 1, :, e].cpu().numpy())
    assert np.array_equal(cols["obs_others_next"][b], ro.obs_others[t + 1, e].cpu().numpy())
    assert np.array_equal(cols["goals"][b], env.goals[e].cpu().numpy())
    rows = ro.as_reference_rows(tt[:5], ee[:5])
    assert rows.shape == (5, 11) and rows.dtype == object
    assert np.array_equal(np.stack(rows[:, 1]), cols["obs_others"][:5])
    g, l = ro.episode_returns()
    assert np.allclose(g.cpu().numpy(), (ro.reward * valid).sum(0).cpu().numpy())
    batch = ro.sample_batch(128, generator=torch.Generator(device="cuda").manual_seed(0))
    assert batch["reward"].shape == (128,)
    from cm3_amd import batch as BR
    n = 0
    for mb in ro.on_policy_minibatches(epochs=24, batch_size=128, generator=torch.Generator(device="cuda").manual_seed(1)):
        out = BR.process_batch(mb)                      # device columns straight into the device reshapers
        assert out[0] == 128 and out[2].shape == (128 * N, 12) and out[4].shape == (128 * N, 5)
        n += 1
    assert n == 24


def test_particle_auto_reset_rollout_keeps_true_terminal_next_state():
    from cm3_amd.rollout import ParticleRollout
    E, N, T = 128, 4, 20
    env = _penv(E, seed=3, auto_reset=True, max_steps=7)
    env.reset()
    ro = ParticleRollout(env, n_ticks=T, use_graph=True).collect()
    ref = _penv(E, seed=3, auto_reset=False, max_steps=7)
    cols = ro.as_reference_batch(numpy=False)
    assert cols["reward"].shape == (T * E,)
    ep_start = 0
    ref.set_state(ro.state[0, :, :, 2:4].permute(1, 0, 2), ro.state[0, :, :, 0:2].permute(1, 0, 2),
                  ro.goals[0].permute(1, 0, 2))
    for t in range(T):
        gs, oo, _, rew, rew_n, done = ref.step(ro.actions[t])
        sel = torch.arange(E, device="cuda") + t * E          # time-major flat index of (t, e)
        assert torch.equal(cols["v_global_next"][sel], gs)    # true next state, also on terminal ticks
        assert torch.equal(cols["obs_others_next"][sel], oo)
        assert torch.equal(cols["reward"][sel], rew) and torch.equal(cols["done"][sel], done)
        assert torch.equal(cols["goals"][sel], ref.goals)
        if bool(done.all()):                                   # all envs end together at max_steps here
            ref.set_state(ro.state[t + 1, :, :, 2:4].permute(1, 0, 2), ro.state[t + 1, :, :, 0:2].permute(1, 0, 2),
                          ro.goals[t + 1].permute(1, 0, 2))
            assert (ro.state[t + 1, :, :, 0:2] == 0).all()     # fresh episode: zero velocity
    ro.close()


def test_checkers_rollout_matches_oracle_and_16_column_layout():
    from cm3_amd.checkers import VecCheckersEnv
    from cm3_amd.rollout import CheckersRollout, CHECKERS_ORDER
    cfg = load_cfg("checkers_stage2.json")
    i = cfg["init"]
    E, N, T = 200, 2, 33
    env = VecCheckersEnv(i, N, T, E, device="cuda:0", seed=8)
    ro = CheckersRollout(env).collect(np.eye(2))
    first = ro.actions.clone()
    ro.collect(np.eye(2))                            # hipGraph replay: a fresh episode (new episode index => new actions)
    assert not torch.equal(first, ro.actions)
    orc = VecCheckersOracle(i["n_rows"], i["n_columns"], i["n_obs"], i["agents_r"], i["agents_c"], N, T, E)
    w0 = orc.reset(np.eye(2))
    assert np.array_equal(ro.grid[0].cpu().numpy().astype(float), w0[0])
    for t in range(T):
        w = orc.step(ro.actions[t].cpu().numpy())
        assert np.array_equal(ro.grid[t + 1].cpu().numpy().astype(float), w[0])
        assert np.array_equal(ro.obs_self_t[t + 1].cpu().numpy().astype(float), w[3])
        assert np.array_equal(ro.reward[t].cpu().numpy(), w[5])
        assert np.array_equal(ro.done[t].cpu().numpy().astype(bool), w[7])
    cols = ro.as_reference_batch()
    assert list(cols) == list(CHECKERS_ORDER)
    B = cols["reward"].shape[0]
    assert cols["grid"].shape == (B, 3, 9, 2) and cols["obs_self_t"].shape == (B, N, 5, 5, 3)
    assert cols["goals"].shape == (B, N, 2) and cols["actions_prev"].shape == (B, N)
    tt, ee = ro.valid_indices()
    first = (tt == 0).cpu().numpy()
    assert (cols["actions_prev"][first] == 0).all()                       # train_onpolicy.py:295
    later = np.nonzero(~first)[0][:50]
    for b in later:
        t, e = int(tt[b]), int(ee[b])
        assert np.array_equal(cols["actions_prev"][b], ro.actions[t - 1, e].cpu().numpy())
    rows = ro.as_reference_rows(tt[:4], ee[:4])
    assert rows.shape == (4, 16)


@pytest.mark.parametrize("kernel", ["env", "pair", "agent"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_fused_rollout_equals_per_tick_rollout(kernel, dtype, auto_reset):
    """CM3_FLAG_FUSED_TICKS (all ticks in one launch, state in registers) is bit-identical to one launch per tick."""
    from cm3_amd.rollout import ParticleRollout
    E, T = 777, 40
    outs = []
    for fused in (False, True):
        env = _penv(E, dtype=dtype, cfg="particle_stage2_cross.json", seed=4, auto_reset=auto_reset, max_steps=9,
                    kernel=kernel)
        env.reset()
        ro = ParticleRollout(env, n_ticks=T, use_graph=False, fused=fused).collect(reset=False)
        outs.append((ro, env))
    a, b = outs[0][0], outs[1][0]
    for name in ("state", "obs_others", "actions", "reward", "reward_n", "done"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    if auto_reset:
        assert torch.equal(a.goals, b.goals)
        d = a.done.bool()
        assert torch.equal(a.term_state.permute(0, 2, 1, 3)[d], b.term_state.permute(0, 2, 1, 3)[d])
        assert torch.equal(a.term_obs_others[d], b.term_obs_others[d])
    ea, eb = outs[0][1], outs[1][1]
    assert torch.equal(ea.steps, eb.steps) and torch.equal(ea.collisions, eb.collisions)
    assert torch.equal(ea.episode, eb.episode) and torch.equal(ea.global_state, eb.global_state)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("C", [1, 4])
def test_returns_moments_and_normalisation_kernels(dtype, C):
    """HIP returns/moments/normalise kernels against the plain torch formulation in cm3_amd/shard.py."""
    from cm3_amd.shard import global_moments, normalize_advantages, normalized_returns, returns_to_go
    T, E = 33, 1000
    g = torch.Generator(device="cuda").manual_seed(1)
    shape = (T, E) if C == 1 else (T, E, C)
    x = torch.randn(shape, generator=g, device="cuda", dtype=dtype) * 2 - 1
    done = torch.rand(T, E, generator=g, device="cuda") < 0.05
    valid = torch.rand(T, E, generator=g, device="cuda") > 0.1
    want_ret = returns_to_go(x, done, gamma=0.97)
    vmask = valid if C == 1 else valid.unsqueeze(-1).expand_as(want_ret)
    want_ret = torch.where(vmask, want_ret, torch.zeros_like(want_ret))
    ret, (mean, std, cnt) = normalized_returns(x, done, valid, gamma=0.97, normalize=False)
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    assert torch.allclose(ret, want_ret, atol=tol, rtol=tol)
    m2, s2, n2 = global_moments(want_ret, valid)
    assert int(cnt) == int(n2) and abs(float(mean) - float(m2)) < 1e-6 and abs(float(std) - float(s2)) < 1e-6
    norm, _ = normalized_returns(x, done, valid, gamma=0.97)
    want = normalize_advantages(want_ret, valid)
    assert torch.allclose(norm, want, atol=10 * tol, rtol=10 * tol)
    # deterministic: two runs give identical bits
    again, (mean_b, std_b, _) = normalized_returns(x, done, valid, gamma=0.97)
    assert torch.equal(again, norm) and float(mean_b) == float(mean) and float(std_b) == float(std)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("E,C", [(1000, 4), (4096, 4), (70000, 8), (333, 1)])
def test_single_rank_advantage_step_equals_moments_then_normalize(dtype, E, C):
    """cm3_returns_normalize_* (returns + per-block partials [+ the slot bookkeeping cm3_copy_shift], then fold + normalisation in
    every block) against cm3_returns_moments_* + cm3_normalize_*, bit for bit, call after call; 70000 x 8 columns = more columns
    than lanes in the grid (the generic walk), the others take the one-round-trip kernel (T <= 40)."""
    import ctypes
    from cm3_amd import _lib
    from cm3_amd.shard import ReturnsNormalizer
    T = 33
    g = torch.Generator(device="cuda").manual_seed(E + C)
    x = torch.randn((T, E, C) if C > 1 else (T, E), generator=g, device="cuda", dtype=dtype) * 3 - 1
    done = (torch.rand(T, E, generator=g, device="cuda") < 0.05).to(torch.uint8)
    valid = (torch.rand(T, E, generator=g, device="cuda") > 0.1).to(torch.uint8)
    lib = _lib.lib()
    sfx = "f32" if dtype == torch.float32 else "f64"
    s = _lib.current_stream_handle(x.device)
    for v8 in (None, valid):
        for apply in (1, 0):
            two, one = ReturnsNormalizer(x, done, 0.97, 1e-8, bool(apply)), ReturnsNormalizer(x, done, 0.97, 1e-8, bool(apply))
            # slot bookkeeping: a <- m, m <- c
            m0 = torch.randn(4096, generator=g, device="cuda")
            c0 = torch.randn(4096, generator=g, device="cuda")
            a, m, c = torch.zeros_like(m0), m0.clone(), c0.clone()
            cs = _lib.CopyShift()
            cs.n = 1
            cs.first_dst[0], cs.mid[0], cs.last_src[0], cs.bytes[0] = a.data_ptr(), m.data_ptr(), c.data_ptr(), m.numel() * 4
            for rep in range(3):
                _lib.check(getattr(lib, "cm3_returns_moments_" + sfx)(
                    x.data_ptr(), done.data_ptr(), _lib.ptr(v8), two.out.data_ptr(), two.scratch.data_ptr(), two.moments.data_ptr(),
                    T, E, C, 0.97, s))
                _lib.check(getattr(lib, "cm3_normalize_" + sfx)(
                    two.out.data_ptr(), _lib.ptr(v8), two.moments.data_ptr(), 1, two.stats.data_ptr(), two.out.numel(), C, 1e-8,
                    apply, s))
                _lib.check(getattr(lib, "cm3_returns_normalize_" + sfx)(
                    x.data_ptr(), done.data_ptr(), _lib.ptr(v8), one.out.data_ptr(), one.scratch.data_ptr(), one.moments.data_ptr(),
                    one.stats.data_ptr(), T, E, C, 0.97, 1e-8, apply, ctypes.byref(cs) if rep == 0 else None, s))
                torch.cuda.synchronize()
                assert torch.equal(one.out, two.out), (rep, apply)
                assert torch.equal(one.buf, two.buf)
            assert torch.equal(a, m0) and torch.equal(m, c0) and torch.equal(c, c0)
    assert float(one.stats[2]) > 0


@pytest.mark.parametrize("cfg_name", ["checkers_stage2.json", "checkers_stage1.json"])
def test_checkers_fused_rollout_equals_per_tick(cfg_name):
    from cm3_amd.checkers import VecCheckersEnv
    from cm3_amd.rollout import CheckersRollout
    cfg = load_cfg(cfg_name)
    N, E, T = cfg["n_agents"], 333, 40
    goals = np.eye(2) if N == 2 else np.array([[0, 1]])
    outs = []
    for fused, graph in ((False, False), (False, True), (True, False)):
        env = VecCheckersEnv(cfg["init"], N, 9, E, device="cuda:0", seed=6, auto_reset=True)
        ro = CheckersRollout(env, n_ticks=T, use_graph=graph, fused=fused).collect(goals)
        outs.append((ro, env))
    a = outs[0][0]
    for ro, env in outs[1:]:
        for name in ("grid", "vec", "obs_others", "obs_self_t", "obs_self_v", "actions", "local_rewards", "reward", "done"):
            assert torch.equal(getattr(a, name), getattr(ro, name)), name
        assert torch.equal(outs[0][1].steps, env.steps) and torch.equal(outs[0][1]._mask, env._mask)
        assert torch.equal(outs[0][1]._episode, env._episode) and torch.equal(outs[0][1]._goals, env._goals)
        ro.close()
    assert int(a.done.sum()) > 0            # episodes of 9 ticks: several auto-resets inside the 40-tick rollout


@pytest.mark.parametrize("E,N,cfg_name", [(4096, 4, "particle_stage2_antipodal.json"), (1000, 4, "particle_stage2_cross.json"),
                                          (512, 8, "particle_merge8.json")])
def test_in_place_rollout_equals_stepwise(E, N, cfg_name):
    """cm3_particle_rollout_f32 over ZERO strides (every tick overwrites the live buffers -- how bench.py steps) with
    in-kernel actions and auto-reset (the same memory row holds this tick's and the next tick's actions).  70 ticks = two
    episode boundaries; must leave exactly the state, counters and last action row of 70 single env.step() launches.
    (Written in round 1 for the "draw wave" that handed action rows forward between launches; that mechanism was measured to
    be a loss in round 2 and removed -- the test keeps guarding the C-level rollout loop against env.step().)"""
    from bench import ParticleStepper
    from tests.helpers import load_cfg
    cfg = load_cfg(cfg_name)
    st = ParticleStepper(cfg, N, E, "cuda:0", seed=77)
    ref = _penv(E, N, dtype=torch.float32, cfg=cfg_name, seed=77, auto_reset=True)
    ref.reset()
    # stagger the step counters so that episodes end on different ticks in different envs (and inside one wave)
    stagger = torch.randint(0, 33, (E,), generator=torch.Generator().manual_seed(5), dtype=torch.int32).to("cuda:0")
    st.env._meta[:, 0] = stagger
    ref._meta[:, 0] = stagger
    for _ in range(2):
        st.run(35)                       # eager C rollout loop: 35 launches, flags first / middle / last
    for _ in range(70):
        ref.step()
    torch.cuda.synchronize()
    a, b = st.env, ref
    assert torch.equal(a._state[0], b._state[b._cur])
    assert torch.equal(a._obs_others[0], b._obs_others[b._cur])
    assert torch.equal(a._goals, b._goals) and torch.equal(a._meta, b._meta) and torch.equal(a._episode, b._episode)
    assert torch.equal(a._actions[0], b.last_actions)
    assert int(b._episode.min()) >= 3                # two resets happened
    st.capture(33)
    st.run(66)                                      # the same through hipGraph replays
    for _ in range(66):
        ref.step()
    torch.cuda.synchronize()
    assert torch.equal(a._state[0], b._state[b._cur]) and torch.equal(a._meta, b._meta)
    assert torch.equal(a._actions[0], b.last_actions)
    st.close()


@pytest.mark.parametrize("E,stage", [(8192, 2), (1000, 2), (777, 1)])
def test_checkers_in_place_rollout_equals_stepwise(E, stage):
    """cm3_checkers_rollout over ZERO strides with in-kernel actions and auto-reset (how bench.py steps C3): launches hand
    their own actions.  70 ticks must leave exactly the compact state,
    counters, outputs and last action row of 70 single env.step() launches."""
    from bench import CheckersStepper
    from cm3_amd.checkers import VecCheckersEnv
    cfg = load_cfg("checkers_stage%d.json" % stage)
    n = cfg["n_agents"]
    goals = np.eye(2) if n > 1 else np.array([[1, 0]])
    st = CheckersStepper(cfg, E, "cuda:0", seed=99)
    ref = VecCheckersEnv(cfg["init"], n, 33, E, device="cuda:0", seed=99, auto_reset=True)
    ref.reset(goals)
    # stagger the step counters so that episodes end on different ticks in different envs (and inside one wave)
    stagger = torch.randint(0, 33, (E,), generator=torch.Generator().manual_seed(6), dtype=torch.int32).to("cuda:0")
    st.env._steps.copy_(stagger)
    ref._steps.copy_(stagger)
    for _ in range(2):
        st.run(35)
    out = None
    for _ in range(70):
        out = ref.step()
    torch.cuda.synchronize()
    a, b = st.env, ref
    assert torch.equal(a._mask, b._mask) and torch.equal(a._agents, b._agents)
    assert torch.equal(a._steps, b._steps) and torch.equal(a._episode, b._episode) and torch.equal(a._goals, b._goals)
    sa, sb = a._slots[0], b._slots[b._cur]
    for key in ("actions", "grid_raw", "obs_self_t_raw", "vec", "obs_others", "obs_self_v", "local_rewards", "reward", "done"):
        assert torch.equal(sa[key], sb[key]), key
    assert int(b._episode.min()) >= 3
    st.capture(33)
    st.run(66)
    for _ in range(66):
        ref.step()
    torch.cuda.synchronize()
    assert torch.equal(a._mask, b._mask) and torch.equal(a._agents, b._agents) and torch.equal(a._episode, b._episode)
    assert torch.equal(a._slots[0]["actions"], b._slots[b._cur]["actions"])
    st.close()


# ---- batched evaluation against the REAL reference evaluators (alg/evaluate.py) -----------------------------------------
def _particle_eval_policy(obs_others, obs_self, goals):
    """the stand-in policy of oracle/gen_golden_evaluate.py (particle_policy), on device tensors"""
    d = goals - obs_self[..., 2:4]
    horiz = d[..., 0].abs() > d[..., 1].abs()
    a = torch.where(horiz, torch.where(d[..., 0] > 0, 2, 1), torch.where(d[..., 1] > 0, 4, 3))
    fast = (obs_self[..., 0] * obs_self[..., 0] + obs_self[..., 1] * obs_self[..., 1]) > 0.36
    return torch.where(fast, torch.zeros_like(a), a).to(torch.int32)


@pytest.mark.parametrize("tag", ["particle_n4", "particle_n1"])
def test_batched_evaluation_equals_reference_test_particle(tag):
    """cm3_amd.evaluate.test_particle == the REAL alg/evaluate.py:test_particle (:87-123), run in the build container on the
    real reference env with a deterministic stand-in for alg.run_actor (oracle/gen_golden_evaluate.py): same reset states
    (injected), same policy function, float64 -- per-agent and global return averages to 1e-9.  Pins the evaluator's
    control flow and accumulation (one episode per env, stop counting at `done`, averages over n_eval), not a network."""
    import os
    from cm3_amd import evaluate as EV
    from tests.helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "evaluate_%s.npz" % tag))
    N, n_eval = int(z["n_agents"]), int(z["n_eval"])
    env = _penv(n_eval, N, cfg=str(z["config"]) + ".json", max_steps=int(z["max_steps"]))
    gs = z["init_gs"]
    env.set_state(gs[..., 2:4], gs[..., 0:2], z["landmarks"])
    local, glob, n = EV.test_particle(env, _particle_eval_policy, reset=False)
    assert n == n_eval
    assert np.abs(local - z["reward_local_avg"]).max() < 1e-9 * np.abs(z["reward_local_avg"]).max()
    assert abs(glob - float(z["reward_global_avg"])) < 1e-9 * abs(float(z["reward_global_avg"]))


@pytest.mark.parametrize("tag", ["checkers_n2", "checkers_n1"])
def test_batched_evaluation_equals_reference_test_checkers(tag):
    """cm3_amd.evaluate.test_checkers == the REAL alg/evaluate.py:test_checkers (:159-203) with the goals it drew and an
    integer-only stand-in policy that also depends on the episode index (= env index here): exact equality of the averages
    up to the float64 summation order (1e-12).  The action histogram the reference only prints is not pinned."""
    import os
    from cm3_amd import evaluate as EV
    from cm3_amd.checkers import VecCheckersEnv
    from tests.helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "evaluate_%s.npz" % tag))
    N, n_eval = int(z["n_agents"]), int(z["n_eval"])
    cfg = load_cfg(str(z["config"]) + ".json")
    env = VecCheckersEnv(cfg["init"], N, int(z["max_steps"]), n_eval, device="cuda:0")
    episode = torch.arange(n_eval, device="cuda:0").unsqueeze(1)
    agent = torch.arange(N, device="cuda:0").unsqueeze(0)

    def policy(actions_prev, obs_others, obs_self_t, obs_self_v, goals):
        t = obs_self_t.to(torch.int64)                                            # [E, N, 5, 5, 3]
        uncollected = (t[..., 0:2] == -1).flatten(2).sum(2)
        walls = (t[..., 2] == 1).flatten(2).sum(2)
        return ((uncollected + 3 * walls + 2 * actions_prev.to(torch.int64) + agent + episode) % 5).to(torch.int32)
    local, glob, n, dist = EV.test_checkers(env, policy, goals=z["goals"].astype(np.int64))
    assert n == n_eval and abs(float(dist.sum()) - 1.0) < 1e-12
    assert np.abs(local - z["reward_local_avg"]).max() < 1e-12
    assert abs(glob - float(z["reward_global_avg"])) < 1e-12


@pytest.mark.parametrize("kernel,N,E", [("env", 4, 1000), ("pair", 4, 1000), ("agent", 4, 1000), ("agent", 8, 520), ("pair", 2, 333)])
@pytest.mark.parametrize("use_graph", [True, False])
def test_live_state_rollout_equals_slot_chained_rollout(kernel, N, E, use_graph):
    """cm3_particle_traj.state_live (ABI 4): stepping in place on the env's own state / goals buffers with a copy into every slot
    gives the trajectory, the terminal captures and the final env state of chaining the ticks through the slots, bit for bit --
    over episode ends (max_steps 7), for every mapping, two collects in a row."""
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    cfg = load_cfg("particle_merge8.json")
    envs, ros = [], []
    for live in (True, False):
        env = VecParticleEnv(cfg, N, 0.2, 7, E, device="cuda:0", dtype=torch.float32, seed=5, auto_reset=True, kernel=kernel)
        env.reset()
        envs.append(env)
        ros.append(ParticleRollout(env, n_ticks=20, use_graph=use_graph, live_state=live))
    for rep in range(2):
        a, b = (ro.collect(reset=False) for ro in ros)
        assert a._live and not b._live
        for name in ("state", "goals", "obs_others", "actions", "reward", "reward_n", "done", "term_state", "term_obs_others",
                     "collisions"):
            x, y = getattr(a, name), getattr(b, name)
            if name in ("term_state", "term_obs_others"):      # defined where an episode ended
                d = a.done.bool()
                x = x.permute(0, 2, 1, 3)[d] if name == "term_state" else x[d]
                y = y.permute(0, 2, 1, 3)[d] if name == "term_state" else y[d]
            assert torch.equal(x, y), (name, rep)
        assert bool(a.done.any())
        for attr in ("global_state", "goals", "steps", "collisions", "episode"):
            assert torch.equal(getattr(envs[0], attr), getattr(envs[1], attr)), (attr, rep)
        assert torch.equal(envs[0].get_obs()[1], envs[1].get_obs()[1])
    for ro in ros:
        ro.close()


@pytest.mark.parametrize("kernel,N,E,cfg", [("env", 4, 3001, "particle_stage2_cross.json"), ("pair", 4, 1000, "particle_stage2_antipodal.json"),
                                            ("agent", 8, 700, "particle_merge8.json"), ("agent", 5, 900, "particle_merge8.json"),
                                            ("env", 2, 5000, "particle_stage2_merge.json")])
@pytest.mark.parametrize("use_graph", [True, False])
def test_sparse_goal_slots_equal_dense_goal_slots(kernel, N, E, cfg, use_graph):
    """cm3_particle_traj.goals_live alone: the goals slot of a tick is written only for the envs that restart in it (landmarks move
    only at an episode start, multi-goal_spread.py:88-91).  Same trajectories as the dense collector; as_reference_batch gathers the
    goals from the last written slot WITHOUT completing the array; `ro.goals` completes it on first access; the env is left alike."""
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    c = load_cfg(cfg)
    T = 40
    envs = [VecParticleEnv(c, N, 0.5, 7, E, device="cuda:0", auto_reset=True, seed=31, kernel=kernel) for _ in range(2)]
    for e in envs:
        e.reset()
    a = ParticleRollout(envs[0], n_ticks=T, use_graph=use_graph, sparse_goals=False, live_state=False)
    b = ParticleRollout(envs[1], n_ticks=T, use_graph=use_graph, sparse_goals=True, live_state=False)
    for rep in range(3):
        a.collect(reset=False)
        b.collect(reset=False)
        assert b._goals_sparse and not a._goals_sparse
        tt = torch.randint(0, T, (4000,), device="cuda:0")
        ee = torch.randint(0, E, (4000,), device="cuda:0")
        ca, cb = a.as_reference_batch(tt, ee, numpy=False), b.as_reference_batch(tt, ee, numpy=False)
        assert b._goals_sparse                                   # the gather did not need the dense array
        for k in ca:
            assert torch.equal(ca[k], cb[k]), (rep, k)
        written = int((b._goals_buf[1:] != a._goals_buf[1:]).any(dim=1).any(dim=-1).sum())
        assert written > 0 or rep == 0                           # the sparse buffer really is incomplete before the fill
        assert torch.equal(a.goals, b.goals), rep                # first access completes it
        assert not b._goals_sparse
        for name in ("state", "obs_others", "actions", "reward_n", "done", "collisions"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (rep, name)
        assert torch.equal(envs[0].goals, envs[1].goals) and torch.equal(envs[0].global_state, envs[1].global_state)
    assert int(a.done.sum()) > E
    a.close()
    b.close()
