"""GPU: the Checkers device actor (cm3_amd/csrc/actor_checkers.hip) against the NumPy restatement of
networks.actor_checkers + the epsilon-mixed sampling of alg_credit_checkers.py:112-113
(oracle/actor_checkers_oracle.py).  float32 tolerance: probabilities within 2e-5 (different summation order on the
matrix cores), sampled actions equal wherever the uniform is not within 1e-4 of a CDF boundary."""
import numpy as np
import pytest
import torch

from oracle import actor_checkers_oracle as AO
from tests.helpers import load_cfg

pytestmark = pytest.mark.gpu


def _env(E, stage, seed=12341, max_steps=33, **kw):
    from cm3_amd.checkers import VecCheckersEnv
    cfg = load_cfg("checkers_stage%d.json" % stage)
    return VecCheckersEnv(cfg["init"], cfg["n_agents"], max_steps, E, device="cuda:0", seed=seed, **kw), cfg["n_agents"]


def _goals(rng, E, N):
    return np.eye(2)[rng.integers(0, 2, (E, N))] if N == 1 else np.broadcast_to(np.eye(N), (E, N, 2)).copy()


def _oracle_probs(w, env, prev, eps):
    (grid, vec), oo, ot, ov = env.get_obs()
    rows = env.E * env.n
    return AO.mixed_probs(AO.actor_probs(
        w, prev.reshape(rows), ot.reshape(rows, 5, 5, 3).cpu().numpy().astype(np.float64),
        ov.reshape(rows, 4).cpu().numpy(), oo.reshape(rows, -1).cpu().numpy(),
        env.goals.reshape(rows, 2).cpu().numpy()), eps)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("stage", [1, 2])
@pytest.mark.parametrize("eps", [0.0, 0.3])
@pytest.mark.parametrize("E", [1000, 33])
def test_actor_probs_and_samples_match_oracle(stage, eps, E, precision):
    """precision "f16x3" (every layer as three float16 MFMAs over hi + lo splits) is held to the float32 bound."""
    from cm3_amd.actor import CheckersActor
    seed = 91
    rng = np.random.default_rng(stage * 7 + E)
    env, N = _env(E, stage, seed=seed)
    w = AO.init_weights(rng, N, stage=stage)
    env.reset(_goals(rng, E, N))
    for _ in range(5):
        env.step()                       # in-kernel uniform actions: agents spread out, pick cells up
    actor = CheckersActor(w, N, stage=stage, device="cuda:0", seed=seed, precision=precision)
    prev = rng.integers(0, 5, (E, N))
    actions, probs = actor.act(env, eps, actions_prev=prev, return_probs=True)
    rows = E * N
    want = _oracle_probs(w, env, prev, eps)
    got = probs.reshape(rows, 5).cpu().numpy()
    assert np.abs(got - want).max() < 2e-5
    assert np.abs(got.sum(1) - 1).max() < 1e-5
    assert np.ptp(want, axis=1).mean() > 0.05
    u = AO.policy_uniforms(seed, np.arange(E), env._episode.cpu().numpy(), env.steps.cpu().numpy(), N).reshape(rows)
    want_a = AO.sample_actions(want, u)
    safe = np.abs(np.cumsum(want, axis=1) - u[:, None]).min(axis=1) > 1e-4
    assert safe.mean() > 0.98
    assert np.array_equal(actions.reshape(rows).cpu().numpy()[safe], want_a[safe])


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_actor_on_a_large_batch_equals_the_same_rows_in_small_batches(precision):
    """32 768 envs x 2 agents = 1024 workgroups (the oracle tests run at most 32): every row must get the bits it gets when its env
    shard is evaluated alone, and a second launch must repeat the first."""
    from cm3_amd.actor import CheckersActor
    seed, eps, E, Es, stage = 23, 0.1, 32768, 512, 2
    rng = np.random.default_rng(5)
    w = AO.init_weights(rng, 2, stage=stage)
    prev_all = rng.integers(0, 5, (E, 2))

    def run(n, base):
        env, N = _env(n, stage, seed=seed, env_id_base=base)
        env.reset(_goals(np.random.default_rng(1), n, N))
        for _ in range(4):
            env.step()
        actor = CheckersActor(w, N, stage=stage, device="cuda:0", seed=seed, precision=precision)
        a, p = actor.act(env, eps, actions_prev=prev_all[base:base + n], return_probs=True)
        a2, p2 = actor.act(env, eps, actions_prev=prev_all[base:base + n], return_probs=True)
        assert torch.equal(a, a2) and torch.equal(p, p2)
        return a, p

    a_big, p_big = run(E, 0)
    for base in (0, E // 2 - Es, E - Es):
        a, p = run(Es, base)
        assert torch.equal(p, p_big[base:base + Es]), base
        assert torch.equal(a, a_big[base:base + Es]), base


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_unpadded_env_records_take_the_byte_path(precision):
    """obs_self_t records of 150 bytes (not dword-aligned) go through the generic staging loop: same probabilities."""
    from cm3_amd.actor import CheckersActor
    rng = np.random.default_rng(9)
    w = AO.init_weights(rng, 2)
    out = []
    for padded in (True, False):
        env, N = _env(500, 2, seed=4, padded_records=padded)
        assert (env.obst_stride % 4 == 0) == padded
        env.reset(np.eye(2))
        for _ in range(6):
            env.step()
        prev = np.random.default_rng(1).integers(0, 5, (500, N))
        a, pr = CheckersActor(w, N, device="cuda:0", seed=4, precision=precision).act(env, 0.05, actions_prev=prev,
                                                                                      return_probs=True)
        assert np.abs(pr.reshape(-1, 5).cpu().numpy() - _oracle_probs(w, env, prev, 0.05)).max() < 2e-5
        out.append((a, pr))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_actions_prev_none_means_zeros_and_inputs_matter():
    from cm3_amd.actor import CheckersActor
    rng = np.random.default_rng(5)
    env, N = _env(256, 2)
    env.reset(np.eye(2))
    w = AO.init_weights(rng, N)
    actor = CheckersActor(w, N, device="cuda:0")
    _, p_none = actor.act(env, 0.0, return_probs=True)
    _, p_zero = actor.act(env, 0.0, actions_prev=np.zeros((256, N), int), return_probs=True)
    _, p_four = actor.act(env, 0.0, actions_prev=np.full((256, N), 4), return_probs=True)
    assert torch.equal(p_none, p_zero)
    assert not torch.equal(p_zero, p_four)
    want = _oracle_probs(w, env, np.zeros((256, N), int), 0.0)
    assert np.abs(p_none.reshape(-1, 5).cpu().numpy() - want).max() < 2e-5


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_each_weight_tensor_reaches_the_output(precision):
    """Perturbing any one weight tensor changes the probabilities exactly as the oracle says (guards the packing of every
    layer, the Toeplitz expansion of the convolution included; the split-float16 path has its own packed copies)."""
    from cm3_amd.actor import CheckersActor
    rng = np.random.default_rng(11)
    env, N = _env(128, 2)
    env.reset(np.eye(2))
    for _ in range(4):
        env.step()
    w = AO.init_weights(rng, N)
    prev = rng.integers(0, 5, (128, N))
    for name in sorted(w):
        w2 = dict(w)
        w2[name] = (w[name] + rng.standard_normal(w[name].shape).astype(np.float32) * 0.3).astype(np.float32)
        _, probs = CheckersActor(w2, N, device="cuda:0", precision=precision).act(env, 0.1, actions_prev=prev, return_probs=True)
        want = _oracle_probs(w2, env, prev, 0.1)
        base = _oracle_probs(w, env, prev, 0.1)
        assert np.abs(want - base).max() > 1e-3, name
        assert np.abs(probs.reshape(-1, 5).cpu().numpy() - want).max() < 3e-5, name


def test_actor_rejects_bad_weights():
    from cm3_amd import Cm3Error
    from cm3_amd.actor import CheckersActor
    w = AO.init_weights(np.random.default_rng(0), 2)
    bad = dict(w)
    del bad["conv_linear/bias"]
    with pytest.raises(Cm3Error):
        CheckersActor(bad, 2, device="cuda:0")
    bad = dict(w)
    bad["W_self_h2"] = np.zeros((64, 64), np.float32)
    with pytest.raises(Cm3Error):
        CheckersActor(bad, 2, device="cuda:0")


@pytest.mark.parametrize("use_graph", [True, False])
def test_policy_rollout_on_device_equals_host_driven_policy(use_graph):
    """CheckersRollout.collect(goals, policy=actor): alternate actor/step launches (one hipGraph) == calling the actor and
    the env from the host tick by tick with actions_prev rolled forward (train_onpolicy.py:295,309-345)."""
    from cm3_amd.actor import CheckersActor
    from cm3_amd.rollout import CheckersRollout
    E, seed = 300, 17
    env_a, N = _env(E, 2, seed=seed)
    w = AO.init_weights(np.random.default_rng(2), N)
    actor = CheckersActor(w, N, device="cuda:0", seed=seed)
    ro = CheckersRollout(env_a, use_graph=use_graph)
    ro.collect(np.eye(2), policy=actor, epsilon=0.2)
    ro.collect(np.eye(2), policy=actor, epsilon=0.2)            # second replay = a fresh episode
    env_b, _ = _env(E, 2, seed=seed)
    env_b.reset(np.eye(2))
    env_b.reset(np.eye(2))
    assert torch.equal(env_b._episode, env_a._episode)
    prev = None
    for t in range(33):
        a = actor.act(env_b, 0.2, actions_prev=prev)
        assert torch.equal(a, ro.actions[t]), t
        (grid, vec), oo, ot, ov, rew, lrew, done = env_b.step(a)
        assert torch.equal(grid, ro.grid[t + 1]) and torch.equal(vec, ro.vec[t + 1])
        assert torch.equal(ot, ro.obs_self_t[t + 1]) and torch.equal(ov, ro.obs_self_v[t + 1])
        assert torch.equal(rew, ro.reward[t]) and torch.equal(lrew, ro.local_rewards[t])
        assert torch.equal(done, ro.done[t].bool())
        prev = a
    ro.close()


def test_policy_rollout_against_oracle_env_and_oracle_actor():
    """Whole policy-driven episodes against the two oracles chained on the host: oracle actor probabilities + the
    build's Philox uniforms -> actions -> VecCheckersOracle.step.  Envs whose uniform ever falls within 1e-4 of a CDF
    boundary are excluded from the exact comparison."""
    from cm3_amd.actor import CheckersActor
    from cm3_amd.rollout import CheckersRollout
    from oracle.checkers_oracle import VecCheckersOracle
    E, seed, eps = 200, 23, 0.1
    env, N = _env(E, 2, seed=seed)
    cfg = load_cfg("checkers_stage2.json")
    w = AO.init_weights(np.random.default_rng(4), N)
    actor = CheckersActor(w, N, device="cuda:0", seed=seed)
    ro = CheckersRollout(env).collect(np.eye(2), policy=actor, epsilon=eps)
    i = cfg["init"]
    orc = VecCheckersOracle(i["n_rows"], i["n_columns"], i["n_obs"], i["agents_r"], i["agents_c"], N, 33, E)
    goals = np.broadcast_to(np.eye(2), (E, 2, 2))
    grid, vec, oo, ot, ov = orc.reset(goals)
    prev = np.zeros((E, N), int)
    episode = env._episode.cpu().numpy()                       # constant over the rollout (no auto-reset)
    ok = np.ones(E, bool)
    alive = np.ones(E, bool)
    rows = E * N
    for t in range(33):
        p = AO.mixed_probs(AO.actor_probs(w, prev.reshape(rows), ot.reshape(rows, 5, 5, 3), ov.reshape(rows, 4),
                                          oo.reshape(rows, -1), goals.reshape(rows, 2)), eps)
        u = AO.policy_uniforms(seed, np.arange(E), episode, np.full(E, t), N).reshape(rows)
        a = AO.sample_actions(p, u).reshape(E, N)
        near = (np.abs(np.cumsum(p, axis=1) - u[:, None]).min(axis=1) <= 1e-4).reshape(E, N).any(1)
        ok &= ~(near & alive)
        sel = ok & alive
        assert np.array_equal(ro.actions[t].cpu().numpy()[sel], a[sel]), t
        grid, vec, oo, ot, ov, rew, lrew, done = orc.step(a)
        assert np.array_equal(ro.grid[t + 1].cpu().numpy()[sel], grid[sel])
        assert np.array_equal(ro.obs_self_t[t + 1].cpu().numpy()[sel], ot[sel])
        assert np.array_equal(ro.reward[t].cpu().numpy()[sel], rew[sel])
        assert np.array_equal(ro.done[t].cpu().numpy().astype(bool)[sel], done[sel])
        alive &= ~done
        prev = a
    assert ok.mean() > 0.9
    ro.close()


def test_batched_evaluation_matches_stepwise_sums():
    """cm3_amd.evaluate.test_checkers (evaluate.py:159-203 for E episodes at once) == stepping the env with the actor
    from the host and summing rewards until each env's done."""
    from cm3_amd.actor import CheckersActor
    from cm3_amd.evaluate import test_checkers
    E, seed = 256, 31
    env, N = _env(E, 2, seed=seed)
    w = AO.init_weights(np.random.default_rng(6), N)
    actor = CheckersActor(w, N, device="cuda:0", seed=seed)
    r_local, r_global, n, dist = test_checkers(env, actor, n_rounds=1)
    assert n == E and dist.shape == (N, 5) and abs(dist.sum() - 1) < 1e-12
    env_b, _ = _env(E, 2, seed=seed)
    env_b.reset(np.eye(2))
    tot_l = torch.zeros(E, N, dtype=torch.float64, device="cuda:0")
    tot_g = torch.zeros(E, dtype=torch.float64, device="cuda:0")
    alive = torch.ones(E, dtype=torch.bool, device="cuda:0")
    prev = None
    for t in range(33):
        a = actor.act(env_b, 0.0, actions_prev=prev)
        _, _, _, _, rew, lrew, done = env_b.step(a)
        tot_l += lrew * alive[:, None]
        tot_g += rew * alive
        alive &= ~done
        prev = a
    assert np.allclose(r_local, tot_l.mean(0).cpu().numpy(), atol=1e-12)
    assert abs(r_global - float(tot_g.mean())) < 1e-12


def test_batched_evaluation_single_agent_random_goals():
    from cm3_amd.actor import CheckersActor
    from cm3_amd.evaluate import test_checkers
    env, N = _env(128, 1, seed=3)
    actor = CheckersActor(AO.init_weights(np.random.default_rng(8), 1, stage=1), 1, stage=1, device="cuda:0", seed=3)
    g = torch.Generator(device="cuda:0").manual_seed(0)
    r_local, r_global, n, dist = test_checkers(env, actor, n_rounds=2, generator=g)
    assert n == 256 and r_local.shape == (1,) and abs(r_local[0] - r_global) < 1e-12
    assert 0 < int(env._goals.sum()) < 128               # both goals occur


@pytest.mark.parametrize("stage", [1, 2])
def test_bf16_layers_are_close_to_float32(stage):
    """Opt-in precision="bf16": the two 256x256 layers on the bf16 matrix cores.  Not the parity path -- activations and
    those weights are rounded to bf16 (relative 2^-9), so probabilities move by up to a few 1e-2; the tolerance documents it."""
    from cm3_amd.actor import CheckersActor
    rng = np.random.default_rng(21)
    env, N = _env(2000, stage, seed=8)
    env.reset(_goals(rng, 2000, N))
    for _ in range(5):
        env.step()
    w = AO.init_weights(rng, N, stage=stage)
    prev = rng.integers(0, 5, (2000, N))
    a32, p32 = CheckersActor(w, N, stage=stage, device="cuda:0", seed=8).act(env, 0.1, actions_prev=prev, return_probs=True)
    a16, p16 = CheckersActor(w, N, stage=stage, device="cuda:0", seed=8, precision="bf16").act(
        env, 0.1, actions_prev=prev, return_probs=True)
    d = (p32 - p16).abs()
    assert float(d.max()) < 0.1 and float(d.mean()) < 5e-3
    assert float(d.max()) > 0                                     # it is a different kernel
    assert (p16.sum(-1) - 1).abs().max() < 1e-5
    assert float((a32 == a16).float().mean()) > 0.97              # same uniforms, nearly the same CDFs


@pytest.mark.parametrize("tag,N,stage", [("n2_stage2", 2, 2), ("n1_stage1", 1, 1)])
def test_device_actor_matches_vectors_from_the_reference_function_body(tag, N, stage):
    """tests/golden/actor_checkers.npz: probabilities obtained by executing the reference's networks.actor_checkers (and
    convnet_1) under the NumPy TF stand-in (oracle/gen_golden_actor.py)."""
    from cm3_amd.actor import CheckersActor
    from tests.test_oracle_actor_golden import load_cases
    w, inp, want = load_cases("actor_checkers")[tag]
    rows = want.shape[0]
    E = rows // N
    dev = "cuda:0"
    stride = (75 * N + 3) // 4 * 4
    raw = torch.zeros(E, stride, dtype=torch.int8)
    raw[:, :75 * N] = torch.as_tensor(inp["obs_self_t"]).to(torch.int8).reshape(E, N * 75)
    raw = raw.to(dev)
    ov = torch.as_tensor(inp["obs_self_v"]).to(torch.float64).reshape(E, N, 4).contiguous().to(dev)
    oo = torch.as_tensor(inp["obs_others"]).to(torch.float64).reshape(E, N, -1).contiguous().to(dev)
    goals = torch.as_tensor(inp["goals"].argmax(1)).to(torch.uint8).reshape(E, N).contiguous().to(dev)
    prev = torch.as_tensor(inp["a_prev"]).to(torch.int32).reshape(E, N).contiguous().to(dev)
    steps = torch.zeros(E, dtype=torch.int32, device=dev)
    episode = torch.zeros(E, dtype=torch.int32, device=dev)
    actions = torch.empty(E, N, dtype=torch.int32, device=dev)
    probs = torch.empty(E, N, 5, dtype=torch.float32, device=dev)
    CheckersActor(w, N, stage=stage, device=dev).enqueue(E, raw, stride, ov, oo, goals, prev, steps, episode, actions,
                                                         0.0, probs)
    assert np.abs(probs.reshape(rows, 5).cpu().numpy() - want).max() < 2e-5


@pytest.mark.parametrize("stage", [1, 2])
def test_split_float16_layers_stay_in_the_float32_error_class(stage):
    """precision="f16x3" against the float64-accumulated oracle next to precision="f32": worst error of the same order (the
    parity bound 2e-5 holds for both), a different kernel, and the same sampled actions off CDF boundaries."""
    from cm3_amd.actor import CheckersActor
    rng = np.random.default_rng(33)
    E = 2000
    env, N = _env(E, stage, seed=8)
    env.reset(_goals(rng, E, N))
    for _ in range(5):
        env.step()
    w = AO.init_weights(rng, N, stage=stage)
    prev = rng.integers(0, 5, (E, N))
    want = _oracle_probs(w, env, prev, 0.0)
    out = {}
    for prec in ("f32", "f16x3"):
        a, p = CheckersActor(w, N, stage=stage, device="cuda:0", seed=8, precision=prec).act(env, 0.0, actions_prev=prev,
                                                                                            return_probs=True)
        out[prec] = (a.reshape(-1).cpu().numpy(), p.reshape(E * N, 5).cpu().numpy())
    e32, e16 = np.abs(out["f32"][1] - want).max(), np.abs(out["f16x3"][1] - want).max()
    print("worst |p - oracle|: f32 %.2e, f16x3 %.2e" % (e32, e16))
    assert e32 < 2e-5 and e16 < 2e-5
    assert np.abs(out["f32"][1] - out["f16x3"][1]).max() > 0          # it is a different kernel
    u = AO.policy_uniforms(8, np.arange(E), env._episode.cpu().numpy(), env.steps.cpu().numpy(), N).reshape(E * N)
    safe = np.abs(np.cumsum(want, axis=1) - u[:, None]).min(axis=1) > 1e-4
    assert np.array_equal(out["f32"][0][safe], out["f16x3"][0][safe])


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("w2_scale,w1_scale", [(1e-3, 100.0), (1.0, 1.0)])
def test_split_precision_small_weights_large_activations(precision, w2_scale, w1_scale):
    """ADVICE r3 (the Checkers actor keeps the UNSCALED float16 split x = hi + lo in every layer: the residual of a value below
    0.125 falls into float16's subnormals, an absolute error of ~3e-8 per factor): second-layer weights of ~1e-3 / sqrt(256)
    against first-layer activations of ~1e2 -- logits still O(1) -- must stay inside the 2e-5 bound of the float32 path.  The
    float32 oracle's own rounding is part of the budget here (both paths are held to the same bar against it)."""
    from cm3_amd.actor import CheckersActor
    seed, E, stage = 91, 600, 2
    rng = np.random.default_rng(123)
    env, N = _env(E, stage, seed=seed)
    w = AO.init_weights(rng, N, stage=stage)
    for k in ("branch_self/kernel", "branch_self/bias", "stage-2/branch_others/kernel", "stage-2/branch_others/bias"):
        w[k] = (w[k] * w1_scale).astype(np.float32)
    for k in ("W_self_h2", "stage-2/W_others_h2"):
        w[k] = (w[k] * w2_scale).astype(np.float32)
    env.reset(_goals(rng, E, N))
    for _ in range(5):
        env.step()
    actor = CheckersActor(w, N, stage=stage, device="cuda:0", seed=seed, precision=precision)
    prev = rng.integers(0, 5, (E, N))
    _, probs = actor.act(env, 0.0, actions_prev=prev, return_probs=True)
    rows = E * N
    want = _oracle_probs(w, env, prev, 0.0)
    got = probs.reshape(rows, 5).cpu().numpy()
    assert np.abs(got - want).max() < 2e-5, float(np.abs(got - want).max())
    assert np.ptp(want, axis=1).mean() > 0.02


# ---- round 6: the whole policy-driven rollout in ONE launch (csrc/policy_checkers.hip) ------------------------------------------------
_TRAJ = ("actions", "grid", "vec", "obs_others", "obs_self_t", "obs_self_v", "local_rewards", "reward", "done", "probs")
_TERM = ("term_grid", "term_vec", "term_obs_others", "term_obs_self_t", "term_obs_self_v", "goal_slots")


def _collect_twice(stage, E, seed, auto_reset, mode, max_steps=33, T=None, eps=0.15, collects=2):
    from cm3_amd.actor import CheckersActor
    from cm3_amd.rollout import CheckersRollout
    env, N = _env(E, stage, seed=seed, max_steps=max_steps, auto_reset=auto_reset)
    w = AO.init_weights(np.random.default_rng(5), N)
    actor = CheckersActor(w, N, stage=stage, device="cuda:0", seed=seed, precision="f16x3")
    assert actor.fused_rollout_ok(env)
    ro = CheckersRollout(env, n_ticks=T, policy_mode=mode, record_probs=True)
    rng = np.random.default_rng(1)
    snaps = []
    for _ in range(collects):
        ro.collect(_goals(rng, E, N), policy=actor, epsilon=eps)
        torch.cuda.synchronize()
        snap = {k: getattr(ro, k).clone() for k in _TRAJ}
        if auto_reset:
            snap.update({k: getattr(ro, k).clone() for k in _TERM})
        snap.update(mask=env._mask.clone(), agents=env._agents.clone(), steps=env._steps.clone(), episode=env._episode.clone(),
                    goals=env._goals.clone(), prev0=ro.prev0.clone())
        snaps.append(snap)
    from cm3_amd import _lib
    name = _lib.last_kernel_variant() if hasattr(_lib, "last_kernel_variant") else ""
    ro.close()
    return snaps, name


@pytest.mark.parametrize("stage,E,auto_reset,max_steps,T", [
    (2, 8192, False, 33, None),       # C3: config_checkers_stage2, 8192 envs, one reference episode per env and collect()
    (2, 1000, True, 9, 33),           # continuous collection: every env restarts three times inside the launch (terminal capture,
                                      #   fresh record, actions_prev = zeros after a restart), ragged last workgroup
    (1, 4096, False, 33, None),       # stage 1: one agent, no others branch
    (1, 777, True, 7, 20),            # ... with restarts: a fresh random goal per episode picks the start row
])
def test_checkers_policy_rollout_equals_launch_per_tick(stage, E, auto_reset, max_steps, T):
    """The Checkers twin of test_fused_policy_rollout_equals_launch_per_tick: every array of the rollout -- actions, the mixed
    probabilities they were drawn from, all observation slots, rewards, done flags, terminal captures, goal slots -- and the live
    state the launch leaves behind equal, bit for bit, what alternating cm3_actor_checkers_f32 / cm3_checkers_step launches write.
    The one-launch kernel builds the network's inputs in LDS from the env state and reads the others branch from the table
    cm3_actor_checkers_pack made: this is the test that those are the same numbers."""
    fused, kname = _collect_twice(stage, E, 31, auto_reset, "auto", max_steps, T)
    ticks, _ = _collect_twice(stage, E, 31, auto_reset, "tick", max_steps, T)
    assert "k_ck_policy_rollout" in kname or kname == ""
    for k, (a, b) in enumerate(zip(fused, ticks)):
        for name in a:
            assert torch.equal(a[name], b[name]), "collect %d: %s differs" % (k, name)
    if auto_reset:
        assert int(fused[-1]["episode"].min()) >= 2          # restarts did happen inside the launches


def test_checkers_policy_rollout_one_launch_against_the_two_oracles():
    """The one-launch rollout (split float16) against the oracle actor + oracle env chained on the host, as
    test_policy_rollout_against_oracle_env_and_oracle_actor does for the alternating float32 path: probabilities within 2e-5
    at every tick, actions and env outputs equal for envs whose uniforms stay 1e-4 clear of a CDF boundary."""
    from cm3_amd.actor import CheckersActor
    from cm3_amd.rollout import CheckersRollout
    from oracle.checkers_oracle import VecCheckersOracle
    E, seed, eps = 256, 29, 0.1
    env, N = _env(E, 2, seed=seed)
    cfg = load_cfg("checkers_stage2.json")
    w = AO.init_weights(np.random.default_rng(6), N)
    actor = CheckersActor(w, N, device="cuda:0", seed=seed, precision="f16x3")
    ro = CheckersRollout(env, record_probs=True).collect(np.eye(2), policy=actor, epsilon=eps)
    i = cfg["init"]
    orc = VecCheckersOracle(i["n_rows"], i["n_columns"], i["n_obs"], i["agents_r"], i["agents_c"], N, 33, E)
    goals = np.broadcast_to(np.eye(2), (E, 2, 2))
    grid, vec, oo, ot, ov = orc.reset(goals)
    prev = np.zeros((E, N), int)
    episode = env._episode.cpu().numpy()
    ok, alive, rows, worst = np.ones(E, bool), np.ones(E, bool), E * N, 0.0
    for t in range(33):
        p = AO.mixed_probs(AO.actor_probs(w, prev.reshape(rows), ot.reshape(rows, 5, 5, 3), ov.reshape(rows, 4),
                                          oo.reshape(rows, -1), goals.reshape(rows, 2)), eps)
        u = AO.policy_uniforms(seed, np.arange(E), episode, np.full(E, t), N).reshape(rows)
        a = AO.sample_actions(p, u).reshape(E, N)
        near = (np.abs(np.cumsum(p, axis=1) - u[:, None]).min(axis=1) <= 1e-4).reshape(E, N).any(1)
        ok &= ~(near & alive)
        sel = ok & alive
        worst = max(worst, float(np.abs(ro.probs[t].cpu().numpy().reshape(rows, 5) - p).reshape(E, N * 5)[sel].max()))
        assert np.array_equal(ro.actions[t].cpu().numpy()[sel], a[sel]), t
        grid, vec, oo, ot, ov, rew, lrew, done = orc.step(a)
        assert np.array_equal(ro.grid[t + 1].cpu().numpy()[sel], grid[sel])
        assert np.array_equal(ro.obs_self_t[t + 1].cpu().numpy()[sel], ot[sel])
        assert np.array_equal(ro.obs_others[t + 1].cpu().numpy()[sel], oo[sel])
        assert np.array_equal(ro.reward[t].cpu().numpy()[sel], rew[sel])
        assert np.array_equal(ro.done[t].cpu().numpy().astype(bool)[sel], done[sel])
        alive &= ~done
        prev = a
    assert worst < 2e-5, worst          # the stated float32 tolerance of the actor rows
    assert ok.mean() > 0.9
    ro.close()


def test_checkers_policy_rollout_short_soak():
    """tools/ck_policy_soak.py for ten seconds per stage: identical rollouts (restarts inside the launch) reproduce the first one bit for
    bit, launch after launch -- the kernel runs two waves per SIMD through float16 matrix phases next to integer / float64 code, the
    setting in which round 5 found a chip fault (profiles/r05_policy_fault.txt).  The long soak is profiles/r06_ck_policy_soak.txt."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ck_policy_soak
    for stage, E in ((2, 8192), (1, 4096)):
        r = ck_policy_soak.soak(10.0, E, stage=stage, verbose=False)
        assert r["launches"] >= 30 and r["mismatching_launches"] == 0, r
