"""cm3_amd.batch (device batch reformatting) against arrays produced by the REAL reference functions
alg_credit.process_batch / process_actions / process_goals / process_global_state and the n x n repeats of
train_step (tests/golden/batch_particle.npz, recorded by oracle/gen_golden_batch.py).  Bit-exact.
Runs on CPU tensors here and on the GPU under -m gpu."""
import os

import numpy as np
import pytest
import torch

from cm3_amd import batch as B

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_particle.npz")


def _check(device):
    z = np.load(GOLD)
    cols = {k[3:]: torch.as_tensor(z[k]).to(device) for k in z.files if k.startswith("in_")}
    out = B.process_batch(cols)
    names = ("n_steps", "v_global", "obs_others", "v_local", "actions_1hot", "actions_others_1hot", "reward",
             "reward_local", "v_global_next", "obs_others_next", "v_local_next", "done", "goals")
    assert out[0] == int(z["pb_n_steps"])
    for name, got in zip(names[1:], out[1:]):
        want = z["pb_" + name]
        g = got.cpu().numpy()
        assert g.shape == want.shape, name
        assert np.array_equal(g, want), name
        assert g.dtype == want.dtype, (name, g.dtype, want.dtype)
    gs, go = B.process_goals(out[12])
    assert np.array_equal(gs.cpu().numpy(), z["goals_self"]) and np.array_equal(go.cpu().numpy(), z["goals_others"])
    one, others, state = B.process_global_state(out[1])
    assert np.array_equal(one.cpu().numpy(), z["vg_one"])
    assert np.array_equal(others.cpu().numpy(), z["vg_others"])
    assert np.array_equal(state.cpu().numpy(), z["vg_state"])
    N = 4
    assert np.array_equal(B.repeat_indexed_by_n(one, N).cpu().numpy(), z["s_n_rep"])
    assert np.array_equal(B.repeat_indexed_by_n(others, N).cpu().numpy(), z["s_others_rep"])
    assert np.array_equal(B.repeat_indexed_by_m(out[4], N).cpu().numpy(), z["actions_self_rep"])
    assert np.array_equal(B.repeat_indexed_by_m(one, N).cpu().numpy(), z["s_m_rep"])
    assert np.array_equal(B.repeat_indexed_by_n(out[7].unsqueeze(1), N).squeeze(1).cpu().numpy(), z["reward_local_rep"])


def test_batch_reformatting_matches_reference_cpu_tensors():
    _check("cpu")


@pytest.mark.gpu
def test_batch_reformatting_matches_reference_on_device():
    _check("cuda:0")


GOLD_CK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_checkers.npz")


def _check_checkers(device):
    """alg_credit_checkers.Alg.process_batch / process_goals / process_global_state (recorded from the REAL reference)."""
    z = np.load(GOLD_CK)
    cols = {k[3:]: torch.as_tensor(z[k]).to(device) for k in z.files if k.startswith("in_")}
    out = B.process_batch_checkers(cols)
    assert out[0] == int(z["pb_n_steps"]) and len(out) == len(B.CHECKERS_BATCH_NAMES)
    for name, got in zip(B.CHECKERS_BATCH_NAMES[1:], out[1:]):
        want = z["pb_" + name]
        g = got.cpu().numpy()
        assert g.shape == want.shape, (name, g.shape, want.shape)
        assert np.array_equal(g, want), name
        assert g.dtype == want.dtype, (name, g.dtype, want.dtype)
    gs, go = B.process_goals(out[17])
    assert np.array_equal(gs.cpu().numpy(), z["goals_self"]) and np.array_equal(go.cpu().numpy(), z["goals_others"])
    one, others, state = B.process_global_state(out[2])
    assert np.array_equal(one.cpu().numpy(), z["vg_one"])
    assert np.array_equal(others.cpu().numpy(), z["vg_others"])
    assert np.array_equal(state.cpu().numpy(), z["vg_state"])


def test_checkers_batch_reformatting_matches_reference_cpu_tensors():
    _check_checkers("cpu")


@pytest.mark.gpu
def test_checkers_batch_reformatting_matches_reference_on_device():
    _check_checkers("cuda:0")


@pytest.mark.gpu
def test_checkers_rollout_columns_feed_process_batch():
    """Device columns of a real CheckersRollout go through process_batch_checkers with the documented shapes/dtypes."""
    from cm3_amd.checkers import VecCheckersEnv
    from cm3_amd.rollout import CheckersRollout
    from tests.helpers import load_cfg
    cfg = load_cfg("checkers_stage2.json")
    env = VecCheckersEnv(cfg["init"], 2, 33, 64, device="cuda:0", seed=5)
    ro = CheckersRollout(env).collect(np.eye(2))
    cols = ro.as_reference_batch(numpy=False)
    out = B.process_batch_checkers(cols)
    n = out[0]
    assert n == int(ro.valid.sum())
    assert out[1].shape == (2 * n, 3, 9, 2) and out[4].shape == (2 * n, 5, 5, 3) and out[9].shape == (n,)
    assert out[6].dtype == torch.int64 and out[8].dtype == torch.float64 and out[16].dtype == torch.bool
    assert torch.equal(out[7].argmax(1), cols["actions"].reshape(-1).long())


@pytest.mark.parametrize("tag", ["particle_n4", "particle_n1", "checkers_n2", "checkers_n1"])
def test_train_step_feeds_equal_the_real_train_step(tag):
    """cm3_amd.batch.train_step_feeds against every feed_dict the REAL reference train_step built (alg_credit.py:558-800 /
    alg_credit_checkers.py:536-780, incl. the n x n credit repeats and the n x n x l_action counterfactual tiling of
    :730-751), recorded by oracle/gen_golden_trainstep.py from the reference code under a recording session.  The session's
    return values are pseudo-random stand-ins for the networks: this pins the data movement and the TD-target arithmetic,
    not any network."""
    import json
    from cm3_amd import batch as BR
    from tests.helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "trainstep_%s.npz" % tag))
    meta = json.loads(str(z["index"]))
    cols = {k[3:]: torch.as_tensor(z[k]) for k in z.files if k.startswith("in_")}
    it = iter(range(len(meta["calls"])))

    def run(ops, feed):
        c = next(it)
        entry = meta["calls"][c]
        assert ops == entry["ops"], (c, ops, entry["ops"])
        return [torch.as_tensor(z["c%d_res_%s" % (c, op)]) if op in entry["results"] else None for op in ops]

    calls = BR.train_step_feeds(cols, run, meta["gamma"], meta["epsilon"], env=meta["env"])
    assert [c[0] for c in calls] == [e["ops"] for e in meta["calls"]]
    n_arrays = 0
    for c, ((ops, feed), entry) in enumerate(zip(calls, meta["calls"])):
        assert sorted(feed) == entry["feed"], (c, ops, sorted(feed), entry["feed"])
        for k, v in feed.items():
            want = z["c%d_feed_%s" % (c, k)]
            got = v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            assert got.shape == want.shape, (c, ops, k, got.shape, want.shape)
            assert got.dtype == want.dtype, (c, ops, k, got.dtype, want.dtype)      # the reference's own NumPy dtypes
            assert np.array_equal(got, want), (c, ops, k)
            n_arrays += 1
    assert n_arrays >= 30


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["golden_n4", "rollout_n4", "rollout_n8", "rollout_n2"])
def test_device_tiling_equals_the_torch_composition(source):
    """train_step_feeds with the static feeds from two launches of cm3_rows_tile (float32 columns on the GPU: process_actions /
    process_global_state, the n x n credit repeats, the n x n x l_action counterfactual tiling) against the torch composition that
    tests above pin to the arrays the REAL reference train_step fed: every feed of every sess.run, values, shapes and dtypes."""
    import json
    from cm3_amd import batch as BR
    from tests.helpers import GOLDEN, load_cfg
    dev = "cuda:0"
    if source == "golden_n4":
        z = np.load(os.path.join(GOLDEN, "trainstep_particle_n4.npz"))
        cols = {k[3:]: torch.as_tensor(z[k]).to(dev) for k in z.files if k.startswith("in_")}
        cols = {k: (v.to(torch.float32) if v.dtype == torch.float64 else v) for k, v in cols.items()}
    else:
        from cm3_amd.particle import VecParticleEnv
        from cm3_amd.rollout import ParticleRollout
        N = int(source[-1])
        cfg = {8: "particle_merge8.json", 4: "particle_stage2_cross.json", 2: "particle_stage2_merge.json"}[N]
        env = VecParticleEnv(load_cfg(cfg), N, 0.2, 33, 64, device=dev, dtype=torch.float32, auto_reset=True, seed=9)
        env.reset()
        ro = ParticleRollout(env, n_ticks=40, use_graph=False).collect()
        cols = ro.sample_batch(333, generator=torch.Generator(device=dev).manual_seed(2), numpy=False)
        ro.close()
    N = cols["v_global"].shape[1]

    def make_run():
        g = torch.Generator(device=dev).manual_seed(5)

        def run(ops, feed):
            rows = max([v.shape[0] for v in feed.values() if torch.is_tensor(v) and v.dim() > 0] or [1])
            outs = []
            for op in ops:
                if op.endswith("_op") or op == "list_update_target_ops":
                    outs.append(None)
                elif op == "action_samples_target":
                    outs.append(torch.randint(0, 5, (cols["v_global"].shape[0] * N,), generator=g, device=dev))
                elif op == "probs":
                    outs.append(torch.rand(rows, 5, generator=g, device=dev, dtype=torch.float64))
                elif op == "Q_credit":
                    outs.append(torch.rand(rows, generator=g, device=dev, dtype=torch.float64))
                else:
                    outs.append(torch.rand(rows, generator=g, device=dev, dtype=torch.float64))
            return outs
        return run
    a = BR.train_step_feeds(cols, make_run(), 0.99, 0.1, device_tiling=True)
    b = BR.train_step_feeds(cols, make_run(), 0.99, 0.1, device_tiling=False)
    assert [c[0] for c in a] == [c[0] for c in b] and len(a) >= 9
    n = 0
    for (ops, fa), (_, fb) in zip(a, b):
        assert sorted(fa) == sorted(fb), ops
        for k in fb:
            if torch.is_tensor(fb[k]):
                assert fa[k].dtype == fb[k].dtype and fa[k].shape == fb[k].shape, (ops, k, fa[k].dtype, fb[k].dtype, fa[k].shape, fb[k].shape)
                assert torch.equal(fa[k], fb[k]), (ops, k)
                n += 1
            else:
                assert fa[k] == fb[k]
    assert n >= 40


@pytest.mark.gpu
def test_phase_feeds_equal_per_minibatch_feeds():
    """Round 6: ParticleRollout.on_policy_phase hands out, per minibatch, views of ONE export and of ONE pair of tiling launches over the
    whole phase (cm3_amd.batch.phase_static_feeds), and train_step_feeds(static=...) builds the TD targets / sampled-action one-hots with
    single launches (cm3_td_target_f64, cm3_rows_tile).  Every feed of every sess.run of every minibatch equals the torch
    composition (device_tiling=False) on the same minibatch -- which tests above pin to the arrays the REAL train_step fed."""
    from cm3_amd import batch as BR
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    from tests.helpers import load_cfg
    dev = "cuda:0"
    env = VecParticleEnv(load_cfg("particle_stage2_antipodal.json"), 4, 0.2, 33, 256, device=dev, dtype=torch.float32, auto_reset=True, seed=3)
    env.reset()
    ro = ParticleRollout(env, n_ticks=66, use_graph=False).collect()
    pairs = ro.on_policy_phase(epochs=5, batch_size=37, generator=torch.Generator(device=dev).manual_seed(4))
    assert len(pairs) == 5 and all(s is not None for _, s in pairs)

    def make_run(n_rows):
        g = torch.Generator(device=dev).manual_seed(6)

        def run(ops, feed):
            rows = max([v.shape[0] for v in feed.values() if torch.is_tensor(v) and v.dim() > 0] or [1])
            outs = []
            for op in ops:
                if op.endswith("_op") or op == "list_update_target_ops":
                    outs.append(None)
                elif op == "action_samples_target":
                    outs.append(torch.randint(0, 5, (n_rows * 4,), generator=g, device=dev))
                elif op == "probs":
                    outs.append(torch.rand(rows, 5, generator=g, device=dev, dtype=torch.float64))
                else:
                    outs.append(torch.rand(rows, generator=g, device=dev, dtype=torch.float64))
            return outs
        return run
    checked = 0
    for cols, static in pairs:
        assert cols["v_global"].shape[0] == 37
        a = BR.train_step_feeds(cols, make_run(37), 0.99, 0.1, static=static)
        b = BR.train_step_feeds(cols, make_run(37), 0.99, 0.1, device_tiling=False)
        assert [c[0] for c in a] == [c[0] for c in b]
        for (ops, fa), (_, fb) in zip(a, b):
            assert sorted(fa) == sorted(fb), ops
            for k in fb:
                if torch.is_tensor(fb[k]):
                    assert fa[k].dtype == fb[k].dtype and fa[k].shape == fb[k].shape, (ops, k)
                    assert torch.equal(fa[k], fb[k]), (ops, k)
                    checked += 1
    assert checked >= 5 * 40
    ro.close()


@pytest.mark.gpu
def test_phase_plan_replays_equal_eager_feeds():
    """OnPolicyPhasePlan: the data movement of every train_step of a phase as ONE hipGraph replay over persistent tensors.  Two
    phases (the second re-uses every tensor and every graph): each replayed minibatch's feeds equal the eager torch composition
    on that phase's columns; the session stand-in writes pseudo-random outputs that depend on the phase into buffers it keeps."""
    from cm3_amd import batch as BR
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    from tests.helpers import load_cfg
    dev = "cuda:0"
    env = VecParticleEnv(load_cfg("particle_stage2_cross.json"), 4, 0.2, 33, 128, device=dev, dtype=torch.float32, auto_reset=True, seed=8)
    env.reset()
    ro = ParticleRollout(env, n_ticks=50, use_graph=False)
    M, K, N = 4, 29, 4
    R = K * N
    bufs = {"acts": torch.zeros(R, dtype=torch.int64, device=dev), "probs": torch.zeros(R, 5, dtype=torch.float64, device=dev),
            "q": torch.zeros(R * N * 5, dtype=torch.float64, device=dev)}

    def run(ops, feed):                      # (outputs live in fixed buffers: a captured consumer reads the same addresses every replay)
        rows = max([v.shape[0] for v in feed.values() if torch.is_tensor(v) and v.dim() > 0] or [1])
        outs = []
        for op in ops:
            if op.endswith("_op") or op == "list_update_target_ops":
                outs.append(None)
            elif op == "action_samples_target":
                outs.append(bufs["acts"])
            elif op == "probs":
                outs.append(bufs["probs"])
            else:
                outs.append(bufs["q"][:rows])
        return outs

    plan = BR.OnPolicyPhasePlan(ro, run, 0.99, 0.1, epochs=M, batch_size=K)
    g = torch.Generator(device=dev).manual_seed(9)
    ptrs = None
    for phase in range(2):
        ro.collect()
        bufs["acts"].copy_(torch.randint(0, 5, (R,), generator=g, device=dev))
        bufs["probs"].copy_(torch.rand(R, 5, generator=g, device=dev, dtype=torch.float64))
        bufs["q"].copy_(torch.rand(R * N * 5, generator=g, device=dev, dtype=torch.float64))
        plan.refresh(g)
        if ptrs is None:
            ptrs = {k: v.data_ptr() for k, v in plan.cols.items()}
        assert ptrs == {k: v.data_ptr() for k, v in plan.cols.items()}          # persistent
        checked = 0
        for k in range(M):
            calls = plan.step(k)
            torch.cuda.synchronize()
            cols, _ = plan.minibatch(k)
            want = BR.train_step_feeds({n: v.clone() for n, v in cols.items()}, run, 0.99, 0.1, device_tiling=False)
            assert [c[0] for c in calls] == [c[0] for c in want]
            for (ops, fa), (_, fb) in zip(calls, want):
                for name in fb:
                    if torch.is_tensor(fb[name]):
                        assert fa[name].dtype == fb[name].dtype and fa[name].shape == fb[name].shape, (phase, k, ops, name)
                        assert torch.equal(fa[name], fb[name]), (phase, k, ops, name)
                        checked += 1
        assert checked >= M * 40
    plan.close()
    ro.close()
