"""GPU: the NON-TEMPORAL instantiations of the step kernels.  cm3_particle_rollout_* / cm3_checkers_rollout pick them only
when a rollout's observation slots total >= 128 MB (a stream), which no other parity test reaches -- so each mapping is run
here at such a size and every slot is compared, bit for bit, with the same envs advanced tick by tick through env.step()
(always the plain kernels, re-used double buffers)."""
import numpy as np
import pytest
import torch

from tests.helpers import load_cfg

pytestmark = pytest.mark.gpu
MB128 = 128 << 20


def _penv(E, N, cfg, **kw):
    from cm3_amd.particle import VecParticleEnv
    return VecParticleEnv(load_cfg(cfg), N, 0.2, kw.pop("max_steps", 9), E, device="cuda:0", dtype=torch.float32, **kw)


@pytest.mark.parametrize("kernel,N,E,T,cfg", [("pair", 4, 4096, 180, "particle_stage2_antipodal.json"),
                                              ("agent", 8, 8192, 20, "particle_merge8.json"),
                                              ("env", 4, 4096, 180, "particle_stage2_cross.json"),
                                              ("env", 2, 65536, 140, "particle_stage2_merge.json")])
@pytest.mark.parametrize("mode", ["graph", "fused"])
def test_particle_streaming_rollout_equals_stepwise(kernel, N, E, T, cfg, mode):
    from cm3_amd.rollout import ParticleRollout
    L = 4 * max(N - 1, 1)
    assert E * N * L * 4 * T >= MB128, "this size must select the non-temporal kernels"
    env = _penv(E, N, cfg, seed=17, auto_reset=True, kernel=kernel)
    ref = _penv(E, N, cfg, seed=17, auto_reset=True, kernel=kernel)
    ref.enable_terminal_capture()
    env.reset()
    ref.reset()
    ro = ParticleRollout(env, n_ticks=T, use_graph=(mode == "graph"), fused=(mode == "fused")).collect(reset=False)
    assert torch.equal(ro.state[0].permute(1, 0, 2), ref.global_state)
    n_done = 0
    for t in range(T):
        gs, oo, _, rew, rew_n, done = ref.step()                      # plain kernels, in-kernel actions (same Philox keys)
        assert torch.equal(ref.last_actions, ro.actions[t]), t
        assert torch.equal(rew, ro.reward[t]) and torch.equal(rew_n, ro.reward_n[t]) and torch.equal(done, ro.done[t].bool()), t
        assert torch.equal(gs, ro.state[t + 1].permute(1, 0, 2)), t
        assert torch.equal(oo, ro.obs_others[t + 1]), t               # the rows the non-temporal stores wrote
        assert torch.equal(ref.collisions_after_last_step, ro.collisions[t]), t
        if bool(done.any()):
            n_done += int(done.sum())
            assert torch.equal(ref.terminal_obs_others[done], ro.term_obs_others[t][done])
            assert torch.equal(ref.terminal_state[done], ro.term_state[t].permute(1, 0, 2)[done])
    assert n_done >= E
    ro.close()


@pytest.mark.parametrize("stage,E,T", [(2, 8192, 52), (1, 16384, 60)])
@pytest.mark.parametrize("mode", ["graph", "fused"])
def test_checkers_streaming_rollout_equals_stepwise(stage, E, T, mode):
    from cm3_amd.checkers import VecCheckersEnv
    from cm3_amd.rollout import CheckersRollout
    cfg = load_cfg("checkers_stage%d.json" % stage)
    N = cfg["n_agents"]
    goals = np.eye(2) if N == 2 else np.array([[0, 1]])
    env = VecCheckersEnv(cfg["init"], N, 9, E, device="cuda:0", seed=23, auto_reset=True)
    ref = VecCheckersEnv(cfg["init"], N, 9, E, device="cuda:0", seed=23, auto_reset=True)
    per_env = env.grid_stride + env.obst_stride + N * 4 * 4 + N * env.Lo * 8 + N * 4 * 8
    assert per_env * E * T >= MB128, "this size must select the non-temporal kernels"
    ref.enable_terminal_capture()
    ref.reset(goals)
    ro = CheckersRollout(env, n_ticks=T, use_graph=(mode == "graph"), fused=(mode == "fused")).collect(goals)
    n_done = 0
    for t in range(T):
        (g, v), oo, ot, ov, rew, local, done = ref.step()
        assert torch.equal(ref.last_actions, ro.actions[t]), t
        assert torch.equal(rew, ro.reward[t]) and torch.equal(local, ro.local_rewards[t]) and torch.equal(done, ro.done[t].bool())
        assert torch.equal(g, ro.grid[t + 1]) and torch.equal(v, ro.vec[t + 1]) and torch.equal(oo, ro.obs_others[t + 1]), t
        assert torch.equal(ot, ro.obs_self_t[t + 1]) and torch.equal(ov, ro.obs_self_v[t + 1]), t
        assert torch.equal(ref._goals, ro.goal_slots[t + 1]), t
        if bool(done.any()):
            n_done += int(done.sum())
            (tg, tv), too, tot, tov = ref.terminal_obs()
            assert torch.equal(tg[done], ro.term_grid[t][done]) and torch.equal(tv[done], ro.term_vec[t][done])
            assert torch.equal(too[done], ro.term_obs_others[t][done]) and torch.equal(tot[done], ro.term_obs_self_t[t][done])
            assert torch.equal(tov[done], ro.term_obs_self_v[t][done])
    assert n_done >= 3 * E
    ro.close()


def test_headline_kernel_vs_f64_oracle_teacher_forced():
    """bench.py's headline instantiation -- k_particle_step_pairs<float, 4, 4, FUSED = false, NT = true, LIVE = true> (C2:
    antipodal, 4 agents, 4096 envs, streaming-size trajectory -> non-temporal observation stores + live-state stepping) --
    compared DIRECTLY with the float64 oracle (the restatement of environment.py:81-123 pinned against the reference's own
    outputs), not with another kernel: every tick the oracle is given slot t of the device trajectory (state, goals, counters)
    and the kernel's actions, and slot t + 1 / the terminal capture / rewards / done / collision counts must equal its step
    within the float32 tolerance of the north-star (1e-5).  In-kernel actions and same-launch resets are checked against the
    Philox specification (oracle/philox.py).  T = 198 ticks = six 33-tick episodes per env."""
    from oracle import philox
    from oracle.particle_oracle import VecParticleOracle
    from cm3_amd.rollout import ParticleRollout
    E, N, T, TOL, EDGE = 4096, 4, 198, 1e-5, 2e-6
    cfg = load_cfg("particle_stage2_antipodal.json")
    env = _penv(E, N, "particle_stage2_antipodal.json", seed=12341, auto_reset=True, max_steps=33)     # kernel="auto", as bench.py
    env.reset()
    ro = ParticleRollout(env, n_ticks=T, use_graph=True).collect(reset=False)
    assert ro._live, "this size must select the live-state rollout (bench.py's headline)"
    assert E * N * 12 * 4 * T >= MB128, "this size must select the non-temporal kernels"
    f64 = lambda x: x.detach().cpu().numpy().astype(np.float64)     # noqa: E731
    state, goals = f64(ro.state).transpose(0, 2, 1, 3), f64(ro.goals).transpose(0, 2, 1, 3)       # [T+1, E, N, .]
    obs, term_state, term_obs = f64(ro.obs_others), f64(ro.term_state).transpose(0, 2, 1, 3), f64(ro.term_obs_others)
    acts, rew, rew_n = ro.actions.cpu().numpy(), f64(ro.reward), f64(ro.reward_n)
    done, coll = ro.done.cpu().numpy().astype(bool), ro.collisions.cpu().numpy()
    ro.close()
    orc = VecParticleOracle(N, cfg, 0.2, 33, E)
    ids = np.arange(E)
    steps, prev_coll, episode = np.zeros(E, np.int64), np.zeros(E, np.int64), np.ones(E, np.int64)   # after the first reset
    skipped = n_done = 0
    worst = 0.0
    for t in range(T):
        # in-kernel uniform actions (train_onpolicy.py:305-307) = the Philox stream keyed (seed, env, episode, step)
        for ep in np.unique(episode):
            for st in np.unique(steps[episode == ep]):
                m = (episode == ep) & (steps == st)
                assert np.array_equal(acts[t][m], philox.expected_actions(12341, ids[m], int(ep), int(st), N)), t
        orc.set_from_global_state(state[t], goals[t], steps=steps, collisions=prev_coll)
        w_gs, w_oo, _, w_rew, w_rn, w_done = orc.step(acts[t])
        m_col, m_reach = orc.pair_margins()
        safe = (m_col > EDGE) & (m_reach > EDGE)
        skipped += int((~safe).sum())
        d = done[t]
        got_gs = np.where(d[:, None, None], term_state[t], state[t + 1])
        got_oo = np.where(d[:, None, None], term_obs[t], obs[t + 1])
        worst = max(worst, float(np.abs(got_gs - w_gs).max()), float(np.abs(got_oo - w_oo).max()))
        assert np.abs(got_gs - w_gs).max() < TOL and np.abs(got_oo - w_oo).max() < TOL, t
        assert np.abs(rew_n[t][safe] - w_rn[safe]).max() < TOL and np.abs(rew[t][safe] - w_rew[safe]).max() < 4 * TOL, t
        assert np.array_equal(d[safe], w_done[safe]), t
        assert np.array_equal(coll[t][safe], orc.collisions[safe]), t
        # every slot's observation rows (the non-temporal stores) are the observation of that slot's state
        orc.set_from_global_state(state[t + 1], goals[t + 1])
        assert np.abs(orc.observe()[1] - obs[t + 1]).max() < 1e-6, t
        if d.any():      # same-launch resets (multi-goal_spread.py:65-93): fresh episodes from the Philox reset stream
            n_done += int(d.sum())
            episode = episode + d
            for ep in np.unique(episode[d]):
                m = d & (episode == ep)
                pos, lm, _ = philox.expected_reset(12341, ids[m], int(ep), cfg, N, 0.2)
                assert np.abs(state[t + 1][m][..., 2:4] - pos).max() < 1e-6 and np.abs(goals[t + 1][m] - lm).max() < 1e-6
                assert np.abs(state[t + 1][m][..., 0:2]).max() == 0.0
        steps = np.where(d, 0, steps + 1)
        prev_coll = np.where(d, 0, coll[t])
    assert n_done >= 6 * E and skipped <= 400, (n_done, skipped)      # ~1 near-threshold sample per 100 k pair-steps
    assert worst < TOL
