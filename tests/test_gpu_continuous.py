"""GPU: round-2 additions of the collection loop (train_onpolicy.py:281-377).

* rollouts over sub-batches of the envs (env_offset / env_count) produce the whole batch's trajectory;
* env.step() keeps the reference's counters (they go on counting if an env is stepped after `done`), and the COLLECTOR reads
  scenario.collisions of each episode from the per-tick trajectory slot at the tick that ends it (train_onpolicy.py:302, :356);
  the continuous mode captures that count before the same-launch reset zeroes it;
* Checkers continuous collection: the terminal transition keeps the true post-step next_* (train_onpolicy.py:336-347),
  goals are recorded per slot, actions_prev restarts at zeros (:295);
* a captured actor/step graph follows the annealed epsilon (:369) without re-capture;
* cm3_normalize_* with n_parts > 1 (the rank-ordered sum of the all-gathered moments).
"""
import numpy as np
import pytest
import torch

from oracle import philox
from oracle.checkers_oracle import VecCheckersOracle
from tests.helpers import load_cfg

pytestmark = pytest.mark.gpu


def _penv(E, N=4, dtype=torch.float32, cfg="particle_stage2_antipodal.json", **kw):
    from cm3_amd.particle import VecParticleEnv
    c = cfg if isinstance(cfg, dict) else load_cfg(cfg)
    return VecParticleEnv(c, N, kw.pop("prob_random", 0.2), kw.pop("max_steps", 33), E, device="cuda:0", dtype=dtype, **kw)


TRAJ = ("state", "obs_others", "actions", "reward", "reward_n", "done", "goals", "collisions")


@pytest.mark.parametrize("kernel,N,cfg", [("pair", 4, "particle_stage2_cross.json"), ("env", 4, "particle_stage2_cross.json"),
                                          ("agent", 8, "particle_merge8.json"), ("auto", 2, "particle_stage2_merge.json")])
@pytest.mark.parametrize("mode", ["eager", "fused"])
def test_sub_batch_launches_equal_the_whole_batch(kernel, N, cfg, mode):
    """cm3_particle_desc.env_offset / env_count: three rollouts over sub-batches of the envs (the last one ragged) leave exactly the
    trajectory, terminal captures and live counters of one rollout over the whole batch -- envs never interact and every RNG draw
    is keyed by the global env id.  (Until ABI 7 cm3_particle_rollout_chains_* ran such sub-batches on several streams; that
    entry point was a measured regression and is gone, tools/chains_diag.py reproduces it with these two fields.)"""
    from cm3_amd import _lib
    from cm3_amd.rollout import ParticleRollout
    E, T = 1000, 25
    outs = []
    for parts in (1, 3):
        env = _penv(E, N, cfg=cfg, seed=21, auto_reset=True, max_steps=9, kernel=kernel)
        env.reset()
        ro = ParticleRollout(env, n_ticks=T, use_graph=False, fused=(mode == "fused"), live_state=False, sparse_goals=False)
        chunk = ((E + parts - 1) // parts + 255) // 256 * 256          # whole workgroups per sub-batch
        for _ in range(2):                                             # second phase: continued episodes
            ro._load_slot0()
            flags = _lib.FLAG_AUTO_RESET | _lib.FLAG_GEN_ACTIONS | env.kernel_flags | (_lib.FLAG_FUSED_TICKS if mode == "fused" else 0)
            for c in range(parts):
                lo = c * chunk
                if lo >= E:
                    break
                if parts > 1:
                    env._desc.env_offset, env._desc.env_count = lo, min(chunk, E - lo)
                ro._enqueue(0, T, flags)
                env._desc.env_offset, env._desc.env_count = 0, 0
            ro._store_back(False)
        torch.cuda.synchronize()
        outs.append((ro, env))
    a, b = outs[0][0], outs[1][0]
    for name in ("state", "obs_others", "actions", "reward", "reward_n", "done", "_goals_buf", "collisions"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    d = a.done.bool()
    assert int(d.sum()) > 0
    assert torch.equal(a.term_state.permute(0, 2, 1, 3)[d], b.term_state.permute(0, 2, 1, 3)[d])
    assert torch.equal(a.term_obs_others[d], b.term_obs_others[d])
    ea, eb = outs[0][1], outs[1][1]
    assert torch.equal(ea._meta, eb._meta) and torch.equal(ea._episode, eb._episode)
    assert torch.equal(ea.global_state, eb.global_state)
    for ro, _ in outs:
        ro.close()


def _two_agent_cfg(gap):
    # two agents standing ON their landmarks, `gap` apart: reached at the first tick
    return dict(n_agents=2, agents_x=[-gap / 2, gap / 2], agents_y=[0.0, 0.0], landmarks_x=[-gap / 2, gap / 2],
                landmarks_y=[0.0, 0.0], initial_std=0.0)


@pytest.mark.parametrize("kernel", ["env", "pair", "agent"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_post_done_collisions_do_not_make_an_episode_bad(kernel, dtype):
    """Env A's episode ends at tick 1 (both agents on their landmarks, 0.31 apart: no collision); the policy then drives
    the agents into each other.  env.step() is the reference's MultiAgentEnv.step and keeps counting steps and collisions
    when it is called after `done` (environment.py:93; multi-goal_spread.py:137) -- but the reference's LOOP stops at
    `done` (train_onpolicy.py:302) and reads scenario.collisions there (:356), so the collector's episode_is_bad() must be
    False for A.  Control env B (same motion, far landmarks) runs to max_steps and IS bad.  (ADVICE r1, medium.)"""
    from cm3_amd.rollout import ParticleRollout
    E, T = 130, 7
    ctl_cfg = _two_agent_cfg(0.31)
    ctl_cfg["landmarks_x"], ctl_cfg["landmarks_y"] = [0.9, -0.9], [0.9, -0.9]
    fin = _penv(E, 2, dtype=dtype, cfg=_two_agent_cfg(0.31), prob_random=0.0, kernel=kernel, max_steps=T)
    ctl = _penv(E, 2, dtype=dtype, cfg=ctl_cfg, prob_random=0.0, kernel=kernel, max_steps=T)
    tick = {"t": 0}

    def policy(obs_others, obs_self, goals):          # tick 0: stay; afterwards agent 0 right, agent 1 left
        a = torch.zeros(E, 2, dtype=torch.int32, device="cuda:0")
        if tick["t"] > 0:
            a[:, 0], a[:, 1] = 2, 1
        tick["t"] += 1
        return a
    ro_f = ParticleRollout(fin, n_ticks=T, use_graph=False).collect(policy=policy)
    tick["t"] = 0
    ro_c = ParticleRollout(ctl, n_ticks=T, use_graph=False).collect(policy=policy)
    assert bool(ro_f.done[0].all()) and not bool(ro_c.done[:T - 1].any()) and bool(ro_c.done[T - 1].all())
    assert torch.equal(ro_f.state, ro_c.state)                         # identical physics: the collisions did happen
    assert int(fin.collisions.min()) > 0 and torch.equal(fin.collisions, ctl.collisions)    # the reference's counter
    assert int(fin.steps.min()) == T                                   # ... and its step counter keep counting
    assert torch.equal(ro_f.valid.sum(0), torch.ones(E, dtype=torch.long, device="cuda:0"))  # one valid transition each
    assert not bool(ro_f.episode_is_bad().any())                       # A: ended clean at tick 1
    assert bool(ro_c.episode_is_bad().all())                           # B: collided inside its episode
    assert int(ro_f.collisions[0].max()) == 0 and int(ro_f.collisions[T - 1].min()) > 0


def test_episode_synchronous_rollout_is_bad_and_valid_across_collects():
    """ParticleRollout without auto-reset: episode_is_bad() is scenario.collisions != 0 of each env's episode, and a
    collect(reset=False) after the episodes ended yields no valid transition."""
    from cm3_amd.rollout import ParticleRollout
    E, N = 256, 2
    env = _penv(E, N, cfg="particle_stage2_merge.json", seed=13, max_steps=12)
    ro = ParticleRollout(env, n_ticks=12, use_graph=False).collect()
    assert bool(ro._finished.all()) and bool(ro.valid.all())
    # per-episode collision count from the trajectory itself: every ordered colliding pair costs its agent 1.0 of reward
    # beyond -dist (multi-goal_spread.py:121-138), so reward_n + dist is minus the agent's collision count
    pos = ro.state[1:, :, :, 2:4]                                  # [T, N, E, 2]
    dist = (pos - env._goals.unsqueeze(0)).norm(dim=-1)            # [T, N, E]
    hits = (-(ro.reward_n.permute(0, 2, 1) + dist)).round().clamp(min=0).sum((0, 1))
    assert torch.equal(ro.episode_is_bad(), hits > 0)
    assert 0 < int(ro.episode_is_bad().sum()) < E                  # both kinds occur
    assert torch.equal(ro.collisions[-1].to(hits.dtype), hits) and torch.equal(env.collisions, ro.collisions[-1])
    ro.collect(reset=False)                                        # nothing restarted them: all invalid
    assert int(ro.valid.sum()) == 0
    ro.collect(reset=True)
    assert bool(ro.valid.all())


@pytest.mark.parametrize("kernel", ["env", "pair", "agent"])
@pytest.mark.parametrize("fused", [False, True])
def test_continuous_rollout_captures_terminal_collisions(kernel, fused):
    """AUTO_RESET zeroes scenario.collisions in the launch that ends the episode; the trajectory's per-tick `collisions`
    slot keeps the pre-reset value (the is_bad flag of train_onpolicy.py:356 in continuous mode).  Every tick of every env
    is checked against a second env stepped tick by tick without auto-reset and re-injected at every episode start."""
    from cm3_amd.rollout import ParticleRollout
    E, N, T, S = 192, 2, 30, 8
    env = _penv(E, N, dtype=torch.float64, cfg="particle_stage2_merge.json", seed=4, auto_reset=True, max_steps=S,
                kernel=kernel)
    env.reset()
    ro = ParticleRollout(env, n_ticks=T, use_graph=False, fused=fused).collect(reset=False)
    ref = _penv(E, N, dtype=torch.float64, cfg="particle_stage2_merge.json", seed=4, max_steps=S, kernel="env")
    ref.set_state(ro.state[0, :, :, 2:4].permute(1, 0, 2), ro.state[0, :, :, 0:2].permute(1, 0, 2), ro.goals[0].permute(1, 0, 2))
    n_bad = 0
    for t in range(T):
        _, _, _, _, _, done = ref.step(ro.actions[t])
        assert torch.equal(done, ro.done[t].bool())
        assert torch.equal(ro.collisions[t], ref.collisions)          # running count, every env, every tick
        if bool(done.any()):
            assert torch.equal(ro.episode_is_bad()[t][done], ref.collisions[done] != 0)
            n_bad += int((ref.collisions[done] != 0).sum())
            assert bool(done.all())                                # every env ends at max_steps here
            ref.set_state(ro.state[t + 1, :, :, 2:4].permute(1, 0, 2), ro.state[t + 1, :, :, 0:2].permute(1, 0, 2),
                          ro.goals[t + 1].permute(1, 0, 2))
    assert n_bad > 0
    assert not bool(ro.episode_is_bad()[~ro.done.bool()].any())


# ---- Checkers continuous collection ---------------------------------------------------------------------------------
def _ck(stage, E, max_steps, seed, auto_reset=True, **kw):
    from cm3_amd.checkers import VecCheckersEnv
    cfg = load_cfg("checkers_stage%d.json" % stage)
    return cfg, VecCheckersEnv(cfg["init"], cfg["n_agents"], max_steps, E, device="cuda:0", seed=seed, auto_reset=auto_reset, **kw)


def _f64(x):
    return x.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("stage", [2, 1])
@pytest.mark.parametrize("mode", ["eager", "graph", "fused", "generic-kernel"])
def test_checkers_continuous_rollout_matches_oracle_with_terminal_capture(stage, mode):
    """CheckersRollout on an auto-reset env, two consecutive collects: every slot equals the oracle driven through the same
    actions and restarted where `done` fires; the terminal transition's next_* are the TRUE post-step observation
    (train_onpolicy.py:336-347), slot t+1 the fresh episode's; goals follow the per-episode draw of a single-agent env
    (:288-291); actions_prev restarts at zeros (:295)."""
    from cm3_amd.rollout import CheckersRollout
    E, T, S, seed = 150, 21, 6, 31
    cfg, env = _ck(stage, E, S, seed, padded_records=(None if mode != "generic-kernel" else False))
    N, i = cfg["n_agents"], cfg["init"]
    goals0 = np.eye(2) if N == 2 else np.array([[0, 1]])
    ro = CheckersRollout(env, n_ticks=T, use_graph=(mode == "graph"), fused=(mode == "fused"))
    orc = VecCheckersOracle(i["n_rows"], i["n_columns"], i["n_obs"], i["agents_r"], i["agents_c"], N, S, E)
    orc.reset(goals0)
    episode = np.ones(E, np.int64)                      # episodes started so far (reset() -> 1)
    prev = np.zeros((E, N), np.int64)
    n_term = 0
    for phase in range(2):
        ro.collect(goals0 if phase == 0 else None)
        cols = ro.as_reference_batch()
        B = T * E
        assert cols["reward"].shape == (B,) and bool(ro.valid.all())
        for t in range(T):
            sel = slice(t * E, (t + 1) * E)             # time-major flat index of (t, e)
            acts = ro.actions[t].cpu().numpy()
            assert np.array_equal(cols["actions_prev"][sel], prev)
            assert np.array_equal(cols["goals"][sel], np.eye(2)[orc.goal])
            before = orc.outputs()
            assert np.array_equal(cols["grid"][sel], before[0]) and np.array_equal(cols["obs_self_t"][sel], before[3])
            assert np.array_equal(cols["vec"][sel], before[1]) and np.array_equal(cols["obs_self_v"][sel], before[4])
            w = orc.step(acts)
            assert np.array_equal(cols["reward"][sel], w[5]) and np.array_equal(cols["local_rewards"][sel], w[6])
            assert np.array_equal(cols["done"][sel], w[7])
            for k, name in enumerate(("next_grid", "next_vec", "next_obs_others", "next_obs_self_t", "next_obs_self_v")):
                assert np.array_equal(cols[name][sel], w[k]), (name, t)          # true post-step values, terminal or not
            done = w[7]
            prev = np.where(done[:, None], 0, acts)
            if done.any():
                n_term += int(done.sum())
                episode = episode + done
                new_goal = None
                if N == 1:                               # Philox draw of the restarted episode (csrc/checkers.hip)
                    word = philox.reset_words(seed, np.arange(E), episode, 0)[0]
                    new_goal = (np.asarray(word) & 1).astype(np.int64).reshape(E, 1)
                fresh = orc.reset_envs(done, new_goal)
                # slot t+1 holds the fresh episode's observation for the restarted envs
                assert np.array_equal(_f64(ro.grid[t + 1])[done], fresh[0][done])
                assert np.array_equal(_f64(ro.obs_self_t[t + 1])[done], fresh[3][done])
                assert np.array_equal(_f64(ro.vec[t + 1])[done], fresh[1][done])
        # the env's own buffers hold the last slot (get_obs / act(env) after a rollout)
        (g, v), oo, ot, ov = env.get_obs()
        assert torch.equal(g, ro.grid[T]) and torch.equal(ot, ro.obs_self_t[T]) and torch.equal(ov, ro.obs_self_v[T])
        assert torch.equal(env.last_actions, ro.actions[T - 1])
    assert n_term >= 3 * E
    if N == 1:
        assert 0 < int(ro.goal_slots.sum()) < ro.goal_slots.numel()       # both goals were drawn
    ro.close()


def test_checkers_step_terminal_capture():
    """VecCheckersEnv.step under auto_reset with enable_terminal_capture(): terminal_obs() is the post-step observation of
    the finished episode, the returned observation the fresh one."""
    cfg, env = _ck(2, 64, 4, 5)
    _, ref = _ck(2, 64, 4, 5, auto_reset=False)
    env.enable_terminal_capture()
    env.reset(np.eye(2))
    ref.reset(np.eye(2))
    fresh = [x.clone() for x in (ref.get_obs()[0][0], ref.get_obs()[2])]
    for t in range(4):
        a = env.step()
        b = ref.step(env.last_actions)
    assert bool(a[6].all()) and bool(b[6].all())
    (tg, tv), too, tot, tov = env.terminal_obs()
    assert torch.equal(tg, b[0][0]) and torch.equal(tv, b[0][1]) and torch.equal(too, b[1])
    assert torch.equal(tot, b[2]) and torch.equal(tov, b[3])
    assert torch.equal(a[0][0], fresh[0]) and torch.equal(a[2], fresh[1])


# ---- captured actor graphs follow epsilon ---------------------------------------------------------------------------
def _particle_actor(N, seed=0):
    from cm3_amd.actor import ParticleActor
    rng = np.random.default_rng(seed)
    Lo = 4 * max(N - 1, 1)
    shapes = {"actor_branch_self/kernel": (6, 64), "actor_branch_self/bias": (64,), "W_branch_self_h2": (64, 64),
              "stage-2/actor_others/kernel": (Lo, 128), "stage-2/actor_others/bias": (128,),
              "stage-2/W_others_h2": (128, 64), "b": (64,), "actor_out/kernel": (64, 5), "actor_out/bias": (5,)}
    w = {k: (rng.standard_normal(v) * 0.3).astype(np.float32) for k, v in shapes.items()}
    return ParticleActor(w, N, stage=2, device="cuda:0", seed=12341)


def test_actor_graph_follows_annealed_epsilon_without_recapture():
    """train_onpolicy.py:369 anneals epsilon after every training phase.  The captured (actor, step) x T graph reads
    epsilon from a device float: replays with a new epsilon equal eager collection with that epsilon, and the graph
    handle is the one captured first."""
    from cm3_amd.rollout import ParticleRollout
    E, N, T = 200, 4, 12
    outs = []
    for graph in (True, False):
        env = _penv(E, N, seed=12341, auto_reset=True, max_steps=7)
        env.reset()
        actor = _particle_actor(N)
        ro = ParticleRollout(env, n_ticks=T, use_graph=graph, policy_mode="tick")      # the captured actor / step graph
        handles, acts = [], []
        for eps in (0.5, 0.3, 0.05):
            ro.collect(policy=actor, epsilon=eps, reset=False)
            acts.append(ro.actions.clone())
            handles.append(ro._actor_graph.graph.value if graph else None)
        outs.append((acts, handles, ro))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b)
    assert len(set(outs[0][1])) == 1
    assert not torch.equal(outs[0][0][0], outs[0][0][2])
    for _, _, ro in outs:
        ro.close()


# ---- the collective's consumer with n_parts > 1 -----------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_normalize_with_eight_rank_moments(dtype):
    """cm3_normalize_*(n_parts = 8): the rank-ordered sum of eight (sum, sum of squares, count) triples -- what the
    all-gather hands every rank on an 8-GPU node -- gives the float64 torch statistics of the whole batch and the same
    normalised values as the host formulation, bit for bit."""
    import ctypes
    from cm3_amd import _lib
    from cm3_amd.shard import local_moments
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    world, T, E, C = 8, 33, 512, 4
    shards = [torch.randn(T, E, C, generator=g, device="cuda", dtype=dtype) * (1 + r) + r for r in range(world)]
    valids = [torch.rand(T, E, generator=g, device="cuda") > 0.2 for _ in range(world)]
    parts = torch.cat([local_moments(x, v) for x, v in zip(shards, valids)]).contiguous()       # [8 * 3] float64
    tot = parts.view(world, 3)[0].clone()
    for r in range(1, world):
        tot = tot + parts.view(world, 3)[r]
    mean = tot[0] / tot[2]
    std = (tot[1] / tot[2] - mean * mean).clamp(min=0).sqrt()
    fn = lib.cm3_normalize_f32 if dtype == torch.float32 else lib.cm3_normalize_f64
    stats = torch.zeros(3, dtype=torch.float64, device="cuda")
    stream = _lib.current_stream_handle(torch.device("cuda:0"))
    for r in (0, 5):
        x = shards[r].clone()
        v8 = valids[r].to(torch.uint8).contiguous()
        _lib.check(fn(x.data_ptr(), v8.data_ptr(), parts.data_ptr(), world, stats.data_ptr(), x.numel(), C, 1e-8, 1, stream))
        assert float(stats[0]) == float(mean) and float(stats[2]) == float(tot[2])
        assert abs(float(stats[1]) - float(std)) <= 1e-14 * float(std)
        want = (shards[r] - mean.to(dtype)) / (std + 1e-8).to(dtype)
        want = torch.where(valids[r].unsqueeze(-1), want, torch.zeros_like(want))
        if dtype == torch.float64:
            assert torch.equal(x, want)                 # one expression on both paths (cm3_amd.shard.normalize_advantages)
        else:                                           # torch's float32 division kernel need not be correctly rounded
            assert torch.allclose(x, want, rtol=3e-7, atol=0.0)
    # statistics only
    stats.zero_()
    _lib.check(fn(0, 0, parts.data_ptr(), world, stats.data_ptr(), 1, 1, 1e-8, 0, stream))
    assert float(stats[0]) == float(mean)
    whole = torch.cat([s[v] for s, v in zip(shards, valids)]).double()
    assert abs(float(mean) - float(whole.mean())) < 1e-9 * (1 + abs(float(whole.mean())))
    assert abs(float(std) - float(whole.std(unbiased=False))) < 1e-7 * float(whole.std())


def test_external_reset_clears_the_finished_flags_of_a_reused_rollout():
    """a reused ParticleRollout on an env WITHOUT auto-reset carries `_finished` across collect(reset=False)
    calls; re-seeding the env outside the collector (env.reset(), env.reset(mask), env.set_state()) must clear it, or
    every later transition stays invalid and episode_returns() is all zero."""
    from cm3_amd.rollout import ParticleRollout
    E, T = 256, 9
    env = _penv(E, seed=41, auto_reset=False, max_steps=T)
    env.reset()
    ro = ParticleRollout(env, n_ticks=T, use_graph=False)
    ro.collect(reset=False)
    assert bool(ro.valid.all()) and bool(ro._finished.all())          # every episode ended at max_steps
    ro.collect(reset=False)
    assert not bool(ro.valid.any())                                    # stepped on past `done`: nothing valid
    env.reset()                                                        # external re-seed of every env
    ro.collect(reset=False)
    assert bool(ro.valid.all())
    half = torch.zeros(E, dtype=torch.bool, device=env.device)
    half[: E // 2] = True
    env.reset(half)                                                    # ... of a subset
    ro.collect(reset=False)
    v = ro.valid
    assert bool(v[:, : E // 2].all()) and not bool(v[:, E // 2:].any())
    st = env.get_state()
    env.set_state(st["pos"], st["vel"], st["landmarks"])               # state injection restarts the step counters
    ro.collect(reset=False)
    assert bool(ro.valid.all())
    g, _ = ro.episode_returns()
    assert float(g.abs().min()) > 0.0
