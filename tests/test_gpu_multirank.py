"""GPU: the N > 1 plumbing on a one-GPU box -- bench.py under a launcher (RCCL process group, barrier, all-gather of the
per-rank times and of the advantage moments), the self-spawning entry, and a genuine world-size-2 execution of the
product's collective path (both ranks on cuda:0, gloo group)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _torchrun(nproc, script_args, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("workload", ["c4", "c2", "c5", "c3"])
def test_bench_under_torchrun_one_rank(workload, tmp_path):
    """The driver's N > 1 command shape with N = 1: nccl (= RCCL) init, barriers, the all-gather of the per-rank clocks and
    (c4) of the advantage moments all execute on the box.  The LAST stdout line is the driver's compact record (< 4 KB); the full
    record is the extras file."""
    extras = str(tmp_path / "extras.json")
    r = _torchrun(1, ["bench.py", "--gpus", "1", "--workload", workload, "--steps", "3", "--warmup", "1", "--no-extras",
                      "--no-sweep", "--no-cpu-baseline", "--extras-file", extras])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert len(r.stdout.strip().splitlines()[-1]) < 4096
    assert line["n_gpus"] == 1 and line["value"] > 1e6 and line["roofline"]["frac"] > 0 and "workload" in line["config"]
    assert line["rccl"]["rccl_world_size"] == 1 and line["rccl"]["all_reduce_ok"] is True
    with open(extras) as fh:
        out = json.load(fh)
    assert out["value"] == pytest.approx(line["value"], rel=1e-5)
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["value"] > 1e6
    assert len(out["per_rank"]) == 1 and out["per_rank"][0]["rank"] == 0
    # the line says by itself what RCCL saw: world size from the process group, a checked all-reduce, the P2P matrix
    assert out["rccl_world_size"] == 1 and out["rccl"]["backend"] == "nccl" and out["rccl"]["all_reduce_ok"] is True
    assert out["rccl"]["p2p_access"][0]["can_access_peer"][0] == 1
    assert abs(out["per_rank"][0]["avg_launch_us"] - out["roofline"]["avg_launch_us"]) < 1e-6
    if workload == "c4":
        assert "all-gather" in out["config"]["parallelism"]
    # one clock: value and roofline agree
    bytes_per_tick = out["roofline"]["algorithmic_bytes_per_launch"]
    assert abs(out["roofline"]["achieved"] * 1e9 / bytes_per_tick * out["config"]["envs_per_gpu"] / out["value"] - 1) < 1e-9


def test_bench_self_spawn_with_too_few_gpus_fails_after_spawning():
    """`python bench.py --gpus N` re-executes itself through torch.distributed.run; with fewer GPUs than ranks the ranks
    without a device say so (after spawning, not before)."""
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "1", "--no-extras",
                        "--no-sweep", "--no-cpu-baseline"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "needs %d GPUs on this node, only %d visible" % (n, n - 1) in (r.stderr + r.stdout)


def test_two_rank_product_path_on_one_gpu(tmp_path):
    """world size 2: each rank collects its shard of the envs (RNG keyed by global env id), computes its float64 moments on
    the device, all-gathers the triples and normalises with n_parts = 2.  Both ranks must hold bit-identical statistics,
    equal to the float64 statistics of the concatenated batch, and the shards must equal the single-process run."""
    from cm3_amd.shard import normalized_returns, returns_to_go
    import cm3_amd
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    E = 1000 + 24                                   # ragged split: 512 + 512
    r = _torchrun(2, [os.path.join("tests", "workers", "two_rank_adv_worker.py"), str(tmp_path), str(E)])
    assert r.returncode == 0, r.stderr[-3000:]
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % k)) for k in range(2)]
    assert parts[0]["mean"] == parts[1]["mean"] and parts[0]["std"] == parts[1]["std"]
    assert parts[0]["count"] == parts[1]["count"] == float(33 * E * 4)
    # single process, all envs: the same trajectories (shard invariance) and the same statistics
    cfg = cm3_amd.load_config("particle_stage2_cross")
    env = VecParticleEnv(cfg, 4, 0.2, 33, E, device="cuda:0", auto_reset=True, seed=12341)
    env.reset()
    ro = ParticleRollout(env, n_ticks=33, use_graph=True).collect(reset=False)
    whole_r = torch.cat([p["reward_n"] for p in parts], dim=1).cuda()
    assert torch.equal(whole_r, ro.reward_n)
    assert torch.equal(torch.cat([p["done"] for p in parts], dim=1).cuda(), ro.done)
    ret = returns_to_go(ro.reward_n.double(), ro.done, gamma=0.99)
    assert abs(parts[0]["mean"] - float(ret.mean())) < 1e-5 * abs(float(ret.mean()))        # float32 returns vs float64
    assert abs(parts[0]["std"] - float(ret.std(unbiased=False))) < 1e-5 * float(ret.std())
    norm1, (m1, s1, _) = normalized_returns(ro.reward_n, ro.done, None, gamma=0.99)
    assert abs(float(m1) - parts[0]["mean"]) < 1e-12 * abs(float(m1))      # 1 part vs 2 parts: only the summation tree differs
    whole_norm = torch.cat([p["norm"] for p in parts], dim=1).cuda()
    assert torch.allclose(whole_norm, norm1, rtol=1e-6, atol=1e-6)
    ro.close()


@pytest.mark.parametrize("workload", ["c5", "c3", "c2"])
def test_shards_reproduce_the_single_process_trajectory(workload):
    """What `bench.py --gpus G` relies on for every workload: rank r steps envs [r E, (r + 1) E) with env_id_base = r E, and
    because every random stream is keyed by the GLOBAL env id the G shards together are, bit for bit, the trajectory of one
    process stepping G E envs (here G = 2 shards on one GPU against the whole batch; in-kernel actions, auto-reset)."""
    import numpy as np
    import cm3_amd
    from cm3_amd.rollout import CheckersRollout, ParticleRollout
    T, E = 40, 1024 + 64
    halves = [(0, 512 + 64), (512 + 64, 512)]
    if workload == "c3":
        from cm3_amd.checkers import VecCheckersEnv
        cfg = cm3_amd.load_config("checkers_stage2")
        goals = np.eye(2)
        def run(base, n):
            env = VecCheckersEnv(cfg["init"], 2, 9, n, device="cuda:0", auto_reset=True, env_id_base=base, seed=12341)
            ro = CheckersRollout(env, n_ticks=T, use_graph=True).collect(goals)
            out = {k: getattr(ro, k).clone() for k in ("actions", "reward", "local_rewards", "done", "vec", "obs_self_v", "obs_others")}
            out["grid"], out["obs_self_t"] = ro.grid.clone(), ro.obs_self_t.clone()
            ro.close()
            return out
        env_axis = {k: 1 for k in ("actions", "reward", "local_rewards", "done", "vec", "obs_self_v", "obs_others", "grid", "obs_self_t")}
    else:
        from cm3_amd.particle import VecParticleEnv
        name, N = ("particle_merge8", 8) if workload == "c5" else ("particle_stage2_antipodal", 4)
        cfg = cm3_amd.load_config(name)
        def run(base, n):
            env = VecParticleEnv(cfg, N, 0.2, 9, n, device="cuda:0", auto_reset=True, env_id_base=base, seed=12341)
            env.reset()
            ro = ParticleRollout(env, n_ticks=T, use_graph=True).collect(reset=False)
            out = {k: getattr(ro, k).clone() for k in ("actions", "reward", "reward_n", "done", "obs_others", "collisions", "state", "goals",
                                                       "term_state", "term_obs_others")}
            ro.close()
            return out
        env_axis = dict(actions=1, reward=1, reward_n=1, done=1, obs_others=1, collisions=1, state=2, goals=2, term_state=2,
                        term_obs_others=1)
    whole = run(0, E)
    parts = [run(b, n) for b, n in halves]
    for k, ax in env_axis.items():
        assert torch.equal(torch.cat([p[k] for p in parts], dim=ax), whole[k]), k
    assert int(whole["done"].sum()) >= 3 * E


@pytest.mark.parametrize("use_graph", [True, False])
def test_collect_normalized_is_collect_followed_by_normalized_returns(use_graph):
    """ParticleRollout.collect_normalized -- slot copies, T step launches, returns + moments, normalise captured as ONE hipGraph
    (world size 1: the all-gather is the identity) -- equals collect() followed by cm3_amd.shard.normalized_returns, bit for
    bit, rollout after rollout (graph replays)."""
    import cm3_amd
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    from cm3_amd.shard import normalized_returns
    cfg = cm3_amd.load_config("particle_stage2_cross")
    E, T = 777, 33
    envs = [VecParticleEnv(cfg, 4, 0.2, 33, E, device="cuda:0", auto_reset=True, seed=12341) for _ in range(2)]
    for e in envs:
        e.reset()
    a, b = (ParticleRollout(e, n_ticks=T, use_graph=use_graph) for e in envs)
    for _ in range(3):
        a.collect(reset=False)
        want, (m, s, n) = normalized_returns(a.reward_n, a.done, None, gamma=0.99)
        got, (m2, s2, n2), t_coll = b.collect_normalized(gamma=0.99, time_collective=True)
        assert t_coll == 0.0
        for name in ("state", "obs_others", "actions", "reward_n", "done", "goals", "term_state"):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert torch.equal(want, got)
        assert float(m) == float(m2) and float(s) == float(s2) and float(n) == float(n2) == float(T * E * 4)
        assert torch.equal(envs[0].global_state, envs[1].global_state)
    raw, _ = b.collect_normalized(gamma=0.99, normalize=False)          # a different key re-captures
    assert float(raw.abs().max()) > 1.0
    a.close()
    b.close()


@pytest.mark.parametrize("E", [777, 1031])
def test_collect_normalized_after_env_step_flips_the_buffers(E):
    """VecParticleEnv.step() flips the env's double buffer; the graph captured by collect_normalized holds the addresses of the
    buffers that were current at capture time, so a collect_normalized -> env.step() -> collect_normalized sequence must
    re-capture (it replayed on the stale buffer).  E = 1031: N * E is odd for the 8-byte goal rows, so the slot
    copies are not 16-byte sized and take the torch-copy path INSIDE the capture."""
    import cm3_amd
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    from cm3_amd.shard import normalized_returns
    N = 4 if E == 777 else 3
    cfg = cm3_amd.load_config("particle_stage2_cross" if N == 4 else "particle_merge8")
    T = 12
    envs = [VecParticleEnv(cfg, N, 0.2, 7, E, device="cuda:0", auto_reset=True, seed=77) for _ in range(2)]
    for e in envs:
        e.reset()
    a, b = (ParticleRollout(e, n_ticks=T, use_graph=True) for e in envs)
    for k in range(4):
        a.collect(reset=False)
        want, (m, s, n) = normalized_returns(a.reward_n, a.done, None, gamma=0.99)
        got, (m2, s2, n2) = b.collect_normalized(gamma=0.99)
        for name in ("state", "obs_others", "actions", "reward_n", "done", "goals", "term_state"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (k, name)
        assert torch.equal(want, got), k
        assert float(m) == float(m2) and float(s) == float(s2)
        assert torch.equal(envs[0].global_state, envs[1].global_state), k
        assert torch.equal(envs[0].get_obs()[1], envs[1].get_obs()[1]), k
        for e in envs:                       # an odd number of env.step() calls between two collections
            for _ in range(1 if k % 2 == 0 else 3):
                e.step()
        assert torch.equal(envs[0].global_state, envs[1].global_state), k
    a.close()
    b.close()


def test_c4_shards_moments_combine_to_the_single_process_statistics():
    """BASELINE configs[3] over G ranks = every rank's collect_normalized up to its moments, ONE all-gather of the (sum, sum of
    squares, count) triples, cm3_normalize_* over the G triples in rank order.  Here G = 2 shards on one GPU (env_id_base = the
    shard's first global env id) against one process collecting all envs through the one-graph path: the RAW returns of the
    shards are, bit for bit, the columns of the whole batch; the triples add up to the whole batch's triple (the count exactly,
    the sums to the last bits of their different summation trees); normalising the shards with the combined triples gives the
    single process's normalised returns to float32 rounding."""
    import cm3_amd
    from cm3_amd import _lib
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    cfg = cm3_amd.load_config("particle_stage2_cross")
    T, E = 33, 1024 + 64
    halves = [(0, 512 + 64), (512 + 64, 512)]

    def run(base, n, normalize):
        env = VecParticleEnv(cfg, 4, 0.2, 33, n, device="cuda:0", auto_reset=True, env_id_base=base, seed=12341)
        env.reset()
        ro = ParticleRollout(env, n_ticks=T, use_graph=True)
        out, stats = ro.collect_normalized(gamma=0.99, normalize=normalize)
        res = dict(out=out.clone(), moments=ro._norm.moments.clone(), stats=[float(x) for x in stats], reward_n=ro.reward_n.clone())
        ro.close()
        return res

    whole = run(0, E, True)
    whole_raw = run(0, E, False)
    parts = [run(b, n, False) for b, n in halves]
    assert torch.equal(torch.cat([p["reward_n"] for p in parts], dim=1), whole["reward_n"])
    assert torch.equal(torch.cat([p["out"] for p in parts], dim=1), whole_raw["out"])           # raw returns: shard-invariant
    gathered = torch.cat([p["moments"] for p in parts])                                         # what the all-gather delivers
    tot = gathered.view(2, 3)[0] + gathered.view(2, 3)[1]
    assert float(tot[2]) == float(whole["moments"][2]) == float(T * E * 4)
    assert abs(float(tot[0]) - float(whole["moments"][0])) <= 1e-12 * abs(float(whole["moments"][0]))
    assert abs(float(tot[1]) - float(whole["moments"][1])) <= 1e-12 * abs(float(whole["moments"][1]))
    lib = _lib.lib()
    stats = torch.zeros(3, dtype=torch.float64, device="cuda:0")
    normed = []
    for p in parts:                         # every rank: cm3_normalize_f32 over the G gathered triples, rank order
        x = p["out"].clone()
        _lib.check(lib.cm3_normalize_f32(x.data_ptr(), 0, gathered.data_ptr(), 2, stats.data_ptr(), x.numel(), 4, 1e-8, 1,
                                         _lib.current_stream_handle(x.device)))
        normed.append(x)
    assert abs(float(stats[0]) - whole["stats"][0]) <= 1e-12 * abs(whole["stats"][0])
    assert abs(float(stats[1]) - whole["stats"][1]) <= 1e-12 * abs(whole["stats"][1])
    assert torch.allclose(torch.cat(normed, dim=1), whole["out"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("E,N,cfg_name", [(777, 4, "particle_stage2_cross"), (300, 2, "particle_stage2_merge")])
def test_collect_normalized_segments_equal_rollout_by_rollout(E, N, cfg_name):
    """K rollouts of one collection phase in ONE hipGraph replay (collect_normalized(segments=K): K T step launches + the two
    launches of cm3_returns_normalize_segments_*) against K replays of the single-rollout graph on a twin env: trajectories,
    per-rollout normalised returns and statistics, the env left behind -- bit for bit, phase after phase."""
    import cm3_amd
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    cfg = cm3_amd.load_config(cfg_name)
    K, S = 5, 9
    envs = [VecParticleEnv(cfg, N, 0.2, 7, E, device="cuda:0", auto_reset=True, seed=4242) for _ in range(2)]
    for e in envs:
        e.reset()
    a = ParticleRollout(envs[0], n_ticks=K * S, use_graph=True)
    b = ParticleRollout(envs[1], n_ticks=S, use_graph=True)
    for phase in range(3):
        out, (mean, std, cnt) = a.collect_normalized(gamma=0.97, segments=K)
        assert tuple(mean.shape) == (K,)
        for k in range(K):
            o1, (m1, s1, n1) = b.collect_normalized(gamma=0.97)
            sl = slice(k * S, (k + 1) * S)
            assert torch.equal(out[sl], o1), (phase, k)
            assert float(mean[k]) == float(m1) and float(std[k]) == float(s1) and float(cnt[k]) == float(n1) == float(S * E * N)
            for name in ("actions", "reward_n", "reward", "done", "collisions"):
                assert torch.equal(getattr(a, name)[sl], getattr(b, name)), (phase, k, name)
            d = b.done.bool()                  # terminal captures are written where an episode ended (stale elsewhere)
            assert int(d.sum()) > 0
            assert torch.equal(a.term_state[sl].permute(0, 2, 1, 3)[d], b.term_state.permute(0, 2, 1, 3)[d]), (phase, k)
            assert torch.equal(a.term_obs_others[sl][d], b.term_obs_others[d]), (phase, k)
            for name in ("state", "obs_others", "goals"):
                assert torch.equal(getattr(a, name)[k * S:(k + 1) * S + 1], getattr(b, name)), (phase, k, name)
        assert torch.equal(envs[0].global_state, envs[1].global_state), phase
        assert torch.equal(envs[0].get_obs()[1], envs[1].get_obs()[1]) and torch.equal(envs[0].goals, envs[1].goals)
        assert torch.equal(envs[0].steps, envs[1].steps) and torch.equal(envs[0].episode, envs[1].episode)
    raw, _ = a.collect_normalized(gamma=0.97, segments=K, normalize=False)        # (another key re-captures)
    assert float(raw.abs().max()) > 1.0
    a.close()
    b.close()


def test_two_rank_segmented_phase_on_one_gpu(tmp_path):
    """world size 2, K = 3 rollouts per phase: each rank's graph ends at its K moment triples, ONE all-gather carries them
    (gloo group, both ranks on cuda:0), cm3_normalize_segments_* applies the global statistics per rollout.  Both ranks must hold
    identical statistics, equal to the single process's over all envs (to the last bits of the different summation trees), and the
    normalised shards must be the single process's columns."""
    import cm3_amd
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    E, K = 1000 + 24, 3
    r = _torchrun(2, [os.path.join("tests", "workers", "two_rank_adv_worker.py"), str(tmp_path), str(E), str(K)])
    assert r.returncode == 0, r.stderr[-3000:]
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % k)) for k in range(2)]
    for key in ("mean", "std", "count"):
        assert torch.equal(parts[0][key], parts[1][key]), key
    assert parts[0]["count"].tolist() == [float(11 * E * 4)] * K
    cfg = cm3_amd.load_config("particle_stage2_cross")
    env = VecParticleEnv(cfg, 4, 0.2, 33, E, device="cuda:0", auto_reset=True, seed=12341)
    env.reset()
    ro = ParticleRollout(env, n_ticks=11 * K, use_graph=True)
    for rep in range(2):
        out, (mean, std, cnt) = ro.collect_normalized(gamma=0.99, segments=K)
    assert torch.equal(torch.cat([p["reward_n"] for p in parts], dim=1).cuda(), ro.reward_n)
    assert torch.allclose(parts[0]["mean"].cuda(), mean, rtol=1e-12, atol=0) and torch.allclose(parts[0]["std"].cuda(), std, rtol=1e-12, atol=0)
    assert torch.allclose(torch.cat([p["norm"] for p in parts], dim=1).cuda(), out, rtol=1e-6, atol=1e-6)
    tot = parts[0]["moments"] + parts[1]["moments"]
    assert torch.allclose(tot.cuda(), ro._norm.moments, rtol=1e-12, atol=0)
    ro.close()


@pytest.mark.parametrize("stage", [2, 1])
def test_policy_driven_checkers_shards_reproduce_the_single_process_rollout(stage):
    """The one-launch Checkers policy rollout under env sharding (round 6): rank r holds envs [r E, (r + 1) E) with env_id_base = r E;
    the sampling uniforms and, for one agent, the per-episode goals are keyed by the GLOBAL env id (actor and env share seed and
    env_id_base), so two shards on one GPU are, bit for bit, the rollout of one process holding all envs -- actions, probabilities,
    observations, rewards, terminal captures."""
    import numpy as np
    import cm3_amd
    from cm3_amd.actor import CheckersActor
    from cm3_amd.checkers import VecCheckersEnv
    from cm3_amd.rollout import CheckersRollout
    from oracle import actor_checkers_oracle as AO
    N = 2 if stage == 2 else 1
    cfg = cm3_amd.load_config("checkers_stage%d" % stage)
    w = AO.init_weights(np.random.default_rng(12), N)
    T, E = 33, 1024 + 96
    halves = [(0, 512 + 96), (512 + 96, 512)]
    goals = np.eye(2) if N > 1 else None
    names = ("actions", "probs", "reward", "local_rewards", "done", "vec", "obs_self_v", "obs_others", "term_vec", "term_obs_self_v", "goal_slots")

    def run(base, n, all_goals):
        env = VecCheckersEnv(cfg["init"], N, 9, n, device="cuda:0", auto_reset=True, env_id_base=base, seed=4242)
        actor = CheckersActor(w, N, stage=stage, device="cuda:0", seed=4242, env_id_base=base, precision="f16x3")
        assert actor.fused_rollout_ok(env)
        ro = CheckersRollout(env, n_ticks=T, record_probs=True)
        g = goals if N > 1 else all_goals[base:base + n]
        ro.collect(g, policy=actor, epsilon=0.2)
        ro.collect(g, policy=actor, epsilon=0.2)
        torch.cuda.synchronize()
        out = {k: getattr(ro, k).clone() for k in names}
        out["grid"], out["obs_self_t"] = ro.grid.clone(), ro.obs_self_t.clone()
        ro.close()
        return out
    first_goals = np.eye(2)[np.random.default_rng(1).integers(0, 2, (E, 1))]          # stage 1: one random one-hot goal per env
    whole = run(0, E, first_goals)
    parts = [run(b, n, first_goals) for b, n in halves]
    for k in list(names) + ["grid", "obs_self_t"]:
        assert torch.equal(torch.cat([p[k] for p in parts], dim=1), whole[k]), k
    assert int(whole["done"].sum()) >= 3 * E
