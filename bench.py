#!/usr/bin/env python
"""bench.py -- env-steps/s of the CM3 rollout hot path on MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 ...            # re-executes itself through torch.distributed.run with 8 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: one on-policy COLLECTION PHASE of the trajectory-collection loop
(alg/train_onpolicy.py:302-377: episodes_per_train = 10 episodes of max_steps = 33 ticks = 330 transitions per env between
two training steps) for E = 4096 environments x 4 agents of config_particle_stage2_antipodal (BASELINE.json configs[1])
per GPU.  Every tick is ONE launch of the particle step kernel (csrc/particle.hip) that reads trajectory slot t and
writes slot t+1 of a [331, E, ...] device trajectory (state, obs_others, goals, actions, reward, reward_n, done, plus the
terminal next-state / observation / collision count of envs that restart in the same launch) -- the product's own
collection object (cm3_amd.rollout.ParticleRollout), replayed as one hipGraph per phase; float32, uniform random actions
drawn in-kernel (train_onpolicy.py:305-307), auto-reset at max_steps.  --steps K times exactly K phases after --warmup W.
Env instances shard across ranks with NO data-path collective => "weak" scaling.

Rank 0 prints the driver's line LAST and keeps it under 4 KB (compact_line(): contract keys + numbers); everything else this
run measured goes to bench_extras.json and, as bare numbers, to the stdout line before it (extras_summary()).
The line:  metric = env-steps/s summed over all GPUs (max-over-ranks wall clock between two barriers);
roofline  = algorithmic bytes of the timed launches (400 B x E per launch for N=4) / the SAME wall
            clock, against the 8 TB/s HBM3E peak (the HIP-event time of the same region is reported beside it);
cpu_baseline = the reference-shaped scalar NumPy port (oracle.particle_oracle.ParticleEnvOracle) timed on the host
            cores of this box (rank 0, at every N; the other ranks wait at the closing barrier).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense (= the FP32 vector peak)
MFMA_F16_PEAK_TFLOPS = 2500.0  # same guide: FP16 / BF16 MFMA, dense (the 5 PF figure is 2:1 sparsity)


def particle_actor_mfma_work(n_agents, precision):
    """EXECUTED matrix-core work of one agent row and tick of the particle actor (csrc/actor.hip), per pipe, in MACs -- the
    padded tiles as the instructions run them, not the network's own 6*64 + L*128 + 192*64 + 64*5:
      first layers   v_mfma_f32_16x16x4_f32: K = 6 -> 8 for branch_self (64 units), K = L for actor_others (128 units) -- f16x3 with
                     16 <= L <= 32 (N = 5..9, ActorFirstB::kF16Oth): actor_others as THREE float16 passes with K = 32
      second layer   f32: 192 x 64 on the f32 pipe; f16x3: THREE float16 passes (hi hi + hi lo + lo hi) on the f16 pipe; bf16: one
      output layer   v_mfma_f32_16x16x4_f32: 64 x 16 (5 actions padded to a 16-column tile)
    -> (f32 MACs, f16/bf16 MACs, network MACs)"""
    L = 4 * max(n_agents - 1, 1)
    f32 = 64 * 8 + 128 * L + 64 * 16
    second = 192 * 64
    net = 6 * 64 + L * 128 + second + 64 * 5
    if precision == "f32":
        return f32 + second, 0, net
    if precision == "f16x3" and 16 <= L <= 32:
        return 64 * 8 + 64 * 16, (second + 128 * 32) * 3, net
    return f32, second * (3 if precision == "f16x3" else 1), net


def checkers_actor_mfma_work(n_agents, precision, others_from_table=False):
    """Executed matrix-core MACs per agent row of the Checkers actor (csrc/actor_checkers.hip: tile widths of its layers) ->
    (f32 MACs, f16/bf16 MACs, network MACs).  f32: every layer on v_mfma_f32_16x16x4_f32 (K padded to 16s: conv 80 x 160 Toeplitz,
    conv_linear 160 x 32, branch_self 48 x 256, branch_others 16 x 256, the two 256 x 256, out 256 x 16); f16x3: every layer as
    float16 passes with K padded to 32s -- two passes for the conv (its int8 input is exact in float16), three for the others."""
    net = 25 * 6 * 27 + 150 * 32 + 43 * 256 + (2 * max(n_agents - 1, 1) * 256 + 256 * 256 if n_agents > 1 else 0) + 256 * 256 + 256 * 5
    small_f32 = 80 * 160 + 160 * 32 + 48 * 256 + 16 * 256 + 256 * 16
    big = 2 * 256 * 256
    if precision == "f32":
        return small_f32 + big, 0, net
    if precision == "bf16":
        return small_f32, big, net
    # others_from_table: the one-launch rollout (csrc/policy_checkers.hip) reads branch_others W_others_h2 from a table -- neither the
    # 32 x 256 first layer nor its 256 x 256 second layer is executed (stage 1 has no others branch at all)
    others = 0 if (others_from_table or n_agents == 1) else 32 * 256 + 256 * 256
    return 0, 96 * 160 * 2 + (160 * 32 + 64 * 256 + 256 * 256 + others + 256 * 16) * 3, net


def mfma_roofline(rows_per_tick, us_per_tick, f32_macs, f16_macs, net_macs):
    """The matrix-core roofline of a policy tick: every pass divided by the peak of the pipe it runs on.  `matrix_time_us` is the
    time the executed instructions need at those peaks; frac = matrix_time / tick time (what share of the tick the matrix cores
    are the bound of).  The network's own FLOP rate is reported beside it as `network_TFLOPs` (never against a peak: its
    padded / split passes run on two different pipes)."""
    t = us_per_tick * 1e-6
    f32_tf = 2.0 * f32_macs * rows_per_tick / t / 1e12
    f16_tf = 2.0 * f16_macs * rows_per_tick / t / 1e12
    matrix_us = (2.0 * f32_macs * rows_per_tick / (MFMA_F32_PEAK_TFLOPS * 1e12) +
                 2.0 * f16_macs * rows_per_tick / (MFMA_F16_PEAK_TFLOPS * 1e12)) * 1e6
    return {"bound": "mfma", "achieved": f32_tf + f16_tf, "unit": "TFLOP/s", "frac": matrix_us / us_per_tick,
            "peak": (f32_tf + f16_tf) / max(matrix_us / us_per_tick, 1e-12),
            "executed_f32_mfma_TFLOPs": f32_tf, "f32_peak": MFMA_F32_PEAK_TFLOPS, "frac_of_f32_peak": f32_tf / MFMA_F32_PEAK_TFLOPS,
            "executed_f16_mfma_TFLOPs": f16_tf, "f16_peak": MFMA_F16_PEAK_TFLOPS, "frac_of_f16_peak": f16_tf / MFMA_F16_PEAK_TFLOPS,
            "matrix_time_us": matrix_us, "network_TFLOPs": 2.0 * net_macs * rows_per_tick / t / 1e12,
            "note": "executed MFMA FLOPs per pipe (padded tiles, three float16 passes for f16x3) against that pipe's dense peak; "
                    "frac = time the executed matrix instructions need at peak / time per tick; `peak` = the blended peak "
                    "that makes achieved / peak equal frac.  A small frac means the tick is latency-bound, not MFMA-bound"}

EP_TICKS = 33                 # ticks of one episode (config.json max_steps)
PHASE_EPISODES = 10           # episodes_per_train (alg/config.json; train_onpolicy.py:359): one collection phase
PHASE_TICKS = EP_TICKS * PHASE_EPISODES


def algorithmic_bytes_per_env_step(n_agents):
    """16 + 48 N + 16 N max(N-1,1)  (f32/int32): 80 B (N=1), 400 B (N=4), 1296 B (N=8)."""
    n = n_agents
    return 16 + 48 * n + 16 * n * max(n - 1, 1)


# Checkers N=2 -- compact state 16 r + 16 w, actions 8, outputs 360
CHECKERS_BYTES_PER_ENV_STEP = 400


class GraphStepper(object):
    """Common replay plumbing: `enqueue(n_ticks, stream)` is captured once as a hipGraph of `graph_ticks` ticks."""
    graph = None
    graph_ticks = 0
    launches_per_tick = 1

    def stream(self):
        return self._lib_mod.current_stream_handle(self.device)

    def capture(self, n_ticks):
        self.graph = self._lib_mod.capture_graph(self.device, lambda s: self.enqueue(n_ticks, s))
        self.graph_ticks = n_ticks

    def run(self, n_ticks):
        """Enqueue n_ticks ticks: whole graph replays plus an eager remainder."""
        s = self.stream()
        if self.graph is not None:
            while n_ticks >= self.graph_ticks:
                self._lib_mod.check(self.lib.cm3_graph_launch(self.graph, s))
                n_ticks -= self.graph_ticks
        if n_ticks > 0:
            self.enqueue(n_ticks)

    def close(self):
        if self.graph is not None:
            self.torch.cuda.synchronize(self.device)
            self.lib.cm3_graph_destroy(self.graph)
            self.graph = None


class ParticleStepper(GraphStepper):
    """IN-PLACE stepping of E envs through the C ABI (zero strides = every tick overwrites the same live buffers; no
    trajectory is stored).  Used for the E-sweep and as the labelled `in_place` extra."""

    def __init__(self, cfg, n_agents, n_envs, device, seed=12341, env_id_base=0, max_steps=33, prob_random=0.2,
                 kernel="auto", fused=False, tensor_actions=False):
        import torch
        from cm3_amd import _lib
        from cm3_amd.particle import VecParticleEnv
        self.torch, self._lib_mod, self.lib = torch, _lib, _lib.lib()
        self.env = VecParticleEnv(cfg, n_agents, prob_random, max_steps, n_envs, device=device, seed=seed,
                                  dtype=torch.float32, auto_reset=True, env_id_base=env_id_base, kernel=kernel)
        self.env.reset()
        e = self.env
        e._desc.flags = _lib.FLAG_AUTO_RESET | _lib.FLAG_GEN_ACTIONS | e.kernel_flags
        if tensor_actions:      # the policy-provided-actions branch (train_onpolicy.py:311-313): the step launch READS its actions
            e._desc.flags &= ~_lib.FLAG_GEN_ACTIONS
            e._actions[0].random_(0, 5)
        self.fused = bool(fused)
        if self.fused:       # all ticks of an enqueue() in ONE launch (state in registers)
            e._desc.flags |= _lib.FLAG_FUSED_TICKS
        t = self.traj = _lib.ParticleTraj()
        t.state = e._state[0].data_ptr()
        t.goals = e._goals.data_ptr()
        t.obs_others = e._obs_others[0].data_ptr()
        t.actions = e._actions[0].data_ptr()
        t.reward_n = e._reward_n[0].data_ptr()
        t.reward = e._reward[0].data_ptr()
        t.done = e._done[0].data_ptr()
        t.meta = e._meta.data_ptr()
        t.episode = e._episode.data_ptr()      # all strides stay 0: in place
        self.device = e.device
        self.launches_per_tick = 1

    def enqueue(self, n_ticks, stream=None):
        s = self.stream() if stream is None else stream
        self._lib_mod.check(self.lib.cm3_particle_rollout_f32(ctypes.byref(self.env._desc), ctypes.byref(self.traj), int(n_ticks), s))


class TrajectoryStepper(object):
    """The headline: cm3_amd.rollout.ParticleRollout over a [phase_ticks + 1]-slot device trajectory with terminal capture
    (continuous collection, train_onpolicy.py:302-350); one collect() = slot copy in, ONE hipGraph replay of phase_ticks
    step launches, slot copy out."""

    def __init__(self, cfg, n_agents, n_envs, device, env_id_base=0, kernel="auto", phase_ticks=PHASE_TICKS,
                 use_graph=True, fused=False, live_state=None):
        import torch
        from cm3_amd.particle import VecParticleEnv
        from cm3_amd.rollout import ParticleRollout
        self.torch = torch
        self.env = VecParticleEnv(cfg, n_agents, 0.2, EP_TICKS, n_envs, device=device, dtype=torch.float32, auto_reset=True,
                                  env_id_base=env_id_base, kernel=kernel)
        self.env.reset()
        self.ro = ParticleRollout(self.env, n_ticks=phase_ticks, use_graph=use_graph, fused=fused, live_state=live_state)
        self.device = self.env.device
        self.phase_ticks = int(phase_ticks)
        self.launches_per_tick = 1

    def capture(self, n_ticks):
        pass                                    # ParticleRollout captures its own graph on first use

    def run(self, n_ticks):
        assert n_ticks % self.phase_ticks == 0, "trajectory mode runs whole collection phases"
        for _ in range(n_ticks // self.phase_ticks):
            self.ro.collect(reset=False)

    def close(self):
        self.ro.close()


class CheckersStepper(GraphStepper):
    """IN-PLACE stepping of E Checkers envs through cm3_checkers_rollout (labelled extra)."""

    def __init__(self, cfg, n_envs, device, seed=12341, env_id_base=0, max_steps=33, fused=False):
        import numpy as np
        import torch
        from cm3_amd import _lib
        from cm3_amd.checkers import VecCheckersEnv
        self.torch, self._lib_mod, self.lib = torch, _lib, _lib.lib()
        self.env = VecCheckersEnv(cfg["init"], cfg["n_agents"], max_steps, n_envs, device=device, seed=seed,
                                  auto_reset=True, env_id_base=env_id_base)
        n = cfg["n_agents"]
        self.env.reset(np.eye(2) if n > 1 else np.array([[1, 0]]))
        self.env._desc.flags = _lib.FLAG_AUTO_RESET | _lib.FLAG_GEN_ACTIONS
        self.bufs = self.env._bufs(0)
        self.device = self.env.device
        self.fused = bool(fused)
        b, t = self.bufs, _lib.CheckersTraj()
        for name in ("mask", "agents", "steps", "episode", "goals", "actions", "grid", "vec", "obs_others",
                     "obs_self_t", "obs_self_v", "local_rewards", "reward", "done"):
            setattr(t, name, getattr(b, name))
        self.traj = t
        if self.fused:
            self.env._desc.flags |= _lib.FLAG_FUSED_TICKS

    def enqueue(self, n_ticks, stream=None):
        s = self.stream() if stream is None else stream
        self._lib_mod.check(self.lib.cm3_checkers_rollout(ctypes.byref(self.env._desc), ctypes.byref(self.traj),
                                                          int(n_ticks), s))


class CheckersTrajectoryStepper(object):
    """C3 headline: cm3_amd.rollout.CheckersRollout in continuous mode (auto-reset + terminal capture + per-slot goals)
    over a [phase_ticks + 1]-slot trajectory, one hipGraph replay per phase."""

    def __init__(self, cfg, n_envs, device, env_id_base=0, phase_ticks=PHASE_TICKS, fused=False):
        import numpy as np
        import torch
        from cm3_amd.checkers import VecCheckersEnv
        from cm3_amd.rollout import CheckersRollout
        self.torch = torch
        n = cfg["n_agents"]
        self.env = VecCheckersEnv(cfg["init"], n, EP_TICKS, n_envs, device=device, auto_reset=True, env_id_base=env_id_base)
        self.goals = np.eye(2) if n > 1 else np.array([[1, 0]])
        self.ro = CheckersRollout(self.env, n_ticks=phase_ticks, use_graph=True, fused=fused)
        self.device = self.env.device
        self.phase_ticks = int(phase_ticks)
        self.launches_per_tick = 1

    def capture(self, n_ticks):
        pass

    def run(self, n_ticks):
        assert n_ticks % self.phase_ticks == 0
        for _ in range(n_ticks // self.phase_ticks):
            self.ro.collect(self.goals)

    def close(self):
        self.ro.close()


class RolloutAdvStepper(object):
    """BASELINE configs[3] (C4): 33-tick trajectory collection followed by the advantage-normalisation step -- discounted
    returns + moments, ONE all-gather of 3 float64 per rank and rollout (RCCL), normalisation.  The rollouts of one collection
    phase (episodes_per_train = 10, train_onpolicy.py:359) are ONE hipGraph replay: 330 step launches + the two launches of
    cm3_returns_normalize_segments_* for the ten rollouts' ten advantage steps (round 4; with several ranks the graph ends at the
    ten moment triples, one all-gather carries them all).  A bench step stays one rollout + its normalisation; a number of steps
    that is not a multiple of ten finishes on the single-rollout graph of round 3."""

    def __init__(self, cfg, n_agents, n_envs, device, env_id_base=0, kernel="auto", fused=False,
                 phase_rollouts=PHASE_EPISODES):
        import torch
        from cm3_amd.particle import VecParticleEnv
        from cm3_amd.rollout import ParticleRollout
        self.torch = torch
        self.env = VecParticleEnv(cfg, n_agents, 0.2, EP_TICKS, n_envs, device=device, dtype=torch.float32, auto_reset=True,
                                  env_id_base=env_id_base, kernel=kernel)
        self.env.reset()
        self.K = int(phase_rollouts)
        self.ro = ParticleRollout(self.env, n_ticks=EP_TICKS * self.K, use_graph=True, fused=fused)
        self.ro1 = ParticleRollout(self.env, n_ticks=EP_TICKS, use_graph=True, fused=fused)
        # set-up, not a step: both graphs are captured here, so that no capture (milliseconds) lands in a timed region whatever
        # the warm-up / step counts are
        self.ro.collect_normalized(gamma=0.99, segments=self.K)
        self.ro1.collect_normalized(gamma=0.99)
        self.device = self.env.device
        self.last = None
        self.launches_per_tick = 1
        self.collective_s, self.rollouts = 0.0, 0     # host time inside the all-gather call, summed over rollouts

    def capture(self, n_ticks):
        pass

    def run(self, n_ticks):
        assert n_ticks % EP_TICKS == 0, "c4 runs whole 33-tick rollouts"
        phases, rest = divmod(n_ticks // EP_TICKS, self.K)
        for _ in range(phases):
            *self.last, t_coll = self.ro.collect_normalized(gamma=0.99, time_collective=True, segments=self.K)
            self.collective_s += t_coll
            self.rollouts += self.K
        for _ in range(rest):
            *self.last, t_coll = self.ro1.collect_normalized(gamma=0.99, time_collective=True)
            self.collective_s += t_coll
            self.rollouts += 1

    def close(self):
        self.ro.close()
        self.ro1.close()


WORKLOADS = {
    # name: (kind, config file, envs per GPU, description)
    "c2": ("particle", "particle_stage2_antipodal", 4096, "config_particle_stage2_antipodal.json: 4 agents"),
    "c3": ("checkers", "checkers_stage2", 8192, "config_checkers_stage2.json: 2 agents, integer grid"),
    "c4": ("particle_adv", "particle_stage2_cross", 4096,
           "config_particle_stage2_cross.json: 4 agents, trajectory + advantage normalisation (1 moments all-gather per rollout)"),
    "c5": ("particle", "particle_merge8", 8192, "build-defined merge8: 8 agents (SURVEY.md section 8d C5)"),
}


def timed_ticks(stepper, n_ticks):
    """HIP-event time (ms) of n_ticks ticks on the launch stream."""
    torch = stepper.torch
    with torch.cuda.device(stepper.device):
        stream = torch.cuda.current_stream()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(stream)
        stepper.run(n_ticks)
        stop.record(stream)
        stop.synchronize()
        return start.elapsed_time(stop)


def measure_bandwidth(device, gib=4.0, reps=5):
    """(read GB/s, copy GB/s): streaming 16 B/lane read of a buffer far larger than the 256 MiB Infinity Cache, and a
    16 B/lane copy of half of it into the other half (2 x bytes / time)."""
    import torch
    from cm3_amd import _lib
    lib = _lib.lib()
    nbytes = int(gib * (1 << 30)) // 32 * 32
    buf = torch.empty(nbytes // 4, dtype=torch.int32, device=device)
    buf.random_(0, 1 << 30)
    sink = torch.zeros(lib.cm3_hbm_bench_sink_words(), dtype=torch.int32, device=device)
    s = _lib.current_stream_handle(device)
    half = nbytes // 2

    def timed(fn):
        fn()
        torch.cuda.synchronize(device)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(reps):
            fn()
        stop.record()
        stop.synchronize()
        return start.elapsed_time(stop) / reps * 1e-3

    t_read = timed(lambda: _lib.check(lib.cm3_hbm_read_bench(buf.data_ptr(), nbytes, sink.data_ptr(), s)))
    t_copy = timed(lambda: _lib.check(lib.cm3_hbm_copy_bench(buf.data_ptr() + half, buf.data_ptr(), half, s)))
    del buf
    return nbytes / t_read / 1e9, 2.0 * half / t_copy / 1e9


def measure_launch_floor(device, read_bytes, write_bytes, blocks, threads=256, nodes=PHASE_TICKS):
    """us per launch of (a) the load -> store skeleton with the step launch's traffic and (b) an empty launch of the same
    shape, each as a hipGraph of `nodes` launches replayed like the bench's phase (cm3_traffic_floor_bench)."""
    import torch
    from cm3_amd import _lib
    lib = _lib.lib()
    src = torch.zeros(max(read_bytes, 16) // 4, dtype=torch.int32, device=device)
    dst = torch.zeros(max(write_bytes, 16) // 4, dtype=torch.int32, device=device)
    res = {}
    for name, (rb, wb) in (("same_traffic_us", (read_bytes, write_bytes)), ("empty_launch_us", (0, 0))):
        def enqueue(s, rb=rb, wb=wb):
            for _ in range(nodes):
                _lib.check(lib.cm3_traffic_floor_bench(src.data_ptr(), rb, dst.data_ptr(), wb, blocks, threads, s))
        graph = _lib.capture_graph(device, enqueue)
        s = _lib.current_stream_handle(device)
        for _ in range(3):
            _lib.check(lib.cm3_graph_launch(graph, s))
        torch.cuda.synchronize(device)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(5, 1650 // nodes)
        start.record()
        for _ in range(reps):
            _lib.check(lib.cm3_graph_launch(graph, s))
        stop.record()
        stop.synchronize()
        res[name] = start.elapsed_time(stop) * 1e3 / (reps * nodes)
        torch.cuda.synchronize(device)
        lib.cm3_graph_destroy(graph)
    res["note"] = ("%d x %d lanes, %d B read then %d B written per launch, no arithmetic, %d launches per hipGraph replay; "
                   "same_traffic_us divided by the step kernel's time per launch is frac_of_floor"
                   % (blocks, threads, read_bytes, write_bytes, nodes))
    return res


def step_traffic_bytes(kind, n_agents, n_envs):
    """(read, written) algorithmic bytes of one step launch, rounded up to 16: the skeleton of the floor"""
    if kind == "checkers":
        rd, wr = 24 * n_envs, 376 * n_envs
    else:
        n = n_agents
        rd, wr = (28 * n + 4) * n_envs, (20 * n + 16 * n * max(n - 1, 1) + 12) * n_envs
    return (rd + 15) // 16 * 16, (wr + 15) // 16 * 16


def launch_ceiling(device, kind, n_agents, n_envs, bytes_per_launch, launch_us):
    """What ONE launch per tick allows at this batch: the same number of workgroups reading and writing the same algorithmic bytes
    with no arithmetic, and an empty launch, each replayed as a hipGraph like the bench's phase (measure_launch_floor).
    ceiling_frac = algorithmic bytes / same_traffic_us / peak: the roofline fraction a perfect step kernel would show with one launch
    per tick; frac_of_floor = same_traffic_us / the step kernel's time per launch."""
    rd, wr = step_traffic_bytes(kind, n_agents, n_envs)
    floor = measure_launch_floor(device, rd, wr, blocks=max(1, min(2048, (n_envs * 16 + 255) // 256)), nodes=PHASE_TICKS)
    floor["frac_of_floor"] = floor["same_traffic_us"] / launch_us
    floor["ceiling_frac"] = bytes_per_launch / (floor["same_traffic_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
    return floor


def cfg_name_of(cfg):
    """Name of the cm3_amd/configs file a loaded particle config came from (for the worker command line)."""
    import cm3_amd
    for name in ("particle_stage1", "particle_stage2_antipodal", "particle_stage2_cross", "particle_stage2_merge",
                 "particle_merge8"):
        if cm3_amd.load_config(name) == cfg:
            return name
    raise ValueError("unknown particle config")


def _scalar_port_worker(args):
    """One host process of the all-cores CPU line: whole episodes of the scalar port for `budget_s` seconds."""
    cfg, n_agents, budget_s, seed = args
    import random
    import numpy as np
    from oracle.particle_oracle import ParticleEnvOracle
    env = ParticleEnvOracle(n_agents, cfg, 0.2, 33)
    py_rng, np_rng = random.Random(seed), np.random.RandomState(seed)
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        env.reset(py_rng, np_rng)
        done = False
        while not done:
            *_, done = env.step(np_rng.randint(0, 5, n_agents))
            steps += 1
    return steps / (time.perf_counter() - t0)


def cpu_baseline(cfg, n_agents, budget_s=12.0):
    """Reference-shaped scalar NumPy port on ONE host core: whole 33-tick episodes with uniform random
    actions, exactly the reference's pretrain loop (train_onpolicy.py:281-350 without the buffer)."""
    import random
    import numpy as np
    from oracle.particle_oracle import ParticleEnvOracle
    env = ParticleEnvOracle(n_agents, cfg, 0.2, 33)
    py_rng, np_rng = random.Random(12341), np.random.RandomState(12341)
    steps, episodes = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        env.reset(py_rng, np_rng)
        done = False
        while not done:
            acts = np_rng.randint(0, 5, n_agents)
            *_, done = env.step(acts)
            steps += 1
        episodes += 1
    dt = time.perf_counter() - t0
    out = dict(value=steps / dt, unit="env-steps/s", cores=1, kind="port",
               sample="%d episodes (%d env-steps) of the same workload (N=%d, 33 ticks, uniform actions) "
                      "in %.1f s on 1 of %d host cores; scalar per-env NumPy port of the reference's call structure"
                      % (episodes, steps, n_agents, dt, os.cpu_count()))
    # second, stronger CPU line: the vectorised [E,N,...] NumPy restatement, one process
    try:
        from oracle.particle_oracle import VecParticleOracle
        E = 4096
        vec = VecParticleOracle(n_agents, cfg, 0.2, 33, E)
        rng = np.random.default_rng(0)
        pos0 = np.stack([np.array(cfg["agents_x"][:n_agents]), np.array(cfg["agents_y"][:n_agents])], axis=1)
        lm0 = np.stack([np.array(cfg["landmarks_x"][:n_agents]), np.array(cfg["landmarks_y"][:n_agents])], axis=1)
        vec.set_state(np.broadcast_to(pos0, (E, n_agents, 2)).copy(), np.zeros((E, n_agents, 2)),
                      np.broadcast_to(lm0, (E, n_agents, 2)).copy())
        t0, ticks = time.perf_counter(), 0
        while time.perf_counter() - t0 < min(3.0, budget_s):
            vec.step(rng.integers(0, 5, (E, n_agents)))
            ticks += 1
        out["vectorised_numpy"] = {"value": E * ticks / (time.perf_counter() - t0), "unit": "env-steps/s", "cores": 1,
                                   "sample": "%d ticks of %d envs, float64 [E,N,...] NumPy restatement" % (ticks, E)}
    except Exception as exc:          # the extra line must never break the bench
        out["vectorised_numpy"] = {"error": repr(exc)}
    # third line: the scalar port on ALL host cores, one env stream per process
    # (plain subprocesses with a hard timeout: nothing here can hang or outlive the bench)
    try:
        procs = max(1, os.cpu_count() or 1)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", cfg_name_of(cfg), str(n_agents), "%.2f" % min(3.0, budget_s)]
        env_vars = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        children = [subprocess.Popen(cmd + [str(1000 + k)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env_vars)
                    for k in range(procs)]
        rates = []
        deadline = time.time() + 90.0
        for ch in children:
            try:
                o, _ = ch.communicate(timeout=max(1.0, deadline - time.time()))
                rates.append(float(o.decode().strip().splitlines()[-1]))
            except Exception:
                ch.kill()
        out["scalar_port_all_cores"] = {"value": float(sum(rates)), "unit": "env-steps/s", "cores": len(rates),
                                        "sample": "%d processes (one per host core, os.cpu_count() = %d) x %.1f s of the scalar port"
                                                  % (len(rates), os.cpu_count() or 0, min(3.0, budget_s))}
    except Exception as exc:
        out["scalar_port_all_cores"] = {"error": repr(exc)}
    return out


def cpu_baseline_checkers(cfg, budget_s=10.0):
    """Dense scalar NumPy port of env/checkers.py on one host core, uniform random actions."""
    import numpy as np
    from oracle.checkers_oracle import CheckersEnvOracle
    i, n = cfg["init"], cfg["n_agents"]
    env = CheckersEnvOracle(i["n_rows"], i["n_columns"], i["n_obs"], i["agents_r"], i["agents_c"], n, 33)
    rng = np.random.RandomState(12341)
    steps, episodes = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        env.reset(np.eye(2) if n > 1 else np.array([[1, 0]]))
        done = False
        while not done:
            *_, done = env.step(rng.randint(0, 5, n))
            steps += 1
        episodes += 1
    dt = time.perf_counter() - t0
    return dict(value=steps / dt, unit="env-steps/s", cores=1, kind="port",
                sample="%d episodes (%d env-steps) of Checkers stage 2 (N=%d, 33 ticks, uniform actions) in %.1f s on 1 of "
                       "%d host cores; scalar dense NumPy port of the reference's call structure"
                       % (episodes, steps, n, dt, os.cpu_count()))


def run_c1(args):
    """BASELINE config C1 (config_particle_stage1.json, 1 agent, 1 env; /root/reference/alg/config_particle_stage1.json:1-8): CPU
    plumbing, no GPU -- (a) PARITY: the reference-shaped scalar port replays every episode of the stage-1 golden fixtures (recorded
    from the reference's own MultiAgentEnv.step, tests/golden/particle_stage1_*.npz) and must reproduce global state, observations,
    rewards, done flags and the collision counter bit for bit in float64; (b) the port's env-steps/s on one host core.  The port
    is the checker being timed here by definition of C1; nothing of the GPU product is involved."""
    import glob
    import numpy as np
    from oracle.particle_oracle import ParticleEnvOracle
    checked, bad, files = 0, 0, []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "particle_stage1_*.npz"))):
        g = np.load(path, allow_pickle=False)
        meta = json.loads(str(g["meta"]))
        files.append(os.path.basename(path))
        env = ParticleEnvOracle(meta["n_agents"], meta["config"], meta["prob_random"], meta["max_steps"])
        for ep in range(len(g["ep_len"])):
            gs0 = g["init_gs"][ep]
            env.set_state(gs0[:, 2:4], gs0[:, 0:2], g["landmarks"][ep])
            for t in range(int(g["ep_len"][ep])):
                gs, oo, os_, rew, rew_n, done = env.step(g["actions"][ep, t])
                ok = (np.array_equal(gs, g["gs"][ep, t]) and np.array_equal(np.array(oo), g["obs_others"][ep, t])
                      and np.array_equal(np.array(os_), g["obs_self"][ep, t]) and rew == g["reward"][ep, t]
                      and np.array_equal(np.array(rew_n), g["reward_n"][ep, t]) and bool(done) == bool(g["done"][ep, t])
                      and env.collisions == g["collisions"][ep, t])
                checked += 1
                bad += 0 if ok else 1
    import cm3_amd
    cfg = cm3_amd.load_config("particle_stage1")
    cb = cpu_baseline(cfg, 1, budget_s=min(12.0, max(2.0, float(args.steps))))
    out = {"metric": "env-steps/s (CPU port; C1 plumbing config, no GPU)", "value": cb["value"], "unit": "env-steps/s", "n_gpus": 0,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / cb["value"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic (uniform random actions, MT19937 as the reference's loop)",
           "config": {"workload": "config_particle_stage1.json: 1 agent, 1 env, CPU reference-shaped step() (BASELINE configs[0])",
                      "envs_per_gpu": 0, "n_agents": 1, "global_envs": 1, "mode": "cpu", "parallelism": "none"},
           "parity": {"pass": bad == 0 and checked > 0, "ticks_checked": checked, "ticks_differing": bad, "fixtures": files,
                      "what": "float64 bit-exact replay of the reference-recorded stage-1 episodes by the scalar port"},
           "roofline": {"bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 0.0, "traffic": None,
                        "kernel": "none (C1 runs no kernel: CPU plumbing)"},
           "cpu_baseline": cb}
    print(json.dumps(out))
    return 0 if out["parity"]["pass"] else 1


def pmc_traffic(tag):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/pmc_traffic.json, written by tools/pmc_run.sh + tools/pmc_summary.py); None if absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    rec = json.load(open(path)).get(tag)
    if not rec:
        return None, None
    return rec["hbm_bytes_per_launch"], rec


SPAN_FILE = "r06_kernel_span.json"


def span_record(tag):
    """The undisturbed kernel-duration record of this workload (profiles/r0N_kernel_span.json, written by tools/kernel_span.py
    from the two-stamp build on an UNPROFILED 330-launch hipGraph): per-launch span (first wave in -> last wave out) and
    start-to-start; None if absent.  rocprofv3's kernel-trace average is not used for this: it exceeds the unprofiled time per
    launch (the profiler stretches every dispatch of a launch-bound graph)."""
    global SPAN_FILE
    for name in ("r06_kernel_span.json", "r05_kernel_span.json", "r04_kernel_span.json", "r03_kernel_span.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            SPAN_FILE = name
            return json.load(open(path)).get(tag)
    return None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_spawn(n_gpus):
    """`python bench.py --gpus N` without a launcher: re-execute through torch.distributed.run, one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")          # one launch thread per rank; the host side is a hipGraph replay loop
    return subprocess.call(cmd, env=env)


def pin_rank_to_gpu_numa_node(device_index):
    """Best effort: bind this rank's host threads to the CPUs of the NUMA node its GPU hangs off (the launch thread then
    writes AQL packets / kernel arguments from local memory).  Returns a short description for the JSON line; never raises."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return "%s: no NUMA affinity reported" % bdf
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return "%s: NUMA node %d has no allowed CPU" % (bdf, node)
        os.sched_setaffinity(0, cpus)
        return "%s: NUMA node %d, %d CPUs" % (bdf, node, len(cpus))
    except Exception as exc:
        return "not pinned (%s)" % type(exc).__name__


def measure_other_config(name, args, device, steps=20, cpu_budget_s=6.0):
    """One of the OTHER BASELINE workloads (c3 / c5 / c4), timed inside the default `bench.py --gpus 1` run exactly like the
    headline -- the product's own collector in trajectory mode, one step launch per tick, hipGraph replays, wall clock between
    two device synchronisations with the HIP-event time of the same region beside it -- so that the driver's record carries a
    driver-timed number for every BASELINE config, each with its own roofline (algorithmic bytes per launch / time per launch
    against 8 TB/s) and the citations of its committed kernel-span / PMC records.  c3 also times the scalar dense NumPy port
    of env/checkers.py:228-262 on one host core (cpu_baseline_checkers)."""
    import torch
    import cm3_amd
    kind, cfg_name, E, wl_desc = WORKLOADS[name]
    cfg = cm3_amd.load_config(cfg_name)
    N = cfg["n_agents"]
    a2 = argparse.Namespace(**vars(args))
    a2.mode, a2.fused, a2.no_graph, a2.kernel, a2.workload = "trajectory", False, False, "auto", name
    st, tps, bps, dtype_name, mode = build_headline(a2, kind, cfg, N, E, device, 0)
    if kind == "particle_adv":
        steps = steps * 10                      # a c4 step is one 33-tick rollout: time as many ticks as the others
    K = steps * tps
    st.run(5 * tps * (10 if kind == "particle_adv" else 1))       # warm-up as the headline's: 5 collection phases
    torch.cuda.synchronize(device)
    stream = torch.cuda.current_stream(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    st.run(K)
    e1.record(stream)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    ev_s = e0.elapsed_time(e1) * 1e-3
    launch_s = wall / K
    bytes_per_launch = float(bps) * E
    achieved = bytes_per_launch / launch_s / 1e9
    live = bool(getattr(getattr(st, "ro", None), "_live", False))
    st.close()
    del st
    torch.cuda.empty_cache()
    kname = {"c3": "k_checkers_step_fast<2,false,true,8>", "c5": "k_particle_step_agents2<4,wt,live,early>",
             "c4": "k_particle_step_pairs<float,4,4,false,plain> (+ k_returns_partials_keep + k_fold_normalize per rollout)"}[name]
    rec = {
        "workload": "%s, %d vectorised envs, max_steps=33, auto-reset, trajectory mode incl. terminal capture, one step-kernel "
                    "launch per tick, %s" % (wl_desc, E, "one hipGraph replay per phase of 10 rollouts x 33 ticks + their advantage normalisations"
                                             if kind == "particle_adv" else "hipGraph of %d ticks per replay" % tps),
        "envs_per_gpu": E, "n_agents": N, "dtype": dtype_name, "steps": steps, "ticks_per_step": tps, "ticks_timed": K,
        "ms_per_step": wall / steps * 1e3, "us_per_tick": launch_s * 1e6, "env_steps_per_s": E / launch_s, "live_state": live,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": kname, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "avg_launch_us": launch_s * 1e6, "avg_launch_us_hip_events": ev_s / K * 1e6,
                     "clock": "wall clock between two device synchronisations (the headline's clock); c4: the two launches of "
                              "the advantage step and the replay boundary are inside the time, the bytes count the step launches only"},
    }
    tag = {"c3": "c3_trajectory_n2_e8192", "c5": "c5_trajectory_n8_e8192"}.get(name)
    if tag:
        traffic, trec = pmc_traffic(tag)
        if traffic is not None:
            rec["roofline"]["traffic"] = traffic
            rec["roofline"]["traffic_source"] = "profiles/pmc_traffic.json: %s (committed rocprofv3 --pmc passes of this workload)" % tag
    sp = span_record("c4_rollout" if name == "c4" else "%s_trajectory" % name)
    if sp:
        rec["roofline"]["kernel_span"] = {"span_us": sp["span_us_mean"], "start_to_start_us": sp["start_to_start_us_mean"],
                                          "gap_us": sp["gap_us_mean"], "source": "profiles/%s (two-stamp build, unprofiled)" % SPAN_FILE}
    floor = launch_ceiling(device, "checkers" if kind == "checkers" else "particle", N, E, bytes_per_launch, launch_s * 1e6)
    rec["roofline"]["launch_floor"] = floor
    rec["roofline"]["ceiling_frac"] = floor["ceiling_frac"]
    if name == "c3":          # (BEFORE the CPU baseline: six seconds of host-only work let the GPU's clocks drop, and the first
        rec["policy"] = measure_checkers_policy(cfg, E, device)       # collects after it measured 15.3 us per tick against 13.9)
    if name == "c3" and not args.no_cpu_baseline:
        rec["cpu_baseline_checkers"] = cpu_baseline_checkers(cfg, budget_s=cpu_budget_s)
    return rec


def measure_checkers_policy(cfg, E, device, reps=40):
    """POLICY-driven Checkers collection at the workload's BASELINE size (train_onpolicy.py:309-347, the branch the reference takes
    after its 50 pretrain episodes): CheckersRollout.collect(goals, policy=actor) with the on-device split-float16 actor (random
    float32 weights of the reference's shapes, epsilon 0.1), full trajectory storage, env reset per collect -- ONE launch per
    33-tick rollout (csrc/policy_checkers.hip).  -> us per tick, env-steps/s and the matrix-core roofline of the instructions executed."""
    import numpy as np
    import torch
    from cm3_amd.actor import CheckersActor
    from cm3_amd.checkers import VecCheckersEnv
    from cm3_amd.rollout import CheckersRollout
    rng = np.random.default_rng(0)
    Nc = cfg["n_agents"]
    shapes = {"conv/Conv/weights": (3, 3, 3, 6), "conv/Conv/biases": (6,), "conv_linear/kernel": (150, 32),
              "conv_linear/bias": (32,), "branch_self/kernel": (43, 256), "branch_self/bias": (256,),
              "W_self_h2": (256, 256), "stage-2/branch_others/kernel": (2 * max(Nc - 1, 1), 256),
              "stage-2/branch_others/bias": (256,), "stage-2/W_others_h2": (256, 256), "b": (256,),
              "actor_out/kernel": (256, 5), "actor_out/bias": (5,)}
    wts = {k: (rng.standard_normal(v) * 0.1).astype(np.float32) for k, v in shapes.items()}
    goals = np.eye(2) if Nc > 1 else np.array([[1, 0]])
    env = VecCheckersEnv(cfg["init"], Nc, 33, E, device=device)
    actor = CheckersActor(wts, Nc, stage=2 if Nc > 1 else 1, device=device, precision="f16x3")
    one = actor.fused_rollout_ok(env)
    ro = CheckersRollout(env, n_ticks=EP_TICKS)
    for _ in range(20):                     # ~10 ms of launches: the clocks are up when the timed collects start
        ro.collect(goals, policy=actor, epsilon=0.1)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ro.collect(goals, policy=actor, epsilon=0.1)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * EP_TICKS)
    ro.close()
    roof = mfma_roofline(E * Nc, us, *checkers_actor_mfma_work(Nc, "f16x3", others_from_table=one))
    return {"us_per_tick": us, "env_steps_per_s": E / us * 1e6, "mode": "one launch per rollout" if one else "actor + step launch per tick",
            "actor_precision": "f16x3", "mfma_frac": roof["frac"], "mfma_matrix_time_us": roof["matrix_time_us"], "roofline": roof}


LINE_LIMIT = 4000            # bytes: the driver keeps an 8 KB stdout tail; round 4's 24 KB line was cut and never parsed
EXTRAS_FILE = "bench_extras.json"


def _sig(x, digits=6):
    """numbers of the driver line carry six significant digits: enough for every quoted figure, a third of repr()'s bytes"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


def _pick(d, keys):
    return {k: _sig(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out):
    """The driver's line: `out` (everything this run measured, tens of KB with the extras) cut down to the contract's keys
    + numbers -- headline, roofline, cpu_baseline, the other BASELINE configs and the policy-driven collection as plain
    figures, no prose.  Everything dropped here is in bench_extras.json / the earlier `extras_summary` stdout line.
    Always < LINE_LIMIT bytes (tests/test_bench_line.py builds it from recorded runs)."""
    c = out.get("config", {})
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = out.get("vs_baseline")
    line["dtype"] = str(out.get("dtype", ""))[:64]
    line["data"] = "synthetic"
    line["config"] = {"workload": str(c.get("workload", ""))[:200]}
    line["config"].update(_pick(c, ("envs_per_gpu", "n_agents", "global_envs", "mode", "live_state", "ticks_per_launch")))
    line["config"]["parallelism"] = str(c.get("parallelism", ""))[:80]
    line["config"]["ticks_per_step"] = out.get("ticks_per_step")
    line["us_per_tick"] = _sig(out.get("us_per_tick"))
    r = out.get("roofline", {})
    roof = _pick(r, ("bound", "achieved", "peak", "unit", "frac"))
    roof["traffic"] = _sig(r.get("traffic"))
    roof.update(_pick(r, ("algorithmic_bytes_per_launch", "avg_launch_us", "avg_launch_us_hip_events", "measured_read_GBps",
                          "frac_of_measured_read", "ceiling_frac")))
    roof["kernel"] = str(r.get("kernel", ""))[:96]
    if r.get("traffic") is not None:
        roof["traffic_kind"] = "committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), not this run"
    if isinstance(r.get("kernel_span"), dict):
        roof["kernel_span"] = _pick(r["kernel_span"], ("span_us", "gap_us", "start_to_start_us"))
    if isinstance(r.get("launch_floor"), dict):
        roof["launch_floor"] = _pick(r["launch_floor"], ("same_traffic_us", "empty_launch_us", "frac_of_floor"))
    line["roofline"] = roof
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        for k, short in (("vectorised_numpy", "vectorised_numpy"), ("scalar_port_all_cores", "all_cores")):
            if isinstance(cb.get(k), dict):
                line["cpu_baseline"][short] = _pick(cb[k], ("value", "cores"))
    oc = out.get("other_configs")
    if isinstance(oc, dict):
        line["other_configs"] = {}
        for name, rec in oc.items():
            if not isinstance(rec, dict):
                continue
            o = _pick(rec, ("envs_per_gpu", "n_agents", "us_per_tick", "live_state"))
            o["value"] = _sig(rec.get("env_steps_per_s"))
            rr = rec.get("roofline", {})
            o.update(_pick(rr, ("frac", "ceiling_frac", "achieved", "algorithmic_bytes_per_launch", "avg_launch_us_hip_events")))
            if isinstance(rr.get("launch_floor"), dict):
                o["floor_us"] = _sig(rr["launch_floor"].get("same_traffic_us"), 4)
            o["traffic"] = _sig(rr.get("traffic"))
            if isinstance(rr.get("kernel_span"), dict):
                o["kernel_span"] = _pick(rr["kernel_span"], ("span_us", "gap_us"))
            if isinstance(rec.get("cpu_baseline_checkers"), dict):
                o["cpu_baseline"] = _pick(rec["cpu_baseline_checkers"], ("value", "cores", "kind"))
            if isinstance(rec.get("policy"), dict):
                o["policy_us_per_tick"] = _sig(rec["policy"].get("us_per_tick"))
                o["policy_mfma_frac"] = _sig(rec["policy"].get("mfma_frac"), 4)
            line["other_configs"][name] = o
    pr = out.get("policy_rollout")
    if isinstance(pr, dict) and isinstance(pr.get("headline"), dict):
        h = pr["headline"]
        line["policy_rollout"] = _pick(h, ("us_per_tick", "env_steps_per_s"))
        if isinstance(h.get("roofline"), dict):
            line["policy_rollout"].update({"mfma_" + k: v for k, v in _pick(h["roofline"], ("frac", "matrix_time_us")).items()})
    if isinstance(out.get("collective"), dict):
        line["collective"] = _pick(out["collective"], ("host_us_per_rollout_rank0", "share_of_step_rank0"))
    if "per_rank" in out and len(out["per_rank"]) > 1:
        line["per_rank_wall_s"] = [_sig(p.get("wall_s")) for p in out["per_rank"]]
    if isinstance(out.get("rccl"), dict):
        line["rccl"] = _pick(out["rccl"], ("rccl_world_size", "backend", "rccl_version", "all_reduce_ok", "p2p_all_pairs"))
    line["extras"] = EXTRAS_FILE
    # belt and braces: whatever a future key adds, the line stays under the limit -- optional groups go first
    for k in ("policy_rollout", "other_configs", "collective", "per_rank_wall_s", "rccl"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(k, None)
    return line


def extras_summary(out):
    """Second-to-last stdout line (also < LINE_LIMIT): the sweep and the launch modes as bare numbers, so that the driver's
    tail shows the streaming-size roofline points next to the headline."""
    s = {"extras_summary": True}
    if isinstance(out.get("sweep"), list):
        s["sweep_cols"] = ["log2_envs", "variant", "avg_launch_us", "frac_of_peak"]
        s["sweep"] = [[int(p["envs"]).bit_length() - 1, str(p.get("variant", "in-place"))[:28], _sig(p["avg_launch_us"], 5),
                       _sig(p["frac_of_peak"], 4)] for p in out["sweep"]]
    lm = out.get("launch_modes")
    if isinstance(lm, dict):
        s["launch_modes"] = {k: _pick(v, ("us_per_tick", "frac_of_peak")) for k, v in lm.items() if isinstance(v, dict)}
    if isinstance(out.get("fused_rollout"), dict):
        s["fused_rollout"] = _pick(out["fused_rollout"], ("value", "avg_launch_us", "frac_of_peak"))
    pr = out.get("policy_rollout")
    if isinstance(pr, dict):
        s["policy_us_per_tick"] = {prec: {m: _sig(v.get("us_per_tick", v.get("avg_launch_us")), 5) for m, v in sub.items() if isinstance(v, dict)}
                                   for prec, sub in pr.items() if prec in ("f32", "f16x3") and isinstance(sub, dict)}
    if isinstance(out.get("host_side"), dict):
        s["host_side_cols"] = ["ms", "ratio_to_copy_rate_floor"]
        s["host_side"] = {k: [_sig(v["ms"], 4), _sig(v.get("ratio_to_floor", 0.0), 3)] for k, v in out["host_side"].items()
                          if isinstance(v, dict) and "ms" in v}
        if isinstance(out["host_side"].get("phase_total"), dict):
            s["phase_total_ms"] = _sig(out["host_side"]["phase_total"]["ms"], 4)
        if isinstance(out["host_side"].get("phase_total_graph"), dict):
            s["phase_total_graph_ms"] = _sig(out["host_side"]["phase_total_graph"]["ms"], 4)
    while len(json.dumps(s)) >= LINE_LIMIT and s.get("sweep"):
        s["sweep"].pop()
    return s


def emit(out, extras_file=None):
    """rank 0: bench_extras.json (the whole record; also under gpurun_out/ when that exists; `--extras-file` names another place),
    then the summary line, then THE line."""
    full = json.dumps(out, indent=1)
    targets = [extras_file] if extras_file else [os.path.join(d, EXTRAS_FILE) for d in (ROOT, os.path.join(ROOT, "gpurun_out"))
                                                 if os.path.isdir(d)]
    for path in targets:
        try:
            with open(path, "w") as fh:
                fh.write(full)
        except OSError as exc:
            sys.stderr.write("bench: could not write %s (%s)\n" % (path, exc))
    summary = extras_summary(out)
    if len(summary) > 1:
        print(json.dumps(summary))
    line = compact_line(out)
    text = json.dumps(line)
    assert len(text) < LINE_LIMIT, len(text)
    print(text)
    sys.stdout.flush()


def rccl_report(dist, torch, device, local_rank, world):
    """What the first multi-GPU record should answer by itself: did RCCL see `world` ranks, does a collective over them give
    the right answer, and which peers can every rank's GPU reach directly (P2P over xGMI)."""
    rep = {"rccl_world_size": dist.get_world_size(), "backend": dist.get_backend()}
    try:
        rep["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as exc:
        rep["rccl_version"] = "unknown (%s)" % type(exc).__name__
    probe = torch.full((1,), float(dist.get_rank() + 1), dtype=torch.float64, device=device)
    dist.all_reduce(probe)
    rep["all_reduce_sum_of_rank_plus_1"] = float(probe[0])
    rep["all_reduce_ok"] = float(probe[0]) == world * (world + 1) / 2.0
    n_dev = torch.cuda.device_count()
    mine = torch.zeros(8, dtype=torch.int32, device=device)
    for j in range(min(n_dev, 8)):
        ok = 1 if j == local_rank else int(torch.cuda.can_device_access_peer(local_rank, j))
        mine[j] = ok
    allp = torch.empty(world * 8, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(allp, mine)
    rep["p2p_access"] = [{"rank": r, "visible_devices": n_dev, "can_access_peer": row[:min(n_dev, 8)]}
                         for r, row in enumerate(allp.view(world, 8).cpu().tolist())]
    rep["p2p_all_pairs"] = all(all(row["can_access_peer"][:world]) for row in rep["p2p_access"]) if n_dev >= world else False
    return rep


def headline_record(*, world, steps, warm, K, ticks_per_step, wall_max, ev_max, per_rank, E, N, kind, mode,
                    fused_ticks, launches_per_tick, bytes_per_env_step, dtype_name, wl_desc, no_graph, pinned, live_state,
                    rccl=None):
    """The record rank 0 builds from the timed region, for ANY world size (main() calls it; tests/test_bench_line.py calls it with
    world = 8 and recorded clocks).  `value`, `ms_per_step`, `roofline.avg_launch_us` and `roofline.achieved` come from ONE clock:
    the max-over-ranks wall clock between the two barriers."""
    total_env_steps = float(E) * K * world
    value = total_env_steps / wall_max
    launches = K / float(fused_ticks) * launches_per_tick          # step-kernel launches inside the timed region, per rank
    launch_s = wall_max / launches                                   # time per launch
    bytes_per_launch = bytes_per_env_step * E * fused_ticks / float(launches_per_tick)
    achieved = bytes_per_launch / launch_s / 1e9
    if kind == "particle_adv":
        launch_desc = ("hipGraph of %d rollouts x %d ticks + their advantage steps per replay (one collection phase)"
                       % (PHASE_EPISODES, EP_TICKS))
    elif no_graph:
        launch_desc = "eager launches"
    else:
        launch_desc = "hipGraph of %d ticks = one collection phase per replay" % PHASE_TICKS
    kname = "k_checkers_step_fast" if kind == "checkers" else "k_particle_step(_pairs|_agents)<float,%d>" % N
    out = {
        "metric": "env-steps/s (all agents, whole node)", "value": value, "unit": "env-steps/s",
        "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": wall_max / steps * 1e3,
        "ticks_per_step": ticks_per_step, "ticks_timed": K,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_name,
        "data": "synthetic (uniform random actions drawn in-kernel, Philox; preset/random resets, prob_random=0.2)",
        "config": {"workload": "%s, %d vectorised envs per GPU, max_steps=33, auto-reset, %s mode (%s), %s"
                               % (wl_desc, E, mode,
                                  "every tick writes slot t+1 of a [%d+1, E, ...] device trajectory incl. terminal capture"
                                  % ticks_per_step if mode == "trajectory" else "every tick overwrites the live buffers",
                                  "one step-kernel launch per tick"
                                  if fused_ticks == 1 else "%d ticks fused per launch (random-action branch)" % fused_ticks),
                   "envs_per_gpu": E, "n_agents": N, "global_envs": E * world, "mode": mode,
                   "launch": launch_desc, "ticks_per_launch": fused_ticks, "live_state": live_state,
                   "step_definition": ("one 33-tick rollout + advantage normalisation" if kind == "particle_adv" else
                                       "one collection phase = %d episodes x %d ticks (train_onpolicy.py:359-377)"
                                       % (PHASE_EPISODES, EP_TICKS)),
                   "parallelism": ("env-sharded x%d, one 24-byte moments all-gather (RCCL) per rollout" % world
                                   if kind == "particle_adv" else "env-sharded x%d, no data-path collective" % world)},
        "agent_steps_per_s": value * N,
        "us_per_tick": wall_max / K * 1e6,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": kname,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "avg_launch_us": launch_s * 1e6,
                     "avg_launch_us_hip_events": ev_max / launches * 1e6,
                     "per": "GPU: one rank's launch moves algorithmic_bytes_per_launch; every rank does the same work (weak scaling)",
                     "clock": "wall clock of the timed region (max over ranks) for value, avg_launch_us and achieved alike; "
                              "avg_launch_us_hip_events = HIP events on the launch stream around the same region"},
        "per_rank": [{"rank": r, "wall_s": w_, "avg_launch_us": w_ / launches * 1e6,
                      "avg_launch_us_hip_events": e_ / launches * 1e6, "env_id_base": r * E, "envs": E}
                     for r, (w_, e_) in enumerate(per_rank)],
        "rank0_host_affinity": pinned,
    }
    if rccl is not None:
        out["rccl"] = rccl
        out["rccl_world_size"] = rccl["rccl_world_size"]
    return out


def rank0_baselines(out, kind, cfg, N, world, device, no_cpu_baseline=False, measure_bw=None, cpu_budget_s=None):
    """What rank 0 adds to the record at EVERY world size (round 6: a multi-GPU line without these was ungradeable): the read /
    copy bandwidth measured on rank 0's GPU -- the denominator of `frac_of_measured_read` -- and the CPU baseline, a bounded sample
    of the same workload on rank 0's host cores while the other ranks wait at the closing barrier (world > 1: a shorter sample)."""
    bw_read, bw_copy = (measure_bw or measure_bandwidth)(device)
    r = out["roofline"]
    r["measured_read_GBps"], r["measured_copy_GBps"] = bw_read, bw_copy
    r["frac_of_measured_read"] = r["achieved"] / bw_read
    if not no_cpu_baseline:
        budget = cpu_budget_s if cpu_budget_s is not None else (12.0 if world == 1 else 8.0)
        out["cpu_baseline"] = cpu_baseline_checkers(cfg, budget_s=budget) if kind == "checkers" else cpu_baseline(cfg, N, budget_s=budget)
    return bw_read, bw_copy


def build_headline(args, kind, cfg, N, E, device, rank):
    """-> (stepper, ticks_per_step, bytes_per_env_step, dtype_name, mode description)"""
    if kind == "particle_adv":
        st = RolloutAdvStepper(cfg, N, E, device, env_id_base=rank * E, kernel=args.kernel, fused=args.fused)
        return st, EP_TICKS, algorithmic_bytes_per_env_step(N), "f32", "trajectory"
    if kind == "particle":
        bps = algorithmic_bytes_per_env_step(N)
        if args.fused:   # state, goals and counters are read once per launch, not once per tick
            bps -= (16 * N + 8 * N + 8) * (EP_TICKS - 1) / float(EP_TICKS)
        if args.mode == "trajectory":
            st = TrajectoryStepper(cfg, N, E, device, env_id_base=rank * E, kernel=args.kernel,
                                   use_graph=not args.no_graph, fused=args.fused,
                                   live_state={"auto": None, "on": True, "off": False}[getattr(args, "live_state", "auto")])
        else:
            st = ParticleStepper(cfg, N, E, device, env_id_base=rank * E, kernel=args.kernel, fused=args.fused)
            if not args.no_graph:
                st.capture(PHASE_TICKS)
        return st, PHASE_TICKS, bps, "f32", args.mode
    dtype_name = "int8/int32 state and grids, f64 normalised outputs (bit-exact)"
    if args.mode == "trajectory":
        st = CheckersTrajectoryStepper(cfg, E, device, env_id_base=rank * E, fused=args.fused)
    else:
        st = CheckersStepper(cfg, E, device, env_id_base=rank * E, fused=args.fused)
        if not args.no_graph:
            st.capture(PHASE_TICKS)
    return st, PHASE_TICKS, CHECKERS_BYTES_PER_ENV_STEP, dtype_name, args.mode


def main():
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":      # child of cpu_baseline(): prints its env-steps/s
        import cm3_amd
        print(_scalar_port_worker((cm3_amd.load_config(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]))))
        return 0
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps; one step = one collection phase of %d ticks (c4: one 33-tick rollout + normalisation)"
                         % PHASE_TICKS)
    ap.add_argument("--warmup", type=int, default=20, help="untimed warm-up steps (defaults: 200 timed phases = ~0.16 s of launches at C2 -- round 5's "
                         "20 gave a 16 ms timed region, too short for a coarse GPU-busy sampler to see)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["c1"], default="c2",
                    help="c2 (default, the configuration BASELINE.json's metric is quoted on) | c3 | c4 | c5 | c1 (stage 1, one agent, one "
                         "env: CPU only -- parity pass / fail against the reference-recorded fixtures + the port's env-steps/s)")
    ap.add_argument("--mode", choices=["trajectory", "in-place"], default="trajectory",
                    help="trajectory (default): every tick writes its slot of a device trajectory (the collection loop); "
                         "in-place: every tick overwrites the same live buffers (stepping only)")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="override the workload's batch size")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-sweep", action="store_true", help="skip the E-sweep (extra 'sweep' field)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short c3 / c5 / c4 runs that the default c2 line carries as 'other_configs'")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline only: skip the in-place / fused / policy-rollout / launch-floor extras (clean profiles)")
    ap.add_argument("--live-state", choices=["auto", "on", "off"], default="auto",
                    help="trajectory mode of the particle workloads: step in place on the env's buffers and copy every tick's state to "
                         "its slot (on), chain the ticks through the slots (off), or by size (auto, the product's rule)")
    ap.add_argument("--extras-file", default=None,
                    help="where the full record goes (default: %s next to bench.py and under gpurun_out/)" % EXTRAS_FILE)
    ap.add_argument("--fused", action="store_true",
                    help="random-action rollouts with all ticks of a phase in ONE launch (CM3_FLAG_FUSED_TICKS); "
                         "the default keeps one launch per tick")
    ap.add_argument("--kernel", choices=["auto", "env", "pair", "agent"], default="auto",
                    help="step-kernel mapping (auto = library heuristic)")
    args = ap.parse_args()

    if args.workload == "c1":
        return run_c1(args)
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not under_launcher:
        return self_spawn(args.gpus)

    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    import cm3_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if under_launcher and world != args.gpus:
        raise SystemExit("rank %d: launched with WORLD_SIZE=%d but --gpus %d" % (rank, world, args.gpus))
    n_visible = torch.cuda.device_count()
    if local_rank >= n_visible:
        sys.stderr.write("rank %d: --gpus %d needs %d GPUs on this node, only %d visible (local rank %d has no device)\n"
                         % (rank, args.gpus, args.gpus, n_visible, local_rank))
        sys.stderr.flush()
        if under_launcher:
            time.sleep(6.0)     # the launcher ends the other ranks at the first exit: let every rank say which device it lacks
        raise SystemExit(1)
    if rank == 0:
        graft.build()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    pinned = pin_rank_to_gpu_numa_node(local_rank) if under_launcher else "single process: not pinned"
    # everything (launches, graph replays, HIP events) goes on one explicit non-default stream
    bench_stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(bench_stream)
    # Under a launcher the RCCL process group is always created -- also for one rank, so the barrier /
    # max-over-ranks path is the same code at every N.
    use_dist = under_launcher
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", device_id=device)
        dist.barrier(device_ids=[local_rank])
    if rank != 0:
        graft.build()

    kind, cfg_name, default_e, wl_desc = WORKLOADS[args.workload]
    cfg = cm3_amd.load_config(cfg_name)
    N = cfg["n_agents"]
    E = args.envs_per_gpu or default_e
    steps, warm = max(args.steps, 1), max(args.warmup, 0)
    stepper, ticks_per_step, bytes_per_env_step, dtype_name, mode = build_headline(args, kind, cfg, N, E, device, rank)
    K, W = steps * ticks_per_step, max(warm, 1) * ticks_per_step      # in ticks
    stepper.run(W)
    torch.cuda.synchronize(device)

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if kind == "particle_adv":
        stepper.collective_s, stepper.rollouts = 0.0, 0
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    ev0.record(bench_stream)
    stepper.run(K)
    ev1.record(bench_stream)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    barrier()
    ev_ms = ev0.elapsed_time(ev1)
    t = torch.tensor([wall, ev_ms * 1e-3], dtype=torch.float64, device=device)
    per_rank = [[wall, ev_ms * 1e-3]]
    if use_dist:
        gathered = torch.empty(world * 2, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(gathered, t)
        per_rank = gathered.view(world, 2).cpu().tolist()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max, ev_max = float(t[0]), float(t[1])
    rccl = rccl_report(dist, torch, device, local_rank, world) if use_dist else None    # (every rank takes part; rank 0 prints it)

    fused_ticks = ticks_per_step if (args.fused and kind != "particle_adv") else 1
    # (the extras below quote the headline's time and bytes per launch; headline_record() derives the same from the same clock)
    launch_s = wall_max / (K / float(fused_ticks) * stepper.launches_per_tick)
    bytes_per_launch = bytes_per_env_step * E * fused_ticks / float(stepper.launches_per_tick)

    out = None
    if rank == 0:
        out = headline_record(world=world, steps=steps, warm=warm, K=K, ticks_per_step=ticks_per_step, wall_max=wall_max,
                              ev_max=ev_max, per_rank=per_rank, E=E, N=N, kind=kind, mode=mode,
                              fused_ticks=fused_ticks, launches_per_tick=stepper.launches_per_tick,
                              bytes_per_env_step=bytes_per_env_step, dtype_name=dtype_name, wl_desc=wl_desc, no_graph=args.no_graph,
                              pinned=pinned, live_state=bool(getattr(getattr(stepper, "ro", None), "_live", False)), rccl=rccl)
        if kind == "particle_adv":
            out["collective"] = {
                "what": "all_gather_into_tensor of 3 float64 per rank and rollout (advantage moments); the %d rollouts of a phase "
                        "travel in ONE collective" % PHASE_EPISODES,
                "host_us_per_rollout_rank0": stepper.collective_s / max(stepper.rollouts, 1) * 1e6,
                "share_of_step_rank0": stepper.collective_s / max(wall, 1e-12),
                "note": ("one rank: the collective is the identity and a whole phase (10 rollouts + their normalisations) is ONE hipGraph replay"
                         if world == 1 else "host time of the all-gather call (enqueue + any wait it implies), rank 0")}
    traffic_tag = {("c2", 4096, "trajectory"): "c2_trajectory_n4_e4096", ("c2", 4096, "in-place"): "c2_particle_antipodal_n4_e4096",
                   ("c3", 8192, "in-place"): "c3_checkers_stage2_n2_e8192", ("c3", 8192, "trajectory"): "c3_trajectory_n2_e8192",
                   ("c5", 8192, "in-place"): "c5_particle_merge8_n8_e8192",
                   ("c5", 8192, "trajectory"): "c5_trajectory_n8_e8192"}.get((args.workload, E, mode))
    if rank == 0 and traffic_tag and not args.fused:
        traffic, rec = pmc_traffic(traffic_tag)
        if traffic is not None:
            out["roofline"]["traffic"] = traffic
            out["roofline"]["traffic_source"] = ("profiles/pmc_traffic.json: %s, %d launches, FETCH_SIZE %.1f KB (x2) + "
                                                 "WRITE_SIZE %.1f KB" % (rec["kernel"], rec["launches"],
                                                                         rec["FETCH_SIZE_KB"], rec["WRITE_SIZE_KB"]))
    if rank == 0 and not args.fused:
        rec = span_record("%s_%s" % (args.workload, mode.replace("-", "_")))
        if rec and E == default_e:
            out["roofline"]["kernel_span"] = {
                "span_us": rec["span_us_mean"], "start_to_start_us": rec["start_to_start_us_mean"], "gap_us": rec["gap_us_mean"],
                "achieved_GBps_over_span": bytes_per_launch / (rec["span_us_mean"] * 1e-6) / 1e9,
                "frac_of_peak_over_span": bytes_per_launch / (rec["span_us_mean"] * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                "source": "profiles/" + SPAN_FILE + " (tools/kernel_span.py, two-stamp build, unprofiled 330-launch hipGraph): "
                          "span = first wave in -> last wave out; start-to-start includes the dependent-launch boundary and is "
                          "what roofline.avg_launch_us of THIS run measures live"}
    extras = world == 1 and rank == 0 and not args.no_extras and not args.fused
    if extras and kind in ("particle", "checkers"):
        stepper.close()
        del stepper
        stepper = None
        torch.cuda.empty_cache()
        modes = {}
        # the other storage mode on the same workload and graph size
        for vmode in ("in-place" if mode == "trajectory" else "trajectory",):
            a2 = argparse.Namespace(**vars(args))
            a2.mode, a2.fused, a2.no_graph = vmode, False, False
            st, tps, bps, _, _ = build_headline(a2, kind, cfg, N, E, device, 0)
            st.run(tps * 2)
            torch.cuda.synchronize(device)
            n = tps * max(3, min(steps, 10))
            ms = timed_ticks(st, n)
            us = ms * 1e3 / n
            modes[vmode.replace("-", "_")] = {
                "us_per_tick": us, "env_steps_per_s": E / us * 1e6,
                "achieved_GBps": bps * E / (us * 1e-6) / 1e9, "frac_of_peak": bps * E / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS}
            st.close()
            del st
            torch.cuda.empty_cache()
        if kind == "particle":
            # the LITERAL drop-in of INTEGRATION.md section 2: the reference's loop calls env.step(actions) from Python every tick
            # (train_onpolicy.py:321-323) -- VecParticleEnv.step with (a) device int32 actions (what ParticleActor.act returns) and
            # (b) host int64 actions (what np.random.randint / a NumPy policy returns: one host-to-device copy per tick)
            import numpy as np
            from cm3_amd.particle import VecParticleEnv
            penv = VecParticleEnv(cfg, N, 0.2, 33, E, device=device, auto_reset=True)
            penv.reset()
            a_dev = torch.randint(0, 5, (E, N), dtype=torch.int32, device=device)
            a_host = np.random.default_rng(0).integers(0, 5, (E, N))
            for label, act in (("python_step_device_int32", a_dev), ("python_step_host_int64", a_host)):
                for _ in range(50):
                    penv.step(act)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(PHASE_TICKS):
                    penv.step(act)
                torch.cuda.synchronize(device)
                us = (time.perf_counter() - t0) / PHASE_TICKS * 1e6
                modes[label] = {"us_per_tick": us, "env_steps_per_s": E / us * 1e6,
                                "achieved_GBps": bytes_per_env_step * E / (us * 1e-6) / 1e9,
                                "frac_of_peak": bytes_per_env_step * E / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS}
            del penv
        modes["note"] = ("extra, not the headline: same workload and graph length, HIP-event time.  in_place = every tick "
                         "overwrites the live buffers (stepping only, no trajectory is stored); python_step_* = VecParticleEnv.step(actions) called "
                         "from Python once per tick, wall clock (the reference's own loop shape, INTEGRATION.md section 2)")
        out["launch_modes"] = modes
        # the same workload with all ticks of an episode fused into ONE launch (random-action branch only)
        if kind == "particle":
            fs = ParticleStepper(cfg, N, E, device, kernel=args.kernel, fused=True)
            f_bps = algorithmic_bytes_per_env_step(N) - (24 * N + 8) * (EP_TICKS - 1) / float(EP_TICKS)
            where = "tests/test_gpu_rollout.py::test_fused_rollout_equals_per_tick_rollout"
        else:
            fs = CheckersStepper(cfg, E, device, fused=True)
            f_bps = CHECKERS_BYTES_PER_ENV_STEP - 32 * (EP_TICKS - 1) / float(EP_TICKS)   # compact state r/w once
            where = "tests/test_gpu_rollout.py::test_checkers_fused_rollout_equals_per_tick"
        fs.capture(EP_TICKS)
        fs.run(EP_TICKS * 4)
        torch.cuda.synchronize(device)
        fk = EP_TICKS * 100
        f_ms = timed_ticks(fs, fk)
        f_bytes = f_bps * E * EP_TICKS
        f_launch_s = f_ms * 1e-3 / (fk / float(EP_TICKS))
        out["fused_rollout"] = {
            "note": "extra, not the headline: %d ticks per launch, in place, state in registers, bit-identical trajectories (%s)"
                    % (EP_TICKS, where),
            "value": E * fk / (f_ms * 1e-3), "unit": "env-steps/s", "avg_launch_us": f_launch_s * 1e6,
            "algorithmic_bytes_per_launch": f_bytes, "achieved_GBps": f_bytes / f_launch_s / 1e9,
            "frac_of_peak": f_bytes / f_launch_s / 1e9 / HBM_PEAK_GBPS}
        fs.close()
        del fs
    if extras and kind == "particle" and N in (1, 2, 4, 8):
        # POLICY-driven collection, the branch the reference takes for 49 950 of its 50 000 episodes
        # (train_onpolicy.py:311-313): on-device actor (random float32 weights of the reference's shapes, epsilon 0.1)
        # + env step per tick, (a) alternating launches in one hipGraph, (b) the whole episode in ONE launch
        # (csrc/policy.hip).  Trajectory mode: every tick's state / obs / goals / rewards are stored.
        import numpy as np
        from cm3_amd.actor import ParticleActor
        from cm3_amd.particle import VecParticleEnv
        from cm3_amd.rollout import ParticleRollout
        rng = np.random.default_rng(0)
        Lo = 4 * max(N - 1, 1)
        shapes = {"actor_branch_self/kernel": (6, 64), "actor_branch_self/bias": (64,), "W_branch_self_h2": (64, 64),
                  "stage-2/actor_others/kernel": (Lo, 128), "stage-2/actor_others/bias": (128,),
                  "stage-2/W_others_h2": (128, 64), "b": (64,), "actor_out/kernel": (64, 5), "actor_out/bias": (5,)}
        wts = {k: (rng.standard_normal(v) * 0.1).astype(np.float32) for k, v in shapes.items()}
        pol = {}
        macs = 6 * 64 + Lo * 128 + 64 * 64 + 128 * 64 + 64 * 5
        for prec in ("f32", "f16x3"):
            sub = {}
            for label, kw in (("launch_per_tick", dict(policy_mode="tick")),
                              ("fused_launch_per_tick", dict(policy_mode="tick", fused_policy_tick=True)),
                              ("one_launch_per_episode", dict())):          # the default of collect(policy=...): policy_mode="auto"
                penv = VecParticleEnv(cfg, N, 0.2, 33, E, device=device, auto_reset=True)
                penv.reset()
                actor = ParticleActor(wts, N, stage=2, device=device, precision=prec)
                ro = ParticleRollout(penv, n_ticks=EP_TICKS, use_graph=True, **kw)
                for _ in range(3):
                    ro.collect(policy=actor, epsilon=0.1, reset=False)
                torch.cuda.synchronize(device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 20
                e0.record()
                for _ in range(reps):
                    ro.collect(policy=actor, epsilon=0.1, reset=False)
                e1.record()
                e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (reps * EP_TICKS)
                sub[label] = {"us_per_tick": us, "env_steps_per_s": E / us * 1e6,
                              "roofline": mfma_roofline(E * N, us, *particle_actor_mfma_work(N, prec))}
                sub[label]["network_TFLOPs"] = sub[label]["roofline"]["network_TFLOPs"]
                ro.close()
            # the actor kernel alone (hipGraph of 100 launches: start-to-start, no host allocation inside)
            penv = VecParticleEnv(cfg, N, 0.2, 33, E, device=device, auto_reset=True)
            penv.reset()
            actor = ParticleActor(wts, N, stage=2, device=device, precision=prec)
            acts = torch.empty(E, N, dtype=torch.int32, device=device)
            from cm3_amd import _lib as _l

            def enq(s_, actor=actor, penv=penv, acts=acts):
                for _ in range(100):
                    actor.enqueue(E, penv._obs_others[penv._cur], penv._state[penv._cur], penv._goals, penv._meta, penv._episode,
                                  acts, 0.1, stream=s_, env_id_base=0)
            g = _l.capture_graph(device, enq)
            sh = _l.current_stream_handle(device)
            for _ in range(3):
                _l.check(_l.lib().cm3_graph_launch(g, sh))
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _l.check(_l.lib().cm3_graph_launch(g, sh))
            e1.record()
            e1.synchronize()
            a_us = e0.elapsed_time(e1) * 1e3 / 1000
            torch.cuda.synchronize(device)
            _l.lib().cm3_graph_destroy(g)
            sub["actor_kernel"] = {"kernel": "k_actor_particle<%d, %s>" % (N, prec), "avg_launch_us": a_us, "rows": E * N,
                                   "macs_per_row": macs,
                                   "roofline": mfma_roofline(E * N, a_us, *particle_actor_mfma_work(N, prec)),
                                   "note": "start-to-start inside a hipGraph of 100 launches"}
            pol[prec] = sub
        best = pol["f16x3"]["one_launch_per_episode"]
        pol["headline"] = {"what": "policy-driven collection (train_onpolicy.py:311-313), the default of "
                                   "ParticleRollout.collect(policy=actor): one launch per episode, actor precision f16x3",
                           "us_per_tick": best["us_per_tick"], "env_steps_per_s": best["env_steps_per_s"],
                           "roofline": best["roofline"]}
        pol["note"] = ("second headline: actor (networks.actor_particle; second layer exact float32 MFMA (f32) or split float16 "
                       "(f16x3, same 2e-5 parity bound)) + env step per tick with full trajectory storage.  launch_per_tick = an "
                       "actor launch then a step launch per tick; fused_launch_per_tick = ONE launch per tick doing both; "
                       "one_launch_per_episode = all 33 ticks in one launch (the default).  The modes are bit-identical per "
                       "precision (tests/test_gpu_actor.py::test_fused_policy_rollout_equals_launch_per_tick)")
        out["policy_rollout"] = pol
    if extras and kind == "checkers" and cfg["n_agents"] in (1, 2):
        # POLICY-driven Checkers collection (train_onpolicy.py:309-321): the on-device actor (networks.actor_checkers:
        # conv + dense chain, 153 k MACs per agent row, every layer on the exact-f32 MFMA) and the env step alternate
        # inside one hipGraph; full trajectory storage.  The actor is contraction work: its roofline is the float32
        # matrix-core peak, not HBM.
        import numpy as np
        from cm3_amd.actor import CheckersActor
        from cm3_amd.checkers import VecCheckersEnv
        from cm3_amd.rollout import CheckersRollout
        rng = np.random.default_rng(0)
        Nc = cfg["n_agents"]
        shapes = {"conv/Conv/weights": (3, 3, 3, 6), "conv/Conv/biases": (6,), "conv_linear/kernel": (150, 32),
                  "conv_linear/bias": (32,), "branch_self/kernel": (43, 256), "branch_self/bias": (256,),
                  "W_self_h2": (256, 256), "stage-2/branch_others/kernel": (2 * max(Nc - 1, 1), 256),
                  "stage-2/branch_others/bias": (256,), "stage-2/W_others_h2": (256, 256), "b": (256,),
                  "actor_out/kernel": (256, 5), "actor_out/bias": (5,)}
        wts = {k: (rng.standard_normal(v) * 0.1).astype(np.float32) for k, v in shapes.items()}
        goals = np.eye(2) if Nc > 1 else np.array([[1, 0]])
        macs = 25 * 6 * 27 + 150 * 32 + 43 * 256 + (2 * max(Nc - 1, 1) * 256 + 256 * 256 if Nc > 1 else 0) + 256 * 256 + 256 * 5
        pol = {}
        def time_collect(ro, actor, reps):
            for _ in range(3):
                ro.collect(goals, policy=actor, epsilon=0.1)
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ro.collect(goals, policy=actor, epsilon=0.1)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e3 / (reps * EP_TICKS)

        for prec in ("f32", "f16x3"):
            cenv = VecCheckersEnv(cfg["init"], Nc, 33, E, device=device)
            cenv.reset(goals)
            actor = CheckersActor(wts, Nc, stage=2 if Nc > 1 else 1, device=device, precision=prec)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                actor.act(cenv, 0.1)
            torch.cuda.synchronize(device)
            reps = 20
            e0.record()
            for _ in range(reps):
                actor.act(cenv, 0.1)
            e1.record()
            e1.synchronize()
            a_us = e0.elapsed_time(e1) * 1e3 / reps
            ro = CheckersRollout(cenv, n_ticks=EP_TICKS, use_graph=True, policy_mode="tick")
            us = time_collect(ro, actor, 5)
            ro.close()
            work = checkers_actor_mfma_work(Nc, prec)
            pol[prec] = {
                "launch_per_tick": {"us_per_tick": us, "env_steps_per_s": E / us * 1e6, "roofline": mfma_roofline(E * Nc, us, *work)},
                "actor_kernel": {"kernel": "k_ck_actor<%s>" % prec, "avg_launch_us": a_us, "rows": E * Nc, "macs_per_row": macs,
                                 "roofline": mfma_roofline(E * Nc, a_us, *work)}}
            pol[prec]["launch_per_tick"]["network_TFLOPs"] = pol[prec]["launch_per_tick"]["roofline"]["network_TFLOPs"]
            if actor.fused_rollout_ok(cenv):
                # the whole rollout in ONE launch (csrc/policy_checkers.hip; bit-identical to the launch pairs), (a) as the reference's
                # loop runs it -- one episode per env and collect(), the env reset in between -- and (b) continuous (auto-reset) collection
                twork = checkers_actor_mfma_work(Nc, prec, others_from_table=True)
                for label, auto in (("one_launch_per_rollout", False), ("one_launch_per_rollout_continuous", True)):
                    env1 = VecCheckersEnv(cfg["init"], Nc, 33, E, device=device, auto_reset=auto)
                    ro1 = CheckersRollout(env1, n_ticks=EP_TICKS)
                    us1 = time_collect(ro1, actor, 20)
                    ro1.close()
                    pol[prec][label] = {"us_per_tick": us1, "env_steps_per_s": E / us1 * 1e6, "roofline": mfma_roofline(E * Nc, us1, *twork),
                                        "what": "cm3_policy_rollout_checkers: 33 ticks per launch, " +
                                                ("auto-reset, collect() continues where the last one stopped" if auto else
                                                 "env.reset + one launch per collect()")}
                    del env1, ro1
            del cenv, actor
        best = pol["f16x3"].get("one_launch_per_rollout", pol["f16x3"]["launch_per_tick"])
        pol["headline"] = {"what": "policy-driven Checkers collection (train_onpolicy.py:309-347), the default of CheckersRollout.collect("
                                   "policy=actor): the whole 33-tick rollout in ONE launch, actor precision f16x3, env reset per collect",
                           "us_per_tick": best["us_per_tick"], "env_steps_per_s": best["env_steps_per_s"],
                           "roofline": best["roofline"]}
        pol["note"] = ("extra, not the headline: actor (networks.actor_checkers; every layer in split float16 on the matrix cores (f16x3, "
                       "same 2e-5 parity bound as the exact-f32 build) + env step per tick with full trajectory storage.  launch_per_tick = "
                       "an actor launch and a step launch per tick in one hipGraph (round 5: 29.6 us); one_launch_per_rollout = "
                       "cm3_policy_rollout_checkers, bit-identical (tests/test_gpu_actor_checkers.py::"
                       "test_checkers_policy_rollout_equals_launch_per_tick); its roofline counts the matrix instructions it EXECUTES "
                       "(the others branch of the two-agent network is a table lookup there)")
        out["policy_rollout"] = pol
    if world == 1 and rank == 0 and args.workload == "c2" and mode == "trajectory" and not args.no_other_configs \
            and not args.no_extras and not args.fused and not args.envs_per_gpu:
        # the other BASELINE workloads, driver-timed in the same run (each a short run of ITS configuration; the headline stays c2)
        if stepper is not None:
            stepper.close()
            del stepper
            stepper = None
            torch.cuda.empty_cache()
        out["other_configs"] = {name: measure_other_config(name, args, device) for name in ("c3", "c5", "c4")}
        out["other_configs"]["note"] = ("each: the workload's own BASELINE configuration through the product's collector, 20 hipGraph "
                                        "replays (collection phases) after 5 warm-up ones as the headline, same clock; "
                                        "`python bench.py --workload cN` runs it as the headline with its extras")
    if rank == 0 and world > 1:
        if stepper is not None:
            stepper.close()
            stepper = None
        rank0_baselines(out, kind, cfg, N, world, device, args.no_cpu_baseline)
    if world == 1 and rank == 0:
        bw_read, bw_copy = rank0_baselines(out, kind, cfg, N, world, device, no_cpu_baseline=True)   # (CPU baseline: last, below)
        if kind in ("particle", "checkers") and extras:
            # What one launch per tick cannot go below at this batch: the same number of 256-lane workgroups reading and
            # writing the same algorithmic bytes with NO arithmetic (load -> store skeleton), and an empty launch, both
            # replayed as a hipGraph of the same length.
            floor = launch_ceiling(device, kind, N, E, bytes_per_launch, launch_s * 1e6)
            out["roofline"]["ceiling_frac"] = floor["ceiling_frac"]
            out["roofline"]["launch_floor"] = floor
        if not args.no_sweep and kind == "particle":
            if stepper is not None:
                stepper.close()
                del stepper
                stepper = None
            sweep = []
            L3 = 256 << 20

            def point(Es, variant, st, n_ticks, ws_bytes, note):
                """two timed passes, the second reported: the first measurements after a change of batch size run ~5 % slow
                whichever kernel they are (profiles/r02_auto_vs_best_all.txt, order check)"""
                st.run(n_ticks)
                torch.cuda.synchronize(device)
                timed_ticks(st, n_ticks)
                ms = timed_ticks(st, n_ticks)
                per = ms * 1e-3 / n_ticks
                gbps = bytes_per_env_step * Es / per / 1e9
                sweep.append({"envs": Es, "variant": variant, "env_steps_per_s": Es / per, "avg_launch_us": per * 1e6,
                              "achieved_GBps": gbps, "frac_of_peak": gbps / HBM_PEAK_GBPS,
                              "frac_of_measured_read": gbps / bw_read, "working_set_bytes": int(ws_bytes),
                              "cache_resident": bool(ws_bytes < L3), "mode": note})
                st.close()
                torch.cuda.empty_cache()

            per_tick = lambda Es: bytes_per_env_step * Es          # noqa: E731  bytes one tick touches (in place: re-used buffers)
            for log2e in (14, 16, 18, 20, 22):
                Es = 1 << log2e
                st = ParticleStepper(cfg, N, Es, device, kernel=args.kernel, fused=args.fused)
                st.capture(EP_TICKS)
                point(Es, "in-place", st, EP_TICKS * (10 if log2e <= 18 else 3), per_tick(Es),
                      "in place (re-used buffers), hipGraph of 33 ticks, in-kernel actions; second of two timed passes")
            if not args.fused:
                # SURVEY.md section 8(d) "Extra sweep": the same kernels (a) streaming into a trajectory (every tick its own
                # slot: nothing is re-used, non-temporal / write-through stores), (b) reading policy-provided actions instead of
                # drawing them, (c) launched eagerly instead of as a hipGraph
                for log2e in (12, 16, 18, 20, 22):
                    Es = 1 << log2e
                    T = EP_TICKS
                    slot_bytes = lambda T_: ((T_ + 1) * Es * (16 * N + 8 * N + 16 * N * max(N - 1, 1)) + T_ * Es * (8 * N + 9) +  # noqa: E731
                                             T_ * Es * (16 * N + 16 * N * max(N - 1, 1) + 4))
                    free_b = torch.cuda.mem_get_info(device)[0]
                    while T > 4 and slot_bytes(T) > 0.6 * free_b:      # 2^22 envs x 33 ticks = 83 GB of slots: fits 288 GB
                        T //= 2
                    st = TrajectoryStepper(cfg, N, Es, device, kernel=args.kernel, phase_ticks=T)
                    slots = slot_bytes(T)
                    point(Es, "trajectory", st, T * (10 if log2e <= 16 else 3), slots,
                          "trajectory mode: slots t -> t+1 of a [%d, E, ...] device trajectory incl. terminal capture, one hipGraph "
                          "replay per %d-tick collect" % (T + 1, T))
                for log2e in (12, 16, 20):
                    Es = 1 << log2e
                    st = ParticleStepper(cfg, N, Es, device, kernel=args.kernel, tensor_actions=True)
                    st.capture(EP_TICKS)
                    point(Es, "in-place, tensor actions", st, EP_TICKS * (10 if log2e <= 16 else 3), per_tick(Es),
                          "in place, hipGraph of 33 ticks, actions READ from a device tensor (the policy-provided branch)")
                    st = ParticleStepper(cfg, N, Es, device, kernel=args.kernel)
                    point(Es, "in-place, eager launches", st, EP_TICKS * (10 if log2e <= 16 else 3), per_tick(Es),
                          "in place, in-kernel actions, 33 eager launches per enqueue (no hipGraph)")
            out["sweep"] = sweep
        if extras and args.workload == "c2" and not args.envs_per_gpu:
            # the rows SURVEY.md section 8(f2)-(f4) on one collection phase of this workload: transition export, train_step feeds,
            # replay rings, batched evaluation -- each with the bytes it moves and the time those take at the measured copy rate
            try:
                if stepper is not None:
                    stepper.close()
                    stepper = None
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import host_side_timing
                out["host_side"] = host_side_timing.measure(device, copy_gbps=bw_copy)
            except Exception as exc:                     # (an extra: never costs the run its headline)
                out["host_side"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_checkers(cfg) if kind == "checkers" else cpu_baseline(cfg, N)
    if rank == 0:
        emit(out, args.extras_file)
    if use_dist:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
