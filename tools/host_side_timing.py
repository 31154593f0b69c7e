#!/usr/bin/env python
"""Times the rows SURVEY.md section 8(f2)-(f4) on ONE collection phase of the headline workload (C2: 330 ticks x 4096 envs, N = 4):
exporting the phase's transitions (ParticleRollout.as_reference_batch), cm3_amd.batch.process_batch + the n x n x l_action
counterfactual tiling (alg_credit.py:406-557, :614-658, :730-751), the device replay buffers (replay_buffer.py:11-37,
replay_buffer_dual.py:40-63) and batched evaluation (evaluate.py:87-123).  Every figure comes with the bytes the operation has to
move and the time those bytes take at the chip's measured copy rate: ratio = time / that floor.

    python tools/host_side_timing.py            (prints one JSON object; bench.py imports measure() for bench_extras.json)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _nbytes(x):
    import torch
    if isinstance(x, torch.Tensor):
        return x.numel() * x.element_size()
    if isinstance(x, dict):
        return sum(_nbytes(v) for v in x.values())
    if isinstance(x, (list, tuple)):
        return sum(_nbytes(v) for v in x)
    return 0


def _time(fn, device, reps=5, warm=2):
    import torch
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) / reps, out


def measure(device, copy_gbps=None, n_envs=4096, n_agents=4, cfg_name="particle_stage2_antipodal", ticks=330):
    import numpy as np
    import torch
    import cm3_amd
    from cm3_amd import batch as B
    from cm3_amd.actor import ParticleActor
    from cm3_amd.evaluate import test_particle
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.replay import DeviceDualReplayBuffer, DeviceReplayBuffer
    from cm3_amd.rollout import ParticleRollout
    cfg = cm3_amd.load_config(cfg_name)
    N, E = n_agents, n_envs
    copy = (copy_gbps or 5500.0) * 1e9

    def rec(seconds, moved, what, **kw):
        floor = moved / copy
        r = {"ms": seconds * 1e3, "bytes_moved": int(moved), "floor_ms_at_copy_rate": floor * 1e3, "ratio_to_floor": seconds / max(floor, 1e-12),
             "what": what}
        r.update(kw)
        return r

    out = {"workload": "%s, %d envs x %d agents, one collection phase of %d ticks (auto-reset, random actions)" % (cfg_name, E, N, ticks),
           "copy_rate_GBps": copy / 1e9}
    env = VecParticleEnv(cfg, N, 0.2, 33, E, device=device, auto_reset=True)
    env.reset()
    ro = ParticleRollout(env, n_ticks=ticks, use_graph=True)
    ro.collect()
    torch.cuda.synchronize(device)
    # (f2) export of the whole phase: every column of the reference's transition batch for all T x E transitions
    s, cols = _time(lambda: ro.as_reference_batch(numpy=False), device)
    nb = _nbytes(cols)
    out["export_phase"] = rec(s, 2 * nb, "ParticleRollout.as_reference_batch(numpy=False), %d transitions: read + write of the columns"
                              % cols["reward"].shape[0], transitions=int(cols["reward"].shape[0]), column_bytes=int(nb))
    tt_all, ee_all = ro.valid_indices()
    s, _ = _time(lambda: ro.as_reference_batch_torch(tt_all, ee_all, numpy=False), device, reps=2, warm=1)
    out["export_phase_torch_composition"] = rec(s, 2 * nb, "the same columns as ~25 torch indexing launches (what round 4 shipped)")
    # minibatch export: 128 transitions (batch_size of alg/config.json), as the on-policy cadence samples them
    g = torch.Generator(device=device).manual_seed(0)
    s, mb = _time(lambda: ro.sample_batch(128, generator=g, numpy=False), device, reps=20)
    out["sample_minibatch_128"] = rec(s, 2 * _nbytes(mb), "ParticleRollout.sample_batch(128, numpy=False)")
    s, mbs = _time(lambda: list(ro.on_policy_minibatches(epochs=24, batch_size=128, generator=g)), device, reps=10)
    out["on_policy_minibatches_24x128"] = rec(s, 2 * sum(_nbytes(m) for m in mbs) / 1.0, "ParticleRollout.on_policy_minibatches(24, 128): all "
                                              "minibatches of a phase drawn together and exported by one launch (train_onpolicy.py:359-377)")
    # (f2) process_batch + the feeds of train_step with a stand-in session, on a 128-transition minibatch and on 16 384 transitions
    for label, nb_rows in (("minibatch_128", 128), ("batch_16384", 16384)):
        tt = torch.randint(0, ticks, (nb_rows,), device=device)
        ee = torch.randint(0, E, (nb_rows,), device=device)
        c = ro.as_reference_batch(tt, ee, numpy=False)
        s, pb = _time(lambda: B.process_batch(c), device, reps=20)
        out["process_batch_" + label] = rec(s, _nbytes(c) + _nbytes(pb[1:]), "cm3_amd.batch.process_batch")

        def run(ops, feed, _n=nb_rows):
            rows = next(v.shape[0] for v in feed.values() if hasattr(v, "shape") and v.dim() > 0) if feed else 1
            outs = []
            for op in ops:
                if op == "action_samples_target":
                    outs.append(torch.zeros(_n * N, dtype=torch.int64, device=device))
                elif op in ("probs",):
                    outs.append(torch.full((_n * N, 5), 0.2, dtype=torch.float64, device=device))
                elif op in ("Q_credit",):
                    outs.append(torch.zeros(rows, dtype=torch.float64, device=device))
                elif op.endswith("_op") or op == "list_update_target_ops":
                    outs.append(None)
                else:
                    outs.append(torch.zeros(rows, dtype=torch.float64, device=device))
            return outs
        s, calls = _time(lambda: B.train_step_feeds(c, run, 0.99, 0.1), device, reps=10)
        fb = sum(_nbytes(f) for _, f in calls)
        out["train_step_feeds_" + label] = rec(s, _nbytes(c) + fb, "cm3_amd.batch.train_step_feeds (process_batch, n x n credit repeats, "
                                               "n x n x l_action counterfactual tiling; static feeds from two cm3_rows_tile launches) with a "
                                               "stand-in session", feed_bytes=int(fb), sess_runs=len(calls))
        s, _ = _time(lambda: B.train_step_feeds(c, run, 0.99, 0.1, device_tiling=False), device, reps=10)
        out["train_step_feeds_" + label + "_torch_composition"] = rec(s, _nbytes(c) + fb, "the same feeds as a composition of torch operations")
        s, S = _time(lambda: B.particle_static_feeds_device(c), device, reps=20)
        out["static_feeds_" + label] = rec(s, _nbytes(c) + _nbytes(S), "particle_static_feeds_device alone: 22 feed tensors, two launches")
    # ---- one on-policy phase END TO END (train_onpolicy.py:302-377): collect 330 ticks, draw + export the 24 minibatches of 128, build
    # every feed of the 24 train_steps (stand-in session) -- round 5: 0.79 + 0.44 + 24 x 0.32 = ~9 ms, launch latency of the feeds
    def run128(ops, feed):
        outs = []
        for op in ops:
            if op == "action_samples_target":
                outs.append(_zeros["acts"])
            elif op == "probs":
                outs.append(_zeros["probs"])
            elif op == "Q_credit":
                outs.append(_zeros["qcf"])
            elif op.endswith("_op") or op == "list_update_target_ops":
                outs.append(None)
            else:
                rows = next(v.shape[0] for v in feed.values() if hasattr(v, "shape") and v.dim() > 0)
                outs.append(_zeros["q"][:rows])
        return outs
    _zeros = {"acts": torch.zeros(128 * N, dtype=torch.int64, device=device), "probs": torch.full((128 * N, 5), 0.2, dtype=torch.float64, device=device),
              "qcf": torch.zeros(128 * N * N * 5, dtype=torch.float64, device=device), "q": torch.zeros(128 * N * N * 5, dtype=torch.float64, device=device)}

    def phase(with_collect=True):
        if with_collect:
            ro.collect()
        n = 0
        for c_k, s_k in ro.on_policy_phase(epochs=24, batch_size=128, generator=g):
            n += len(B.train_step_feeds(c_k, run128, 0.99, 0.1, static=s_k))
        return n
    s_feeds, n_calls = _time(lambda: phase(False), device, reps=5)
    s_coll, _ = _time(lambda: ro.collect(), device, reps=5)
    s_phase, _ = _time(phase, device, reps=5)
    out["phase_total"] = {"ms": s_phase * 1e3, "collect_ms": s_coll * 1e3, "export_and_feeds_ms": s_feeds * 1e3, "sess_runs": int(n_calls),
                          "what": "ParticleRollout.collect (330 ticks) + on_policy_phase(24, 128) (one export launch + one pair of tiling "
                                  "launches for all 24 minibatches) + cm3_amd.batch.train_step_feeds x 24 with a stand-in session whose "
                                  "outputs are preallocated; round 5: ~9 ms"}
    # ... and with the train_steps' data movement as hipGraph replays over persistent tensors (cm3_amd.batch.OnPolicyPhasePlan)
    plan = B.OnPolicyPhasePlan(ro, run128, 0.99, 0.1, epochs=24, batch_size=128)

    def phase_graph(with_collect=True):
        if with_collect:
            ro.collect()
        plan.refresh(g)
        n = 0
        for k in range(24):
            n += len(plan.step(k))
        return n
    phase_graph()
    s_pg_feeds, _ = _time(lambda: phase_graph(False), device, reps=5)
    s_pg, n_calls_g = _time(phase_graph, device, reps=5)
    out["phase_total_graph"] = {"ms": s_pg * 1e3, "collect_ms": s_coll * 1e3, "export_and_feeds_ms": s_pg_feeds * 1e3, "sess_runs": int(n_calls_g),
                                "what": "the same phase through OnPolicyPhasePlan: persistent columns / static feeds (refresh: one export launch + "
                                        "one pair of tiling launches) and ONE hipGraph replay per minibatch for everything train_step_feeds "
                                        "enqueues; the stand-in session does no GPU work, a real one would be captured with it"}
    plan.close()
    # (f4) replay: add the whole phase, sample 128
    cols = ro.as_reference_batch(numpy=False)
    B_all = cols["reward"].shape[0]
    buf = DeviceReplayBuffer(size=B_all, device=device)
    s, _ = _time(lambda: buf.add(cols), device, reps=5)
    out["replay_add_phase"] = rec(s, 2 * _nbytes(cols), "DeviceReplayBuffer.add of the phase's %d transitions" % B_all)
    buf2 = DeviceReplayBuffer(size=B_all, device=device)
    s, _ = _time(lambda: buf2.add_rollout(ro), device, reps=5)
    out["replay_add_rollout_phase"] = rec(s, 2 * _nbytes(cols), "DeviceReplayBuffer.add_rollout: the phase's transitions from the trajectory "
                                          "straight into the ring, one launch (export + add)")
    del buf2
    s, b = _time(lambda: buf.sample_batch(128, generator=g), device, reps=20)
    out["replay_sample_128"] = rec(s, 2 * _nbytes(b), "DeviceReplayBuffer.sample_batch(128)")
    bad = ro.episode_is_bad()
    # the flag of an episode belongs to all of its transitions: here per transition = the flag of the tick that ends the episode,
    # spread back over the episode is the trainer's job; for the timing the per-transition flag is any bool [B] column
    flag = (torch.arange(B_all, device=device) % 3) == 0
    dual = DeviceDualReplayBuffer(size=B_all, device=device)
    s, _ = _time(lambda: dual.add(cols, flag), device, reps=5)
    out["dual_replay_add_phase"] = rec(s, 2 * _nbytes(cols), "DeviceDualReplayBuffer.add (split by flag) of %d transitions" % B_all)
    s, b = _time(lambda: dual.sample_batch(128, generator=g), device, reps=20)
    out["dual_replay_sample_128"] = rec(s, 2 * _nbytes(b), "DeviceDualReplayBuffer.sample_batch(128)")
    del buf, dual, cols, bad
    ro.close()
    # (f3) batched evaluation: 4096 greedy episodes of 33 ticks with the on-device actor
    rng = np.random.default_rng(0)
    Lo = 4 * max(N - 1, 1)
    shapes = {"actor_branch_self/kernel": (6, 64), "actor_branch_self/bias": (64,), "W_branch_self_h2": (64, 64),
              "stage-2/actor_others/kernel": (Lo, 128), "stage-2/actor_others/bias": (128,),
              "stage-2/W_others_h2": (128, 64), "b": (64,), "actor_out/kernel": (64, 5), "actor_out/bias": (5,)}
    wts = {k: (rng.standard_normal(v) * 0.1).astype(np.float32) for k, v in shapes.items()}
    eenv = VecParticleEnv(cfg, N, 0.2, 33, E, device=device, auto_reset=False)
    actor = ParticleActor(wts, N, stage=2, device=device, precision="f16x3")
    ero = ParticleRollout(eenv, use_graph=True)
    s, res = _time(lambda: test_particle(eenv, actor, n_rounds=1, rollout=ero), device, reps=5)
    out["evaluate_4096_episodes"] = {"ms": s * 1e3, "episodes": E, "us_per_tick": s * 1e6 / 33.0, "env_steps_per_s": E * 33 / s,
                                     "what": "evaluate.test_particle: reset + 33 policy-driven ticks (one launch per episode) + return accumulation + "
                                             "the two averages read back"}
    ero.close()
    return out


if __name__ == "__main__":
    import torch
    dev = torch.device("cuda", 0)
    print(json.dumps(measure(dev), indent=1))
