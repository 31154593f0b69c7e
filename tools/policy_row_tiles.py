#!/usr/bin/env python
"""Whole-episode policy collection (cm3_policy_rollout_f32, one launch per 33-tick episode) per row-tile count RT of its workgroups
(CM3_POLICY_RT = 1 | 2 | 4: 16 RT agent rows per 4-wave workgroup), us per tick; one subprocess per setting (the override is read at
launch time, but a fresh process keeps the measurements independent)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(cfg_name, N, E, prec):
    import numpy as np
    import torch
    import cm3_amd
    from cm3_amd import _lib
    from cm3_amd.actor import ParticleActor
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    cfg = cm3_amd.load_config(cfg_name)
    rng = np.random.default_rng(0)
    lo = 4 * max(N - 1, 1)
    shapes = {"actor_branch_self/kernel": (6, 64), "actor_branch_self/bias": (64,), "W_branch_self_h2": (64, 64),
              "stage-2/actor_others/kernel": (lo, 128), "stage-2/actor_others/bias": (128,),
              "stage-2/W_others_h2": (128, 64), "b": (64,), "actor_out/kernel": (64, 5), "actor_out/bias": (5,)}
    w = {k: (rng.standard_normal(v) * 0.1).astype(np.float32) for k, v in shapes.items()}
    env = VecParticleEnv(cfg, N, 0.2, 33, E, device=dev, auto_reset=True)
    env.reset()
    actor = ParticleActor(w, N, stage=2 if N > 1 else 1, device=dev, precision=prec)
    ro = ParticleRollout(env, n_ticks=33, use_graph=True)
    for _ in range(3):
        ro.collect(policy=actor, epsilon=0.1, reset=False)
    torch.cuda.synchronize()
    variant = _lib.last_kernel_variant()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        a.record()
        for _ in range(reps):
            ro.collect(policy=actor, epsilon=0.1, reset=False)
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / (reps * 33))
    print(json.dumps({"us_per_tick": round(best, 3), "variant": variant}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    cases = [("particle_stage2_antipodal", 4, 4096), ("particle_stage2_antipodal", 4, 2048), ("particle_stage2_antipodal", 4, 8192),
             ("particle_stage2_antipodal", 4, 16384), ("particle_stage2_antipodal", 4, 65536),
             ("particle_merge8", 8, 8192), ("particle_merge8", 8, 2048), ("particle_stage2_merge", 2, 8192), ("particle_stage1", 1, 16384)]
    print("%-28s %2s %7s %6s | us per tick at RT = 4 / 2 / 1 / auto" % ("config", "N", "envs", "prec"))
    for cfg_name, N, E in cases:
        for prec in os.environ.get("ROW_TILES_PRECS", "f16x3,f32").split(","):
            row = []
            for rt in ("4", "2", "1", ""):
                env = dict(os.environ)
                env.pop("CM3_POLICY_RT", None)
                if rt:
                    env["CM3_POLICY_RT"] = rt
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", cfg_name, str(N), str(E), prec],
                                     env=env, capture_output=True, text=True, timeout=300)
                try:
                    rec = json.loads(out.stdout.strip().splitlines()[-1])
                    row.append("%.2f" % rec["us_per_tick"] + ("(g=%s)" % rec["variant"].split("g=")[1].split(",")[0] if not rt else ""))
                except Exception:
                    row.append("ERR " + out.stderr[-200:])
            print("%-28s %2d %7d %6s | %s" % (cfg_name, N, E, prec, "  ".join(row)), flush=True)


if __name__ == "__main__":
    main()
