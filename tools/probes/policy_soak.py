"""Soak: whole-episode policy launches against alternating actor / step launches, many seeds, BASELINE sizes; counts differing
tensors (expected: none).  tools/probes/policy_soak.py [minutes]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import tests.test_gpu_actor as TA

budget = 60.0 * float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
cases = [(4096, 4, "particle_stage2_antipodal.json"), (8192, 8, "particle_merge8.json"), (8192, 4, "particle_stage2_cross.json"),
         (2048, 8, "particle_merge8.json"), (16384, 2, "particle_stage2_merge.json"), (4096, 8, "particle_merge8.json"),
         (4099, 4, "particle_stage2_cross.json"), (8191, 8, "particle_merge8.json"), (32771, 1, "particle_stage1.json"),
         (65536, 4, "particle_stage2_antipodal.json")]
t0, runs, bad = time.time(), 0, 0
seed = 1000
while time.time() - t0 < budget:
    for E, N, cfg in cases:
        for prec in ("f16x3", "f32"):
            seed += 1
            ref, eref, _ = TA._policy_run(E, N, cfg, prec, 33, "tick", seed=seed)
            ro, env, v = TA._policy_run(E, N, cfg, prec, 33, "episode", seed=seed)
            for name in ("actions", "state", "obs_others", "reward", "reward_n", "done", "collisions"):
                if not torch.equal(getattr(ref, name), getattr(ro, name)):
                    bad += 1
                    print("MISMATCH", E, N, prec, seed, name, v, flush=True)
            ro.close(); ref.close()
            runs += 1
print("soak: %d comparisons of 33-tick collections in %.0f s, %d differing tensors" % (runs, time.time() - t0, bad))
