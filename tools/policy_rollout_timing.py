#!/usr/bin/env python
"""Times policy-driven collection on the device: T x (actor launch + step launch) as one hipGraph replay."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cm3_amd  # noqa: E402
from cm3_amd.actor import ParticleActor  # noqa: E402
from cm3_amd.particle import VecParticleEnv  # noqa: E402
from cm3_amd.rollout import ParticleRollout  # noqa: E402


def init_weights(rng, n_agents):
    """Random float32 weights with the reference's variable names / shapes (networks.py:517-538)."""
    lo = 4 * max(n_agents - 1, 1)
    shapes = {"actor_branch_self/kernel": (6, 64), "actor_branch_self/bias": (64,), "W_branch_self_h2": (64, 64),
              "stage-2/actor_others/kernel": (lo, 128), "stage-2/actor_others/bias": (128,),
              "stage-2/W_others_h2": (128, 64), "b": (64,), "actor_out/kernel": (64, 5), "actor_out/bias": (5,)}
    return {k: (rng.standard_normal(v) * 0.5).astype(np.float32) for k, v in shapes.items()}


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    cfg = cm3_amd.load_config("particle_stage2_antipodal")
    for E, prec, fused in [(e, pr, fu) for e in (4096, 65536, 1 << 20) for pr in ("f32", "bf16") for fu in (False, True)]:
        env = VecParticleEnv(cfg, 4, 0.2, 33, E, device=dev, auto_reset=True)
        env.reset()
        actor = ParticleActor(init_weights(np.random.default_rng(0), 4), 4, device=dev, precision=prec)
        ro = ParticleRollout(env, use_graph=True, fused=fused)
        for _ in range(3):
            ro.collect(policy=actor, epsilon=0.1, reset=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if E <= 65536 else 3
        a.record()
        for _ in range(reps):
            ro.collect(policy=actor, epsilon=0.1, reset=False)
        b.record()
        b.synchronize()
        us = a.elapsed_time(b) * 1e3 / (reps * 33)
        print(json.dumps({"envs": E, "actor_precision": prec, "one_launch_per_episode": fused, "us_per_tick_actor_plus_step": round(us, 2), "env_steps_per_s": E / us * 1e6}))
        ro.close()
        del ro, env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
