#!/usr/bin/env python
"""us per tick of policy-driven collection at C2 (4096 envs x 4 agents, 33-tick rollouts with full trajectory storage, epsilon
0.1, random weights of the reference's shapes) in its three launch modes, plus the actor launch alone.  Used with CM3_AMD_LIB to
compare two builds on one box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import cm3_amd
    from cm3_amd.actor import ParticleActor
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    cfg = cm3_amd.load_config("particle_stage2_antipodal")
    dev, E, N, T = torch.device("cuda:0"), 4096, 4, 33
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    rng = np.random.default_rng(0)
    shapes = {"actor_branch_self/kernel": (6, 64), "actor_branch_self/bias": (64,), "W_branch_self_h2": (64, 64),
              "stage-2/actor_others/kernel": (12, 128), "stage-2/actor_others/bias": (128,), "stage-2/W_others_h2": (128, 64),
              "b": (64,), "actor_out/kernel": (64, 5), "actor_out/bias": (5,)}
    wts = {k: (rng.standard_normal(v) * 0.1).astype(np.float32) for k, v in shapes.items()}
    out = []
    for label, fused, ftick in (("alternating", False, False), ("fused_tick", False, True), ("fused_episode", True, False)):
        env = VecParticleEnv(cfg, N, 0.2, 33, E, device=dev, auto_reset=True)
        env.reset()
        actor = ParticleActor(wts, N, stage=2, device=dev)
        ro = ParticleRollout(env, n_ticks=T, use_graph=True, fused=fused, fused_policy_tick=ftick)
        for _ in range(3):
            ro.collect(policy=actor, epsilon=0.1, reset=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            ro.collect(policy=actor, epsilon=0.1, reset=False)
        b.record()
        b.synchronize()
        out.append("%s %.3f" % (label, a.elapsed_time(b) * 1e3 / (30 * T)))
        ro.close()
    print("  ".join(out))


if __name__ == "__main__":
    main()
