#!/usr/bin/env python
"""Worker of tools/pmc_ck_policy.sh: policy-driven Checkers collection at C3 (config_checkers_stage2, 8192 envs x 2 agents, 33 ticks per
collect, one launch per rollout: cm3_policy_rollout_checkers) -- 3 warm-up + N timed collects; prints us per tick."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cm3_amd  # noqa: E402
from cm3_amd.actor import CheckersActor  # noqa: E402
from cm3_amd.checkers import VecCheckersEnv  # noqa: E402
from cm3_amd.rollout import CheckersRollout  # noqa: E402


def weights(Nc, rng):
    shapes = {"conv/Conv/weights": (3, 3, 3, 6), "conv/Conv/biases": (6,), "conv_linear/kernel": (150, 32),
              "conv_linear/bias": (32,), "branch_self/kernel": (43, 256), "branch_self/bias": (256,),
              "W_self_h2": (256, 256), "stage-2/branch_others/kernel": (2 * max(Nc - 1, 1), 256),
              "stage-2/branch_others/bias": (256,), "stage-2/W_others_h2": (256, 256), "b": (256,),
              "actor_out/kernel": (256, 5), "actor_out/bias": (5,)}
    return {k: (rng.standard_normal(v) * 0.1).astype(np.float32) for k, v in shapes.items()}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    mode = sys.argv[2] if len(sys.argv) > 2 else "auto"          # auto = one launch per rollout | tick = launch pairs
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    dev = torch.device("cuda", 0)
    cfg = cm3_amd.load_config("checkers_stage2")
    env = VecCheckersEnv(cfg["init"], 2, 33, E, device=dev)
    actor = CheckersActor(weights(2, np.random.default_rng(0)), 2, stage=2, device=dev, precision="f16x3")
    ro = CheckersRollout(env, n_ticks=33, policy_mode=mode)
    goals = np.eye(2)
    for _ in range(3):
        ro.collect(goals, policy=actor, epsilon=0.1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ro.collect(goals, policy=actor, epsilon=0.1)
    b.record()
    b.synchronize()
    print(json.dumps({"mode": mode, "envs": E, "collects": reps, "us_per_tick": a.elapsed_time(b) * 1e3 / (reps * 33)}))
    ro.close()


if __name__ == "__main__":
    main()
