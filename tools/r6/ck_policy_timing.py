"""Round 6: policy-driven Checkers collection, us per tick -- the whole rollout in ONE launch (csrc/policy_checkers.hip) against an actor
launch + a step launch per tick inside one hipGraph.  One JSON line per measurement."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cm3_amd
from cm3_amd.actor import CheckersActor
from cm3_amd.checkers import VecCheckersEnv
from cm3_amd.rollout import CheckersRollout


def weights(Nc, rng):
    shapes = {"conv/Conv/weights": (3, 3, 3, 6), "conv/Conv/biases": (6,), "conv_linear/kernel": (150, 32),
              "conv_linear/bias": (32,), "branch_self/kernel": (43, 256), "branch_self/bias": (256,),
              "W_self_h2": (256, 256), "stage-2/branch_others/kernel": (2 * max(Nc - 1, 1), 256),
              "stage-2/branch_others/bias": (256,), "stage-2/W_others_h2": (256, 256), "b": (256,),
              "actor_out/kernel": (256, 5), "actor_out/bias": (5,)}
    return {k: (rng.standard_normal(v) * 0.1).astype(np.float32) for k, v in shapes.items()}



def main():
    dev = torch.device("cuda:0")
    cases = [("checkers_stage2", 2, 8192), ("checkers_stage1", 1, 16384), ("checkers_stage2", 2, 65536), ("checkers_stage2", 2, 2048)]
    if len(sys.argv) > 1:
        cases = cases[:int(sys.argv[1])]
    for cfg_name, Nc, E in cases:
        cfg = cm3_amd.load_config(cfg_name)
        rng = np.random.default_rng(0)
        goals = np.eye(2) if Nc > 1 else np.array([[1, 0]])
        for auto_reset in (False, True):
            for mode in ("tick", "auto"):
                env = VecCheckersEnv(cfg["init"], Nc, 33, E, device=dev, auto_reset=auto_reset)
                actor = CheckersActor(weights(Nc, rng), Nc, stage=2 if Nc > 1 else 1, device=dev, precision="f16x3")
                ro = CheckersRollout(env, n_ticks=33, use_graph=True, policy_mode=mode)
                for _ in range(3):
                    ro.collect(goals, policy=actor, epsilon=0.1)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 10
                e0.record()
                for _ in range(reps):
                    ro.collect(goals, policy=actor, epsilon=0.1)
                e1.record(); e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (reps * 33)
                print(json.dumps({"what": "checkers policy collection", "config": cfg_name, "envs": E, "agents": Nc, "auto_reset": auto_reset,
                                  "mode": "one launch per rollout" if mode == "auto" else "actor + step launch per tick",
                                  "us_per_tick": round(us, 3), "env_steps_per_s": round(E / us * 1e6)}))
                ro.close()
                del env, actor, ro


if __name__ == "__main__":
    main()
