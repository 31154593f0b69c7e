"""Round 6: where a CheckersRollout.collect(policy=actor) spends its time besides the rollout launch (C3, one-launch mode)."""
import json, os, sys, time
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


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    host = (time.perf_counter() - t0) / reps * 1e6
    return round(e0.elapsed_time(e1) * 1e3 / reps, 1), round(host, 1)

dev = torch.device("cuda:0")
cfg = cm3_amd.load_config("checkers_stage2")
rng = np.random.default_rng(0)
goals = np.eye(2)
out = {}
for auto_reset in (False, True):
    env = VecCheckersEnv(cfg["init"], 2, 33, 8192, device=dev, auto_reset=auto_reset)
    actor = CheckersActor(weights(2, rng), 2, stage=2, device=dev, precision="f16x3")
    ro = CheckersRollout(env, n_ticks=33)
    ro.collect(goals, policy=actor, epsilon=0.1)
    tag = "auto_reset" if auto_reset else "episode_sync"
    out[tag + "_collect_us(gpu,host)"] = timed(lambda: ro.collect(goals, policy=actor, epsilon=0.1))
    out[tag + "_rollout_launch_only"] = timed(lambda: ro._enqueue_policy_rollout(actor, 0.1, env._stream()))
    out[tag + "_env_reset"] = timed(lambda: env.reset(goals))
    out[tag + "_load_slot0"] = timed(ro._load_slot0)
    out[tag + "_store_back"] = timed(ro._store_back)
print(json.dumps(out, indent=1))
