#!/usr/bin/env python
"""Times the Checkers device actor alone and policy-driven Checkers collection (T x (actor + step) in one hipGraph)."""
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

MACS_PER_ROW = 80 * 160 + 160 * 32 + 48 * 256 + 16 * 256 + 2 * 256 * 256 + 256 * 16   # as executed (padded tiles)
MACS_PER_ROW_ALG = 25 * 6 * 27 + 150 * 32 + 43 * 256 + 2 * 256 + 2 * 256 * 256 + 256 * 5  # the network itself


def random_weights(rng, n_agents, stage=2):
    lo = 2 * max(n_agents - 1, 1)
    f = lambda *s: (rng.standard_normal(s) / np.sqrt(s[0] if len(s) > 1 else 4)).astype(np.float32)  # noqa: E731
    w = {"conv/Conv/weights": f(3, 3, 3, 6), "conv/Conv/biases": f(6), "conv_linear/kernel": f(150, 32),
         "conv_linear/bias": f(32), "branch_self/kernel": f(43, 256), "branch_self/bias": f(256), "W_self_h2": f(256, 256),
         "b": f(256), "actor_out/kernel": f(256, 5), "actor_out/bias": f(5)}
    if stage > 1:
        w.update({"stage-2/branch_others/kernel": f(lo, 256), "stage-2/branch_others/bias": f(256),
                  "stage-2/W_others_h2": f(256, 256)})
    return w


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    cfg = cm3_amd.load_config("checkers_stage2")
    N = 2
    for prec, E in [(pr, e) for pr in ("f32", "f16x3", "bf16") for e in (8192, 65536)]:
        actor = CheckersActor(random_weights(np.random.default_rng(0), N), N, device=dev, precision=prec)
        env = VecCheckersEnv(cfg["init"], N, 33, E, device=dev)
        env.reset(np.eye(2))
        for _ in range(3):
            actor.act(env, 0.1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50 if E <= 65536 else 10
        a.record()
        for _ in range(reps):
            actor.act(env, 0.1)
        b.record()
        b.synchronize()
        us = a.elapsed_time(b) * 1e3 / reps
        rows = E * N
        print(json.dumps({"what": "actor launch", "precision": prec, "envs": E, "rows": rows, "us": round(us, 2),
                          "tflops_executed": round(2 * MACS_PER_ROW * rows / us * 1e-6, 2),
                          "tflops_network": round(2 * MACS_PER_ROW_ALG * rows / us * 1e-6, 2)}), flush=True)
        if E <= 65536:
            ro = CheckersRollout(env, use_graph=True)
            for _ in range(2):
                ro.collect(np.eye(2), policy=actor, epsilon=0.1)
            torch.cuda.synchronize()
            reps = 10
            a.record()
            for _ in range(reps):
                ro.collect(np.eye(2), policy=actor, epsilon=0.1)
            b.record()
            b.synchronize()
            us = a.elapsed_time(b) * 1e3 / (reps * 33)
            print(json.dumps({"what": "policy rollout (reset + 33 x (actor + step), hipGraph)", "precision": prec, "envs": E,
                              "us_per_tick": round(us, 2), "env_steps_per_s": E / us * 1e6}), flush=True)
            ro.close()
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
