#!/usr/bin/env python
"""Soak of the one-launch Checkers policy rollout (cm3_policy_rollout_checkers): the SAME rollout -- same env seed, same weights, same
epsilon -- launched over and over; every launch must reproduce the first one bit for bit (actions, probabilities, every trajectory
array, the live state).  Round 5 found a chip fault exactly this way (a packed multiply going wrong in lanes 48..63 while the SIMD's
other wave started float16 matrix instructions: profiles/r05_policy_fault.txt); this kernel runs two waves per SIMD through float16
matrix phases next to the env step's integer / float64 code for its whole life.  usage: ck_policy_soak.py [seconds = 240] [envs = 8192]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cm3_amd  # noqa: E402
from cm3_amd.actor import CheckersActor  # noqa: E402
from cm3_amd.checkers import VecCheckersEnv  # noqa: E402
from cm3_amd.rollout import CheckersRollout  # noqa: E402
from ck_policy_worker import weights  # noqa: E402

NAMES = ("actions", "probs", "grid", "vec", "obs_others", "obs_self_t", "obs_self_v", "local_rewards", "reward", "done",
         "term_grid", "term_vec", "term_obs_others", "term_obs_self_t", "term_obs_self_v", "goal_slots")


def soak(seconds, E, stage=2, max_steps=11, verbose=True):
    dev = torch.device("cuda", 0)
    N = 2 if stage == 2 else 1
    cfg = cm3_amd.load_config("checkers_stage%d" % stage)
    actor = CheckersActor(weights(N, np.random.default_rng(0)), N, stage=stage, device=dev, precision="f16x3", seed=77)
    goals = np.eye(2) if N > 1 else np.array([[1, 0]])
    ref, launches, bad, t0 = None, 0, [], time.time()
    while time.time() - t0 < seconds:
        # a fresh env object with the same seed: the same episodes, restarts inside the launch (max_steps 11, 33 ticks)
        env = VecCheckersEnv(cfg["init"], N, max_steps, E, device=dev, seed=77, auto_reset=True)
        ro = CheckersRollout(env, n_ticks=33, record_probs=True)
        assert actor.fused_rollout_ok(env)
        for c in range(3):
            ro.collect(goals, policy=actor, epsilon=0.1)
            torch.cuda.synchronize()
            snap = {n: getattr(ro, n).clone() for n in NAMES}
            snap.update(mask=env._mask.clone(), agents=env._agents.clone(), steps=env._steps.clone(), episode=env._episode.clone())
            launches += 1
            if ref is None or len(ref) <= c:
                ref = (ref or []) + [snap]
            else:
                diff = [n for n in snap if not torch.equal(snap[n], ref[c][n])]
                if diff:
                    bad.append({"launch": launches, "collect": c, "arrays": diff})
        ro.close()
        del env, ro
    out = {"what": "k_ck_policy_rollout<%d> soak: identical rollouts, every launch compared with the first" % N, "envs": E, "seconds": round(time.time() - t0, 1),
           "launches": launches, "ticks": launches * 33, "mismatching_launches": len(bad), "first_mismatches": bad[:5]}
    if verbose:
        print(json.dumps(out))
    return out


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    r = soak(secs, E)
    sys.exit(1 if r["mismatching_launches"] else 0)
