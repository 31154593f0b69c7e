"""TEST INFRASTRUCTURE ONLY -- records what the REAL reference evaluators return (build container only).

Runs alg/evaluate.py:test_particle (:87-123) and :test_checkers (:159-203) from /root/reference -- imported unmodified, with
`env.multicar_simple` (SUMO, needs traci) stubbed out of the import -- on the REAL reference envs, with a deterministic
stand-in for `alg.run_actor` (TensorFlow is not installable here).  The stand-in policies are pure functions of the
observation that cm3_amd's host-policy hook can reproduce exactly, so the fixtures pin the evaluators' control flow and
return accumulation: per-episode reset, epsilon = 0, `while not done`, reward_local / reward_global accumulation, the
averages over n_eval (and, for single-agent Checkers, the per-episode goal draw).  They do NOT pin a network.

    python oracle/gen_golden_evaluate.py  ->  tests/golden/evaluate_{particle_n4,particle_n1,checkers_n2,checkers_n1}.npz
The action histogram test_checkers prints is not returned by the reference and is therefore not in the fixtures.
"""
import contextlib
import io
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def particle_policy(local_others, local_v, goals):
    """walk toward the own landmark along the axis with the larger gap; coast when fast (pure function of obs_self, goals)"""
    v = np.asarray(local_v, dtype=np.float64)
    g = np.asarray(goals, dtype=np.float64)
    d = g - v[:, 2:4]
    horiz = np.abs(d[:, 0]) > np.abs(d[:, 1])
    a = np.where(horiz, np.where(d[:, 0] > 0, 2, 1), np.where(d[:, 1] > 0, 4, 3))
    fast = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) > 0.36
    return np.where(fast, 0, a).astype(np.int64)


def checkers_policy(actions_prev, obs_self_t, episode):
    """integer-only function of the 5x5x3 window, the previous action and the episode index (Checkers stage 2 has a fixed
    start and fixed goals: without the episode index every evaluation episode would be the same trajectory): exact on
    every platform"""
    t = np.asarray(obs_self_t)
    n = t.shape[0]
    uncollected = (t[..., 0:2] == -1).reshape(n, -1).sum(1).astype(np.int64)
    walls = (t[..., 2] == 1).reshape(n, -1).sum(1).astype(np.int64)
    return (uncollected + 3 * walls + 2 * np.asarray(actions_prev, dtype=np.int64) + np.arange(n) + int(episode)) % 5


def main():
    from oracle import _reference_harness as H
    ns = H.load_reference()
    sys.modules.setdefault("env.multicar_simple", types.ModuleType("env.multicar_simple"))
    sys.path.insert(0, os.path.join(H.REFERENCE_ROOT, "alg"))
    import evaluate as ref_eval                       # the reference's own alg/evaluate.py
    import cm3_amd
    out = os.path.join(ROOT, "tests", "golden")

    for tag, cfg_name, N, n_eval in (("particle_n4", "particle_stage2_antipodal", 4, 24), ("particle_n1", "particle_stage1", 1, 16)):
        cfg = cm3_amd.load_config(cfg_name)
        np.random.seed(12341 + N)
        random.seed(12341 + N)
        env, scenario, world = H.make_reference_particle_env(ns, N, cfg, 0.2, 33)
        inits, lms = [], []
        real_reset = env.reset

        def reset(real_reset=real_reset, env=env):
            o = real_reset()
            inits.append(np.array(o[0], dtype=np.float64))
            lms.append(np.array([l.state.p_pos for l in env.world.landmarks], dtype=np.float64))
            return o
        env.reset = reset
        alg = types.SimpleNamespace(run_actor=lambda lo, lv, goals, eps, sess: particle_policy(lo, lv, goals))
        local, glob = ref_eval.test_particle(n_eval, env, None, N, 2, alg)
        np.savez_compressed(os.path.join(out, "evaluate_%s.npz" % tag), init_gs=np.stack(inits), landmarks=np.stack(lms),
                            reward_local_avg=np.asarray(local, np.float64), reward_global_avg=np.float64(glob),
                            n_agents=N, n_eval=n_eval, max_steps=33, config=np.array(cfg_name))
        print(tag, "local", local, "global", glob)

    for tag, cfg_name, N, n_eval in (("checkers_n2", "checkers_stage2", 2, 20), ("checkers_n1", "checkers_stage1", 1, 20)):
        cfg = cm3_amd.load_config(cfg_name)
        np.random.seed(777 + N)
        env = H.make_reference_checkers_env(ns, cfg["init"], N, 33)
        goals_seen = []
        real_reset = env.reset

        def reset(goals, real_reset=real_reset):
            goals_seen.append(np.array(goals, dtype=np.float64))
            return real_reset(goals)
        env.reset = reset
        # episode index = number of resets so far - 1 (the stand-in policy varies with it, see checkers_policy)
        alg = types.SimpleNamespace(run_actor=lambda prev, oo, ot, ov, goals, eps, sess:
                                    checkers_policy(prev, ot, len(goals_seen) - 1))
        with contextlib.redirect_stdout(io.StringIO()):          # the reference prints its action histogram
            local, glob = ref_eval.test_checkers(n_eval, env, None, N, alg)
        np.savez_compressed(os.path.join(out, "evaluate_%s.npz" % tag), goals=np.stack(goals_seen),
                            reward_local_avg=np.asarray(local, np.float64), reward_global_avg=np.float64(glob),
                            n_agents=N, n_eval=n_eval, max_steps=33, config=np.array(cfg_name))
        print(tag, "local", local, "global", glob, "goal mix", np.stack(goals_seen)[:, 0].mean(0))


if __name__ == "__main__":
    main()
