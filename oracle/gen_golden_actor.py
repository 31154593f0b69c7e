"""TEST INFRASTRUCTURE ONLY (build container) -- golden vectors for the two policy networks, produced by executing the
REFERENCE's own function bodies networks.actor_particle / networks.actor_checkers under oracle/tf_numpy_shim.py:
weights under the variable names the reference's code creates, inputs, and the softmax probabilities.
Writes tests/golden/actor_particle.npz and tests/golden/actor_checkers.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import tf_numpy_shim as S  # noqa: E402


def particle_case(rng, n_agents, stage, rows):
    shim = S.Shim(rng=rng)
    net = S.load_networks(shim)
    lo = 4 * max(n_agents - 1, 1)
    oo = rng.uniform(-2, 2, (rows, lo)).astype(np.float32)
    vo = rng.uniform(-1.5, 1.5, (rows, 4)).astype(np.float32)
    vg = rng.uniform(-1, 1, (rows, 2)).astype(np.float32)
    probs = net.actor_particle(S._t(oo), S._t(vo), S._t(vg), n_actions=5, n_h1_self=64, n_h1_others=128, n_h2=64, stage=stage)
    return shim.weights, dict(obs_others=oo, v_obs=vo, v_goal=vg), np.asarray(probs)


def checkers_case(rng, n_agents, stage, rows):
    shim = S.Shim(rng=rng, scale=1.0)
    net = S.load_networks(shim)
    lo = 2 * max(n_agents - 1, 1)
    a_prev = rng.integers(0, 5, rows)
    a1 = np.eye(5, dtype=np.float32)[a_prev]
    t = rng.integers(-1, 2, (rows, 5, 5, 3)).astype(np.float32)
    v = rng.uniform(-0.5, 1.0, (rows, 4)).astype(np.float32)
    oo = rng.uniform(-0.5, 0.5, (rows, lo)).astype(np.float32)
    g = np.eye(2, dtype=np.float32)[rng.integers(0, 2, rows)]
    probs = net.actor_checkers(S._t(a1), S._t(t), S._t(v), S._t(oo), S._t(g), f1=6, k1=[3, 3], n_h1=256, n_h2=256,
                               n_actions=5, stage=stage)
    return shim.weights, dict(a_prev=a_prev, obs_self_t=t, obs_self_v=v, obs_others=oo, goals=g), np.asarray(probs)


def pack(cases):
    rec = {}
    for tag, (w, inputs, probs) in cases.items():
        rec[tag + "/names"] = np.array(sorted(w))
        for k, v in w.items():
            rec[tag + "/w/" + k] = v
        for k, v in inputs.items():
            rec[tag + "/in/" + k] = v
        rec[tag + "/probs"] = probs
    return rec


def main():
    rng = np.random.default_rng(20260929)
    out = os.path.join(ROOT, "tests", "golden")
    pc = {"n4_stage2": particle_case(rng, 4, 2, 96), "n1_stage1": particle_case(rng, 1, 1, 40),
          "n8_stage2": particle_case(rng, 8, 2, 64)}
    np.savez_compressed(os.path.join(out, "actor_particle.npz"), **pack(pc))
    cc = {"n2_stage2": checkers_case(rng, 2, 2, 96), "n1_stage1": checkers_case(rng, 1, 1, 40)}
    np.savez_compressed(os.path.join(out, "actor_checkers.npz"), **pack(cc))
    for name, cases in (("actor_particle", pc), ("actor_checkers", cc)):
        for tag, (w, _, probs) in cases.items():
            print(name, tag, sorted(w), probs.shape, float(np.ptp(probs, axis=1).mean()))


if __name__ == "__main__":
    main()
