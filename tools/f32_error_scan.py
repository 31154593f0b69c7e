#!/usr/bin/env python
"""Largest one-tick error of the float32 particle step against the float64 oracle over crowded random states (teacher-forced:
both start every tick from the same float32-representable state), per agent count.  Run on the GPU box; CM3_AMD_LIB selects the
library.  The test suite asserts these stay below 1e-5 (tests/test_gpu_particle.py); this prints the actual margins."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cm3_amd  # noqa: E402
from oracle.particle_oracle import VecParticleOracle  # noqa: E402  (a measurement tool, like the tests: not product code)


def random_states(rng, E, N, crowd=0.5):
    pos = rng.uniform(-1, 1, (E, N, 2))
    pos[rng.random(E) < crowd] *= 0.25
    vel = rng.normal(0, 0.7, (E, N, 2))
    lm = rng.uniform(-1, 1, (E, N, 2))
    return pos, vel, lm


def main():
    dev = torch.device("cuda:0")
    for name, N, E in (("particle_stage2_antipodal", 4, 4096), ("particle_merge8", 8, 4096), ("particle_stage2_merge", 2, 4096)):
        cfg = cm3_amd.load_config(name)
        rng = np.random.default_rng(99 + N)
        env = cm3_amd.VecParticleEnv(cfg, N, 0.2, 33, E, dev, seed=1, dtype=torch.float32)
        orc = VecParticleOracle(N, cfg, 0.2, 33, E)
        worst = dict(state=0.0, obs=0.0, reward_n=0.0)
        for it in range(20):
            pos, vel, lm = (x.astype(np.float32).astype(np.float64) for x in random_states(rng, E, N))
            acts = rng.integers(0, 5, (E, N))
            orc.set_state(pos, vel, lm)
            w_gs, w_oo, _, w_rew, w_rn, _ = orc.step(acts)
            env.set_state(pos, vel, lm)
            gs, oo, _, rew, rn, _ = env.step(torch.as_tensor(acts))
            m_col, m_reach = orc.pair_margins()
            safe = (m_col > 1e-4) & (m_reach > 1e-4)
            worst["state"] = max(worst["state"], float(np.nanmax(np.abs(gs.cpu().numpy() - w_gs))))
            worst["obs"] = max(worst["obs"], float(np.nanmax(np.abs(oo.cpu().numpy() - w_oo))))
            worst["reward_n"] = max(worst["reward_n"], float(np.nanmax(np.abs(rn.cpu().numpy()[safe] - w_rn[safe]))))
        print("ERR N=%d  state %.3e  obs_others %.3e  reward_n %.3e   (bound 1e-5)" % (N, worst["state"], worst["obs"], worst["reward_n"]))


if __name__ == "__main__":
    main()
