#!/usr/bin/env python
"""Checkers twin of tools/trajectory_gap.py: C3 (2 agents, 8192 envs), 330-tick hipGraph of cm3_checkers_rollout, the slot strides
of cm3_checkers_traj switched on group by group (stride 0 = that output is overwritten in place every tick).  The library picks
its non-temporal 16-lane kernel from the observation strides, so the first rows also show that switch."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cm3_amd  # noqa: E402
from cm3_amd import _lib  # noqa: E402
from cm3_amd.checkers import VecCheckersEnv  # noqa: E402
from cm3_amd.rollout import CheckersRollout  # noqa: E402

T = 330
GROUPS = [("obs_self_t",), ("grid",), ("vec", "obs_others", "obs_self_v"), ("actions", "local_rewards", "reward", "done"),
          ("term_grid", "term_vec", "term_obs_others", "term_obs_self_t", "term_obs_self_v"), ("goals_slots",)]


def main():
    cfg = cm3_amd.load_config("checkers_stage2")
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    lib = _lib.lib()
    env = VecCheckersEnv(cfg["init"], cfg["n_agents"], int(os.environ.get("MAX_STEPS", "33")), 8192, device=dev, seed=1, auto_reset=True)
    env.reset(np.eye(2))
    ro = CheckersRollout(env, n_ticks=T)
    full = ro._traj()

    def stride_field(name):
        return name + ("_slot_stride" if name in ("grid", "obs_self_t", "term_grid", "term_obs_self_t") else "_stride")

    def traj(n_groups):
        t = _lib.CheckersTraj()
        ctypes.memmove(ctypes.byref(t), ctypes.byref(full), ctypes.sizeof(t))
        for g in GROUPS[n_groups:]:
            for name in g:
                setattr(t, stride_field(name), 0)
                if name.startswith("term_") or name == "goals_slots":
                    setattr(t, name, None)
        return t

    graphs = []
    env._desc.flags = _lib.FLAG_AUTO_RESET | _lib.FLAG_GEN_ACTIONS
    for k in range(len(GROUPS) + 1):
        t = traj(k)
        _lib.check(lib.cm3_checkers_rollout(ctypes.byref(env._desc), ctypes.byref(t), T, stream.cuda_stream))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            _lib.check(lib.cm3_checkers_rollout(ctypes.byref(env._desc), ctypes.byref(t), T, stream.cuda_stream))
        graphs.append(g)
    best = [1e9] * len(graphs)
    for rep in range(4):
        for k, g in enumerate(graphs):
            g.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(5):
                g.replay()
            b.record(stream); b.synchronize()
            best[k] = min(best[k], a.elapsed_time(b) * 1e3 / (5 * T))
    print("C3: 2 agents, 8192 envs, us per tick (330-tick hipGraph, best of 4 alternating rounds)")
    print("   %-70s %.3f" % ("in place (all slot strides 0, no terminal capture)", best[0]))
    for k, g in enumerate(GROUPS):
        print("   %-70s %.3f" % ("+ slots: " + ", ".join(g), best[k + 1]))


if __name__ == "__main__":
    main()
