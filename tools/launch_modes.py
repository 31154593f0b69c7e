#!/usr/bin/env python
"""SURVEY.md section 8(d) extra sweep at the C2 size: per-tick cost of the particle step with / without hipGraph
capture of 33 ticks, with in-kernel vs tensor actions, through the C rollout entry vs the Python step() surface."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cm3_amd  # noqa: E402
from bench import GRAPH_TICKS, ParticleStepper, timed_ticks  # noqa: E402
from cm3_amd import _lib  # noqa: E402
from cm3_amd.particle import VecParticleEnv  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    cfg = cm3_amd.load_config("particle_stage2_antipodal")
    E, N, n = 4096, 4, GRAPH_TICKS * 30
    rows = []

    def add(name, us):
        rows.append({"mode": name, "us_per_tick": round(us, 3), "env_steps_per_s": E / us * 1e6})
        print(json.dumps(rows[-1]), flush=True)

    st = ParticleStepper(cfg, N, E, dev)
    st.capture(GRAPH_TICKS)
    st.run(GRAPH_TICKS * 3)
    torch.cuda.synchronize()
    add("hipGraph(33 ticks), in-kernel actions, C rollout entry", timed_ticks(st, n) * 1e3 / n)
    st.graph_save, st.graph = st.graph, None
    add("eager launches from the C rollout loop, in-kernel actions", timed_ticks(st, n) * 1e3 / n)
    # tensor actions: pre-filled [E,N] int32, read by every tick
    st.env._desc.flags &= ~_lib.FLAG_GEN_ACTIONS
    st.env._actions[0].random_(0, 5)
    st.capture(GRAPH_TICKS)
    st.run(GRAPH_TICKS * 3)
    torch.cuda.synchronize()
    add("hipGraph(33 ticks), tensor actions", timed_ticks(st, n) * 1e3 / n)
    st.close()
    # Python surface: env.step() per tick (ctypes call + tuple of views per tick)
    env = VecParticleEnv(cfg, N, 0.2, 33, E, device=dev, auto_reset=True)
    env.reset()
    for _ in range(100):
        env.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        env.step()
    torch.cuda.synchronize()
    add("Python env.step() per tick, in-kernel actions (host-bound)", (time.perf_counter() - t0) * 1e6 / n)
    acts = torch.randint(0, 5, (E, N), dtype=torch.int32, device=dev)
    t0 = time.perf_counter()
    for _ in range(n):
        env.step(acts)
    torch.cuda.synchronize()
    add("Python env.step(actions) per tick, device int32 actions (host-bound)", (time.perf_counter() - t0) * 1e6 / n)
    fs = ParticleStepper(cfg, N, E, dev, fused=True)
    fs.capture(GRAPH_TICKS)
    fs.run(GRAPH_TICKS * 3)
    torch.cuda.synchronize()
    add("fused: 33 ticks per launch (CM3_FLAG_FUSED_TICKS)", timed_ticks(fs, n) * 1e3 / n)
    fs.close()


if __name__ == "__main__":
    main()
