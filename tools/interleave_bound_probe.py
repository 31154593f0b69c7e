#!/usr/bin/env python
"""Upper bound of what interleaving the per-tick launches over Q hardware queues could buy: the product's own T step launches of a
trajectory rollout captured as Q hipGraphs (tick t in graph t mod Q), replayed concurrently on Q streams WITHOUT any dependency
between the graphs (the results are therefore wrong -- this only measures how far the launches overlap when nothing holds them
back)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import cm3_amd
    from cm3_amd import _lib
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.checkers import VecCheckersEnv
    from cm3_amd.rollout import ParticleRollout, CheckersRollout
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    T, reps = 330, 10
    for name, N, E, cfgname in (("c2", 4, 4096, "particle_stage2_antipodal"), ("c5", 8, 8192, "particle_merge8"), ("c3", 2, 8192, "checkers_stage2")):
        cfg = cm3_amd.load_config(cfgname)
        for Q in (1, 2, 3, 4):
            if name == "c3":
                env = VecCheckersEnv(cfg["init"], 2, 33, E, device=dev, auto_reset=True)
                env.reset(np.eye(2))
                ro = CheckersRollout(env, n_ticks=T, use_graph=False)
                def enq_tick(t, s):
                    env._desc.flags = _lib.FLAG_AUTO_RESET | _lib.FLAG_GEN_ACTIONS
                    b = ro._bufs(t)
                    _lib.check(lib.cm3_checkers_step(__import__("ctypes").byref(env._desc), __import__("ctypes").byref(b), s))
            else:
                env = VecParticleEnv(cfg, N, 0.2, 33, E, device=dev, auto_reset=True)
                env.reset()
                ro = ParticleRollout(env, n_ticks=T, use_graph=False)
                flags = _lib.FLAG_AUTO_RESET | _lib.FLAG_GEN_ACTIONS
                live = name == "c2" or name == "c5"
                def enq_tick(t, s):
                    ro._enqueue(t, 1, flags, s, live=live)
            streams = [torch.cuda.Stream(device=dev) for _ in range(Q)]
            graphs = []
            for q in range(Q):
                def enq(s, q=q):
                    for t in range(q, T, Q):
                        enq_tick(t, s)
                graphs.append(_lib.capture_graph(dev, enq))

            def run():
                for g, s in zip(graphs, streams):
                    _lib.check(lib.cm3_graph_launch(g, s.cuda_stream))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) * 1e6 / (reps * T)
            print("%s  Q=%d  %.3f us per tick (no inter-graph dependencies: an upper bound of the overlap, results invalid for Q > 1)" % (name, Q, us))
            for g in graphs:
                lib.cm3_graph_destroy(g)


if __name__ == "__main__":
    main()
