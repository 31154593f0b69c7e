#!/usr/bin/env python
"""Which particle step-kernel mapping (lane per env / per ordered pair / per agent) is fastest for (n_agents, n_envs): in-place
stepping, 33-tick hipGraphs, HIP-event time per launch, the three mappings alternated, best of `reps` rounds.  Run on the GPU box:
    python tools/mapping_sweep.py [N ...]
Prints one row per (N, E) with the three times, the winner and what `auto` (the library's table in particle.hip, launch_n) takes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cm3_amd  # noqa: E402
from bench import ParticleStepper, timed_ticks  # noqa: E402

GRAPH_TICKS = 33

SIZES = (2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 98304, 131072, 262144, 524288, 1048576)
if os.environ.get("CM3_SWEEP_SIZES"):       # e.g. CM3_SWEEP_SIZES=65536,262144,1048576
    SIZES = tuple(int(x) for x in os.environ["CM3_SWEEP_SIZES"].split(","))


def main():
    ns = [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5, 6, 7, 8]
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    cfg = cm3_amd.load_config("particle_merge8")
    print("%2s %8s %9s %9s %9s %9s  %-6s %s" % ("N", "envs", "env", "pair", "agent", "auto", "best", "auto vs best"))
    for N in ns:
        for E in SIZES:
            if N * (N - 1) * 16 * E > (3 << 30):
                continue
            kinds = ["env", "pair", "agent", "auto"] if N >= 2 else ["env", "auto"]
            if E > 65536 and N >= 4:
                kinds.remove("pair")       # far behind there (round 2), and its 32-bit offsets run out first
            st = {k: ParticleStepper(cfg, N, E, dev, kernel=k) for k in kinds}
            for s in st.values():
                s.capture(GRAPH_TICKS)
                s.run(GRAPH_TICKS * 2)
            torch.cuda.synchronize()
            n = GRAPH_TICKS * (6 if E <= 65536 else 2)
            best = {k: 1e9 for k in kinds}
            for _ in range(3):
                for k, s in st.items():
                    best[k] = min(best[k], timed_ticks(s, n) * 1e3 / n)
            for s in st.values():
                s.close()
            del st
            torch.cuda.empty_cache()
            three = {k: best[k] for k in kinds if k != "auto"}
            win = min(three, key=three.get)
            print("%2d %8d %9.2f %9.2f %9.2f %9.2f  %-6s %+.1f %%" % (N, E, best.get("env", 0), best.get("pair", 0), best.get("agent", 0),
                                                                      best["auto"], win, (best["auto"] / three[win] - 1) * 100), flush=True)


if __name__ == "__main__":
    main()
