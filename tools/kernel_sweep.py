#!/usr/bin/env python
"""A/B of the two particle step-kernel mappings over batch sizes (GPU box): avg launch time from HIP events
over hipGraph replays of 33 ticks, interleaved, and algorithmic GB/s."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cm3_amd  # noqa: E402
from bench import GRAPH_TICKS, ParticleStepper, algorithmic_bytes_per_env_step, timed_ticks  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    rows = []
    for cfg_name, N in (("particle_stage2_antipodal", 4), ("particle_merge8", 8), ("particle_stage2_merge", 2)):
        cfg = cm3_amd.load_config(cfg_name)
        for log2e in (12, 13, 14, 15, 16, 17, 18, 20, 22):
            E = 1 << log2e
            if N == 8 and log2e > 20:
                continue
            res = {}
            steppers = {k: ParticleStepper(cfg, N, E, dev, kernel=k) for k in ("env", "pair", "agent")}
            for st in steppers.values():
                st.capture(GRAPH_TICKS)
                st.run(GRAPH_TICKS * 2)
            torch.cuda.synchronize()
            reps = 5
            n = GRAPH_TICKS * (10 if log2e <= 16 else 3)
            for k in steppers:
                res[k] = []
            for _ in range(reps):
                for k, st in steppers.items():
                    res[k].append(timed_ticks(st, n) * 1e3 / n)        # us per launch
            row = dict(N=N, E=E)
            for k in steppers:
                us = sorted(res[k])[len(res[k]) // 2]
                row[k + "_us"] = round(us, 3)
                row[k + "_GBps"] = round(algorithmic_bytes_per_env_step(N) * E / us / 1e3, 1)
            rows.append(row)
            print(json.dumps(row), flush=True)
            for st in steppers.values():
                st.close()
            del steppers
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
