#!/bin/bash
# N = 8 (merge8), in-place stepping, 33-tick hipGraph: us per tick and algorithmic GB/s for the lane-per-env and lane-per-agent
# mappings at large batches (the library's crossover was tuned in round 1, before the exact squared-distance thresholds).
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import cm3_amd
from bench import ParticleStepper, timed_ticks, algorithmic_bytes_per_env_step
cfg = cm3_amd.load_config("particle_merge8")
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
print("%9s %-6s %10s %10s" % ("envs", "kernel", "us/tick", "GB/s"))
for log2e in (16, 17, 18, 19, 20):
    E = 1 << log2e
    for kernel in ("env", "agent"):
        vals = []
        for rep in range(2):
            st = ParticleStepper(cfg, 8, E, dev, kernel=kernel)
            st.capture(33); st.run(66); torch.cuda.synchronize(dev)
            n = 33 * (6 if log2e <= 18 else 3)
            vals.append(timed_ticks(st, n) * 1e3 / n)
            st.close(); del st; torch.cuda.empty_cache()
        us = min(vals)
        print("%9d %-6s %10.2f %10.0f   (runs: %s)" % (E, kernel, us, algorithmic_bytes_per_env_step(8) * E / us / 1e3, " ".join("%.2f" % v for v in vals)))
PY
