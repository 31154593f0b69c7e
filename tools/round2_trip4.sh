#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; O="$R/gpurun_out"; mkdir -p "$O"; rm -f "$O/f32_free_running_drift.txt"; cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_particle.py -m gpu -q -k "free_running_drift" > "$O/pytest_drift.log" 2>&1; tail -1 "$O/pytest_drift.log"
awk '$2=="env"' "$O/f32_free_running_drift.txt" | sort -k4 -g -r
timeout 300 python tools/headline_ab.py 2>/dev/null | tee "$O/headline_ab.txt"
