#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
timeout 2400 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; grep -E "^FAILED|^ERROR" "$O/pytest_gpu.log" | sed 's/ - .*//' | head; tail -1 "$O/pytest_gpu.log"
echo "workload obs-store  us/tick (3 alternating runs, bench.py --no-extras, wall clock)" | tee "$O/obs_store_by_workload.txt"
for rep in 1 2 3; do for wl in c2 c5 c4; do for pol in 0 1; do
  v=$(CM3_EXPERIMENT_OBS_STORE=$pol timeout 300 python bench.py --workload $wl --no-extras --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readline())['us_per_tick'])")
  echo "$wl $([ $pol = 0 ] && echo plain || echo nt) $v" | tee -a "$O/obs_store_by_workload.txt"
done; done; done
