#!/bin/bash
# GPU trip 1 of round 2: full GPU test suite, default bench, probe sweep, kernel-trace profile of the headline.
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"
O="$R/gpurun_out"
mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -x --deselect tests/test_gpu_multirank.py > "$O/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$O/pytest_gpu.log"
tail -5 "$O/pytest_gpu.log"
timeout 900 python -m pytest tests/test_gpu_multirank.py -q > "$O/pytest_multirank.log" 2>&1
echo "multirank rc=$?"; tail -15 "$O/pytest_multirank.log"
timeout 600 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench rc=$?"; head -c 1500 "$O/bench_default.json"; tail -3 "$O/bench_default.err"
timeout 300 python tools/hbm_probe_sweep.py > "$O/hbm_probe_sweep.txt" 2>&1
tail -4 "$O/hbm_probe_sweep.txt"
