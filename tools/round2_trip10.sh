#!/bin/bash
# same-box A/B: baseline library (commit 6a3c072) vs the current build with COMPILE-TIME plain / nt observation stores
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
timeout 2400 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR" "$O/pytest_gpu.log" | head; tail -1 "$O/pytest_gpu.log"
echo "workload mode build  us/tick   (bench.py --no-extras, wall clock; 3 alternating runs)" | tee "$O/base_vs_new.txt"
for rep in 1 2 3; do for spec in "c2 trajectory" "c5 trajectory" "c5 in-place" "c4 trajectory" "c2 in-place"; do set -- $spec; for b in base new; do
  lib=""; [ $b = base ] && lib="$R/cm3_amd/libcm3_hip_base.so"
  v=$(CM3_AMD_LIB=$lib timeout 300 python bench.py --workload $1 --mode $2 --no-extras --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readline())['us_per_tick'])")
  echo "$1 $2 $b $v" | tee -a "$O/base_vs_new.txt"
done; done; done
