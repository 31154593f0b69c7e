#!/bin/bash
# (1) parity tests that reach the non-temporal kernels, (2) full suite, (3) same-box timing of Checkers: HEAD build vs working tree
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
timeout 1200 python -m pytest tests/test_gpu_nt_paths.py -m gpu -q > "$O/pytest_nt.log" 2>&1; echo "nt-path tests rc=$?"; grep -E "^FAILED|^ERROR" "$O/pytest_nt.log" | sed 's/ - .*//' | head -20; tail -1 "$O/pytest_nt.log"
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_nt_paths.py > "$O/pytest_gpu.log" 2>&1; echo "suite rc=$?"; grep -E "^FAILED|^ERROR" "$O/pytest_gpu.log" | head; tail -1 "$O/pytest_gpu.log"
echo "workload mode build  us/tick   (3 alternating runs)" | tee "$O/ck_base_vs_new.txt"
for rep in 1 2 3; do for spec in "c3 trajectory" "c3 in-place"; do set -- $spec; for b in base new; do
  lib=""; [ $b = base ] && lib="$R/cm3_amd/libcm3_hip_base.so"
  v=$(CM3_AMD_LIB=$lib timeout 300 python bench.py --workload $1 --mode $2 --no-extras --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readline())['us_per_tick'])")
  echo "$1 $2 $b $v" | tee -a "$O/ck_base_vs_new.txt"
done; done; done
