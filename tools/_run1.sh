set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; O="$R/gpurun_out/r3a"; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_nt_paths.py -q -m gpu -k headline > "$O/test_headline.log" 2>&1; echo "headline test rc=$?"; tail -3 "$O/test_headline.log"
CM3_AMD_LIB=$R/cm3_amd/libcm3_hip_span.so timeout 900 python tools/kernel_span.py c2 c3 c5 floor --json > "$O/span.txt" 2> "$O/span.err"; echo "span rc=$?"; grep -v "^{" "$O/span.txt"
ub() { CM3_AMD_LIB=$1 timeout 300 python bench.py --workload $2 --mode $3 --no-extras --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readline())['us_per_tick'])"; }
echo "waves-per-WG A/B (us per tick; lib w2 / product(4) / w8), 3 alternating rounds"
for rep in 1 2 3; do for spec in "c2 trajectory" "c2 in-place" "c5 trajectory" "c4 trajectory"; do set -- $spec
  echo "$1 $2 w2=$(ub $R/cm3_amd/libcm3_hip_w2.so $1 $2) w4=$(ub "" $1 $2) w8=$(ub $R/cm3_amd/libcm3_hip_w8.so $1 $2) span=$(ub $R/cm3_amd/libcm3_hip_span.so $1 $2)"
done; done
