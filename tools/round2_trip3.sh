#!/bin/bash
# GPU trip 3: whole GPU suite (no -x) after the freeze revert, default bench, kernel-trace profile of the headline.
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"
O="$R/gpurun_out"
mkdir -p "$O"; rm -f "$O/f32_free_running_drift.txt"
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
timeout 2400 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; grep -E "^FAILED|^ERROR" "$O/pytest_gpu.log" | sed 's/ - .*//' | head -40; tail -2 "$O/pytest_gpu.log"
sort -k4 -g -r "$O/f32_free_running_drift.txt" | awk '$2=="env"'
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$O/smoke.log"
timeout 600 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench rc=$?"
python - "$O/bench_default.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print({k: d[k] for k in ("value", "us_per_tick", "ms_per_step")})
print({k: r[k] for k in ("frac", "avg_launch_us", "avg_launch_us_hip_events", "measured_read_GBps", "measured_copy_GBps", "frac_of_measured_read")}, r.get("launch_floor", {}).get("frac_of_floor"))
print({k: round(v["us_per_tick"], 2) for k, v in d.get("launch_modes", {}).items() if isinstance(v, dict)})
for s in d.get("sweep", []): print(s["envs"], round(s["avg_launch_us"], 2), round(s["achieved_GBps"]), round(s["frac_of_measured_read"], 3))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c2" -o c2 -- python "$R/bench.py" --no-extras --no-sweep --no-cpu-baseline > "$O/prof_c2.log" 2>&1; echo "rocprof rc=$?"
tail -1 "$O/prof_c2.log" | head -c 600; echo
cd "$R"; python tools/rocprof_summary.py "$O/prof_c2" > "$O/prof_c2_kernel_stats.txt" 2>&1; head -8 "$O/prof_c2_kernel_stats.txt"
