#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; O="$R/gpurun_out"; mkdir -p "$O"; rm -f "$O/f32_free_running_drift.txt"; cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
timeout 2400 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; grep -E "^FAILED|^ERROR" "$O/pytest_gpu.log" | sed 's/ - .*//' | head -20; tail -1 "$O/pytest_gpu.log"
for wl in c3 c4 c5; do
  timeout 600 python bench.py --workload $wl --no-sweep > "$O/bench_$wl.json" 2> "$O/bench_$wl.err"; echo "bench $wl rc=$?"
  python - "$O/bench_$wl.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print({k: round(d[k], 3) if isinstance(d[k], float) else d[k] for k in ("value", "us_per_tick")}, "frac", round(r["frac"], 4),
      "floor", r.get("launch_floor", {}).get("frac_of_floor"), {k: round(v["us_per_tick"], 2) for k, v in d.get("launch_modes", {}).items() if isinstance(v, dict)},
      "cpu", round(d.get("cpu_baseline", {}).get("value", 0)))
PY
done
