#!/bin/bash
# GPU trip 2 of round 2: the whole GPU suite (no -x), chains diagnostic, read-probe repeats, C3/C4/C5 bench lines.
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"
O="$R/gpurun_out"
mkdir -p "$O"; rm -f "$O/f32_free_running_drift.txt"
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
timeout 2400 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; tail -25 "$O/pytest_gpu.log"
sort -k3 -g -r "$O/f32_free_running_drift.txt" | head -12
timeout 600 python tools/chains_diag.py > "$O/chains_diag.txt" 2>&1; cat "$O/chains_diag.txt"
python - > "$O/hbm_probe_repeat.txt" 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
from cm3_amd import _lib
lib = _lib.lib(); dev = torch.device("cuda:0"); nbytes = 4 << 30
buf = torch.empty(nbytes // 4, dtype=torch.int32, device=dev); buf.random_(0, 1 << 30)
sink = torch.zeros(lib.cm3_hbm_bench_sink_words(), dtype=torch.int32, device=dev)
s = _lib.current_stream_handle(dev)
def t(u, wg, nt, reps=20):
    f = lambda: _lib.check(lib.cm3_hbm_read_bench_cfg(buf.data_ptr(), nbytes, sink.data_ptr(), u, wg, nt, s))
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); b.synchronize()
    return nbytes / (a.elapsed_time(b) / reps * 1e-3) / 1e9
print("# read probe, 20 reps per measurement, 3 measurements per configuration (unroll, wg_per_cu, nt): GB/s")
for cfg in ((1, 8, 1), (2, 4, 1), (16, 32, 1), (8, 32, 1), (1, 4, 1), (8, 16, 1), (4, 8, 0), (1, 8, 0)):
    print(cfg, ["%.1f" % t(*cfg) for _ in range(3)])
PY
cat "$O/hbm_probe_repeat.txt"
for wl in c5 c3 c4; do
  timeout 600 python bench.py --workload $wl --no-sweep --no-cpu-baseline > "$O/bench_$wl.json" 2> "$O/bench_$wl.err"
  echo "bench $wl rc=$?"; python - "$O/bench_$wl.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "us_per_tick")}, d["roofline"]["frac"], d["roofline"].get("launch_floor", {}).get("frac_of_floor"))
print({k: round(v["us_per_tick"], 2) for k, v in d.get("launch_modes", {}).items() if isinstance(v, dict)})
PY
done
