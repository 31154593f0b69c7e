#!/bin/bash
# Reference artefacts of a round (ROUND=6 by default) on ONE build: full GPU suite, smoke, two-stamp kernel-span record, bench lines for every BASELINE
# config, PMC passes, rocprofv3 kernel-trace summary.  Run on the GPU box: gpurun -- 'bash tools/round_final.sh'
# (tools/variants/libcm3_hip_span.so [+ _marks.so] must have been built from the same sources:
#   CM3_EXTRA_FLAGS=-DCM3_SPAN_STAMPS CM3_OUT=$PWD/tools/variants/libcm3_hip_span.so CM3_OBJ_DIR=/tmp/obj_span CM3_SKIP_ISA_LINT=1 bash cm3_amd/csrc/build.sh)
set -u
RN="${ROUND:-6}"; R="${GRAFT_REPO_ROOT:-$PWD}"; O="$R/gpurun_out/r${RN}final"; mkdir -p "$O"; cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
timeout 2400 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; echo "suite rc=$?"; grep -E "^FAILED|^ERROR" "$O/pytest_gpu.log" | head; tail -1 "$O/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$O/smoke.log"
# undisturbed kernel durations (two stamps per wave, unprofiled graph replays)
CM3_AMD_LIB=$R/tools/variants/libcm3_hip_span.so timeout 900 python tools/kernel_span.py c2 c3 c4 c5 floor --json > "$O/kernel_span.txt" 2> "$O/kernel_span.err"; echo "span rc=$?"
grep "^{" "$O/kernel_span.txt" | tail -1 > "$O/kernel_span.json"; cp "$O/kernel_span.json" "$R/gpurun_out/r0${RN}_kernel_span.json"
if [ -f "$R/tools/variants/libcm3_hip_marks.so" ]; then CM3_AMD_LIB=$R/tools/variants/libcm3_hip_marks.so timeout 600 python tools/kernel_span.py c2 c3 c5 > "$O/kernel_span_marks.txt" 2>&1; echo "marks rc=$?"; fi
# (since round 5 the LAST stdout line is the driver's compact record, < 4 KB; the full record goes to --extras-file)
timeout 1200 python bench.py --extras-file "$O/bench_c2.json" > "$O/bench_c2.stdout" 2> "$O/bench_c2.err"; echo "bench c2 rc=$?"
tail -1 "$O/bench_c2.stdout" > "$O/driver_line_c2.json"; echo "driver line: $(wc -c < "$O/driver_line_c2.json") bytes"
for wl in c3 c4 c5; do timeout 900 python bench.py --workload $wl --no-sweep --extras-file "$O/bench_$wl.json" > "$O/bench_$wl.stdout" 2> "$O/bench_$wl.err"; echo "bench $wl rc=$?"; tail -1 "$O/bench_$wl.stdout" > "$O/driver_line_$wl.json"; done
for wl in c2 c3 c4 c5; do python - "$O/bench_$wl.json" $wl <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "us/tick %.3f" % d["us_per_tick"], "value %.4g" % d["value"], "frac %.4f" % r["frac"], "events %.3f" % r["avg_launch_us_hip_events"],
      "floor", r.get("launch_floor", {}).get("frac_of_floor"), {k: round(v["us_per_tick"], 2) for k, v in d.get("launch_modes", {}).items() if isinstance(v, dict)},
      "span", r.get("kernel_span", {}).get("span_us"), "read_GBps %.0f" % r.get("measured_read_GBps", 0), "cpu", round(d.get("cpu_baseline", {}).get("value", 0)))
if "policy_rollout" in d and "headline" in d["policy_rollout"]:
    h = d["policy_rollout"]["headline"]; print("   policy headline: %.3f us/tick, %.3g env-steps/s, %.1f TFLOP/s" % (h["us_per_tick"], h["env_steps_per_s"], h["roofline"]["achieved"]))
PY
done
# PMC passes (separate --pmc passes, kernel-trace only) for the six headline / in-place lines
bash tools/pmc_all.sh > "$O/pmc_all.log" 2>&1; echo "pmc rc=$?"
# C2 trajectory, the two ways of chaining ticks: stepping in place + a slot copy of the state (live, the product's
# choice at this size: 64 B per env more written) against chaining through the slots -- time and HBM traffic of both
for ls in on off; do
  bash tools/pmc_run.sh c2_trajectory_live_$ls --workload c2 --mode trajectory --live-state $ls --no-extras --steps 6 --warmup 2 > gpurun_out/pmc_c2_trajectory_live_$ls.runlog 2>&1
  cd "$R"
  python tools/pmc_summary.py gpurun_out/pmc_c2_trajectory_live_$ls k_particle_step c2_trajectory_live_$ls gpurun_out/pmc_traffic_new.json > gpurun_out/pmc_c2_trajectory_live_${ls}_summary.txt 2>&1
  rm -rf gpurun_out/pmc_c2_trajectory_live_$ls
done
for rep in 1 2 3; do for ls in on off; do
  v=$(timeout 300 python bench.py --workload c2 --live-state $ls --no-extras --no-sweep --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readline())['us_per_tick'])")
  echo "c2 trajectory live-state $ls: $v us per tick"
done; done > "$O/live_state_ab.txt" 2>&1; cat "$O/live_state_ab.txt"
# the one-launch Checkers policy rollout: kernel-trace stats + PMC groups
bash tools/pmc_ck_policy.sh > "$O/pmc_ck_policy.log" 2>&1; echo "ck policy pmc rc=$?"
timeout 120 python bench.py --workload c1 --steps 5 > "$O/bench_c1.stdout" 2>&1; echo "c1 rc=$?"; tail -1 "$O/bench_c1.stdout" | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c2" -o c2 -- python "$R/bench.py" --no-extras --no-sweep --no-cpu-baseline > "$O/prof_c2.log" 2>&1; echo "rocprof rc=$?"
cd "$R"; python tools/rocprof_summary.py "$O/prof_c2" > "$O/prof_c2_kernel_stats.txt" 2>&1; head -4 "$O/prof_c2_kernel_stats.txt"
grep "^{" "$O/prof_c2.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench under rocprofv3: us/tick %.3f ms/step %.3f' % (d['us_per_tick'], d['ms_per_step']))"
rm -rf "$O/prof_c2"
