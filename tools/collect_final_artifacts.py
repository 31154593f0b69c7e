#!/usr/bin/env python
"""Copies what tools/round_final.sh left under gpurun_out/ into profiles/ (the tracked, judged place).

    python tools/collect_final_artifacts.py [round = 6]   # after gpurun -- 'bash tools/round_final.sh'

bench lines -> profiles/rNN_bench_c{2..5}.json, the two-stamp record -> profiles/rNN_kernel_span_c{2,3,4,5}.txt (+ .json),
PMC summaries -> profiles/rNN_pmc_*.txt + profiles/pmc_traffic.json, the rocprofv3 kernel-trace summary ->
profiles/rNN_bench_c2_kernel_stats.txt, the test / smoke tails -> profiles/rNN_pytest_gpu_summary.txt.
"""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
import sys
RN = int(sys.argv[1]) if len(sys.argv) > 1 else 6
RP = "r%02d_" % RN
O = os.path.join(G, "r%dfinal" % RN)

HEAD = """# FINAL build of the round: undisturbed kernel-duration record (tools/kernel_span.py; two-stamp build libcm3_hip_span.so =
# csrc/build.sh with -DCM3_SPAN_STAMPS, csrc/common.h): every wave stamps s_memrealtime (100 MHz) + s_memtime at its first
# instruction and after its last store has been acknowledged; bench.py's own 330-launch hipGraph, NOT profiled; stamps of the
# last of 20 timed replays.  span = first wave in -> last wave out; start-to-start = consecutive first-wave-in; gap = the
# dependent-launch boundary.  The stamp store itself (one more 32-byte store per wave into its own 128-byte record, after the
# wave's last store has been acknowledged) lengthens the BOUNDARY in this build -- by 0.1-0.2 us at 1024 waves per launch, by
# ~0.6 us in the C5 trajectory launches (2048 waves); the spans are not affected.  The product's time per launch is bench.py's
# us_per_tick of the same box, quoted below: read span from this record, and gap as (product time per launch - span).
"""


def first_json(path):
    with open(path) as fh:
        return json.load(fh)            # (round 5: the full record is bench.py's --extras-file, one JSON document)


def main():
    bench = {}
    for wl in ("c2", "c3", "c4", "c5"):
        d = bench[wl] = first_json(os.path.join(O, "bench_%s.json" % wl))
        with open(os.path.join(P, RP + "bench_%s.json" % wl), "w") as f:
            f.write(json.dumps(d) + "\n")
    # the span record: header lines (#) + one block per "== name" section
    lines = open(os.path.join(O, "kernel_span.txt")).read().splitlines()
    tool_head = [l for l in lines if l.startswith("#")]
    blocks, cur = {}, None
    for l in lines:
        if l.startswith("== "):
            cur = l[3:].split(":")[0]
            blocks[cur] = []
        if cur and not l.startswith("{") and not l.startswith("#"):
            blocks[cur].append(l)
    for wl in ("c2", "c3", "c4", "c5"):
        d = bench[wl]
        inpl = d.get("launch_modes", {}).get("in_place", d.get("launch_modes", {}).get("in_place_chains1", {})).get("us_per_tick")
        out = [HEAD.rstrip("\n")] + tool_head
        out.append("# product library, same box, same call: bench.py --workload %s: trajectory mode %.3f us per tick, in place %s"
                   % (wl, d["us_per_tick"], "%.3f" % inpl if inpl else "n/a"))
        for name, b in blocks.items():
            if name.startswith(wl + "_") or (wl == "c2" and name.startswith("floor_")):
                out += b
        with open(os.path.join(P, RP + "kernel_span_%s.txt" % wl), "w") as f:
            f.write("\n".join(out) + "\n")
    shutil.copy(os.path.join(O, "kernel_span.json"), os.path.join(P, RP + "kernel_span.json"))
    if os.path.exists(os.path.join(O, "kernel_span_marks.txt")):
        with open(os.path.join(P, RP + "kernel_span_marks.txt"), "w") as f:
            f.write("# FINAL build of the round with -DCM3_SPAN_STAMPS -DCM3_SPAN_MARKS (tools/kernel_span.py c2 c3 c5): per-wave shader-clock marks inside\n"
                    "# the step kernels (the marks pin the schedule around them and lengthen the kernels a little: read the two-stamp record for\n"
                    "# durations, this one for where a wave's time goes).\n")
            f.write("".join(l for l in open(os.path.join(O, "kernel_span_marks.txt")) if "amdgpu.ids" not in l))
    for s in glob.glob(os.path.join(G, "pmc_*_summary.txt")):
        name = os.path.basename(s)[len("pmc_"):-len("_summary.txt")]
        shutil.copy(s, os.path.join(P, RP + "pmc_%s.txt" % name))
    if os.path.exists(os.path.join(G, "pmc_traffic_new.json")):
        shutil.copy(os.path.join(G, "pmc_traffic_new.json"), os.path.join(P, "pmc_traffic.json"))
    shutil.copy(os.path.join(O, "prof_c2_kernel_stats.txt"), os.path.join(P, RP + "bench_c2_kernel_stats.txt"))
    with open(os.path.join(P, RP + "pytest_gpu_summary.txt"), "w") as f:
        f.write("# python -m pytest tests -m gpu -q on the MI355X box (tools/round_final.sh), then __graft_entry__.smoke()\n")
        f.write("".join(open(os.path.join(O, "pytest_gpu.log")).readlines()[-12:]))
        f.write("".join(open(os.path.join(O, "smoke.log")).readlines()[-1:]))
    for wl in bench:
        shutil.copy(os.path.join(O, "driver_line_%s.json" % wl), os.path.join(P, RP + "driver_line_%s.json" % wl))
    if os.path.exists(os.path.join(O, "live_state_ab.txt")):
        shutil.copy(os.path.join(O, "live_state_ab.txt"), os.path.join(P, RP + "live_state_ab_c2.txt"))
    for wl, d in bench.items():
        print(wl, "%.3f us/tick" % d["us_per_tick"], "%.4g %s" % (d["value"], d["unit"]))


if __name__ == "__main__":
    main()
