#!/usr/bin/env python
"""Undisturbed kernel-duration record of the headline step launches.

Runs on the GPU box with the TWO-STAMP build of the library (csrc/common.h CM3_SPAN_STAMPS; built by
`CM3_EXTRA_FLAGS=-DCM3_SPAN_STAMPS CM3_OUT=tools/variants/libcm3_hip_span.so bash cm3_amd/csrc/build.sh`):

    CM3_AMD_LIB=$PWD/tools/variants/libcm3_hip_span.so python tools/kernel_span.py c2 [c3 c5 floor]

Every wave of a step launch records s_memrealtime (100 MHz, constant) + s_memtime (shader clock) at its first instruction and
again after its last store has been acknowledged.  The workload is bench.py's own headline stepper (330 step launches per
hipGraph replay, NOT profiled); the stamps of the LAST replay are reduced per launch to
    span           = last wave out - first wave in            (the kernel itself)
    start-to-start = first wave in of launch k+1 - of launch k (kernel + dependent-launch boundary)
    gap            = first wave in of launch k+1 - last wave out of launch k   (the boundary alone)
and the sum of the start-to-start intervals is checked against the HIP-event time of the same replays.
`floor` does the same for the load -> store skeleton with C2's traffic and for an empty launch of the same shape.
Prints a text report (copied to profiles/r03_kernel_span_<workload>.txt) and, with --json, one JSON object per workload.
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import cm3_amd  # noqa: E402
from cm3_amd import _lib  # noqa: E402

RT_NS = 10.0          # one s_memrealtime tick (100 MHz)
MAX_WAVES = 2048      # per launch, upper bound of the record area
REC = 16               # 64-bit words per wave record: rt_in, ck_in, rt_out, ck_out, 8 optional marks (ck), pad


def span_config(buf, n_slots):
    h = _lib.lib()
    try:
        fn = h.cm3_span_config
    except AttributeError:
        raise SystemExit("this library has no span stamps: build with -DCM3_SPAN_STAMPS and select it with CM3_AMD_LIB")
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    return fn(buf.data_ptr() if buf is not None else None, n_slots, MAX_WAVES * REC * 8)


def reduce_slots(rec):
    """rec: int64 [n_launches, MAX_WAVES, 4] -> dict of per-launch arrays (ticks of 10 ns / shader cycles)"""
    used = rec[:, :, 0] != 0
    waves = used.sum(1)
    big = np.iinfo(np.int64).max
    rt_in = np.where(used, rec[:, :, 0], big)
    rt_out = np.where(used, rec[:, :, 2], 0)
    start, last_in, end = rt_in.min(1), np.where(used, rec[:, :, 0], 0).max(1), rt_out.max(1)
    life_rt = np.where(used, rec[:, :, 2] - rec[:, :, 0], 0).sum(1) / np.maximum(waves, 1)
    life_ck = np.where(used, rec[:, :, 3] - rec[:, :, 1], 0).sum(1) / np.maximum(waves, 1)
    marks = []
    for k in range(8):     # optional -DCM3_SPAN_MARKS marks: mean over waves and launches of (mark - wave in), shader cycles
        m = used & (rec[:, :, 4 + k] != 0)
        marks.append(float((rec[:, :, 4 + k] - rec[:, :, 1])[m].mean()) if m.any() else None)
    return dict(marks=marks, waves=waves, start=start, end=end, span=end - start, skew_in=last_in - start, life_rt=life_rt, life_ck=life_ck)


def stats(x, scale=1.0):
    x = np.asarray(x, dtype=np.float64) * scale
    return "mean %.3f  median %.3f  min %.3f  p95 %.3f  max %.3f" % (x.mean(), np.median(x), x.min(), np.percentile(x, 95), x.max())


def report(name, desc, r, ev_us_per_launch, n_launches, out):
    us = RT_NS * 1e-3
    s2s = np.diff(r["start"])
    gap = r["start"][1:] - r["end"][:-1]
    lines = ["== %s: %s" % (name, desc),
             "   launches per replay %d, waves per launch %d (min %d)" % (n_launches, int(np.median(r["waves"])), int(r["waves"].min())),
             "   span (first wave in -> last wave out)      us: " + stats(r["span"], us),
             "   start-to-start (consecutive launches)      us: " + stats(s2s, us),
             "   gap (last wave out -> next first wave in)  us: " + stats(gap, us),
             "   first wave in -> last wave IN (dispatch)   us: " + stats(r["skew_in"], us),
             "   life of one wave (mean over waves)         us: " + stats(r["life_rt"], us) +
             "   | shader cycles: mean %.0f  (= %.2f GHz)" % (r["life_ck"].mean(), r["life_ck"].mean() / max(r["life_rt"].mean() * RT_NS, 1e-9)),
             "   sum of start-to-start over the replay: %.3f us per launch; HIP events around the same replays: %.3f us per launch "
             "(ratio %.4f)" % (s2s.mean() * us, ev_us_per_launch, s2s.mean() * us / ev_us_per_launch)]
    if any(m is not None for m in r["marks"]):
        lines.append("   marks (shader cycles after the wave's first instruction, mean): " +
                     "  ".join("m%d %.0f" % (k, m) for k, m in enumerate(r["marks"]) if m is not None) +
                     "  | out %.0f" % r["life_ck"].mean())
    print("\n".join(lines))
    out[name] = dict(marks_cycles=r["marks"], desc=desc, launches=n_launches, waves=int(np.median(r["waves"])),
                     span_us_mean=float(r["span"].mean() * us), span_us_median=float(np.median(r["span"]) * us),
                     span_us_min=float(r["span"].min() * us),
                     start_to_start_us_mean=float(s2s.mean() * us), start_to_start_us_median=float(np.median(s2s) * us),
                     gap_us_mean=float(gap.mean() * us), dispatch_skew_us_mean=float(r["skew_in"].mean() * us),
                     wave_life_us_mean=float(r["life_rt"].mean() * us), wave_life_cycles_mean=float(r["life_ck"].mean()),
                     hip_event_us_per_launch=float(ev_us_per_launch))


def run_stepper(name, make, n_launches, steps, warm, desc, out):
    dev = torch.device("cuda", 0)
    buf = torch.zeros(n_launches * MAX_WAVES * REC, dtype=torch.int64, device=dev)
    span_config(buf, n_launches)          # the next n_launches step launches (= the graph capture) take slots 0..n-1
    st = make()
    st.run(warm * n_launches)
    torch.cuda.synchronize(dev)
    ms = bench.timed_ticks(st, steps * n_launches)
    torch.cuda.synchronize(dev)
    used = span_config(None, 0)
    rec = buf.cpu().numpy().reshape(n_launches, MAX_WAVES, REC)
    st.close()
    if used != n_launches:
        print("   (note: %d slots were handed out, expected %d)" % (used, n_launches))
    report(name, desc, reduce_slots(rec), ms * 1e3 / (steps * n_launches), n_launches, out)


class FloorStepper(bench.GraphStepper):
    """the load -> store skeleton (cm3_traffic_floor_bench) as a hipGraph of n launches"""

    def __init__(self, rd, wr, blocks, threads=256):
        self.torch, self._lib_mod, self.lib = torch, _lib, _lib.lib()
        self.device = torch.device("cuda", 0)
        self.src = torch.zeros(max(rd, 16) // 4, dtype=torch.int32, device=self.device)
        self.dst = torch.zeros(max(wr, 16) // 4, dtype=torch.int32, device=self.device)
        self.rd, self.wr, self.blocks, self.threads = rd, wr, blocks, threads

    def enqueue(self, n_ticks, stream=None):
        s = self.stream() if stream is None else stream
        for _ in range(n_ticks):
            _lib.check(self.lib.cm3_traffic_floor_bench(self.src.data_ptr(), self.rd, self.dst.data_ptr(), self.wr, self.blocks,
                                                        self.threads, s))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want_json = "--json" in sys.argv
    steps, warm = 20, 5
    out = {}
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    print("# tools/kernel_span.py %s   library %s" % (" ".join(args), os.path.basename(_lib.LIB_PATH)))
    print("# two stamps per wave (s_memrealtime 100 MHz + s_memtime), unprofiled hipGraph replays, stamps of the last of %d timed "
          "replays; device %s" % (steps, torch.cuda.get_device_name(0)))
    T = bench.PHASE_TICKS
    for wl in (args or ["c2"]):
        if wl == "floor":
            N, E = 4, 4096
            rd, wr = (28 * N + 4) * E, (20 * N + 16 * N * max(N - 1, 1) + 12) * E
            rd, wr = (rd + 15) // 16 * 16, (wr + 15) // 16 * 16
            for nm, (a, b) in (("floor_same_traffic_c2", (rd, wr)), ("floor_empty_launch", (0, 0))):
                def make(a=a, b=b):
                    st = FloorStepper(a, b, 256)
                    st.capture(T)
                    return st
                run_stepper(nm, make, T, steps, warm,
                            "256 x 256 lanes, %d B read then %d B written per launch, no arithmetic" % (a, b), out)
            continue
        kind, cfg_name, E, desc = bench.WORKLOADS[wl]
        cfg = cm3_amd.load_config(cfg_name)
        N = cfg["n_agents"]
        if kind == "particle_adv":
            # round 4: one replay = one collection phase = 330 step launches (ten 33-tick rollouts) + the slot copies + the two launches
            # of cm3_returns_normalize_segments_* for the ten advantage steps: the stamps cover the 330 step launches, the rest of
            # the replay period (HIP events) is the tail launches + the replay boundary.  (The stepper also captures its
            # single-rollout graph at set-up: 33 more slots are handed out, none of them runs here.)
            TT = bench.PHASE_TICKS
            run_stepper("%s_rollout" % wl, lambda: bench.RolloutAdvStepper(cfg, N, E, dev), TT, steps, warm,
                        "%s, %d envs, one hipGraph replay per phase of 10 rollouts x 33 ticks + their normalisations" % (desc, E), out)
            r = out["%s_rollout" % wl]
            period = r["hip_event_us_per_launch"] * TT
            inside = r["start_to_start_us_mean"] * (TT - 1) + r["span_us_mean"]
            print("   replay period %.2f us; first step wave in -> last step wave out %.2f us; outside the step launches (slot copies, the "
                  "two advantage launches, the replay boundary) %.2f us" % (period, inside, period - inside))
            r["replay_period_us"], r["step_launches_us"] = period, inside
            continue
        for mode in ("trajectory", "in-place"):
            if kind == "particle":
                if mode == "trajectory":
                    make = lambda: bench.TrajectoryStepper(cfg, N, E, dev)  # noqa: E731
                else:
                    def make():
                        st = bench.ParticleStepper(cfg, N, E, dev)
                        st.capture(T)
                        return st
            elif kind == "checkers":
                if mode == "trajectory":
                    make = lambda: bench.CheckersTrajectoryStepper(cfg, E, dev)  # noqa: E731
                else:
                    def make():
                        st = bench.CheckersStepper(cfg, E, dev)
                        st.capture(T)
                        return st
            else:
                continue
            run_stepper("%s_%s" % (wl, mode.replace("-", "_")), make, T, steps, warm,
                        "%s, %d envs, %s mode, one step launch per tick" % (desc, E, mode), out)
    if want_json:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
