"""The driver keeps an 8 KB tail of bench.py's stdout and parses the LAST line: round 4's 24 KB line was cut in the middle and
the round lost its measurement row.  The line is now built by bench.compact_line() from the full record; here it is built from
every recorded run under profiles/ and must stay under 4 KB with the contract's keys in it."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDED = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_c*.json")))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _full(path):
    with open(path) as fh:
        return json.load(fh)


@pytest.mark.parametrize("path", RECORDED, ids=[os.path.basename(p) for p in RECORDED])
def test_driver_line_is_small_and_complete(path):
    import bench
    out = _full(path)
    if "roofline" not in out:                 # a compact line recorded since round 5: already what the driver saw
        pytest.skip("not a full record")
    line = bench.compact_line(out)
    text = json.dumps(line)
    assert len(text) < 4096 and len(text) < bench.LINE_LIMIT, len(text)
    assert "\n" not in text
    for k in CONTRACT:
        if k == "cpu_baseline" and k not in out:
            continue
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == pytest.approx(out["roofline"]["frac"], rel=1e-5)
    assert line["value"] == pytest.approx(out["value"], rel=1e-5)
    assert "workload" in line["config"] and "model" not in line["config"]
    if "cpu_baseline" in out:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in line["cpu_baseline"], k
    if "other_configs" in out:
        for name in ("c3", "c4", "c5"):
            rec = line["other_configs"][name]
            assert rec["us_per_tick"] > 0 and rec["value"] > 0 and 0 < rec["frac"] < 1
    summary = json.dumps(bench.extras_summary(out))
    assert len(summary) < bench.LINE_LIMIT


def test_there_is_a_recorded_run_with_every_extra():
    assert any("other_configs" in _full(p) and "sweep" in _full(p) and "policy_rollout" in _full(p) for p in RECORDED)


def test_line_survives_eight_ranks_and_a_collective():
    import bench
    out = _full(os.path.join(ROOT, "profiles", "r04_bench_c4.json"))
    out["n_gpus"] = 8
    out["per_rank"] = [dict(out["per_rank"][0], rank=r) for r in range(8)]
    out["rccl"] = {"rccl_world_size": 8, "backend": "nccl", "rccl_version": "2.22.3", "all_reduce_ok": True, "p2p_all_pairs": True,
                   "p2p_access": [{"rank": r, "visible_devices": 8, "can_access_peer": [1] * 8} for r in range(8)]}
    line = bench.compact_line(out)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert len(line["per_rank_wall_s"]) == 8 and line["rccl"]["rccl_world_size"] == 8
