"""CPU: device replay buffers / CSV rows against the reference's own classes (container) and their documented
semantics.  The buffers are torch-device agnostic; CPU tensors here, the GPU rollout tests feed them device columns."""
import os
import random
import sys

import pytest
import torch

from cm3_amd.replay import CsvLog, DeviceDualReplayBuffer, DeviceReplayBuffer

REF_ALG = "/root/reference/alg"


def _cols(start, n):
    i = torch.arange(start, start + n)
    return {"id": i.clone(), "x": torch.stack([i.float(), -i.float()], dim=1)}


def test_ring_semantics_match_reference_buffer():
    ours = DeviceReplayBuffer(size=10, device="cpu")
    start = 0                                      # 18 transitions into a ring of 10
    for chunk in (4, 5, 6, 3):
        ours.add(_cols(start, chunk))
        start += chunk
    if os.path.isdir(REF_ALG):
        sys.path.insert(0, REF_ALG)
        sys.dont_write_bytecode = True
        import replay_buffer
        ref = replay_buffer.Replay_Buffer(size=10)
        for t in range(18):
            ref.add(t)
        assert sorted(ours.all()["id"].tolist()) == sorted(ref.memory)
        assert [ours.cols["id"][k].item() for k in range(10)] == ref.memory       # same slots, too
    assert len(ours) == 10 and sorted(ours.all()["id"].tolist()) == list(range(8, 18))
    b = ours.sample_batch(4, generator=torch.Generator().manual_seed(0))
    assert b["id"].shape == (4,) and len(set(b["id"].tolist())) == 4
    assert torch.equal(b["x"][:, 0], b["id"].float())
    assert ours.sample_batch(100)["id"].shape == (10,)                            # len <= size -> everything


def test_add_larger_than_capacity_keeps_the_newest():
    ours = DeviceReplayBuffer(size=5, device="cpu")
    ours.add(_cols(0, 12))
    assert sorted(ours.all()["id"].tolist()) == [7, 8, 9, 10, 11]


@pytest.mark.parametrize("n1,n2,size,want1,want2", [(100, 100, 20, 10, 10), (100, 3, 20, 17, 3), (4, 100, 20, 4, 16),
                                                    (4, 5, 20, 4, 5), (12, 3, 20, 12, 3)])
def test_dual_buffer_split_rule(n1, n2, size, want1, want2):
    """replay_buffer_dual.py:40-63."""
    buf = DeviceDualReplayBuffer(size=1000, device="cpu")
    cols = _cols(0, n1 + n2)
    bad = torch.arange(n1 + n2) < n1
    buf.add(cols, bad)
    b = buf.sample_batch(size)
    got1 = int((b["id"] < n1).sum())
    got2 = int((b["id"] >= n1).sum())
    assert (got1, got2) == (want1, want2)
    if os.path.isdir(REF_ALG):
        sys.path.insert(0, REF_ALG)
        import replay_buffer_dual
        ref = replay_buffer_dual.Replay_Buffer(size=1000)
        ref.add(list(range(n1)), is_bad=True)
        ref.add(list(range(n1, n1 + n2)), is_bad=False)
        random.seed(0)
        r = ref.sample_batch(size)
        assert (int((r < n1).sum()), int((r >= n1).sum())) == (got1, got2)


def test_csv_rows_have_the_reference_format(tmp_path):
    log = CsvLog(str(tmp_path / "log.csv"), str(tmp_path / "log_century.csv"), 2)
    log.log_episode(66, 2, -12.3456, [-6.1, -6.2456])
    log.log_century(3300, 100, -10.0, [-5.0, -5.0], -9.5, [-4.75, -4.75], 12.9, 80.2)
    assert open(tmp_path / "log.csv").read() == "Step,Episode,r_global,r_0,r_1\n66,2,-12.35,-6.10,-6.25\n"
    c = open(tmp_path / "log_century.csv").read().split("\n")
    assert c[0] == "Step,Century,r_global_avg,r_avg_0,r_avg_1,r_global_eval,r_eval_0,r_eval_1,r_eval_local,t_env (s),t_train(s)"
    assert c[1] == "3300,100,-10.00,-5.00,-5.00,-9.50,-4.75,-4.75,-9.50,12,80"


# ---- against outputs recorded from the REAL reference code (oracle/gen_golden_replay_csv.py) -------------------------------
def _fixture():
    import json
    from tests.helpers import GOLDEN
    return json.load(open(os.path.join(GOLDEN, "replay_csv.json")))


def test_csv_headers_and_rows_equal_what_the_reference_writes(tmp_path):
    """log.csv / log_century.csv: headers and rows byte for byte equal to the strings produced by the reference's own
    statements (alg/train_onpolicy.py:201-215, :399-404, :422-427, executed unmodified at fixture-generation time), for 1, 2 and
    4 agents and 4 value sets each, including %.2f rounding edge cases."""
    fx = _fixture()
    n_rows = 0
    for case in fx["csv"]:
        n = case["n_agents"]
        log = CsvLog(str(tmp_path / ("log_%d.csv" % n)), str(tmp_path / ("century_%d.csv" % n)), n)
        for row in case["rows"]:
            v = row["in"]
            log.log_episode(v["step"], v["idx_episode"], v["reward_global"], v["reward_local"])
            period = float(v["period"])
            log.log_century(v["step"], v["idx_episode"], v["reward_global_century"] / period,
                            [x / period for x in v["reward_local_century"]], v["r_global_eval"], v["r_local_eval"],
                            v["t_env"], v["t_train"])
            n_rows += 1
        assert open(log.log_path).read() == case["header"] + "".join(r["episode_row"] for r in case["rows"]), n
        assert open(log.century_path).read() == case["header_century"] + "".join(r["century_row"] for r in case["rows"]), n
    assert n_rows == 12


def test_ring_buffer_equals_recorded_reference_memory():
    """DeviceReplayBuffer slot contents after chunked adds == replay_buffer.Replay_Buffer.memory after the same sequence of
    single adds (alg/replay_buffer.py:11-16), as recorded from the real class."""
    for case in _fixture()["ring"]:
        ours = DeviceReplayBuffer(size=case["size"], device="cpu")
        start = 0
        for chunk in case["chunks"]:
            ours.add(_cols(start, chunk))
            start += chunk
        assert sorted(ours.all()["id"].tolist()) == sorted(case["memory"]), case
        if max(case["chunks"]) <= case["size"]:                      # same slots too (an over-long chunk keeps the newest)
            assert [ours.cols["id"][k].item() for k in range(len(case["memory"]))] == case["memory"], case


def test_dual_buffer_split_equals_recorded_reference_counts():
    """DeviceDualReplayBuffer.sample_batch takes as many `bad` / `good` transitions as replay_buffer_dual.Replay_Buffer does
    (alg/replay_buffer_dual.py:40-63), as recorded from the real class, including empty halves."""
    for case in _fixture()["dual"]:
        n1, n2 = case["n_bad"], case["n_good"]
        buf = DeviceDualReplayBuffer(size=1000, device="cpu")
        buf.add(_cols(0, n1 + n2), torch.arange(n1 + n2) < n1)
        b = buf.sample_batch(case["size"])
        assert (int((b["id"] < n1).sum()), int((b["id"] >= n1).sum())) == (case["taken_bad"], case["taken_good"]), case
