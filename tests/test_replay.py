"""CPU: the host logic of the device replay buffers (ring bookkeeping, the dual buffer's split rule) and the CSV rows against the
reference's own classes (container) and against outputs recorded from them (tests/golden/replay_csv.json).  The buffers themselves
live on the GPU: tests/test_gpu_replay.py runs the same fixtures, and a real rollout, through DeviceReplayBuffer /
DeviceDualReplayBuffer on cuda:0."""
import os
import random
import sys

import pytest

from cm3_amd.replay import CsvLog, RingIndex, dual_take

REF_ALG = "/root/reference/alg"


class HostRing(object):
    """RingIndex driving a Python list: what the ring positions mean (slot k of the device column tensors = memory[k])."""

    def __init__(self, size):
        self.ring, self.memory = RingIndex(size), [None] * int(size)

    def add(self, items):
        skip, start, kept = self.ring.plan_add(len(items))
        for b in range(kept):
            self.memory[(start + b) % self.ring.maxsize] = items[skip + b]

    def stored(self):
        return self.memory[:self.ring.len]


def test_ring_semantics_match_reference_buffer():
    ours = HostRing(10)
    start = 0                                      # 18 transitions into a ring of 10
    for chunk in (4, 5, 6, 3):
        ours.add(list(range(start, start + chunk)))
        start += chunk
    if os.path.isdir(REF_ALG):
        sys.path.insert(0, REF_ALG)
        sys.dont_write_bytecode = True
        import replay_buffer
        ref = replay_buffer.Replay_Buffer(size=10)
        for t in range(18):
            ref.add(t)
        assert ours.stored() == ref.memory                                       # same slots
    assert ours.ring.len == 10 and sorted(ours.stored()) == list(range(8, 18))


def test_add_larger_than_capacity_keeps_the_newest():
    ours = HostRing(5)
    ours.add(list(range(12)))
    assert sorted(ours.stored()) == [7, 8, 9, 10, 11]
    assert ours.ring.idx == 12 % 5


@pytest.mark.parametrize("n1,n2,size,want1,want2", [(100, 100, 20, 10, 10), (100, 3, 20, 17, 3), (4, 100, 20, 4, 16),
                                                    (4, 5, 20, 4, 5), (12, 3, 20, 12, 3)])
def test_dual_buffer_split_rule(n1, n2, size, want1, want2):
    """replay_buffer_dual.py:40-63."""
    k1, all1, k2, all2 = dual_take(n1, n2, size)
    assert (k1, k2) == (want1, want2)
    assert (not all1 or k1 == n1) and (not all2 or k2 == n2)
    if os.path.isdir(REF_ALG):
        import numpy as np      # noqa: F401
        sys.path.insert(0, REF_ALG)
        import replay_buffer_dual
        ref = replay_buffer_dual.Replay_Buffer(size=1000)
        ref.add(list(range(n1)), is_bad=True)
        ref.add(list(range(n1, n1 + n2)), is_bad=False)
        random.seed(0)
        r = ref.sample_batch(size)
        assert (int((r < n1).sum()), int((r >= n1).sum())) == (k1, k2)


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


def test_ring_bookkeeping_equals_recorded_reference_memory():
    """RingIndex positions after chunked adds == replay_buffer.Replay_Buffer.memory after the same sequence of single adds
    (alg/replay_buffer.py:11-16), as recorded from the real class."""
    for case in _fixture()["ring"]:
        ours = HostRing(case["size"])
        start = 0
        for chunk in case["chunks"]:
            ours.add(list(range(start, start + chunk)))
            start += chunk
        assert sorted(ours.stored()) == sorted(case["memory"]), case
        if max(case["chunks"]) <= case["size"]:                      # same slots too (an over-long chunk keeps the newest)
            assert ours.stored() == case["memory"], case


def test_dual_split_rule_equals_recorded_reference_counts():
    """dual_take takes as many `bad` / `good` transitions as replay_buffer_dual.Replay_Buffer.sample_batch does
    (alg/replay_buffer_dual.py:40-63), as recorded from the real class, including empty halves."""
    for case in _fixture()["dual"]:
        k1, _, k2, _ = dual_take(case["n_bad"], case["n_good"], case["size"])
        assert (k1, k2) == (case["taken_bad"], case["taken_good"]), case


def test_device_buffers_have_no_host_path():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: tests/test_gpu_replay.py")
    from cm3_amd._lib import Cm3Error
    from cm3_amd.replay import DeviceReplayBuffer
    with pytest.raises(Cm3Error):
        DeviceReplayBuffer(size=4, device="cpu")
