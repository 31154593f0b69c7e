"""bench.py --gpus N without a launcher re-executes itself through torch.distributed.run: here, with
no GPU at all, every spawned rank must report the missing device -- after spawning, not before."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_spawns_ranks_that_report_missing_devices():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered by tests/test_gpu_multirank.py on a GPU box")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    text = r.stderr + r.stdout
    assert "needs 2 GPUs on this node, only 0 visible" in text
    assert "rank 0:" in text and "rank 1:" in text           # both ranks were started
