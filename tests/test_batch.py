"""cm3_amd.batch (device batch reformatting) against arrays produced by the REAL reference functions
alg_credit.process_batch / process_actions / process_goals / process_global_state and the n x n repeats of
train_step (tests/golden/batch_particle.npz, recorded by oracle/gen_golden_batch.py).  Bit-exact.
Runs on CPU tensors here and on the GPU under -m gpu."""
import os

import numpy as np
import pytest
import torch

from cm3_amd import batch as B

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_particle.npz")


def _check(device):
    z = np.load(GOLD)
    cols = {k[3:]: torch.as_tensor(z[k]).to(device) for k in z.files if k.startswith("in_")}
    out = B.process_batch(cols)
    names = ("n_steps", "v_global", "obs_others", "v_local", "actions_1hot", "actions_others_1hot", "reward",
             "reward_local", "v_global_next", "obs_others_next", "v_local_next", "done", "goals")
    assert out[0] == int(z["pb_n_steps"])
    for name, got in zip(names[1:], out[1:]):
        want = z["pb_" + name]
        g = got.cpu().numpy()
        assert g.shape == want.shape, name
        assert np.array_equal(g, want), name
        assert g.dtype == want.dtype, (name, g.dtype, want.dtype)
    gs, go = B.process_goals(out[12])
    assert np.array_equal(gs.cpu().numpy(), z["goals_self"]) and np.array_equal(go.cpu().numpy(), z["goals_others"])
    one, others, state = B.process_global_state(out[1])
    assert np.array_equal(one.cpu().numpy(), z["vg_one"])
    assert np.array_equal(others.cpu().numpy(), z["vg_others"])
    assert np.array_equal(state.cpu().numpy(), z["vg_state"])
    N = 4
    assert np.array_equal(B.repeat_indexed_by_n(one, N).cpu().numpy(), z["s_n_rep"])
    assert np.array_equal(B.repeat_indexed_by_n(others, N).cpu().numpy(), z["s_others_rep"])
    assert np.array_equal(B.repeat_indexed_by_m(out[4], N).cpu().numpy(), z["actions_self_rep"])
    assert np.array_equal(B.repeat_indexed_by_m(one, N).cpu().numpy(), z["s_m_rep"])
    assert np.array_equal(B.repeat_indexed_by_n(out[7].unsqueeze(1), N).squeeze(1).cpu().numpy(), z["reward_local_rep"])


def test_batch_reformatting_matches_reference_cpu_tensors():
    _check("cpu")


@pytest.mark.gpu
def test_batch_reformatting_matches_reference_on_device():
    _check("cuda:0")
