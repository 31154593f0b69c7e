"""CPU: the NumPy restatement of networks.actor_checkers (oracle/actor_checkers_oracle.py) against an independent
PyTorch float32 implementation of the same graph (conv2d NCHW with padding 1 == TF "SAME" 3x3 stride 1; NHWC flatten;
dense chain).  TensorFlow itself is not installable here, so this is the strongest pin available for that oracle."""
import numpy as np
import torch

from oracle import actor_checkers_oracle as AO


def _torch_probs(w, a_prev, t, v, oo, g):
    T = lambda k: torch.as_tensor(w[k])  # noqa: E731
    x = torch.as_tensor(t, dtype=torch.float32).permute(0, 3, 1, 2)                       # NHWC -> NCHW
    k = T("conv/Conv/weights").permute(3, 2, 0, 1)                                       # [kh,kw,cin,cout] -> [cout,cin,kh,kw]
    conv = torch.relu(torch.nn.functional.conv2d(x, k, T("conv/Conv/biases"), padding=1))
    flat = conv.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                               # back to (row, col, filter)
    lin = torch.relu(flat @ T("conv_linear/kernel") + T("conv_linear/bias"))
    a1 = torch.nn.functional.one_hot(torch.as_tensor(a_prev).long(), 5).float()
    cat = torch.cat([lin, torch.as_tensor(v, dtype=torch.float32), a1, torch.as_tensor(g, dtype=torch.float32)], 1)
    acc = torch.relu(cat @ T("branch_self/kernel") + T("branch_self/bias")) @ T("W_self_h2")
    if "stage-2/W_others_h2" in w:
        ho = torch.relu(torch.as_tensor(oo, dtype=torch.float32) @ T("stage-2/branch_others/kernel")
                        + T("stage-2/branch_others/bias"))
        acc = acc + ho @ T("stage-2/W_others_h2")
    h2 = torch.relu(acc + T("b"))
    return torch.softmax(h2 @ T("actor_out/kernel") + T("actor_out/bias"), dim=1).numpy()


def _inputs(rng, rows, n_agents):
    t = rng.integers(-1, 2, (rows, 5, 5, 3)).astype(np.float64)
    v = rng.uniform(-0.5, 1.0, (rows, 4))
    oo = rng.uniform(-0.5, 0.5, (rows, 2 * max(n_agents - 1, 1)))
    a_prev = rng.integers(0, 5, rows)
    g = np.eye(2)[rng.integers(0, 2, rows)]
    return a_prev, t, v, oo, g


def test_oracle_matches_torch_stage2():
    rng = np.random.default_rng(0)
    for n_agents in (2, 3):
        w = AO.init_weights(rng, n_agents, stage=2)
        inp = _inputs(rng, 257, n_agents)
        got, want = AO.actor_probs(w, *inp), _torch_probs(w, *inp)
        assert got.shape == (257, 5) and np.abs(got.sum(1) - 1).max() < 1e-6
        assert np.abs(got - want).max() < 2e-6
        assert np.ptp(got, axis=1).mean() > 0.05          # the random policy is not uniform


def test_oracle_matches_torch_stage1_ignores_others():
    rng = np.random.default_rng(1)
    w = AO.init_weights(rng, 1, stage=1)
    a_prev, t, v, oo, g = _inputs(rng, 64, 1)
    got = AO.actor_probs(w, a_prev, t, v, oo, g)
    assert np.abs(got - _torch_probs(w, a_prev, t, v, oo, g)).max() < 2e-6
    assert np.array_equal(got, AO.actor_probs(w, a_prev, t, v, oo + 1.0, g))


def test_conv_is_cross_correlation_with_zero_same_padding():
    """Hand-checkable case: a single 1 at the top-left corner and a kernel that is 1 only at (dr,dc)=(2,2) moves the
    pixel up-left out of the image except ... out[r,c] = x[r+1,c+1]."""
    x = np.zeros((1, 5, 5, 3), np.float32)
    x[0, 1, 1, 0] = 1
    w = np.zeros((3, 3, 3, 6), np.float32)
    w[2, 2, 0, 4] = 1
    out = AO.conv_same_3x3(x, w, np.zeros(6, np.float32))
    want = np.zeros((1, 5, 5, 6), np.float32)
    want[0, 0, 0, 4] = 1
    assert np.array_equal(out, want)
