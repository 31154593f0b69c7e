"""CPU: every primitive of oracle/tf_numpy_shim.py -- the NumPy stand-in under which the reference's own policy-network bodies
(networks.actor_particle, networks.actor_checkers / convnet_1) were executed to obtain tests/golden/actor_*.npz -- against an
INDEPENDENT implementation that is in the container: PyTorch's CPU `linear`, `conv2d(padding="same")` and `softmax`, with the
NHWC <-> NCHW / [kh, kw, cin, cout] <-> [cout, cin, kh, kw] transposes written out.  TensorFlow itself cannot be imported here
(checked again in round 3), so the actor oracles stay "parity unpinned against TensorFlow"; what these tests remove are the
realistic ways a restatement of the published TF1 semantics goes wrong: a flipped (true convolution) kernel, asymmetric SAME
padding, a transposed dense kernel, softmax over the wrong axis, concat order.  """
import numpy as np
import torch
import torch.nn.functional as F

from oracle import actor_oracle as AO
from oracle.tf_numpy_shim import Shim


def _shim(weights):
    return Shim(weights=dict(weights), rng=np.random.default_rng(0))


def test_dense_is_x_times_kernel_plus_bias_then_activation():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((37, 11)).astype(np.float32)
    k = rng.standard_normal((11, 23)).astype(np.float32)          # TF layout [in, out]
    b = rng.standard_normal(23).astype(np.float32)
    tf = _shim({"fc/kernel": k, "fc/bias": b})
    got = np.asarray(tf.layers.dense(x, 23, activation=tf.nn.relu, use_bias=True, name="fc"))
    want = torch.relu(F.linear(torch.as_tensor(x), torch.as_tensor(k).t(), torch.as_tensor(b))).numpy()   # torch: [out, in]
    assert got.dtype == np.float32 and np.abs(got - want).max() < 2e-6
    got_nb = np.asarray(_shim({"fc/kernel": k}).layers.dense(x, 23, activation=None, use_bias=False, name="fc"))
    assert np.abs(got_nb - F.linear(torch.as_tensor(x), torch.as_tensor(k).t()).numpy()).max() < 2e-6
    # a transposed kernel would not even have the right shape for a non-square layer; for a square one it must differ
    ks = rng.standard_normal((11, 11)).astype(np.float32)
    sq = np.asarray(_shim({"fc/kernel": ks}).layers.dense(x, 11, use_bias=False, name="fc"))
    assert np.abs(sq - x @ ks.T).max() > 0.1


def test_conv2d_same_equals_torch_cross_correlation():
    """tf.contrib.layers.conv2d(padding="SAME", stride 1): NHWC input, [kh, kw, cin, cout] weights, CROSS-correlation (no kernel
    flip), symmetric zero padding for odd kernels -- exactly torch's conv2d(padding="same") after the layout transposes."""
    rng = np.random.default_rng(2)
    for (n, h, w, cin, cout, kh, kw) in ((5, 5, 5, 3, 6, 3, 3), (2, 7, 4, 2, 3, 3, 3), (3, 6, 6, 1, 2, 5, 5), (2, 5, 5, 3, 4, 1, 1)):
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        k = rng.standard_normal((kh, kw, cin, cout)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        tf = _shim({"Conv/weights": k, "Conv/biases": b})
        got = np.asarray(tf.contrib.layers.conv2d(x, cout, [kh, kw], 1, padding="SAME", activation_fn=tf.nn.relu))
        xt = torch.as_tensor(x).permute(0, 3, 1, 2)                              # NHWC -> NCHW
        kt = torch.as_tensor(k).permute(3, 2, 0, 1)                              # [kh,kw,cin,cout] -> [cout,cin,kh,kw]
        want = torch.relu(F.conv2d(xt, kt, torch.as_tensor(b), padding="same")).permute(0, 2, 3, 1).numpy()
        assert got.shape == (n, h, w, cout) and np.abs(got - want).max() < 1e-5, (h, w, kh)
        if kh > 1:      # a flipped kernel (true convolution) is a different function on these random weights
            flipped = torch.relu(F.conv2d(xt, torch.flip(kt, (2, 3)), torch.as_tensor(b), padding="same")).permute(0, 2, 3, 1).numpy()
            assert np.abs(got - flipped).max() > 0.1


def test_softmax_relu_concat_matmul_add_n_reshape():
    rng = np.random.default_rng(3)
    tf = _shim({})
    x = (rng.standard_normal((19, 5)) * 8).astype(np.float32)                    # large logits: the max-shift matters
    got = np.asarray(tf.nn.softmax(x))
    assert np.abs(got - torch.softmax(torch.as_tensor(x), dim=-1).numpy()).max() < 1e-6 and np.abs(got.sum(-1) - 1).max() < 1e-6
    x3 = rng.standard_normal((4, 3, 5)).astype(np.float32)
    assert np.abs(np.asarray(tf.nn.softmax(x3)) - torch.softmax(torch.as_tensor(x3), dim=-1).numpy()).max() < 1e-6
    assert np.array_equal(np.asarray(tf.nn.relu(x)), torch.relu(torch.as_tensor(x)).numpy())
    a, b, c = (rng.standard_normal((7, k)).astype(np.float32) for k in (2, 3, 4))
    assert np.array_equal(np.asarray(tf.concat([a, b, c], axis=1)), torch.cat([torch.as_tensor(v) for v in (a, b, c)], 1).numpy())
    m = rng.standard_normal((3, 6)).astype(np.float32)
    assert np.abs(np.asarray(tf.matmul(b, m)) - (torch.as_tensor(b) @ torch.as_tensor(m)).numpy()).max() < 2e-6
    s1, s2, s3 = (rng.standard_normal((7, 6)).astype(np.float32) for _ in range(3))
    assert np.array_equal(np.asarray(tf.add_n([s1, s2, s3])), ((torch.as_tensor(s1) + torch.as_tensor(s2)) + torch.as_tensor(s3)).numpy())
    assert np.array_equal(np.asarray(tf.reshape(x3, [-1, 15])), torch.as_tensor(x3).reshape(-1, 15).numpy())   # C order, like tf.reshape


def _torch_actor_particle(w, obs_others, v_obs, v_goal, stage):
    """networks.actor_particle (networks.py:517-538) written with torch primitives only."""
    T = lambda k: torch.as_tensor(w[k])  # noqa: E731
    self_in = torch.cat([torch.as_tensor(v_obs, dtype=torch.float32), torch.as_tensor(v_goal, dtype=torch.float32)], 1)
    acc = torch.relu(F.linear(self_in, T("actor_branch_self/kernel").t(), T("actor_branch_self/bias"))) @ T("W_branch_self_h2")
    if stage > 1:
        ho = torch.relu(F.linear(torch.as_tensor(obs_others, dtype=torch.float32), T("stage-2/actor_others/kernel").t(),
                                 T("stage-2/actor_others/bias")))
        acc = acc + ho @ T("stage-2/W_others_h2")
    h2 = torch.relu(acc + T("b"))
    return torch.softmax(F.linear(h2, T("actor_out/kernel").t(), T("actor_out/bias")), dim=1).numpy()


def test_particle_actor_oracle_matches_an_independent_torch_graph():
    """oracle/actor_oracle.py (what the device actor is compared with) == the same graph built from torch primitives: input
    order [v_obs | v_goal] for the self branch, add_n of the two second-layer products, bias, relu, 5-way softmax."""
    rng = np.random.default_rng(4)
    for n_agents, stage in ((4, 2), (1, 1), (8, 2), (2, 2)):
        w = AO.init_weights(rng, n_agents, stage=stage)
        rows = 301
        oo = rng.uniform(-2, 2, (rows, 4 * max(n_agents - 1, 1))).astype(np.float32)
        vo = rng.uniform(-1.5, 1.5, (rows, 4)).astype(np.float32)
        vg = rng.uniform(-1, 1, (rows, 2)).astype(np.float32)
        got = AO.actor_probs(w, oo, vo, vg)
        want = _torch_actor_particle(w, oo, vo, vg, stage)
        assert got.shape == (rows, 5) and np.abs(got - want).max() < 5e-6, (n_agents, stage)
        if stage > 1:      # the others branch is really wired in
            assert np.abs(got - AO.actor_probs(w, oo + 0.5, vo, vg)).max() > 1e-3
