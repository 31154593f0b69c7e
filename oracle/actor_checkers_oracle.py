"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's Checkers actor and its action sampling.

Restates (float32, like the TF1 graph):
  networks.convnet_1                 /root/reference/alg/networks.py:67-75
      tf.contrib.layers.conv2d(num_outputs=f1, kernel_size=[3,3], stride=1, padding="SAME", relu) on NHWC
      [rows, 5, 5, 3]; flattened row-major over (row, column, filter)
  networks.actor_checkers            /root/reference/alg/networks.py:549-578
      conv -> dense 32 relu ("conv_linear") -> concat(conv_linear, v_obs_self[4], a_prev[5], v_goal[2])
      -> dense n_h1 relu ("branch_self") -> x W_self_h2
      stage > 1: v_obs_others -> dense n_h1 relu ("stage-2/branch_others") -> x "stage-2/W_others_h2"
      h2 = relu(sum + b) -> dense 5 ("actor_out") -> softmax
  alg_credit_checkers.Alg            /root/reference/alg/alg_credit_checkers.py:107-113
      probs = (1 - eps) * probs + eps / l_action ; action ~ multinomial(log probs)
  alg_credit_checkers.Alg.run_actor  /root/reference/alg/alg_credit_checkers.py:229-253 (actions_prev -> one-hot)
Widths from config_checkers_stage{1,2}.json "nn": A_conv_f 6, A_conv_k [3,3], A_n_h1 256, A_n_h2 256.

PARITY UNPINNED against TensorFlow itself (TF1 is not installable in the build container).  What IS pinned: the layer
wiring, flatten / concat order, variable names and shapes -- tests/golden/actor_checkers.npz holds vectors produced by executing
the reference's OWN function bodies networks.actor_checkers + convnet_1 under a NumPy stand-in for the TF1 calls they make
(oracle/tf_numpy_shim.py, tests/test_oracle_actor_golden.py); the primitive-op semantics (NHWC "SAME" cross-correlation, dense,
relu, softmax) rest on their published behaviour, and tests/test_oracle_actor_checkers.py cross-checks them against an
independent PyTorch float32 implementation.  Sampling is distributional, exactly as in oracle/actor_oracle.py (one Philox
uniform per agent-step, inverse CDF in action order).
"""
import numpy as np

from oracle.actor_oracle import mixed_probs, policy_uniforms, sample_actions  # noqa: F401  (same sampling rule)

CONV_F, CONV_LIN, H1, H2, N_ACTIONS = 6, 32, 256, 256, 5
K_OBS, C_OBS = 5, 3


def init_weights(rng, n_agents, stage=2, scale=None):
    """Random weights under the reference's variable names / shapes.  (The reference initialises with xavier /
    glorot-uniform / truncated_normal(0, 0.01); tests use larger scales so that the policy is not uniform.)"""
    lo = 2 * max(n_agents - 1, 1)
    s = 1.0 if scale is None else scale

    def f(*shape, fan=None):
        fan = fan or shape[0]
        return (rng.standard_normal(shape) * s / np.sqrt(fan)).astype(np.float32)
    w = {"conv/Conv/weights": f(3, 3, C_OBS, CONV_F, fan=27), "conv/Conv/biases": f(CONV_F, fan=4),
         "conv_linear/kernel": f(K_OBS * K_OBS * CONV_F, CONV_LIN), "conv_linear/bias": f(CONV_LIN, fan=4),
         "branch_self/kernel": f(CONV_LIN + 4 + N_ACTIONS + 2, H1), "branch_self/bias": f(H1, fan=4),
         "W_self_h2": f(H1, H2), "b": f(H2, fan=4),
         "actor_out/kernel": f(H2, N_ACTIONS, fan=16), "actor_out/bias": f(N_ACTIONS, fan=4)}
    if stage > 1:
        w["stage-2/branch_others/kernel"] = f(lo, H1)
        w["stage-2/branch_others/bias"] = f(H1, fan=4)
        w["stage-2/W_others_h2"] = f(H1, H2)
    return w


def conv_same_3x3(x, w, b):
    """x [rows, 5, 5, 3] float32 NHWC, w [3, 3, 3, F] (kh, kw, cin, cout), stride 1, zero "SAME" padding,
    cross-correlation (TF does not flip the kernel): out[r, c] = sum_{dr, dc} x[r + dr - 1, c + dc - 1] . w[dr, dc]."""
    rows, H, W, _ = x.shape
    xp = np.zeros((rows, H + 2, W + 2, x.shape[3]), np.float32)
    xp[:, 1:H + 1, 1:W + 1] = x
    out = np.zeros((rows, H, W, w.shape[3]), np.float32)
    for dr in range(3):
        for dc in range(3):
            out += xp[:, dr:dr + H, dc:dc + W, :] @ w[dr, dc]
    return out + b


def actor_probs(w, a_prev, obs_self_t, obs_self_v, obs_others, goals_onehot):
    """Rows = agents of a batch.  a_prev int [rows]; obs_self_t [rows,5,5,3]; obs_self_v [rows,4]; obs_others
    [rows, 2(N-1)]; goals_onehot [rows,2].  Returns softmax probabilities [rows, 5] in float32."""
    f32 = np.float32
    rows = obs_self_t.shape[0]
    relu = lambda v: np.maximum(v, f32(0))  # noqa: E731
    conv = relu(conv_same_3x3(obs_self_t.astype(f32), w["conv/Conv/weights"], w["conv/Conv/biases"]))
    lin = relu(conv.reshape(rows, -1) @ w["conv_linear/kernel"] + w["conv_linear/bias"])
    a1 = np.zeros((rows, N_ACTIONS), f32)
    a1[np.arange(rows), np.asarray(a_prev).reshape(-1)] = 1
    x = np.concatenate([lin, obs_self_v.astype(f32), a1, goals_onehot.astype(f32)], axis=1)
    h_self = relu(x @ w["branch_self/kernel"] + w["branch_self/bias"])
    acc = h_self @ w["W_self_h2"]
    if "stage-2/branch_others/kernel" in w:
        h_oth = relu(obs_others.astype(f32) @ w["stage-2/branch_others/kernel"] + w["stage-2/branch_others/bias"])
        acc = acc + h_oth @ w["stage-2/W_others_h2"]
    h2 = relu(acc + w["b"])
    out = h2 @ w["actor_out/kernel"] + w["actor_out/bias"]
    out = out - out.max(axis=1, keepdims=True)
    e = np.exp(out)
    return (e / e.sum(axis=1, keepdims=True)).astype(f32)
