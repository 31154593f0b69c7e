"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's particle actor and its action sampling.

Restates (float32, like the TF1 graph):
  networks.actor_particle            /root/reference/alg/networks.py:517-538
      concat(v_obs, v_goal) -> dense 64 relu ("actor_branch_self") -> x W_branch_self_h2
      stage > 1: obs_others -> dense n_h1_others relu ("stage-2/actor_others") -> x W_others_h2
      h2 = relu(sum + b) -> dense 5 ("actor_out") -> softmax
  alg_credit.Alg.create_networks     /root/reference/alg/alg_credit.py:113-122
      probs = (1 - eps) * probs + eps / l_action ; action ~ multinomial(log probs)
  alg_credit.Alg.run_actor           /root/reference/alg/alg_credit.py:249-270   (batch of all agents)

PARITY UNPINNED against TensorFlow itself: TF1 is not installable in the build container (no network).  What IS pinned
(tests/test_oracle_actor_golden.py, tests/golden/actor_particle.npz): the layer wiring, concat order, variable names and
shapes -- the golden vectors come from executing the reference's OWN function body networks.actor_particle under a NumPy
stand-in for the TF1 calls it makes (oracle/tf_numpy_shim.py); only the semantics of those primitive ops (dense = x @ kernel
+ bias, relu, softmax) rest on their published behaviour.  Sampling is distributional: tf.multinomial's generator cannot be
reproduced, the build draws one uniform per agent-step from its Philox stream (oracle/philox.py) and inverts the CDF in
action order.
"""
import numpy as np

from oracle import philox

PURPOSE_POLICY = 0x40000000
H1_SELF, H2, N_ACTIONS = 64, 64, 5


def init_weights(rng, n_agents, n_h1_others=128, stage=2, scale=None):
    """Random weights with the reference's variable names/shapes (networks.py:517-538).  The reference
    initialises kernels with glorot-uniform (tf.layers.dense default) / truncated_normal(0, 0.01)
    (get_variable, networks.py:78-81); tests use a larger scale so that the policy is not uniform."""
    l_others = 4 * max(n_agents - 1, 1)
    s = 0.5 if scale is None else scale
    f = lambda *shape: (rng.standard_normal(shape) * s).astype(np.float32)  # noqa: E731
    w = {"actor_branch_self/kernel": f(6, H1_SELF), "actor_branch_self/bias": f(H1_SELF),
         "W_branch_self_h2": f(H1_SELF, H2), "b": f(H2),
         "actor_out/kernel": f(H2, N_ACTIONS), "actor_out/bias": f(N_ACTIONS)}
    if stage > 1:
        w["stage-2/actor_others/kernel"] = f(l_others, n_h1_others)
        w["stage-2/actor_others/bias"] = f(n_h1_others)
        w["stage-2/W_others_h2"] = f(n_h1_others, H2)
    return w


def actor_probs(w, obs_others, v_obs, v_goal):
    """Rows = agents of a batch.  Returns softmax probabilities [rows, 5] in float32."""
    f32 = np.float32
    x = np.concatenate([v_obs, v_goal], axis=1).astype(f32)
    h_self = np.maximum(x @ w["actor_branch_self/kernel"] + w["actor_branch_self/bias"], f32(0))
    acc = h_self @ w["W_branch_self_h2"]
    if "stage-2/actor_others/kernel" in w:
        h_oth = np.maximum(obs_others.astype(f32) @ w["stage-2/actor_others/kernel"]
                           + w["stage-2/actor_others/bias"], f32(0))
        acc = acc + h_oth @ w["stage-2/W_others_h2"]
    h2 = np.maximum(acc + w["b"], f32(0))
    out = h2 @ w["actor_out/kernel"] + w["actor_out/bias"]
    out = out - out.max(axis=1, keepdims=True)
    e = np.exp(out)
    return (e / e.sum(axis=1, keepdims=True)).astype(f32)


def mixed_probs(probs, epsilon):
    """alg_credit.py:119: (1 - eps) * probs + eps / l_action."""
    return (np.float32(1) - np.float32(epsilon)) * probs + np.float32(epsilon) / np.float32(N_ACTIONS)


def policy_uniforms(seed, env_ids, episode, step, n_agents):
    """float32 uniforms in (0,1) [E, N] the kernel draws for (env, episode, step): the two-stage stream of csrc/philox.h with
    the policy purpose bit -- the agent's word of the Philox block over (env id, call c = i // 4), mixed with the episode / step
    counters (philox.action_word), (word + 0.5) * 2^-32 rounded to float32."""
    env_ids = np.asarray(env_ids)
    out = np.zeros((env_ids.shape[0], n_agents), np.float64)
    lo, hi = philox._split(env_ids)
    for call in range((n_agents + 3) // 4):
        c3 = np.uint64(PURPOSE_POLICY | (call << 24))
        w = philox.philox4x32_10(lo, hi, np.uint64(0), c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for k in range(4):
            i = 4 * call + k
            if i < n_agents:
                out[:, i] = philox.u01(philox.action_word(w[k], episode, step))
    return out.astype(np.float32)


def sample_actions(probs_eps, u):
    """Inverse CDF in action order with float32 partial sums: the first k with u < p_0 + ... + p_k (4 if none)."""
    rows = probs_eps.shape[0]
    cdf = np.zeros(rows, np.float32)
    act = np.full(rows, N_ACTIONS - 1, np.int64)
    chosen = np.zeros(rows, bool)
    for k in range(N_ACTIONS - 1):
        cdf = cdf + probs_eps[:, k]
        pick = (~chosen) & (u < cdf)
        act[pick] = k
        chosen |= pick
    return act
