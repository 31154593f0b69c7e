"""Batch reformatting on the device.

The reference reshapes every sampled batch on the host with NumPy before each of its 6-8 sess.run calls:
  alg_credit.Alg.process_actions       alg/alg_credit.py:406-443
  alg_credit.Alg.process_batch         alg/alg_credit.py:445-499
  alg_credit.Alg.process_goals         alg/alg_credit.py:501-526
  alg_credit.Alg.process_global_state  alg/alg_credit.py:528-557
  the n x n "credit" repeats of train_step   alg/alg_credit.py:614-658
Here the same arrays are produced from the device trajectory columns (ParticleRollout.as_reference_batch(...,
numpy=False)) with gathers / views, so a learner living on the GPU never pulls the batch through the host.
Pure data movement: outputs are bit-identical to the reference's (tests/test_batch.py pins them to arrays produced
by the REAL reference functions, tests/golden/batch_particle.npz).
"""
import torch


def others_index(n_agents, device=None):
    """[N, N-1] long: row n lists the other agents in ascending order (np.arange(N) != n)."""
    idx = [[j for j in range(n_agents) if j != n] for n in range(n_agents)]
    return torch.tensor(idx, dtype=torch.long, device=device).reshape(n_agents, max(n_agents - 1, 0))


def gather_others(x):
    """x [B, N, d] -> [B*N, N-1, d]: row b*N+n holds x[b, m] for the agents m != n in ascending order
    (the reference's `out[n::N] = x[:, np.arange(N) != n]` interleave)."""
    B, N = x.shape[0], x.shape[1]
    idx = others_index(N, x.device)
    return x[:, idx].reshape(B * N, N - 1, *x.shape[2:])


def process_actions(actions, l_action=5):
    """actions int [B, N] -> (actions_1hot int64 [B*N, A], actions_others_1hot float64 [B*N, N-1, A])
    (alg_credit.py:406-443; dtypes as the reference: np.zeros(dtype=int) and np.zeros default float64)."""
    B, N = actions.shape
    one = torch.nn.functional.one_hot(actions.long(), l_action)            # [B, N, A] int64
    others = gather_others(one).to(torch.float64)
    return one.reshape(B * N, l_action), others


def process_goals(goals):
    """goals [B, N, l_goal] -> (goals_self [B*N, l_goal], goals_others [B*N, (N-1)*l_goal]) (alg_credit.py:501-526)."""
    B, N, lg = goals.shape
    return goals.reshape(B * N, lg), gather_others(goals).reshape(B * N, (N - 1) * lg).to(torch.float64)


def process_global_state(v_global):
    """v_global [B, N, l] -> (one_agent [B*N, l], others [B*N, (N-1)*l], state [B*N, N*l]) (alg_credit.py:528-557)."""
    B, N, l = v_global.shape
    one = v_global.reshape(B * N, l)
    others = gather_others(v_global).reshape(B * N, (N - 1) * l).to(torch.float64)
    state = v_global.reshape(B, N * l).repeat_interleave(N, dim=0)
    return one, others, state


def process_batch(cols, l_action=5):
    """Columns of ParticleRollout.as_reference_batch(numpy=False) -> the 13-tuple of alg_credit.process_batch
    (alg_credit.py:445-499): per-agent rows, global quantities repeated N times."""
    v_global = cols["v_global"]
    B, N = v_global.shape[0], v_global.shape[1]
    a1, ao = process_actions(cols["actions"], l_action)
    return (B, v_global, cols["obs_others"].reshape(B * N, -1), cols["v_local"].reshape(B * N, -1), a1, ao,
            cols["reward"].repeat_interleave(N, dim=0), cols["reward_local"].reshape(B * N),
            cols["v_global_next"], cols["obs_others_next"].reshape(B * N, -1), cols["v_local_next"].reshape(B * N, -1),
            cols["done"].repeat_interleave(N, dim=0), cols["goals"])


def repeat_indexed_by_n(x, n_agents):
    """[B*N, d] -> [B*N*N, d]: each time step's block of N rows repeated N times (alg_credit.py:621-623, :648-649)."""
    d = x.shape[1:]
    return x.reshape(-1, n_agents, *d).repeat_interleave(n_agents, dim=0).reshape(-1, *d)


def repeat_indexed_by_m(x, n_agents):
    """[B*N, d] -> [B*N*N, d]: every row repeated N times consecutively (alg_credit.py:628-629, :651-652)."""
    return x.repeat_interleave(n_agents, dim=0)


# ---- Checkers (alg_credit_checkers.py:375-535) ---------------------------------------------------------------------------

def process_batch_checkers(cols, l_action=5):
    """Columns of CheckersRollout.as_reference_batch(numpy=False) -> the 18-tuple of
    alg_credit_checkers.Alg.process_batch (alg_credit_checkers.py:414-482): per-agent rows; grid and done repeated N
    times, `reward` left per time step (unlike the particle variant), both action columns one-hot."""
    vec = cols["vec"]
    B, N = vec.shape[0], vec.shape[1]
    rep = lambda x: x.repeat_interleave(N, dim=0)              # noqa: E731
    rows = lambda x: x.reshape(B * N, *x.shape[2:])            # noqa: E731
    a1, ao = process_actions(cols["actions"], l_action)
    prev1 = torch.nn.functional.one_hot(cols["actions_prev"].long(), l_action).reshape(B * N, l_action)
    return (B, rep(cols["grid"]), vec, rows(cols["obs_others"]), rows(cols["obs_self_t"]), rows(cols["obs_self_v"]),
            prev1, a1, ao, cols["reward"], cols["local_rewards"].reshape(B * N), rep(cols["next_grid"]),
            cols["next_vec"], rows(cols["next_obs_others"]), rows(cols["next_obs_self_t"]),
            rows(cols["next_obs_self_v"]), rep(cols["done"]), cols["goals"])


CHECKERS_BATCH_NAMES = ("n_steps", "state_env", "state_agents", "obs_others", "obs_self_t", "obs_self_v",
                        "actions_prev_1hot", "actions_1hot", "actions_others_1hot", "reward", "reward_local",
                        "state_env_next", "state_agents_next", "obs_others_next", "obs_self_t_next", "obs_self_v_next",
                        "done", "goals")
# process_goals / process_global_state of alg_credit_checkers.py:484-535 are the particle functions above, verbatim
# in behaviour (l_goal = 2, l_state_one_agent = 4): use process_goals(goals) and process_global_state(state_agents).
