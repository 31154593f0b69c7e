"""Batched evaluation.

Reference: evaluate.test_particle (alg/evaluate.py:87-123) and evaluate.test_checkers (alg/evaluate.py:159-203), called
every `period` episodes from train_onpolicy.py:394,396 -- N_eval greedy-policy (epsilon = 0) episodes, per-agent and global returns accumulated
per episode (:117-118) and averaged over the episodes (:123).  Here every env of a VecParticleEnv is one
evaluation episode: reset all, run max_steps ticks with the on-device actor inside one hipGraph, stop counting an
env after its `done`, average over the episodes.  Nothing leaves the GPU until the two averages are read.
"""
import torch

from .rollout import CheckersRollout, ParticleRollout


def test_particle(env, actor, n_rounds=1, epsilon=0.0, rollout=None, reset=True):
    """-> (reward_local_avg [N], reward_global_avg, n_episodes) like evaluate.test_particle, over
    n_rounds x env.n_envs episodes.  `env` must not auto-reset (one episode per env per round).  `actor` is the on-device
    ParticleActor or a host callable policy(obs_others, obs_self, goals) -> [E,N]; reset=False evaluates from the states
    the env currently holds (state injection: how tests replay the episodes the reference evaluator saw)."""
    if env.auto_reset:
        raise ValueError("evaluation runs one episode per env: build the env with auto_reset=False")
    ro = rollout or ParticleRollout(env, use_graph=True)
    local_total = torch.zeros(env.n, dtype=torch.float64, device=env.device)
    global_total = torch.zeros((), dtype=torch.float64, device=env.device)
    for _ in range(int(n_rounds)):
        ro.collect(policy=actor, epsilon=epsilon, reset=reset)
        g, l = ro.episode_returns()
        local_total += l.to(torch.float64).sum(0)
        global_total += g.to(torch.float64).sum()
    n = float(n_rounds * env.E)
    if rollout is None:
        ro.close()
    return (local_total / n).cpu().numpy(), float(global_total / n), int(n)


def test_checkers(env, actor, n_rounds=1, epsilon=0.0, rollout=None, generator=None, goals=None):
    """-> (reward_local_avg [N], reward_global_avg, n_episodes, dist_action [N,5]) like evaluate.test_checkers
    (alg/evaluate.py:159-203) over n_rounds x env.n_envs episodes: goals = eye(N), or one random one-hot goal per episode
    when N == 1 (:167-173); actions_prev starts at zeros (:178); dist_action is the normalised action histogram the
    reference prints (:163,:183-184,:200-201).  goals (optional, one-hot [E,N,2]) replaces the draw (tests replay the goals
    the reference evaluator drew)."""
    if env.auto_reset:
        raise ValueError("evaluation runs one episode per env: build the env with auto_reset=False")
    ro = rollout or CheckersRollout(env, use_graph=True)
    N, dev = env.n, env.device
    local_total = torch.zeros(N, dtype=torch.float64, device=dev)
    global_total = torch.zeros((), dtype=torch.float64, device=dev)
    dist = torch.zeros(N, 5, dtype=torch.float64, device=dev)
    for _ in range(int(n_rounds)):
        if goals is not None:
            g_round = torch.as_tensor(goals, device=dev)
        elif N == 1:
            idx = torch.randint(0, 2, (env.E, 1), device=dev, generator=generator)
            g_round = torch.nn.functional.one_hot(idx, 2)
        else:
            g_round = torch.eye(N, 2, device=dev)
        ro.collect(g_round, policy=actor, epsilon=epsilon)
        g, l = ro.episode_returns()
        local_total += l.sum(0)
        global_total += g.sum()
        valid = ro.valid                                                    # [T, E]
        onehot = torch.nn.functional.one_hot(ro.actions.long(), 5).to(torch.float64)   # [T, E, N, 5]
        dist += (onehot * valid[:, :, None, None]).sum((0, 1))
    n = float(n_rounds * env.E)
    if rollout is None:
        ro.close()
    return (local_total / n).cpu().numpy(), float(global_total / n), int(n), (dist / dist.sum()).cpu().numpy()
