"""Batched evaluation.

Reference: evaluate.test_particle (alg/evaluate.py:87-123), called every `period` episodes from
train_onpolicy.py:394 -- N_eval greedy-policy (epsilon = 0) episodes, per-agent and global returns accumulated
per episode (:117-118) and averaged over the episodes (:123).  Here every env of a VecParticleEnv is one
evaluation episode: reset all, run max_steps ticks with the on-device actor inside one hipGraph, stop counting an
env after its `done`, average over the episodes.  Nothing leaves the GPU until the two averages are read.
"""
import torch

from .rollout import ParticleRollout


def test_particle(env, actor, n_rounds=1, epsilon=0.0, rollout=None):
    """-> (reward_local_avg [N], reward_global_avg, n_episodes) like evaluate.test_particle, over
    n_rounds x env.n_envs episodes.  `env` must not auto-reset (one episode per env per round)."""
    if env.auto_reset:
        raise ValueError("evaluation runs one episode per env: build the env with auto_reset=False")
    ro = rollout or ParticleRollout(env, use_graph=True)
    local_total = torch.zeros(env.n, dtype=torch.float64, device=env.device)
    global_total = torch.zeros((), dtype=torch.float64, device=env.device)
    for _ in range(int(n_rounds)):
        ro.collect(policy=actor, epsilon=epsilon, reset=True)
        g, l = ro.episode_returns()
        local_total += l.to(torch.float64).sum(0)
        global_total += g.to(torch.float64).sum()
    n = float(n_rounds * env.E)
    if rollout is None:
        ro.close()
    return (local_total / n).cpu().numpy(), float(global_total / n), int(n)
