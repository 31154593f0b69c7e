"""TEST INFRASTRUCTURE ONLY -- records what the REAL reference batch reshapers return (build container only).

Imports alg/alg_credit.py from /root/reference with a permissive stub `tensorflow` (TF1 is not installed; only the
pure-NumPy methods run), feeds it one recorded episode and stores inputs + outputs in
tests/golden/batch_particle.npz.  Covers process_batch / process_actions / process_goals / process_global_state
(alg_credit.py:406-557) and the n x n repeats of train_step (alg_credit.py:621-652).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"


def main():
    from cm3_amd.rollout import PARTICLE_ORDER, rows_from_columns      # imports torch BEFORE the tensorflow stub exists

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            return _Any(k)

        def __call__(self, *a, **k):
            return _Any("call")
    sys.modules.setdefault("tensorflow", _Any("tensorflow"))
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "alg"))
    if not hasattr(np, "int"):
        np.int = int
    import alg_credit
    z = np.load(os.path.join(ROOT, "tests", "golden", "particle_cross_greedy.npz"))
    N, ep = 4, 2
    T = int(z["ep_len"][ep])
    gs = np.concatenate([z["init_gs"][ep][None], z["gs"][ep, :T]])
    oo = np.concatenate([z["init_obs_others"][ep][None], z["obs_others"][ep, :T]])
    cols = dict(v_global=gs[:-1], obs_others=oo[:-1], v_local=gs[:-1], actions=z["actions"][ep, :T],
                reward=z["reward"][ep, :T], reward_local=z["reward_n"][ep, :T], v_global_next=gs[1:],
                obs_others_next=oo[1:], v_local_next=gs[1:], done=z["done"][ep, :T],
                goals=np.repeat(z["landmarks"][ep][None], T, axis=0))
    alg = alg_credit.Alg.__new__(alg_credit.Alg)
    alg.n_agents, alg.l_action, alg.experiment = N, 5, "particle"
    alg.l_obs_others, alg.l_obs, alg.l_goal = 12, 4, 2
    alg.l_state_one_agent, alg.l_state, alg.l_state_other_agents = 4, 16, 12
    out = alg.process_batch(rows_from_columns(cols, PARTICLE_ORDER))
    names = ("n_steps", "v_global", "obs_others", "v_local", "actions_1hot", "actions_others_1hot", "reward",
             "reward_local", "v_global_next", "obs_others_next", "v_local_next", "done", "goals")
    rec = {"in_" + k: v for k, v in cols.items()}
    for k, v in zip(names, out):
        rec["pb_" + k] = np.asarray(v)
    n_steps = out[0]
    gself, gothers = alg.process_goals(out[12], n_steps)
    one, others, state = alg.process_global_state(out[1], n_steps)
    rec.update(goals_self=gself, goals_others=gothers, vg_one=one, vg_others=others, vg_state=state)
    # n x n repeats (alg_credit.py:621-652)
    n = N
    rec["s_n_rep"] = np.reshape(np.repeat(np.reshape(one, [-1, n * 4]), n, axis=0), [-1, 4])
    rec["s_others_rep"] = np.reshape(np.repeat(np.reshape(others, [-1, n * 12]), n, axis=0), [-1, 12])
    rec["actions_self_rep"] = np.repeat(out[4], n, axis=0)
    rec["s_m_rep"] = np.repeat(one, n, axis=0)
    rec["reward_local_rep"] = np.reshape(np.repeat(np.reshape(out[7], [-1, n]), n, axis=0), [-1])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "batch_particle.npz"), **rec)
    print("wrote batch_particle.npz:", {k: v.shape for k, v in rec.items() if hasattr(v, "shape")})
    checkers()


def checkers():
    """alg_credit_checkers.Alg.process_batch / process_goals / process_global_state (alg_credit_checkers.py:414-535) on one
    recorded Checkers episode -> tests/golden/batch_checkers.npz."""
    from cm3_amd.rollout import CHECKERS_ORDER, rows_from_columns
    import json
    import alg_credit_checkers as mod
    g = np.load(os.path.join(ROOT, "tests", "golden", "checkers_stage2_uniform.npz"))
    d = json.loads(str(g["meta"]))["config"]["dimensions"]
    ep, N = 1, 2
    T = int(g["ep_len"][ep])

    def seq(init, per_tick):
        return np.concatenate([g[init][ep][None], g[per_tick][ep, :T]])
    grid, vec = seq("init_grid", "grid"), seq("init_vec", "vec")
    oo, ot, ov = seq("init_obs_others", "obs_others"), seq("init_obs_self_t", "obs_self_t"), seq("init_obs_self_v", "obs_self_v")
    acts = g["actions"][ep, :T]
    prev = np.concatenate([np.zeros((1, N), acts.dtype), acts[:-1]])
    cols = dict(grid=grid[:-1], vec=vec[:-1], obs_others=oo[:-1], obs_self_t=ot[:-1], obs_self_v=ov[:-1],
                actions_prev=prev, actions=acts, reward=g["reward"][ep, :T], local_rewards=g["local_rewards"][ep, :T],
                next_grid=grid[1:], next_vec=vec[1:], next_obs_others=oo[1:], next_obs_self_t=ot[1:],
                next_obs_self_v=ov[1:], done=g["done"][ep, :T],
                goals=np.repeat(g["goals"][ep][None], T, axis=0).astype(float))
    alg = mod.Alg.__new__(mod.Alg)
    alg.n_agents, alg.l_action, alg.experiment = N, d["l_action"], "checkers"
    alg.l_obs_others, alg.l_obs_self, alg.l_goal = d["l_obs_others"], d["l_obs_self"], d["l_goal"]
    alg.rows_obs, alg.columns_obs, alg.channels_obs = d["rows_obs"], d["columns_obs"], d["channels_obs"]
    alg.l_state_one_agent, alg.l_state = d["l_state_one"], N * d["l_state_one"]
    out = alg.process_batch(rows_from_columns({k: np.array(v) for k, v in cols.items()}, CHECKERS_ORDER))
    names = ("n_steps", "state_env", "state_agents", "obs_others", "obs_self_t", "obs_self_v", "actions_prev_1hot",
             "actions_1hot", "actions_others_1hot", "reward", "reward_local", "state_env_next", "state_agents_next",
             "obs_others_next", "obs_self_t_next", "obs_self_v_next", "done", "goals")
    rec = {"in_" + k: v for k, v in cols.items()}
    for k, v in zip(names, out):
        rec["pb_" + k] = np.asarray(v)
    gself, gothers = alg.process_goals(out[17], out[0])
    one, others, state = alg.process_global_state(out[2], out[0])
    rec.update(goals_self=gself, goals_others=gothers, vg_one=one, vg_others=others, vg_state=state)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "batch_checkers.npz"), **rec)
    print("wrote batch_checkers.npz:", {k: (v.shape, v.dtype) for k, v in rec.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
