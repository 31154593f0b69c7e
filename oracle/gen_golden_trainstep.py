"""TEST INFRASTRUCTURE ONLY -- records every feed_dict the REAL reference train_step builds (build container only).

alg_credit.Alg.train_step (alg/alg_credit.py:558-800) and alg_credit_checkers.Alg.train_step
(alg/alg_credit_checkers.py:536-780) are NumPy data movement around 6-10 sess.run calls: the n x n "credit" repeats
(:614-658), the TD targets, and the n x n x l_action counterfactual tiling (:730-751).  TensorFlow is not installable
here, so the reference modules are imported with a permissive stub `tensorflow` and train_step is driven with a
RECORDING stand-in for the session: placeholders / ops are their own attribute names, every sess.run returns
deterministic pseudo-random arrays of the shape the real network would return, and each call's (ops, feed_dict, result)
is stored.  The fixtures pin cm3_amd.batch.train_step_feeds bit for bit (tests/test_batch.py).

    python oracle/gen_golden_trainstep.py     ->  tests/golden/trainstep_{particle_n4,particle_n1,checkers_n2,checkers_n1}.npz
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"

PLACEHOLDERS = ("v_state_one_agent", "v_goal", "action_one", "v_state_other_agents", "action_others", "v_state_m",
                "obs_others", "v_obs", "epsilon", "action_taken", "Q_actual", "probs_evaluated", "Q_cf", "V_evaluated",
                "Q_global_td_target", "Q_credit_td_target", "V_td_target", "state_env", "obs_self_t", "obs_self_v",
                "actions_prev", "v_goal_others")
OPS = ("action_samples_target", "Q_global_target", "Q_global_op", "Q_global", "Q_credit_target", "Q_credit_op", "Q_credit",
       "V_target", "V_op", "V", "probs", "policy_op", "list_update_target_ops")


class RecordingSession(object):
    """sess.run(ops, feed_dict) stand-in: returns arrays shaped like the networks' outputs, records the call."""

    def __init__(self, l_action, seed=0):
        self.calls = []
        self.rng = np.random.default_rng(seed)
        self.l_action = l_action

    def _one(self, op, rows):
        if op in ("Q_global_op", "Q_credit_op", "V_op", "policy_op", "list_update_target_ops"):
            return None
        if op == "action_samples_target":
            return self.rng.integers(0, self.l_action, (rows, 1))
        if op == "probs":
            p = self.rng.random((rows, self.l_action)) + 0.05
            return (p / p.sum(1, keepdims=True)).astype(np.float32)
        return self.rng.standard_normal((rows, 1)).astype(np.float32)        # Q / V heads: [rows, 1] float32

    def run(self, ops, feed_dict=None):
        feed = dict(feed_dict or {})
        rows = 0
        for v in feed.values():
            if isinstance(v, np.ndarray) and v.ndim >= 1:
                rows = v.shape[0]
                break
        many = isinstance(ops, (list, tuple))
        names = list(ops) if many else [ops]
        res = [self._one(op, rows) for op in names]
        self.calls.append((names, feed, res))
        return res if many else res[0]


def make_alg(mod, n_agents, dims):
    alg = mod.Alg.__new__(mod.Alg)
    for name in PLACEHOLDERS + OPS:
        setattr(alg, name, name)
    alg.n_agents, alg.l_action, alg.gamma = n_agents, 5, 0.99
    alg.use_Q_credit, alg.use_V = True, True
    alg.actions = np.eye(5)                                            # alg_credit.py:70 / alg_credit_checkers.py:65
    for k, v in dims.items():
        setattr(alg, k, v)
    return alg


def save(path, cols, sess, extra):
    rec = {"in_" + k: np.asarray(v) for k, v in cols.items()}
    index = []
    for c, (names, feed, res) in enumerate(sess.calls):
        entry = {"ops": names, "feed": sorted(feed), "results": []}
        for k, v in feed.items():
            rec["c%d_feed_%s" % (c, k)] = np.asarray(v)
        for name, r in zip(names, res):
            if r is not None:
                rec["c%d_res_%s" % (c, name)] = r
                entry["results"].append(name)
        index.append(entry)
    rec["index"] = np.array(json.dumps({"calls": index, **extra}))
    np.savez_compressed(path, **rec)
    print("wrote", os.path.basename(path), "calls:", [(e["ops"], len(e["feed"])) for e in index])


def particle_cols(name, ep, N):
    z = np.load(os.path.join(ROOT, "tests", "golden", name))
    T = min(int(z["ep_len"][ep]), 12)
    gs = np.concatenate([z["init_gs"][ep][None], z["gs"][ep, :T]])
    oo = np.concatenate([z["init_obs_others"][ep][None], z["obs_others"][ep, :T]])
    return dict(v_global=gs[:-1], obs_others=oo[:-1], v_local=gs[:-1], actions=z["actions"][ep, :T],
                reward=z["reward"][ep, :T], reward_local=z["reward_n"][ep, :T], v_global_next=gs[1:],
                obs_others_next=oo[1:], v_local_next=gs[1:], done=z["done"][ep, :T],
                goals=np.repeat(z["landmarks"][ep][None], T, axis=0))


def checkers_cols(name, ep, N):
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    T = min(int(g["ep_len"][ep]), 10)

    def seq(init, per_tick):
        return np.concatenate([g[init][ep][None], g[per_tick][ep, :T]])
    grid, vec = seq("init_grid", "grid"), seq("init_vec", "vec")
    oo, ot, ov = seq("init_obs_others", "obs_others"), seq("init_obs_self_t", "obs_self_t"), seq("init_obs_self_v", "obs_self_v")
    acts = g["actions"][ep, :T]
    prev = np.concatenate([np.zeros((1, N), acts.dtype), acts[:-1]])
    return dict(grid=grid[:-1], vec=vec[:-1], obs_others=oo[:-1], obs_self_t=ot[:-1], obs_self_v=ov[:-1],
                actions_prev=prev, actions=acts, reward=g["reward"][ep, :T], local_rewards=g["local_rewards"][ep, :T],
                next_grid=grid[1:], next_vec=vec[1:], next_obs_others=oo[1:], next_obs_self_t=ot[1:],
                next_obs_self_v=ov[1:], done=g["done"][ep, :T],
                goals=np.repeat(g["goals"][ep][None], T, axis=0).astype(float)), json.loads(str(g["meta"]))


def main():
    from cm3_amd.rollout import CHECKERS_ORDER, PARTICLE_ORDER, rows_from_columns    # torch BEFORE the tensorflow stub

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
    import alg_credit_checkers
    out = os.path.join(ROOT, "tests", "golden")
    for tag, fixture, N in (("particle_n4", "particle_cross_greedy.npz", 4), ("particle_n1", "particle_stage1_greedy.npz", 1)):
        cols = particle_cols(fixture, 1, N)
        Lo = 4 * max(N - 1, 1)
        alg = make_alg(alg_credit, N, dict(experiment="particle", l_obs_others=Lo, l_obs=4, l_goal=2, l_state_one_agent=4,
                                           l_state=4 * N, l_state_other_agents=4 * (N - 1)))
        sess = RecordingSession(5, seed=N)
        alg.train_step(sess, rows_from_columns(cols, PARTICLE_ORDER), 0.25, 7, summarize=False, writer=None)
        save(os.path.join(out, "trainstep_%s.npz" % tag), cols, sess, dict(n_agents=N, gamma=0.99, epsilon=0.25, env="particle"))
    for tag, fixture, N in (("checkers_n2", "checkers_stage2_uniform.npz", 2), ("checkers_n1", "checkers_stage1_uniform.npz", 1)):
        cols, meta = checkers_cols(fixture, 1, N)
        d = meta["config"]["dimensions"]
        alg = make_alg(alg_credit_checkers, N, dict(
            experiment="checkers", l_obs_others=d["l_obs_others"], l_obs_self=d["l_obs_self"], l_goal=d["l_goal"],
            rows_obs=d["rows_obs"], columns_obs=d["columns_obs"], channels_obs=d["channels_obs"],
            l_state_one_agent=d["l_state_one"], l_state=N * d["l_state_one"], l_state_other_agents=(N - 1) * d["l_state_one"]))
        sess = RecordingSession(5, seed=10 + N)
        alg.train_step(sess, rows_from_columns({k: np.array(v) for k, v in cols.items()}, CHECKERS_ORDER), 0.25, 7,
                       summarize=False, writer=None)
        save(os.path.join(out, "trainstep_%s.npz" % tag), cols, sess, dict(n_agents=N, gamma=0.99, epsilon=0.25, env="checkers"))


if __name__ == "__main__":
    main()
