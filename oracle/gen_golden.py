"""TEST INFRASTRUCTURE ONLY -- records golden vectors from the REAL reference envs.

Run in the build container (where /root/reference exists):

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

The reference ships no tests or golden vectors, so parity is pinned by these
fixtures: every array below is an input to, or an output of, the reference's own
``MultiAgentEnv.step/reset`` (environment.py:81-149) or ``Checkers.step/reset``
(checkers.py:228-291), executed unmodified through ``oracle/_reference_harness.py``.
Fixtures are data only (inputs + expected outputs); no reference source is stored.

Episode driver mirrors alg/train_onpolicy.py:281-350: seed both global generators
(:38-39), build the env (:117-119 / :126 -- ``make_world`` itself performs one
``reset_world``), then per episode ``reset`` followed by ``step`` until ``done``.
Action policies:
  uniform   ``np.random.randint(0, 5, n_agents)`` on the GLOBAL stream, exactly :307
  greedy    own RandomState: move toward the agent's landmark w.p. 0.8 (forces collisions)
  wild      own RandomState: integers in [-1, 6] (exercises out-of-range actions)
  script    explicit action lists
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import _reference_harness as H  # noqa: E402

CFG_DIR = os.path.join(ROOT, "cm3_amd", "configs")
OUT_DIR = os.path.join(ROOT, "tests", "golden")
MAX_STEPS = 33


def load_cfg(name):
    with open(os.path.join(CFG_DIR, name)) as f:
        return json.load(f)


# ------------------------------------------------------------------------------------------
# particle
# ------------------------------------------------------------------------------------------
def greedy_actions(rs, env, n):
    acts = np.zeros(n, dtype=np.int64)
    for i in range(n):
        if rs.rand() < 0.8:
            d = env.world.landmarks[i].state.p_pos - env.world.agents[i].state.p_pos
            if abs(d[0]) > abs(d[1]):
                acts[i] = 2 if d[0] > 0 else 1
            else:
                acts[i] = 4 if d[1] > 0 else 3
        else:
            acts[i] = rs.randint(0, 5)
    return acts


def record_particle(ns, cfg_name, n_agents, prob_random, seed, n_episodes, policy,
                    scripts=None, inject=None, config_override=None):
    cfg = config_override if config_override is not None else load_cfg(cfg_name)
    np.random.seed(seed)
    random.seed(seed)
    env, scenario, world = H.make_reference_particle_env(ns, n_agents, cfg, prob_random, MAX_STEPS)
    rs = np.random.RandomState(seed + 1000)
    N, T = n_agents, MAX_STEPS
    L = max(N - 1, 1) * 4
    Ep = n_episodes
    out = dict(
        init_gs=np.zeros((Ep, N, 4)), init_obs_others=np.zeros((Ep, N, L)), init_obs_self=np.zeros((Ep, N, 4)),
        init_done=np.zeros(Ep, bool), landmarks=np.zeros((Ep, N, 2)), ep_len=np.zeros(Ep, np.int64),
        actions=np.full((Ep, T, N), -99, np.int64), gs=np.full((Ep, T, N, 4), np.nan),
        obs_others=np.full((Ep, T, N, L), np.nan), obs_self=np.full((Ep, T, N, 4), np.nan),
        reward=np.full((Ep, T), np.nan), reward_n=np.full((Ep, T, N), np.nan),
        done=np.zeros((Ep, T), bool), collisions=np.zeros((Ep, T), np.int64),
        reached=np.zeros((Ep, T, N), bool))
    for ep in range(Ep):
        gs, oo, os_, done = env.reset()
        if inject is not None:
            # state injection after reset (parity of special configurations)
            pos, vel, lm = inject[ep]
            for i in range(N):
                world.agents[i].state.p_pos = np.array(pos[i], dtype=float)
                world.agents[i].state.p_vel = np.array(vel[i], dtype=float)
                world.landmarks[i].state.p_pos = np.array(lm[i], dtype=float)
            gs = np.array([np.concatenate([a.state.p_vel, a.state.p_pos]) for a in world.agents])
            obs = [scenario.observation(a, world) for a in world.agents]
            os_ = [o[0] for o in obs]
            oo = [o[1] for o in obs]
        out['init_gs'][ep] = gs
        out['init_obs_others'][ep] = np.array(oo)
        out['init_obs_self'][ep] = np.array(os_)
        out['init_done'][ep] = done
        out['landmarks'][ep] = np.array([l.state.p_pos for l in world.landmarks])
        t = 0
        done = False
        while not done:
            if policy == 'uniform':
                acts = np.random.randint(0, 5, N)           # train_onpolicy.py:307
            elif policy == 'greedy':
                acts = greedy_actions(rs, env, N)
            elif policy == 'wild':
                acts = rs.randint(-1, 7, N)
            elif policy == 'script':
                acts = np.array(scripts[ep][t])
            else:
                raise ValueError(policy)
            gs, oo, os_, rew, rew_n, done = env.step(acts)
            out['actions'][ep, t] = acts
            out['gs'][ep, t] = gs
            out['obs_others'][ep, t] = np.array(oo)
            out['obs_self'][ep, t] = np.array(os_)
            out['reward'][ep, t] = rew
            out['reward_n'][ep, t] = np.array(rew_n)
            out['done'][ep, t] = done
            out['collisions'][ep, t] = scenario.collisions
            out['reached'][ep, t] = [a.reached for a in world.agents]
            t += 1
            if policy == 'script' and t >= len(scripts[ep]):
                break
        out['ep_len'][ep] = t
    out['meta'] = np.array(json.dumps(dict(
        kind='particle', config=cfg, config_name=cfg_name, n_agents=N, prob_random=prob_random,
        max_steps=MAX_STEPS, seed=seed, policy=policy, injected=inject is not None,
        numpy=np.__version__)))
    return out


# ------------------------------------------------------------------------------------------
# checkers
# ------------------------------------------------------------------------------------------
def record_checkers(ns, cfg_name, seed, n_episodes, policy, scripts=None, goals_list=None):
    cfg = load_cfg(cfg_name)
    N = cfg['n_agents']
    init = cfg['init']
    env = H.make_reference_checkers_env(ns, init, N, MAX_STEPS)
    rs = np.random.RandomState(seed)
    T = MAX_STEPS
    R, C, O = init['n_rows'], init['n_columns'], init['n_obs']
    K = 2 * O + 1
    Lo = 2 * max(N - 1, 1)
    Ep = n_episodes

    def blank(shape, fill=np.nan):
        return np.full(shape, fill)

    out = dict(
        goals=np.zeros((Ep, N, 2), np.int64), ep_len=np.zeros(Ep, np.int64),
        init_grid=blank((Ep, R, C + 1, 2)), init_vec=blank((Ep, N, 4)), init_obs_others=blank((Ep, N, Lo)),
        init_obs_self_t=blank((Ep, N, K, K, 3)), init_obs_self_v=blank((Ep, N, 4)),
        actions=np.full((Ep, T, N), -99, np.int64),
        grid=blank((Ep, T, R, C + 1, 2)), vec=blank((Ep, T, N, 4)), obs_others=blank((Ep, T, N, Lo)),
        obs_self_t=blank((Ep, T, N, K, K, 3)), obs_self_v=blank((Ep, T, N, 4)),
        reward=blank((Ep, T)), local_rewards=blank((Ep, T, N)), done=np.zeros((Ep, T), bool))
    for ep in range(Ep):
        if goals_list is not None:
            goals = np.array(goals_list[ep])
        elif N == 1:
            goals = np.array([[1, 0]]) if rs.randint(2) == 0 else np.array([[0, 1]])   # train_onpolicy.py:288-291
        else:
            goals = np.eye(N)                                                          # :293
        out['goals'][ep] = goals
        gs, oo, ot, ov, done = env.reset(goals)
        out['init_grid'][ep] = np.array(gs[0])          # snapshot by value
        out['init_vec'][ep] = np.array(gs[1])
        out['init_obs_others'][ep] = np.array(oo)
        out['init_obs_self_t'][ep] = np.array(ot)
        out['init_obs_self_v'][ep] = np.array(ov)
        t = 0
        while not done:
            if policy == 'uniform':
                acts = rs.randint(0, 5, N)
            elif policy == 'wild':
                acts = rs.randint(-1, 7, N)
            elif policy == 'script':
                acts = np.array(scripts[ep][t])
            gs, oo, ot, ov, total, local, done = env.step(acts)
            out['actions'][ep, t] = acts
            out['grid'][ep, t] = np.array(gs[0])
            out['vec'][ep, t] = np.array(gs[1])
            out['obs_others'][ep, t] = np.array(oo)
            out['obs_self_t'][ep, t] = np.array(ot)
            out['obs_self_v'][ep, t] = np.array(ov)
            out['reward'][ep, t] = total
            out['local_rewards'][ep, t] = np.array(local, dtype=float)
            out['done'][ep, t] = done
            t += 1
            if policy == 'script' and t >= len(scripts[ep]):
                break
        out['ep_len'][ep] = t
    out['meta'] = np.array(json.dumps(dict(
        kind='checkers', config=cfg, config_name=cfg_name, n_agents=N, max_steps=MAX_STEPS,
        seed=seed, policy=policy, numpy=np.__version__)))
    return out


def checkers_scripts_stage2():
    """Scripted 2-agent episodes: (i) bump into each other, (ii) race for one cell,
    (iii) every wall, (iv) agent 0 sweeps all 24 cells (early done), (v) agent 1 sweeps."""
    S, U, D, L, R = 0, 1, 2, 3, 4
    scripts = []
    # (i) agents start at rows 2 and 4 of column 10: meet on row 3, then push into each other
    scripts.append([[D, S], [D, U], [D, U], [S, U], [D, S], [L, L], [R, R], [U, D]] + [[S, S]] * 3)
    # (ii) both go for cell (3,9): agent 0 moves first and wins; then swap priorities
    scripts.append([[L, L], [D, U], [S, U], [D, S], [U, S], [S, U], [S, D], [D, S], [U, U], [D, D]])
    # (iii) walls: up/right at the top-right start, then along the borders
    scripts.append([[U, D], [R, R], [L, L], [U, D], [U, D]] + [[L, L]] * 9 + [[U, D]] * 2 + [[R, R]] * 2)
    # (iv) agent 0 snake sweep; agent 1 idles in the start column
    sweep = [L] * 8 + [D] + [R] * 7 + [D] + [L] * 7
    scripts.append([[a, S] for a in sweep] + [[S, S]] * 4)
    # (v) agent 1 sweeps upward, agent 0 idles
    sweep1 = [L] * 8 + [U] + [R] * 7 + [U] + [L] * 7
    scripts.append([[S, a] for a in sweep1] + [[S, S]] * 4)
    # (vi) both sweep concurrently toward each other (contention in the middle row)
    both = [[L, L]] * 8 + [[D, U]] + [[R, R]] * 7 + [[D, U]] + [[L, L]] * 7 + [[S, S]] * 6
    scripts.append(both)
    return scripts


def checkers_scripts_stage1():
    S, U, D, L, R = 0, 1, 2, 3, 4
    top = [L] * 8 + [D] + [R] * 7 + [D] + [L] * 7 + [S] * 4      # start row 0 (goal green)
    bot = [L] * 8 + [U] + [R] * 7 + [U] + [L] * 7 + [S] * 4      # start row 2 (goal orange)
    walls = [U, U, R, D, D, D, R, L, L, U, U, U, U] + [S] * 3
    scripts = [[[a] for a in top], [[a] for a in bot], [[a] for a in walls], [[a] for a in walls]]
    goals = [[[1, 0]], [[0, 1]], [[1, 0]], [[0, 1]]]
    return scripts, goals


# ------------------------------------------------------------------------------------------
def main():
    ns = H.load_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else None     # `gen_golden.py particle_ring`: rewrite only the fixtures with that prefix
    fixtures = {}
    # --- particle: the four reference configs + the build-defined 8-agent one --------------
    fixtures['particle_stage1_uniform'] = record_particle(ns, 'particle_stage1.json', 1, 0.2, 12341, 12, 'uniform')
    fixtures['particle_stage1_greedy'] = record_particle(ns, 'particle_stage1.json', 1, 0.5, 12342, 8, 'greedy')
    fixtures['particle_antipodal_uniform'] = record_particle(ns, 'particle_stage2_antipodal.json', 4, 0.2, 12341, 16, 'uniform')
    fixtures['particle_antipodal_greedy'] = record_particle(ns, 'particle_stage2_antipodal.json', 4, 0.2, 12343, 24, 'greedy')
    fixtures['particle_antipodal_random1'] = record_particle(ns, 'particle_stage2_antipodal.json', 4, 1.0, 12344, 12, 'greedy')
    fixtures['particle_antipodal_wild'] = record_particle(ns, 'particle_stage2_antipodal.json', 4, 0.2, 12345, 4, 'wild')
    fixtures['particle_cross_greedy'] = record_particle(ns, 'particle_stage2_cross.json', 4, 0.2, 12346, 24, 'greedy')
    fixtures['particle_cross_uniform'] = record_particle(ns, 'particle_stage2_cross.json', 4, 0.2, 12341, 8, 'uniform')
    fixtures['particle_merge_greedy'] = record_particle(ns, 'particle_stage2_merge.json', 2, 0.2, 12347, 24, 'greedy')
    fixtures['particle_merge8_greedy'] = record_particle(ns, 'particle_merge8.json', 8, 0.2, 12348, 10, 'greedy')
    fixtures['particle_merge8_uniform'] = record_particle(ns, 'particle_merge8.json', 8, 0.2, 12341, 4, 'uniform')
    # --- the reference's largest worlds (make_world takes up to 10 agents: its colour table has ten rows,
    #     multi-goal_spread.py:7-16) on the build-defined ring config; 9 agents = its first nine entries ---------
    fixtures['particle_ring10_greedy'] = record_particle(ns, 'particle_ring10.json', 10, 0.2, 12349, 8, 'greedy')
    fixtures['particle_ring10_uniform'] = record_particle(ns, 'particle_ring10.json', 10, 0.2, 12341, 3, 'uniform')
    ring = load_cfg('particle_ring10.json')
    ring9 = dict(ring, n_agents=9, **{k: ring[k][:9] for k in ('agents_x', 'agents_y', 'landmarks_x', 'landmarks_y')})
    fixtures['particle_ring9_greedy'] = record_particle(ns, 'particle_ring10.json[:9]', 9, 0.2, 12350, 6, 'greedy', config_override=ring9)
    # --- KAT-P1: head-on 2-agent collision -----------------------------------
    kat_cfg = dict(n_agents=2, agents_x=[-0.2, 0.2], agents_y=[0, 0], landmarks_x=[0.9, -0.9],
                   landmarks_y=[0, 0], initial_std=0)
    fixtures['particle_kat_headon'] = record_particle(
        ns, 'KAT-P1', 2, 0.0, 1, 1, 'script', scripts=[[[2, 1], [2, 1], [2, 1], [0, 0], [0, 0], [1, 2]]],
        config_override=kat_cfg)
    # --- early termination: agents injected within / around the 0.05 reach radius ------------
    lm = [[0.9, 0.9], [-0.9, -0.9], [0.9, -0.9], [-0.9, 0.9]]
    inj = []
    scripts = []
    # ep0: all four already inside the radius and at rest -> done at t=1
    inj.append(([[0.9 + 0.01, 0.9], [-0.9, -0.9 + 0.02], [0.9 - 0.03, -0.9], [-0.9, 0.9 - 0.04]],
                [[0, 0]] * 4, lm))
    scripts.append([[0, 0, 0, 0]] * 3)
    # ep1: three inside, one outside walking in -> done when it arrives
    inj.append(([[0.9, 0.9], [-0.9, -0.9], [0.9, -0.9], [-0.9 + 0.30, 0.9]],
                [[0, 0]] * 4, lm))
    scripts.append([[0, 0, 0, 1]] * 3 + [[0, 0, 0, 0]] * 12)
    # ep2: all inside but moving fast -> leave the radius, never all-reached again
    inj.append(([[0.9, 0.9], [-0.9, -0.9], [0.9, -0.9], [-0.9, 0.9]],
                [[1.0, 0], [0, 1.0], [-1.0, 0], [0, -1.0]], lm))
    scripts.append([[0, 0, 0, 0]] * 6)
    fixtures['particle_early_done'] = record_particle(
        ns, 'particle_stage2_antipodal.json', 4, 0.0, 2, 3, 'script', scripts=scripts, inject=inj)
    # --- checkers -----------------------------------------------------------------------------
    fixtures['checkers_stage2_uniform'] = record_checkers(ns, 'checkers_stage2.json', 12341, 24, 'uniform')
    fixtures['checkers_stage2_wild'] = record_checkers(ns, 'checkers_stage2.json', 12342, 6, 'wild')
    s2 = checkers_scripts_stage2()
    fixtures['checkers_stage2_script'] = record_checkers(ns, 'checkers_stage2.json', 0, len(s2), 'script', scripts=s2)
    fixtures['checkers_stage1_uniform'] = record_checkers(ns, 'checkers_stage1.json', 12341, 16, 'uniform')
    s1, g1 = checkers_scripts_stage1()
    fixtures['checkers_stage1_script'] = record_checkers(ns, 'checkers_stage1.json', 0, len(s1), 'script',
                                                         scripts=s1, goals_list=g1)
    total = 0
    for name, arrs in fixtures.items():
        if only and not name.startswith(only):
            continue
        path = os.path.join(OUT_DIR, name + '.npz')
        np.savez_compressed(path, **arrs)
        sz = os.path.getsize(path)
        total += sz
        extra = ''
        if 'collisions' in arrs:
            extra = ' collisions(final, summed)=%d early_done=%d' % (
                int(sum(arrs['collisions'][e, arrs['ep_len'][e] - 1] for e in range(len(arrs['ep_len'])))),
                int((arrs['ep_len'] < MAX_STEPS).sum()))
        else:
            extra = ' early_done=%d' % int((arrs['ep_len'] < MAX_STEPS).sum())
        print('%-32s episodes=%3d  %7.1f KiB%s' % (name, len(arrs['ep_len']), sz / 1024.0, extra))
    print('total %.1f KiB' % (total / 1024.0))


if __name__ == '__main__':
    main()
