"""TEST INFRASTRUCTURE ONLY -- container-side loader for the *real* reference envs.

This module imports the reference's own Python env code from ``/root/reference``
(read-only) so that ``oracle/gen_golden.py`` can record golden input/output
vectors.  It is never imported by the product (``cm3_amd``), by ``bench.py``,
by ``__graft_entry__.smoke()`` or by any ``-m gpu`` test: ``/root/reference``
does not exist on the GPU box and the reference may not travel in any form.

Shims (harness-side only; the reference tree is not modified) -- SURVEY.md §8(c):
  1. ``gym`` is not installed: a stub module provides the handful of names the
     reference touches at import/constructor time
     (environment.py:1-3,46-70; multi_discrete.py:9; multiagent/__init__.py:1-18).
  2. ``np.int`` / ``np.float`` aliases (removed in NumPy>=1.24; used at
     checkers.py:117,279 and train_onpolicy.py:295).
  3. ``sys.dont_write_bytecode`` so nothing is written under /root/reference.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CM3_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "env"))


def _install_gym_stub():
    if "gym" in sys.modules:
        return
    gym = types.ModuleType("gym")

    class Env(object):
        pass

    class Space(object):
        pass

    gym.Env = Env
    gym.Space = Space
    spaces = types.ModuleType("gym.spaces")

    class Discrete(object):
        def __init__(self, n):
            self.n = n

    class Box(object):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Tuple(object):
        def __init__(self, members):
            self.spaces = members

    spaces.Discrete, spaces.Box, spaces.Tuple = Discrete, Box, Tuple
    gym.spaces = spaces
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")

    class EnvSpec(object):
        pass

    registration.EnvSpec = EnvSpec
    registration.register = lambda *a, **k: None
    envs.registration = registration
    gym.envs = envs
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.envs": envs,
                        "gym.envs.registration": registration})


def load_reference():
    """Returns a namespace with the reference's MultiAgentEnv, scenarios loader and Checkers."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    import numpy as np
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float
    _install_gym_stub()
    for p in (os.path.join(REFERENCE_ROOT, "env", "multiagent-particle-envs"), REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from multiagent.environment import MultiAgentEnv  # noqa
    import multiagent.scenarios as scenarios  # noqa
    from env import checkers  # noqa  (namespace package, as train_onpolicy.py:29 does)
    ns = types.SimpleNamespace()
    ns.MultiAgentEnv = MultiAgentEnv
    ns.scenarios = scenarios
    ns.checkers = checkers
    ns.root = REFERENCE_ROOT
    return ns


def make_reference_particle_env(ns, n_agents, config, prob_random, max_steps):
    """Mirrors train_onpolicy.py:117-119."""
    scenario = ns.scenarios.load("multi-goal_spread.py").Scenario()
    world = scenario.make_world(n_agents, config, prob_random)
    env = ns.MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation,
                           None, scenario.done, max_steps=max_steps)
    return env, scenario, world


def make_reference_checkers_env(ns, init, n_agents, max_steps):
    """Mirrors train_onpolicy.py:126."""
    return ns.checkers.Checkers(init["n_rows"], init["n_columns"], init["n_obs"],
                                init["agents_r"], init["agents_c"], n_agents, max_steps)
