"""cm3_amd -- MI355X-native vectorised rollout engine for the CM3 hot path.

Only what the path needs: the HIP kernels + C ABI (csrc/, libcm3_hip.so), the ctypes binding
(_lib), and the host-side mirrors of the reference's env / trajectory interfaces.
"""
import json
import os

from ._lib import Cm3Error, lib  # noqa: F401

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def load_config(name):
    """Loads one of the build's own copies of the reference's JSON inputs (cm3_amd/configs)."""
    if not name.endswith(".json"):
        name += ".json"
    with open(os.path.join(CONFIG_DIR, name)) as f:
        return json.load(f)


def __getattr__(name):
    if name == "VecParticleEnv":
        from .particle import VecParticleEnv
        return VecParticleEnv
    if name == "VecCheckersEnv":
        from .checkers import VecCheckersEnv
        return VecCheckersEnv
    if name == "ParticleActor":
        from .actor import ParticleActor
        return ParticleActor
    raise AttributeError(name)
