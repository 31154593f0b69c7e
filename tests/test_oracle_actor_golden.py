"""The actor oracles (oracle/actor_oracle.py, oracle/actor_checkers_oracle.py) against golden vectors produced by executing
the REFERENCE's own function bodies networks.actor_particle / networks.actor_checkers under a NumPy stand-in for the TF1
calls they make (oracle/tf_numpy_shim.py, oracle/gen_golden_actor.py -> tests/golden/actor_*.npz): layer wiring, concat
order, variable names and shapes are the reference's; the primitive-op semantics are the shim's (TensorFlow itself is not
installable here).  Where /root/reference exists the vectors are re-derived and compared with the committed ones."""
import os

import numpy as np
import pytest

from oracle import actor_checkers_oracle as CO
from oracle import actor_oracle as PO
from tests.helpers import GOLDEN

REF = "/root/reference"


def load_cases(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    tags = sorted({k.split("/")[0] for k in z.files})
    cases = {}
    for tag in tags:
        w = {k[len(tag) + 3:]: z[k] for k in z.files if k.startswith(tag + "/w/")}
        inp = {k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(tag + "/in/")}
        cases[tag] = (w, inp, z[tag + "/probs"])
    return cases


@pytest.mark.parametrize("tag", ["n1_stage1", "n4_stage2", "n8_stage2"])
def test_particle_actor_oracle_matches_reference_wiring(tag):
    w, inp, probs = load_cases("actor_particle")[tag]
    got = PO.actor_probs(w, inp["obs_others"], inp["v_obs"], inp["v_goal"])
    assert got.shape == probs.shape and np.abs(got - probs).max() < 1e-6
    assert np.ptp(probs, axis=1).mean() > 0.05
    assert set(w) == set(PO.init_weights(np.random.default_rng(0), int(tag[1]), stage=int(tag[-1])))


@pytest.mark.parametrize("tag", ["n1_stage1", "n2_stage2"])
def test_checkers_actor_oracle_matches_reference_wiring(tag):
    w, inp, probs = load_cases("actor_checkers")[tag]
    got = CO.actor_probs(w, inp["a_prev"], inp["obs_self_t"], inp["obs_self_v"], inp["obs_others"], inp["goals"])
    assert got.shape == probs.shape and np.abs(got - probs).max() < 1e-6
    assert set(w) == set(CO.init_weights(np.random.default_rng(0), int(tag[1]), stage=int(tag[-1])))
    for k, v in CO.init_weights(np.random.default_rng(0), int(tag[1]), stage=int(tag[-1])).items():
        assert v.shape == w[k].shape, k


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "alg")), reason="reference tree not present")
def test_reference_functions_under_the_shim_reproduce_the_committed_vectors():
    from oracle import tf_numpy_shim as S
    for name, fn in (("actor_particle", "actor_particle"), ("actor_checkers", "actor_checkers")):
        for tag, (w, inp, probs) in load_cases(name).items():
            shim = S.Shim(weights=dict(w))
            net = S.load_networks(shim)
            stage = int(tag[-1])
            if name == "actor_particle":
                got = net.actor_particle(S._t(inp["obs_others"]), S._t(inp["v_obs"]), S._t(inp["v_goal"]), n_actions=5,
                                         n_h1_self=64, n_h1_others=128, n_h2=64, stage=stage)
            else:
                a1 = np.eye(5, dtype=np.float32)[inp["a_prev"]]
                got = net.actor_checkers(S._t(a1), S._t(inp["obs_self_t"]), S._t(inp["obs_self_v"]), S._t(inp["obs_others"]),
                                         S._t(inp["goals"]), f1=6, k1=[3, 3], n_h1=256, n_h2=256, n_actions=5, stage=stage)
            assert not shim.created, shim.created          # every variable the reference asks for is in the fixture
            assert np.array_equal(np.asarray(got), probs), (name, tag)
