"""CPU: the C-ABI library builds, loads, and exports every symbol include/cm3_amd.h declares; the
product fails loudly (no CPU fallback) when asked to compute without a GPU.  No compute calls here."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cm3_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cm3_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from cm3_amd import _lib
    return _lib


def test_header_symbols_all_exported(built):
    declared = _declared_symbols()
    assert len(declared) >= 20
    handle = built.lib()
    for name in declared:
        assert hasattr(handle, name), "libcm3_hip.so does not export %s" % name
    assert sorted(built.SYMBOLS) == declared, "ctypes binding and header disagree"
    assert handle.cm3_abi_version() == built.ABI_VERSION


def test_struct_sizes_match_header(built):
    import ctypes
    # sizes computed from include/cm3_amd.h by hand: catches accidental field drift
    assert ctypes.sizeof(built.ParticleDesc) == 4 * 4 + 8 + 8 + 8 + 8 + 4 * 8 * 8
    assert ctypes.sizeof(built.ParticleBufs) == 14 * 8
    assert ctypes.sizeof(built.ParticleTraj) == 20 * 8
    assert ctypes.sizeof(built.CheckersDesc) == 10 * 4 + 8 + 8 + 2 * 8 * 4
    assert ctypes.sizeof(built.CheckersBufs) == 14 * 8
    assert ctypes.sizeof(built.ActorParticleDesc) == 10 * 4 + 8 + 8
    assert ctypes.sizeof(built.ActorParticleWeights) == 10 * 8
    assert ctypes.sizeof(built.ActorParticleBufs) == 7 * 8


def test_invalid_arguments_return_error_codes_without_a_gpu(built):
    import ctypes
    handle = built.lib()
    d = built.ParticleDesc()
    b = built.ParticleBufs()
    d.n_envs, d.n_agents, d.max_steps = 0, 4, 33
    assert handle.cm3_particle_step_f32(ctypes.byref(d), ctypes.byref(b), None) == -1
    assert b"n_envs" in handle.cm3_last_error()
    d.n_envs, d.n_agents = 16, 9
    assert handle.cm3_particle_step_f32(ctypes.byref(d), ctypes.byref(b), None) == -1
    cd = built.CheckersDesc()
    cb = built.CheckersBufs()
    cd.n_envs, cd.n_agents, cd.n_rows, cd.n_columns, cd.n_obs, cd.max_steps = 8, 2, 4, 8, 2, 33
    assert handle.cm3_checkers_step(ctypes.byref(cd), ctypes.byref(cb), None) == -1
    assert b"odd" in handle.cm3_last_error()


def test_product_has_no_cpu_fallback():
    import torch
    import cm3_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cm3_amd.Cm3Error):
        cm3_amd.VecParticleEnv(cm3_amd.load_config("particle_stage1"), 1, 0.2, 33, 4, device="cuda:0")
    with pytest.raises(cm3_amd.Cm3Error):
        cm3_amd.VecParticleEnv(cm3_amd.load_config("particle_stage1"), 1, 0.2, 33, 4, device="cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cm3_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
