"""GPU: a plain C++ program (examples/kat_p1.cpp) that links libcm3_hip.so and uses only the C ABI + the HIP runtime -- no
Python, no PyTorch -- replays the reference's known-answer vector KAT-P1 in float64."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_consumer_replays_kat_p1_and_kat_c1():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    import cm3_amd._lib as L
    L.lib()                                                    # fails loudly if the library is missing
    exe = os.path.join(ROOT, "examples", "kat_p1")
    libdir = os.path.join(ROOT, "cm3_amd")
    subprocess.run([hipcc, "-std=c++17", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "kat_p1.cpp"), "-L", libdir, "-lcm3_hip", "-Wl,-rpath," + libdir,
                    "-o", exe], check=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "KAT-P1 through the C ABI" in out.stdout and "KAT-C1 through the C ABI: exact" in out.stdout


@pytest.mark.parametrize("kernel", ["pair", "agent"])
def test_forced_shared_env_mapping_beyond_4gib_is_refused_before_any_launch(kernel):
    """The lane-per-pair / lane-per-agent kernels index with 32-bit byte offsets (include/cm3_amd.h): a FORCED mapping whose
    obs_others array would reach 4 GiB must come back as CM3_ERR_INVALID with a message -- checked before anything is launched, so
    a descriptor that merely claims 5 M envs over small buffers is safe to probe with."""
    import torch
    from cm3_amd import _lib
    from cm3_amd.particle import VecParticleEnv
    from tests.helpers import load_cfg
    env = VecParticleEnv(load_cfg("particle_merge8.json"), 8, 0.2, 33, 64, device="cuda:0", dtype=torch.float32, kernel=kernel)
    env.reset()
    env.step()                                              # the honest descriptor works
    env._desc.n_envs = 5_000_000                            # 5 M x 8 x 7 x 16 B = 4.5 GB of obs_others
    with pytest.raises(_lib.Cm3Error, match="4 GiB"):
        env.step()
    env._desc.n_envs = 64
    env.step()
    torch.cuda.synchronize()
