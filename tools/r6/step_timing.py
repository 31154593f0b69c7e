"""Round 6: VecParticleEnv.step(actions) called from Python once per tick at C2, us per tick by wall clock."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cm3_amd
from cm3_amd.particle import VecParticleEnv
dev = torch.device("cuda:0")
cfg = cm3_amd.load_config("particle_stage2_antipodal")
E, N = 4096, 4
env = VecParticleEnv(cfg, N, 0.2, 33, E, device=dev, auto_reset=True)
env.reset()
a_dev = torch.randint(0, 5, (E, N), dtype=torch.int32, device=dev)
a_host = np.random.default_rng(0).integers(0, 5, (E, N))
out = {}
for label, act in (("device_int32", a_dev), ("host_int64", a_host), ("none_in_kernel", None)):
    for _ in range(100):
        env.step(act)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        env.step(act)
    torch.cuda.synchronize()
    out[label] = round((time.perf_counter() - t0) / 2000 * 1e6, 2)
print(json.dumps(out))
