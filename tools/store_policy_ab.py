#!/usr/bin/env python
"""EXPERIMENT: does the way the observation rows are stored explain the 0.85 us/tick that trajectory mode costs over in-place
stepping at C2 (same bytes per tick; only the destination lines differ: fresh HBM vs re-used)?  CM3_EXPERIMENT_OBS_STORE selects
the store flavour of the pair mapping's obs_others rows (0 plain, 1 nt, 2 sc1 write-through, 3 sc0 sc1); each flavour runs in
its own process (the variable is read when launches are built), alternating, 3 repeats; us per tick by HIP events."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import sys, torch
sys.path.insert(0, %r)
import cm3_amd
from cm3_amd.particle import VecParticleEnv
from cm3_amd.rollout import ParticleRollout
from bench import ParticleStepper
cfg = cm3_amd.load_config("particle_stage2_antipodal")
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
def t(fn, n):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / (n * 330)
env = VecParticleEnv(cfg, 4, 0.2, 33, 4096, device=dev, auto_reset=True); env.reset()
ro = ParticleRollout(env, n_ticks=330, use_graph=True)
for _ in range(3): ro.collect(reset=False)
st = ParticleStepper(cfg, 4, 4096, dev); st.capture(330); st.run(990)
print("%%.3f %%.3f" %% (t(lambda: ro.collect(reset=False), 20), t(lambda: st.run(330), 20)))
# parity guard: the flavour must not change a single stored value
print(int(ro.obs_others.view(torch.int32).sum().item()), int(ro.state.view(torch.int32).sum().item()))
''' % ROOT


def main():
    names = {0: "plain", 1: "nt", 2: "sc1 (write-through)", 3: "sc0 sc1"}
    res = {k: [] for k in names}
    sums = {}
    for rep in range(3):
        for pol in names:
            env = dict(os.environ, CM3_EXPERIMENT_OBS_STORE=str(pol))
            out = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=300)
            lines = [l for l in out.stdout.strip().splitlines() if l and l[0].isdigit() or l.startswith("-")]
            if out.returncode != 0 or len(lines) < 2:
                print("policy %d failed: %s" % (pol, out.stderr[-400:]))
                continue
            res[pol].append(tuple(float(x) for x in lines[0].split()))
            sums.setdefault(pol, set()).add(lines[1])
    print("%-22s %-34s %-34s" % ("obs_others store", "trajectory us/tick (3 runs)", "in-place us/tick (3 runs)"))
    for pol, name in names.items():
        print("%-22s %-34s %-34s" % (name, " ".join("%.3f" % r[0] for r in res[pol]), " ".join("%.3f" % r[1] for r in res[pol])))
    ref = sums.get(0)
    print("stored values identical across flavours:", all(sums.get(p) == ref for p in names), sums)


if __name__ == "__main__":
    main()
