import sys, torch
sys.path.insert(0, ".")
import cm3_amd
from bench import ParticleStepper, timed_ticks
cfg = cm3_amd.load_config("particle_stage2_antipodal")
dev = torch.device("cuda:0"); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
out = []
for E in (32768, 65536, 131072, 262144, 1048576):
    st = ParticleStepper(cfg, 4, E, dev, kernel="env"); st.capture(33); st.run(132); torch.cuda.synchronize()
    n = 330 if E <= 262144 else 99
    timed_ticks(st, n); out.append("%d:%.2f" % (E, min(timed_ticks(st, n) * 1e3 / n for _ in range(2)))); st.close(); del st; torch.cuda.empty_cache()
print(" ".join(out))
