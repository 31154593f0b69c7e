import sys, torch, numpy as np
sys.path.insert(0, ".")
import cm3_amd
from bench import ParticleStepper, CheckersStepper, TrajectoryStepper, CheckersTrajectoryStepper, timed_ticks
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
cfg8 = cm3_amd.load_config("particle_merge8")
out = []
for N, E in ((4, 1024), (4, 2048), (4, 4096), (5, 2048), (6, 2048), (8, 1024)):
    st = ParticleStepper(cfg8, N, E, dev, kernel="pair"); st.capture(330); st.run(660); torch.cuda.synchronize()
    a = min(timed_ticks(st, 3300) * 1e3 / 3300 for _ in range(2)); st.close(); del st
    tr = TrajectoryStepper(cfg8, N, E, dev, kernel="pair"); tr.run(660); torch.cuda.synchronize()
    b = min(timed_ticks(tr, 3300) * 1e3 / 3300 for _ in range(2)); tr.close(); del tr
    out.append("particle N=%d E=%d in-place %.3f trajectory %.3f" % (N, E, a, b))
for stage, E in ((2, 8192), (2, 4096), (1, 8192)):
    cfg = cm3_amd.load_config("checkers_stage%d" % stage)
    st = CheckersStepper(cfg, E, dev); st.capture(330); st.run(660); torch.cuda.synchronize()
    a = min(timed_ticks(st, 3300) * 1e3 / 3300 for _ in range(2)); st.close(); del st
    out.append("checkers stage%d E=%d in-place %.3f" % (stage, E, a))
print("\n".join(out))
