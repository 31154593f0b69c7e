"""Where do the whole-episode policy launch and the alternating actor / step launches part ways?  (debug aid)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import tests.test_gpu_actor as TA


def first_diff(a, b):
    for name in ("actions", "state", "obs_others", "reward_n", "reward", "done", "collisions"):
        x, y = getattr(a, name), getattr(b, name)
        ne = (x != y)
        if ne.any():
            t = int(ne.reshape(ne.shape[0], -1).any(1).nonzero()[0])
            idx = ne[t].nonzero()[:3].tolist()
            print("   %-10s first differs at slot %d, %d elements there, e.g. %s: %s vs %s" % (name, t, int(ne[t].sum()), idx[0], x[t][tuple(idx[0])].item(), y[t][tuple(idx[0])].item()))
        else:
            pass


for E, N, cfg, prec in [(8192, 8, "particle_merge8.json", "f16x3"), (8192, 8, "particle_merge8.json", "f32"), (4096, 8, "particle_merge8.json", "f16x3"),
                        (2048, 8, "particle_merge8.json", "f16x3"), (16384, 4, "particle_stage2_cross.json", "f16x3"), (16384, 2, "particle_stage2_merge.json", "f16x3")]:
    from cm3_amd import _lib as _L
    _L.lib().cm3_policy_force_row_tiles(0)
    ref, eref, v0 = TA._policy_run(E, N, cfg, prec, 9, "tick")
    print(E, N, prec, "tick mode:", v0)
    for rt in ("", "4", "2", "1"):
        _L.lib().cm3_policy_force_row_tiles(int(rt) if rt else 0)
        ro, env, v = TA._policy_run(E, N, cfg, prec, 9, "episode")
        print("  RT=%s %s" % (rt or "auto", v))
        first_diff(ref, ro)
        ro.close()
    _L.lib().cm3_policy_force_row_tiles(0)
    ro, env, v = TA._policy_run(E, N, cfg, prec, 9, "tick", fused_policy_tick=True)
    print("  fused tick", v)
    first_diff(ref, ro)
    ro.close()
    ref.close()
