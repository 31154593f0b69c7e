#!/usr/bin/env python
"""A/B with repeats: C2 trajectory-mode headline (330-tick phases, one chain) with and without the per-tick collisions slot,
us per tick by HIP events, 5 alternating repeats each -- to attribute a 3.69 -> 3.94 us/tick difference between two single
bench runs (noise, or the cost of the extra 4 B/env-step store)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cm3_amd
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    cfg = cm3_amd.load_config("particle_stage2_antipodal")
    dev = torch.device("cuda:0")
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ros = {}
    for rec in (True, False):
        env = VecParticleEnv(cfg, 4, 0.2, 33, 4096, device=dev, auto_reset=True)
        env.reset()
        ros[rec] = ParticleRollout(env, n_ticks=330, use_graph=True, record_collisions=rec)
        for _ in range(3):
            ros[rec].collect(reset=False)
    torch.cuda.synchronize()
    print("%-22s %s" % ("variant", "us/tick per repeat (20 phases each)"))
    res = {True: [], False: []}
    for rep in range(5):
        for rec in (True, False):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                ros[rec].collect(reset=False)
            b.record()
            b.synchronize()
            res[rec].append(a.elapsed_time(b) * 1e3 / (20 * 330))
    for rec in (True, False):
        print("%-22s %s" % ("collisions slot " + ("ON" if rec else "OFF"), " ".join("%.3f" % x for x in res[rec])))


if __name__ == "__main__":
    main()
