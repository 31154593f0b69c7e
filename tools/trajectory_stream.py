#!/usr/bin/env python
"""Trajectory-mode streaming at large batches: the product's collector (one launch per tick, slots t -> t + 1 of a
[T + 1, E, ...] device trajectory incl. terminal capture) at E = 2^18 .. 2^22 envs x 4 agents, us per tick and algorithmic GB/s --
  full      : every goals slot written every tick (sparse_goals=False: rounds 1-3)
  sparse    : the product's default at these sizes since round 4 -- a goals slot is written only where an env restarts
  no-goals  : no goals slots at all (a live goals array): the bound of what the goals can give
Usage: trajectory_stream.py [log2E ...]   (with --once LOG2E VARIANT: a single short run for rocprofv3 --pmc passes)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(log2e, variant, T=33):
    import torch
    import cm3_amd
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    dev = torch.device("cuda:0")
    cfg = cm3_amd.load_config("particle_stage2_antipodal")
    E = 1 << log2e
    env = VecParticleEnv(cfg, 4, 0.2, 33, E, device=dev, auto_reset=True)
    env.reset()
    ro = ParticleRollout(env, n_ticks=T, use_graph=True, sparse_goals=(variant == "sparse"))
    if variant == "no-goals":
        ro.goals = None          # the kernel then keeps ONE live goals array (stride 0) and writes it only for restarting envs
    return env, ro, E, T


def main():
    import torch
    torch.cuda.set_stream(torch.cuda.Stream(device="cuda:0"))
    if len(sys.argv) > 1 and sys.argv[1] == "--once":
        env, ro, E, T = build(int(sys.argv[2]), sys.argv[3])
        for _ in range(3):
            ro.collect(reset=False)
        torch.cuda.synchronize()
        return
    sizes = [int(a) for a in sys.argv[1:]] or [18, 20, 22]
    print("%5s %-9s %10s %10s %8s" % ("log2E", "variant", "us/tick", "GB/s(alg)", "of 8TB/s"))
    for log2e in sizes:
        for variant in ("full", "sparse", "no-goals"):
            env, ro, E, T = build(log2e, variant)
            from cm3_amd import _lib
            for _ in range(2):
                ro.collect(reset=False)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(3):
                    ro.collect(reset=False)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / (3 * T) * 1e6)
            gbs = 400.0 * E / (best * 1e-6) / 1e9
            print("%5d %-9s %10.2f %10.0f %8.3f   %s" % (log2e, variant, best, gbs, gbs / 8000.0, _lib.last_kernel_variant()), flush=True)
            ro.close()
            del ro, env
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
