#!/usr/bin/env python
"""N = 8 at 2^18 .. 2^20 envs in place against the load -> store skeleton with the SAME bytes ("N = 8 at
2^20 >= 5.4 TB/s algorithmic").  A step of 8 agents reads 228 B and writes 1068 B per env: 82 % of the traffic is stores, so the
1:1 copy rate of the chip is not its roof -- the skeleton (cm3_traffic_floor_bench: all loads, then stores of a value that depends
on them, nothing else) with the same read / write bytes is.  Prints us per launch and algorithmic TB/s for every mapping and for
the skeleton at several grid shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import cm3_amd  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    print("# tools/n8_large_floor.py on %s" % torch.cuda.get_device_name(0))
    T = bench.EP_TICKS
    for cfg_name, N in (("particle_merge8", 8), ("particle_stage2_antipodal", 4)):
        cfg = cm3_amd.load_config(cfg_name)
        bps = bench.algorithmic_bytes_per_env_step(N)
        for log2e in ((18, 20) if N == 8 else (20,)):
            E = 1 << log2e
            rd, wr = (28 * N + 4) * E, (20 * N + 16 * N * (N - 1) + 12) * E
            print("N = %d, 2^%d envs: %d B read + %d B written per env (%.0f %% stores), %.0f MB per tick" %
                  (N, log2e, rd // E, wr // E, 100.0 * wr / (rd + wr), (rd + wr) / 2.0 ** 20))
            for kernel in ("auto", "env", "agent"):
                st = bench.ParticleStepper(cfg, N, E, dev, kernel=kernel)
                st.capture(T)
                st.run(T)
                torch.cuda.synchronize(dev)
                bench.timed_ticks(st, T * 2)
                us = bench.timed_ticks(st, T * 3) * 1e3 / (T * 3)
                print("    step kernel %-5s  %8.2f us   %5.2f TB/s algorithmic" % (kernel, us, bps * E / us / 1e6))
                st.close()
                del st
                torch.cuda.empty_cache()
            for blocks, threads in ((2048, 256), (4096, 256), (8192, 256), (2048, 1024), (E * 8 // 256, 256)):
                fl = bench.measure_launch_floor(dev, (rd + 15) // 16 * 16, (wr + 15) // 16 * 16, blocks, threads, nodes=T)
                us = fl["same_traffic_us"]
                print("    skeleton %7d x %4d  %8.2f us   %5.2f TB/s" % (blocks, threads, us, (rd + wr) / us / 1e6))
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
