#!/usr/bin/env python
"""Sweeps the configuration of the streaming read / copy probes (cm3_hbm_read_bench_cfg / cm3_hbm_copy_bench_cfg) on the GPU
box and prints GB/s per configuration; the best read configuration becomes cm3_hbm_read_bench's default
(kBenchUnroll / kBenchWgPerCu / kBenchNt in csrc/util.hip).  Output is committed as profiles/r02_hbm_probe_sweep.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from cm3_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    nbytes = 4 << 30
    buf = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    buf.random_(0, 1 << 30)
    sink = torch.zeros(lib.cm3_hbm_bench_sink_words(), dtype=torch.int32, device=dev)
    s = _lib.current_stream_handle(dev)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    print("# read probe: 4 GiB, 16 B per lane per load; GB/s")
    print("%6s %9s %3s %10s" % ("unroll", "wg_per_cu", "nt", "GB/s"))
    best = (0.0, None)
    for nt in (0, 1):
        for wg in (4, 8, 16, 32):
            for u in (1, 2, 4, 8, 16):
                t = timed(lambda: _lib.check(lib.cm3_hbm_read_bench_cfg(buf.data_ptr(), nbytes, sink.data_ptr(), u, wg, nt, s)))
                gbps = nbytes / t / 1e9
                print("%6d %9d %3d %10.1f" % (u, wg, nt, gbps))
                if gbps > best[0]:
                    best = (gbps, (u, wg, nt))
    print("# best read: %.1f GB/s at unroll=%d wg_per_cu=%d nt=%d" % ((best[0],) + best[1]))
    t = timed(lambda: _lib.check(lib.cm3_hbm_read_bench(buf.data_ptr(), nbytes, sink.data_ptr(), s)))
    print("# cm3_hbm_read_bench (library default): %.1f GB/s" % (nbytes / t / 1e9))
    half = nbytes // 2
    print("# copy probe: 2 GiB -> 2 GiB; GB/s = 2 x bytes / time")
    bestc = (0.0, None)
    for nt in (0, 1):
        for wg in (4, 8, 16, 32):
            for u in (1, 2, 4, 8):
                t = timed(lambda: _lib.check(lib.cm3_hbm_copy_bench_cfg(buf.data_ptr() + half, buf.data_ptr(), half, u, wg, nt, s)))
                gbps = 2.0 * half / t / 1e9
                print("%6d %9d %3d %10.1f" % (u, wg, nt, gbps))
                if gbps > bestc[0]:
                    bestc = (gbps, (u, wg, nt))
    print("# best copy: %.1f GB/s at unroll=%d wg_per_cu=%d nt=%d" % ((bestc[0],) + bestc[1]))
    # torch's own device-to-device copy for reference
    dst = torch.empty(half // 4, dtype=torch.int32, device=dev)
    t = timed(lambda: dst.copy_(buf[:half // 4]))
    print("# torch copy_: %.1f GB/s" % (2.0 * half / t / 1e9))


if __name__ == "__main__":
    main()
