#!/usr/bin/env python
"""Why the in-place E-sweep peaks at 2^18..2^20 envs and falls back at 2^22.

Hypothesis: in place every tick re-reads and overwrites the SAME buffers; while they total less than a few times the 256 MiB
Infinity Cache a fraction of every tick's traffic is served on-die, so the "algorithmic GB/s" of those points is not all HBM
traffic.  At 2^22 envs (N = 4: 1.7 GB per tick) nothing survives from tick to tick and the number is the pure streaming rate.
Three measurements on one box:
  (1) the copy / read probes (cm3_hbm_copy_bench / cm3_hbm_read_bench) over working sets of 64 MB .. 4 GB, the SAME buffer
      passed over 20 times: the on-die share shows as bandwidth above the large-buffer plateau;
  (2) the in-place step kernel at 2^18 .. 2^23 envs (per tick: 400 B x envs of algorithmic traffic);
  (3) the same kernel STREAMING (trajectory mode: every tick its own slots, nothing re-used) at 2^18 and 2^20 envs.
If (3) at 2^20 is about what (2) gives at 2^22, and (1) plateaus below its small-buffer values, the fall-back is the cache
dropping out, not a property of the kernel.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import cm3_amd  # noqa: E402
from cm3_amd import _lib  # noqa: E402


def probe(dev, mb, reps=20):
    lib = _lib.lib()
    nbytes = (mb << 20) // 32 * 32
    buf = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    buf.random_(0, 1 << 30)
    sink = torch.zeros(lib.cm3_hbm_bench_sink_words(), dtype=torch.int32, device=dev)
    s = _lib.current_stream_handle(dev)
    half = nbytes // 2

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    t_read = timed(lambda: _lib.check(lib.cm3_hbm_read_bench(buf.data_ptr(), nbytes, sink.data_ptr(), s)))
    t_copy = timed(lambda: _lib.check(lib.cm3_hbm_copy_bench(buf.data_ptr() + half, buf.data_ptr(), half, s)))
    del buf
    torch.cuda.empty_cache()
    return nbytes / t_read / 1e9, 2.0 * half / t_copy / 1e9


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    print("# tools/sweep_2p22.py on %s" % torch.cuda.get_device_name(0))
    print("(1) probes, the same buffer passed over 20 times: working set MB -> read GB/s, copy GB/s (read + write bytes)")
    for mb in (64, 128, 256, 512, 1024, 2048, 4096):
        r, c = probe(dev, mb)
        print("    %5d MB   read %6.0f   copy %6.0f" % (mb, r, c))
    cfg = cm3_amd.load_config("particle_stage2_antipodal")
    N, bps = 4, bench.algorithmic_bytes_per_env_step(4)
    print("(2) in-place step kernel, N = 4, hipGraph of 33 ticks, second of two timed passes: envs -> us per tick, algorithmic GB/s "
          "(working set per tick = %d B x envs)" % bps)
    for log2e in (18, 19, 20, 21, 22, 23):
        Es = 1 << log2e
        st = bench.ParticleStepper(cfg, N, Es, dev)
        st.capture(bench.EP_TICKS)
        st.run(bench.EP_TICKS)
        torch.cuda.synchronize(dev)
        n = bench.EP_TICKS * 3
        bench.timed_ticks(st, n)
        us = bench.timed_ticks(st, n) * 1e3 / n
        print("    2^%d  %8.2f us  %6.0f GB/s  (%.2f of 8 TB/s)   working set %5.0f MB" %
              (log2e, us, bps * Es / us / 1e3, bps * Es / us / 1e3 / 8000.0, bps * Es / 2.0 ** 20))
        st.close()
        del st
        torch.cuda.empty_cache()
    print("(3) the same kernel streaming into a [34, E, ...] trajectory (nothing re-used from tick to tick)")
    for log2e in (18, 20):
        Es = 1 << log2e
        st = bench.TrajectoryStepper(cfg, N, Es, dev, phase_ticks=bench.EP_TICKS)
        st.run(bench.EP_TICKS)
        torch.cuda.synchronize(dev)
        n = bench.EP_TICKS * 3
        bench.timed_ticks(st, n)
        us = bench.timed_ticks(st, n) * 1e3 / n
        print("    2^%d  %8.2f us  %6.0f GB/s  (%.2f of 8 TB/s)" % (log2e, us, bps * Es / us / 1e3, bps * Es / us / 1e3 / 8000.0))
        st.close()
        del st
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
