#!/usr/bin/env python
"""Do two hardware queues run their own chains of small DEPENDENT launches at full rate side by side?
Each chain = a hipGraph of `nodes` launches of the load -> store skeleton (cm3_traffic_floor_bench: C2's traffic or empty),
replayed on its own stream.  Printed: us per launch of one chain alone, and of each chain while K chains run concurrently
(perfect concurrency keeps the per-chain figure; a shared or time-sliced queue multiplies it)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from cm3_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    nodes, reps = 330, 10
    n_streams = 6
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    src = [torch.zeros(1 << 20, dtype=torch.int32, device=dev) for _ in range(n_streams)]
    dst = [torch.zeros(1 << 20, dtype=torch.int32, device=dev) for _ in range(n_streams)]
    for label, rb, wb, blocks in (("empty", 0, 0, 256), ("c2-traffic", 475136, 1163264, 256), ("half-c2", 237568, 581632, 128)):
        graphs = []
        for k in range(n_streams):
            def enq(s, k=k):
                for _ in range(nodes):
                    _lib.check(lib.cm3_traffic_floor_bench(src[k].data_ptr(), rb, dst[k].data_ptr(), wb, blocks, 256, s))
            graphs.append(_lib.capture_graph(dev, enq))

        def run(idx):
            for _ in range(2):
                for i in idx:
                    _lib.check(lib.cm3_graph_launch(graphs[i], streams[i].cuda_stream))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                for i in idx:
                    _lib.check(lib.cm3_graph_launch(graphs[i], streams[i].cuda_stream))
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6 / (reps * nodes)

        print("%-11s alone: %s" % (label, " ".join("%.2f" % run([i]) for i in range(n_streams))))
        for idx in ([0, 1], [0, 2], [1, 2], [2, 3], [0, 1, 2], [0, 1, 2, 3], [0, 1, 2, 3, 4, 5]):
            print("%-11s chains %-18s us per launch per chain: %.2f" % (label, idx, run(idx)))
        torch.cuda.synchronize()
        for g in graphs:
            lib.cm3_graph_destroy(g)


if __name__ == "__main__":
    main()
