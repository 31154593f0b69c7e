#!/usr/bin/env python
"""Diagnostic for sub-batch chains at C2, in place, us per tick of the WHOLE batch -- K sub-batches of the envs (cm3_particle_desc.env_offset /
env_count) each advancing through its ticks on a stream of its own.  The library had an entry point for this until ABI 6
(cm3_particle_rollout_chains_*); it measured 1.2-5x SLOWER than one chain in every form (profiles/r02_chains_diag.txt) and was removed
in round 6 -- this tool rebuilds the three forms from the two descriptor fields to reproduce that result:
  graph-branches : one hipGraph, K parallel branches (fork / join by events inside the capture)
  eager-streams  : the same fork / join enqueued eagerly on K plain streams (no graph)
  graph-per-chain: K independent hipGraphs (one per sub-batch, env_offset/env_count), each replayed on its own stream
Separates "hipGraph branch overhead" from "what K hardware queues do with small dependent kernels"."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cm3_amd
    from cm3_amd import _lib
    from bench import ParticleStepper
    lib = _lib.lib()
    cfg = cm3_amd.load_config("particle_stage2_antipodal")
    E, N, T, reps = 4096, 4, 330, 10
    dev = torch.device("cuda:0")
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main_stream)
        for _ in range(reps):
            fn()
        b.record(main_stream)
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / (reps * T)

    def fork_join(st, K, s0, side):
        """K sub-batch rollouts, chain c on stream c; side streams forked from / joined into s0 by events"""
        chunk = ((E + K - 1) // K + 255) // 256 * 256
        fork = torch.cuda.Event()
        fork.record(torch.cuda.ExternalStream(s0, device=dev))
        for c in range(K):
            lo = c * chunk
            if lo >= E:
                break
            sc = torch.cuda.ExternalStream(s0, device=dev) if c == 0 else side[c - 1]
            if c:
                sc.wait_event(fork)
            st.env._desc.env_offset, st.env._desc.env_count = (lo, min(chunk, E - lo)) if K > 1 else (0, 0)
            _lib.check(lib.cm3_particle_rollout_f32(ctypes.byref(st.env._desc), ctypes.byref(st.traj), T, sc.cuda_stream))
            st.env._desc.env_offset, st.env._desc.env_count = 0, 0
            if c:
                done = torch.cuda.Event()
                done.record(sc)
                torch.cuda.ExternalStream(s0, device=dev).wait_event(done)

    print("%-16s %3s %10s" % ("mode", "K", "us/tick"))
    for K in (1, 2, 4):
        side = [torch.cuda.Stream(device=dev) for _ in range(K - 1)]
        st = ParticleStepper(cfg, N, E, dev)
        g = _lib.capture_graph(dev, lambda s: fork_join(st, K, s, side))
        print("%-16s %3d %10.2f" % ("graph-branches", K, timed(lambda: _lib.check(lib.cm3_graph_launch(g, main_stream.cuda_stream)))))
        torch.cuda.synchronize()
        lib.cm3_graph_destroy(g)
        print("%-16s %3d %10.2f" % ("eager-streams", K, timed(lambda: fork_join(st, K, main_stream.cuda_stream, side))))
        # K independent graphs, each over its own env range, each on its own stream; main stream forks / joins by events
        chunk = (E + K - 1) // K
        chunk = (chunk + 255) // 256 * 256
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        graphs = []
        for c in range(K):
            lo = c * chunk
            cnt = min(chunk, E - lo)
            if cnt <= 0:
                break

            def enq(s, lo=lo, cnt=cnt):
                st.env._desc.env_offset, st.env._desc.env_count = lo, cnt
                _lib.check(lib.cm3_particle_rollout_f32(ctypes.byref(st.env._desc), ctypes.byref(st.traj), T, s))
                st.env._desc.env_offset, st.env._desc.env_count = 0, 0
            graphs.append(_lib.capture_graph(dev, enq))

        def run_per_chain():
            ev = torch.cuda.Event()
            ev.record(main_stream)
            for g, s in zip(graphs, streams):
                s.wait_event(ev)
                _lib.check(lib.cm3_graph_launch(g, s.cuda_stream))
                done = torch.cuda.Event()
                done.record(s)
                main_stream.wait_event(done)
        print("%-16s %3d %10.2f" % ("graph-per-chain", K, timed(run_per_chain)))
        torch.cuda.synchronize()
        for g in graphs:
            lib.cm3_graph_destroy(g)


if __name__ == "__main__":
    main()
