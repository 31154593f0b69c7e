"""Round 5: runs the withdrawn-layer builds of tools/probes/policy_fault_variants.sh on the GPU box.  For every library given
(cm3_amd/libcm3_hip_pf_<variant>.so) a child process compares whole-episode policy launches (k_policy_rollout from THAT build)
with alternating actor / step launches (HEAD's kernels, linked into every variant) on the batches that showed the fault, and
prints the number of differing tensors / elements plus the build's divergence counters when it exports them.

    python tools/probes/policy_fault_probe.py [--reps 3] variant [variant ...]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(8192, 8, "particle_merge8.json", "f16x3"), (4096, 8, "particle_merge8.json", "f16x3"), (8192, 8, "particle_merge8.json", "f32")]
if os.environ.get("PF_CASES") == "short":
    CASES = CASES[:1]
NAMES = ("actions", "state", "obs_others", "reward_n", "reward", "done", "collisions")


def worker(reps):
    sys.path.insert(0, ROOT)
    import torch
    import tests.test_gpu_actor as TA
    from cm3_amd import _lib
    h = _lib.lib()
    dbg = h.cm3_debug_counters if hasattr(h, "cm3_debug_counters") else None
    buf = (ctypes.c_uint * 32)()
    for E, N, cfg, prec in CASES:
        for rep in range(reps):
            seed = 21 + rep
            ref, eref, _ = TA._policy_run(E, N, cfg, prec, 9, "tick", seed=seed)
            if dbg:
                dbg(buf, 1)
            ro, env, v = TA._policy_run(E, N, cfg, prec, 9, "episode", seed=seed)
            torch.cuda.synchronize()
            bad = {}
            for name in NAMES:
                x, y = getattr(ref, name), getattr(ro, name)
                n = int((x != y).sum())
                if n:
                    t = int((x != y).reshape(x.shape[0], -1).any(1).nonzero()[0])
                    bad[name] = (n, t)
            tile = [s.strip() for s in v.split(",") if "g=" in s]
            line = "%5d x %d %-5s seed %d %s: %s" % (E, N, prec, seed, tile[0] if tile else v[:40],
                                                    "SAME" if not bad else "DIFF " + " ".join("%s:%d@slot%d" % (k, a, b) for k, (a, b) in bad.items()))
            if dbg:
                dbg(buf, 0)
                c = list(buf)
                line += "   | counters by fours (16-lane rows 0..3): " + " ".join(str(c[k:k + 4]) for k in range(0, 28, 4)) + " alive %d" % max(c[15], c[31])
            print(line, flush=True)
            if hasattr(h, "cm3_debug_records") and bad:
                import struct
                rec = (ctypes.c_uint * 128)()
                h.cm3_debug_records(rec)
                f = lambda u: struct.unpack("f", struct.pack("I", u))[0]     # noqa: E731
                for k in range(16):
                    r = rec[8 * k:8 * k + 8]
                    if r[0] == 0 and r[1] == 0:
                        continue
                    if os.environ.get("PF_REC") == "2":
                        print("      lane %2d pair %d tick %d wave %d  qx %.9g qy %.9g pen %.9g  f = (%.9g, %.9g)  row lane f = (%.9g, %.9g)  qx*pen = %.9g"
                              % (r[0] & 255, (r[0] >> 8) & 255, (r[0] >> 16) & 255, r[0] >> 24, f(r[1]), f(r[2]), f(r[7]), f(r[3]), f(r[4]), f(r[5]), f(r[6]),
                                 f(r[1]) * f(r[7])), flush=True)
                        continue
                    print("      lane %2d pair %d tick %d wave %d block %5d  dx %.9g dy %.9g (d %.6f)  f = (%.9g, %.9g)  row lane f = (%.9g, %.9g)  bits %08x %08x | %08x %08x"
                          % (r[0] & 255, (r[0] >> 8) & 255, (r[0] >> 16) & 255, r[0] >> 24, r[7], f(r[1]), f(r[2]), (f(r[1]) ** 2 + f(r[2]) ** 2) ** 0.5,
                             f(r[3]), f(r[4]), f(r[5]), f(r[6]), r[3], r[4], r[5], r[6]), flush=True)
            ro.close()
            ref.close()


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(int(sys.argv[2]))
        sys.exit(0)
    args = sys.argv[1:]
    reps = 3
    if args[0] == "--reps":
        reps, args = int(args[1]), args[2:]
    for v in args:
        path = v if os.path.exists(v) else os.path.join(ROOT, "cm3_amd", "libcm3_hip_pf_%s.so" % v)
        print("==== %s" % os.path.basename(path), flush=True)
        env = dict(os.environ, CM3_AMD_LIB=path)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(reps)], env=env, cwd=ROOT, capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode:
            print("   child failed:", r.stderr[-1500:])
