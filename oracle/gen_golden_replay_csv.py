"""TEST INFRASTRUCTURE ONLY -- records what the REAL reference writes to log.csv / log_century.csv and how its replay buffers
behave (build container only).

* CSV rows: the reference builds them inline in alg/train_onpolicy.py (headers :201-215, per-century row :399-404, per-episode row
  :422-427), a script that cannot be imported (TensorFlow, SUMO).  The three statement runs are taken from the real file BY LINE
  NUMBER at generation time (their first lines are checked), executed unmodified in a namespace holding the chosen inputs, and the
  strings they produce are recorded.
* Buffers: alg/replay_buffer.py and alg/replay_buffer_dual.py are imported and driven directly.

    python oracle/gen_golden_replay_csv.py  ->  tests/golden/replay_csv.json
"""
import json
import os
import random
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


def run_slice(lines, first, last, must_start_with, ns):
    src = textwrap.dedent("".join(lines[first - 1:last]))
    assert src.lstrip().startswith(must_start_with), (first, src[:60])
    exec(compile(src, "train_onpolicy.py:%d-%d" % (first, last), "exec"), ns)
    return ns


def main():
    lines = open(os.path.join(REF, "alg", "train_onpolicy.py")).readlines()
    cases = []
    rng = np.random.default_rng(5)
    for n_agents in (1, 2, 4):
        ns = {"n_agents": n_agents, "experiment": "particle", "np": np}
        run_slice(lines, 201, 215, 'header = "Step,Episode,r_global"', ns)
        rows = []
        for k in range(4):
            vals = dict(step=int(rng.integers(1, 10 ** 6)), idx_episode=int(rng.integers(1, 50001)),
                        reward_global=float(rng.normal(-40, 30)), reward_local=rng.normal(-10, 8, n_agents),
                        reward_global_century=float(rng.normal(-4000, 900)), reward_local_century=rng.normal(-1000, 300, n_agents),
                        period=100, r_global_eval=float(rng.normal(-30, 10)), r_local_eval=rng.normal(-8, 3, n_agents),
                        t_env=float(rng.uniform(0, 500)), t_train=float(rng.uniform(0, 2000)))
            if k == 0:      # rounding edge cases of %.2f / '{:.2f}'
                vals.update(reward_global=-12.345, reward_local=np.array([-6.245, 0.005, 2.675, -0.004][:n_agents]))
            ep = run_slice(lines, 422, 427, "s = '%d,%d,%.2f,' % (step, idx_episode, reward_global)",
                           dict(vals, n_agents=n_agents, experiment="particle", np=np))["s"]
            ce = run_slice(lines, 399, 404, "s = '%d,%d,%.2f,' % (step, idx_episode, reward_global_century/float(period))",
                           dict(vals, n_agents=n_agents, np=np))["s"]
            rows.append({"in": {k2: (v.tolist() if hasattr(v, "tolist") else v) for k2, v in vals.items()},
                         "episode_row": ep, "century_row": ce})
        cases.append({"n_agents": n_agents, "header": ns["header"], "header_century": ns["header_c"], "rows": rows})

    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "alg"))
    import replay_buffer
    import replay_buffer_dual
    ring = []
    for size, chunks in ((10, (4, 5, 6, 3)), (5, (12,)), (7, (1, 1, 1, 9, 2))):
        ref = replay_buffer.Replay_Buffer(size=size)
        t = 0
        for c in chunks:
            for _ in range(c):
                ref.add(t)
                t += 1
        ring.append({"size": size, "chunks": list(chunks), "memory": [int(x) for x in ref.memory]})
    dual = []
    for n1, n2, size in ((100, 100, 20), (100, 3, 20), (4, 100, 20), (4, 5, 20), (12, 3, 20), (0, 50, 16), (50, 0, 16)):
        ref = replay_buffer_dual.Replay_Buffer(size=1000)
        if n1:
            ref.add(list(range(n1)), is_bad=True)
        if n2:
            ref.add(list(range(n1, n1 + n2)), is_bad=False)
        random.seed(0)
        r = np.asarray(ref.sample_batch(size))
        dual.append({"n_bad": n1, "n_good": n2, "size": size, "taken_bad": int((r < n1).sum()), "taken_good": int((r >= n1).sum())})
    out = {"csv": cases, "ring": ring, "dual": dual,
           "source": "alg/train_onpolicy.py:201-215,399-404,423-428 executed; alg/replay_buffer.py, alg/replay_buffer_dual.py imported"}
    path = os.path.join(ROOT, "tests", "golden", "replay_csv.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, "|", cases[1]["header"].strip(), "|", cases[1]["rows"][0]["episode_row"].strip(), "|", dual)


if __name__ == "__main__":
    main()
