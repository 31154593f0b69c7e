"""Round 6: the split-float16 Checkers actor with 8 waves per 64-row workgroup against round 3's 4-wave build -- same bits, time per launch."""
import sys, os, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cm3_amd
from cm3_amd import _lib
from cm3_amd.actor import CheckersActor
from cm3_amd.checkers import VecCheckersEnv

def weights(Nc, rng):
    shapes = {"conv/Conv/weights": (3, 3, 3, 6), "conv/Conv/biases": (6,), "conv_linear/kernel": (150, 32),
              "conv_linear/bias": (32,), "branch_self/kernel": (43, 256), "branch_self/bias": (256,),
              "W_self_h2": (256, 256), "stage-2/branch_others/kernel": (2 * max(Nc - 1, 1), 256),
              "stage-2/branch_others/bias": (256,), "stage-2/W_others_h2": (256, 256), "b": (256,),
              "actor_out/kernel": (256, 5), "actor_out/bias": (5,)}
    return {k: (rng.standard_normal(v) * 0.1).astype(np.float32) for k, v in shapes.items()}

def main():
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    out = {}
    for cfg_name, Nc, E in (("checkers_stage2", 2, 8192), ("checkers_stage1", 1, 16384), ("checkers_stage2", 2, 65536)):
        cfg = cm3_amd.load_config(cfg_name)
        rng = np.random.default_rng(0)
        env = VecCheckersEnv(cfg["init"], Nc, 33, E, device=dev)
        goals = np.eye(2) if Nc > 1 else np.array([[1, 0]])
        env.reset(goals)
        for _ in range(7):
            env.step(torch.as_tensor(rng.integers(0, 5, (E, Nc))))
        actor = CheckersActor(weights(Nc, rng), Nc, stage=2 if Nc > 1 else 1, device=dev, precision="f16x3")
        res = {}
        for waves in (4, 8):
            _lib.check(lib.cm3_actor_checkers_force_waves(waves))
            a, p = actor.act(env, 0.1, return_probs=True)
            torch.cuda.synchronize()
            res[waves] = (a.clone(), p.clone())
            for _ in range(5):
                actor.act(env, 0.1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            acts = torch.empty(E, Nc, dtype=torch.int32, device=dev)
            s = env._slots[env._cur]
            def enq(st):
                for _ in range(50):
                    actor.enqueue(E, s["obs_self_t_raw"], env.obst_stride, s["obs_self_v"], s["obs_others"], env._goals, None,
                                  env._steps, env._episode, acts, 0.1, stream=st, env_id_base=0)
            g = _lib.capture_graph(dev, enq)
            sh = _lib.current_stream_handle(dev)
            for _ in range(2):
                _lib.check(lib.cm3_graph_launch(g, sh))
            torch.cuda.synchronize()
            e0.record()
            for _ in range(4):
                _lib.check(lib.cm3_graph_launch(g, sh))
            e1.record(); e1.synchronize()
            out["%s_E%d_waves%d_us" % (cfg_name, E, waves)] = e0.elapsed_time(e1) * 1e3 / 200
            lib.cm3_graph_destroy(g)
        same = bool(torch.equal(res[4][0], res[8][0]) and torch.equal(res[4][1], res[8][1]))
        out["%s_E%d_bit_identical" % (cfg_name, E)] = same
        # soak: 200 launches of each, all equal to the first
        bad = 0
        for waves in (8,):
            _lib.check(lib.cm3_actor_checkers_force_waves(waves))
            for _ in range(200):
                a, p = actor.act(env, 0.1, return_probs=True)
                if not (torch.equal(a, res[4][0]) and torch.equal(p, res[4][1])):
                    bad += 1
        out["%s_E%d_soak_mismatches" % (cfg_name, E)] = bad
    _lib.check(lib.cm3_actor_checkers_force_waves(8))
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
