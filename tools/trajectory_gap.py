#!/usr/bin/env python
"""Where does trajectory mode lose its time to in-place stepping (C2: 2.85 vs 2.6 us per tick)?  Same kernel, same 330-tick
hipGraph, the trajectory features switched on one at a time through the strides / optional pointers of cm3_particle_traj.
Run on the GPU box: python tools/trajectory_gap.py [c2|c5]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cm3_amd  # noqa: E402
from cm3_amd import _lib  # noqa: E402
from cm3_amd.particle import VecParticleEnv  # noqa: E402

T = 330


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    if wl in ("c2", "c5"):
        name, N, E = {"c2": ("particle_stage2_antipodal", 4, 4096), "c5": ("particle_merge8", 8, 8192)}[wl]
    else:       # <n_agents> <n_envs>: the merge8 geometry with its first N agents
        name, N, E = "particle_merge8", int(sys.argv[1]), int(sys.argv[2])
        wl = "N=%d E=%d" % (N, E)
    only = [int(x) for x in os.environ.get("LEVELS", "").split(",") if x] or list(range(16))
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    lib = _lib.lib()
    env = VecParticleEnv(cm3_amd.load_config(name), N, 0.2, 33, E, device=dev, dtype=torch.float32, auto_reset=True)
    env.reset()
    env._desc.flags = _lib.FLAG_AUTO_RESET | _lib.FLAG_GEN_ACTIONS | env.kernel_flags
    L = 4 * (N - 1)
    z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)  # noqa: E731
    state, goals, obs = z(T + 1, N, E, 4), z(T + 1, N, E, 2), z(T + 1, E, N, L)
    actions, reward_n, reward = z(T, E, N, dt=torch.int32), z(T, E, N), z(T, E)
    done, coll = z(T, E, dt=torch.uint8), z(T, E, dt=torch.int32)
    term_state, term_obs = z(T, N, E, 4), z(T, E, N, L)
    slab = z(T * (E * N * 2 + E * 2 + (E + 15) // 16 * 4), dt=torch.int32)
    state[0].copy_(env._state[0]); goals[0].copy_(env._goals); obs[0].copy_(env._obs_others[0])

    def traj(level):
        t = _lib.ParticleTraj()
        t.state, t.goals, t.obs_others = state.data_ptr(), goals.data_ptr(), obs.data_ptr()
        t.actions, t.reward_n, t.reward, t.done = actions.data_ptr(), reward_n.data_ptr(), reward.data_ptr(), done.data_ptr()
        t.meta, t.episode = env._meta.data_ptr(), env._episode.data_ptr()
        if level >= 1:
            t.obs_others_stride = E * N * L * 4
        if level >= 2:
            t.state_stride = N * E * 16
        if level >= 3:
            t.goals_stride = N * E * 8
        if level >= 4:
            t.actions_stride, t.reward_n_stride, t.reward_stride, t.done_stride = E * N * 4, E * N * 4, E * 4, E
        if level >= 5:
            t.term_state, t.term_state_stride = term_state.data_ptr(), N * E * 16
            t.term_obs_others, t.term_obs_others_stride = term_obs.data_ptr(), E * N * L * 4
        if level >= 6:
            t.collisions, t.collisions_stride = coll.data_ptr(), E * 4
        if level >= 7:
            t.state_live, t.goals_live = env._state[0].data_ptr(), env._goals.data_ptr()
        if level == 8:      # the live rollout with the five small per-tick outputs packed into ONE slab per tick
            per = [E * N * 4, E * N * 4, E * 4, (E + 15) // 16 * 16, E * 4]
            off = [sum(per[:k]) for k in range(5)]
            st = sum(per)
            t.actions, t.reward_n, t.reward, t.done, t.collisions = [slab.data_ptr() + o for o in off]
            t.actions_stride = t.reward_n_stride = t.reward_stride = t.done_stride = t.collisions_stride = st
        if level == 9:      # the live rollout, per-AGENT outputs in slots, per-ENV outputs (reward, done, collisions) in place
            t.reward_stride = t.done_stride = t.collisions_stride = 0
        if level == 10:     # the live rollout, per-ENV outputs in slots, per-AGENT outputs (actions, reward_n) in place
            t.actions_stride = t.reward_n_stride = 0
        if level == 11:     # level 9 + reward slots
            t.done_stride = t.collisions_stride = 0
        if level == 12:     # level 9 + done slots
            t.reward_stride = t.collisions_stride = 0
        if level == 13:     # level 9 + collisions slots
            t.reward_stride = t.done_stride = 0
        if level == 14:     # live, no collisions pointer at all
            t.collisions, t.collisions_stride = 0, 0
        if level == 15:     # live, no terminal capture
            t.term_state = t.term_obs_others = 0
            t.term_state_stride = t.term_obs_others_stride = 0
        return t

    names = ["in place (all strides 0)", "+ obs_others slots (non-temporal from 128 MB)", "+ state slots", "+ goals slots",
             "+ actions / reward_n / reward / done slots", "+ terminal capture", "+ collisions slot (= the full trajectory, ticks chained through the slots)",
             "same, stepping in place on live state / goals + slot copies (state_live; what ParticleRollout uses)",
             "live, the five small per-tick outputs packed into one slab per tick",
             "live, per-agent outputs (actions, reward_n) in slots, per-env outputs (reward, done, collisions) in place",
             "live, per-env outputs in slots, per-agent outputs in place",
             "live, per-agent outputs + reward in slots (done, collisions in place)",
             "live, per-agent outputs + done in slots",
             "live, per-agent outputs + collisions in slots",
             "live, everything in slots but no collisions array",
             "live, everything in slots but no terminal capture"]
    graphs = []
    for level in only:
        t = traj(level)
        g = torch.cuda.CUDAGraph()
        _lib.check(lib.cm3_particle_rollout_f32(ctypes.byref(env._desc), ctypes.byref(t), T, stream.cuda_stream))
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            _lib.check(lib.cm3_particle_rollout_f32(ctypes.byref(env._desc), ctypes.byref(t), T, stream.cuda_stream))
        graphs.append((g, t))
    best = [1e9] * 16
    for rep in range(4):
        for level, (g, _) in zip(only, graphs):
            g.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(5):
                g.replay()
            b.record(stream); b.synchronize()
            best[level] = min(best[level], a.elapsed_time(b) * 1e3 / (5 * T))
    print("%s: N = %d, %d envs, us per tick (330-tick hipGraph, best of 4 alternating rounds)" % (wl, N, E))
    for level in only:
        print("   %-100s %.3f" % (names[level], best[level]))


if __name__ == "__main__":
    main()
