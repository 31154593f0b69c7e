"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the build's own counter-based RNG
(cm3_amd/csrc/philox.h: Philox4x32-10, Salmon et al. SC'11; since round 4 the per-tick action stream is a
Philox block per (env, call) followed by a MurmurHash3-finaliser mix of the episode / step counters) and of
the draws the kernels make with it.  This has NO counterpart in the reference (which uses global MT19937 streams,
multi-goal_spread.py:75-91, train_onpolicy.py:307): it pins the build-defined stream so tests can
check in-kernel random actions / resets exactly and prove shard-invariance.

Known-answer check: Random123's published vector for Philox4x32-10, counter = key = 0 ->
6627e8d5 e169c58d bc57ac4c 9b00dbd8 (tests/test_philox.py).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
PURPOSE_ACTION = 0x00000000
PURPOSE_RESET = 0x80000000
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c = [np.asarray(x, dtype=np.uint64) & MASK32 for x in (c0, c1, c2, c3)]
    c = list(np.broadcast_arrays(*c))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[0]
            p1 = M1 * c[2]
            hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
            hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
            c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return [x.astype(np.uint32) for x in c]


def _split(v):
    v = np.asarray(v, dtype=np.uint64)
    return v & MASK32, v >> np.uint64(32)


def fmix32(h):
    """MurmurHash3's 32-bit finaliser (a bijection with full avalanche), on uint32 arrays."""
    h = np.asarray(h, dtype=np.uint64) & MASK32
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & MASK32
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & MASK32
    h = h ^ (h >> np.uint64(16))
    return h.astype(np.uint32)


def action_block(seed, env, call):
    """stage 1 of the action stream (csrc/philox.h): Philox4x32-10 over (env id, call), keyed by the seed."""
    lo, hi = _split(env)
    c3 = np.uint64(PURPOSE_ACTION | (int(call) << 24))
    return philox4x32_10(lo, hi, np.uint64(0), c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def action_word(a, episode, step):
    """stage 2: fmix32((a ^ step) + episode * 0x9E3779B1) in uint32 arithmetic."""
    a = np.asarray(a, dtype=np.uint64) & MASK32
    st = np.asarray(step, dtype=np.uint64) & MASK32
    ep = np.asarray(episode, dtype=np.uint64) & MASK32
    return fmix32(((a ^ st) + ((ep * np.uint64(0x9E3779B1)) & MASK32)) & MASK32)


def action_words(seed, env, episode, step, call):
    return [action_word(w, episode, step) for w in action_block(seed, env, call)]


def action_words_direct(seed, env, episode, step, call):
    """the ONE-stage draw the Checkers step kernel used until round 4 (no longer in csrc/philox.h): a Philox block over
    (env, episode, step | call)."""
    lo, hi = _split(env)
    w = PURPOSE_ACTION | (int(call) << 24)
    c3 = (np.asarray(step, dtype=np.uint64) & np.uint64(0x00FFFFFF)) | np.uint64(w)
    return philox4x32_10(lo, hi, np.asarray(episode, np.uint64), c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def reset_words(seed, env, episode, call):
    lo, hi = _split(env)
    c3 = np.uint64(PURPOSE_RESET | int(call))
    return philox4x32_10(lo, hi, np.asarray(episode, np.uint64), c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def u01(r):
    return (r.astype(np.float64) + 0.5) * (1.0 / 4294967296.0)


def rand5(r):
    return ((r.astype(np.uint64) * np.uint64(5)) >> np.uint64(32)).astype(np.int64)


def expected_actions(seed, env_ids, episode, step, n_agents, checkers=False):
    """int64 [E, N]: what CM3_FLAG_GEN_ACTIONS writes for (env, episode, step) -- the two-stage stream of csrc/philox.h, the same for
    the particle and (since round 5 / ABI 6) the Checkers kernels; checkers=True is accepted for old callers and changes nothing
    (action_words_direct, the one-stage draw the Checkers kernel used until round 4, stays here as a restatement only)."""
    env_ids = np.asarray(env_ids)
    out = np.zeros((env_ids.shape[0], n_agents), np.int64)
    for call in range((n_agents + 3) // 4):
        w = action_words(seed, env_ids, episode, step, call)
        for k in range(4):
            i = 4 * call + k
            if i < n_agents:
                out[:, i] = rand5(w[k])
    return out


def expected_reset(seed, env_ids, episode, config, n_agents, prob_random):
    """(pos [E,N,2], landmarks [E,N,2], is_random [E]) in float64 -- init_episode() of particle.hip.
    cos/sin/log/sqrt go through libm here and ocml on the device: compare with ~1e-12 tolerance."""
    env_ids = np.asarray(env_ids)
    E, N = env_ids.shape[0], n_agents
    w0 = reset_words(seed, env_ids, episode, 0)
    rnd = u01(w0[0]) < prob_random
    pos = np.zeros((E, N, 2))
    lm = np.zeros((E, N, 2))
    std = float(config['initial_std'])
    for i in range(N):
        a = reset_words(seed, env_ids, episode, 1 + i)
        ux, uy = 2.0 * u01(a[0]) - 1.0, 2.0 * u01(a[1]) - 1.0
        px = np.full(E, float(config['agents_x'][i]))
        py = np.full(E, float(config['agents_y'][i]))
        if std != 0.0:
            rad = np.sqrt(-2.0 * np.log(u01(a[2])))
            ang = 6.283185307179586476925286766559 * u01(a[3])
            px = px + std * (rad * np.cos(ang))
            py = py + std * (rad * np.sin(ang))
        pos[:, i, 0] = np.where(rnd, ux, px)
        pos[:, i, 1] = np.where(rnd, uy, py)
        l = reset_words(seed, env_ids, episode, 1 + N + i)
        lm[:, i, 0] = np.where(rnd, 2.0 * u01(l[0]) - 1.0, float(config['landmarks_x'][i]))
        lm[:, i, 1] = np.where(rnd, 2.0 * u01(l[1]) - 1.0, float(config['landmarks_y'][i]))
    return pos, lm, rnd
