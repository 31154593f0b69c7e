"""CPU: the NumPy restatement of the build's Philox stream (oracle/philox.py)."""
import numpy as np

from oracle import philox


def test_random123_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds
    out = philox.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    out = philox.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(x) for x in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    out = philox.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in out] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_action_distribution_uniform():
    a = philox.expected_actions(12341, np.arange(50000), 0, 3, 4)
    assert a.min() == 0 and a.max() == 4
    freq = np.bincount(a.ravel(), minlength=5) / a.size
    assert np.abs(freq - 0.2).max() < 0.01


def test_reset_distribution_matches_reference_semantics():
    """One Bernoulli(prob_random) per episode shared by agents AND landmarks; uniform(-1,1) in the random
    branch; preset (+N(0,std) on agents only) otherwise (multi-goal_spread.py:75-91)."""
    cfg = dict(agents_x=[-0.9, -0.9], agents_y=[0.2, -0.2], landmarks_x=[0.9, 0.9], landmarks_y=[-0.2, 0.2],
               initial_std=0.05)
    pos, lm, rnd = philox.expected_reset(7, np.arange(40000), 1, cfg, 2, 0.2)
    assert abs(rnd.mean() - 0.2) < 0.01
    pre = ~rnd
    assert np.all(lm[pre] == np.array([[0.9, -0.2], [0.9, 0.2]]))
    noise = pos[pre] - np.array([[-0.9, 0.2], [-0.9, -0.2]])
    assert abs(noise.std() - 0.05) < 0.002 and abs(noise.mean()) < 0.002
    assert np.all(np.abs(pos[rnd]) < 1) and np.all(np.abs(lm[rnd]) < 1)
    assert abs(pos[rnd].std() - (1 / 3) ** 0.5) < 0.02


def test_two_stage_action_stream_is_uniform_and_uncorrelated_over_time():
    """The action stream of round 4 (Philox block per env + fmix32 over the episode / step counters): per-agent marginals,
    the joint distribution of CONSECUTIVE steps of one agent (25 cells), of two agents of one env, and of the same (agent, step)
    in consecutive episodes -- chi-square against uniform at the 99.9 % level."""
    E, N, T = 4096, 4, 64
    ids = np.arange(E) + 1000
    a = np.stack([philox.expected_actions(99, ids, 7, t, N) for t in range(T)])          # [T, E, N]
    b = philox.expected_actions(99, ids, 8, 5, N)

    def chi2(counts):
        exp = counts.sum() / counts.size
        return float(((counts - exp) ** 2 / exp).sum())

    assert chi2(np.bincount(a.ravel(), minlength=5)) < 18.5                                 # 4 dof
    pair_t = np.bincount((a[:-1] * 5 + a[1:]).ravel(), minlength=25)
    assert chi2(pair_t) < 51.2                                                              # 24 dof
    pair_ag = np.bincount((a[..., 0] * 5 + a[..., 1]).ravel(), minlength=25)
    assert chi2(pair_ag) < 51.2
    pair_ep = np.bincount((a[5] * 5 + b).ravel(), minlength=25)
    assert chi2(pair_ep) < 51.2
    # lag-k autocorrelation of one agent's action sequence, averaged over envs
    x = a[..., 0].astype(np.float64) - 2.0
    for lag in (1, 2, 3, 8):
        r = float((x[:-lag] * x[lag:]).mean() / x.var())
        assert abs(r) < 0.01, (lag, r)
    # a different seed / env id gives an unrelated stream
    c = philox.expected_actions(100, ids, 7, 3, N)
    assert 0.15 < float((c == a[3]).mean()) < 0.25


def test_fmix32_known_answers():
    # MurmurHash3 fmix32 reference values
    got = [int(x) for x in philox.fmix32(np.array([0, 1, 0xFFFFFFFF, 0xDEADBEEF], dtype=np.uint64))]
    assert got == [0x00000000, 0x514E28B7, 0x81F16F39, 0x0DE5C6A9]
