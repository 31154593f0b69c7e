"""GPU: the device replay buffers and the single-launch transition export (f2).

* DeviceReplayBuffer / DeviceDualReplayBuffer on cuda:0 against what the REAL reference classes hold after the same adds
  (tests/golden/replay_csv.json, recorded from alg/replay_buffer.py:11-37 and alg/replay_buffer_dual.py:40-63): slot contents,
  wrap-around, over-long batches, the dual buffer's split counts;
* filled from a real ParticleRollout.as_reference_batch(numpy=False) with `scenario.collisions != 0` of the transition's episode as
  the split flag (alg/train_onpolicy.py:356): every stored transition is found again in the rollout, in the right memory;
* cm3_transitions_gather_f32 (one launch) against the torch composition it replaces, dense / sparse goal slots, terminal capture."""
import json
import os

import pytest
import torch

from tests.helpers import GOLDEN, load_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fixture():
    return json.load(open(os.path.join(GOLDEN, "replay_csv.json")))


def _cols(start, n):
    i = torch.arange(start, start + n, device=DEV)
    return {"id": i.clone(), "x": torch.stack([i.float(), -i.float()], dim=1), "flag": (i % 2 == 0), "w": i.to(torch.float64).view(n, 1, 1).repeat(1, 3, 2)}


def test_ring_buffer_on_the_gpu_equals_recorded_reference_memory():
    from cm3_amd.replay import DeviceReplayBuffer
    for case in _fixture()["ring"]:
        ours = DeviceReplayBuffer(size=case["size"], device=DEV)
        start = 0
        for chunk in case["chunks"]:
            ours.add(_cols(start, chunk))
            start += chunk
        got = ours.all()
        assert sorted(got["id"].tolist()) == sorted(case["memory"]), case
        if max(case["chunks"]) <= case["size"]:                      # same slots too (an over-long chunk keeps the newest)
            assert got["id"].tolist() == case["memory"], case
        # every column moved with its row (16-, 8-, 4- and 1-byte row sizes)
        assert torch.equal(got["x"][:, 0], got["id"].float()) and torch.equal(got["x"][:, 1], -got["id"].float())
        assert torch.equal(got["flag"], got["id"] % 2 == 0)
        assert torch.equal(got["w"], got["id"].to(torch.float64).view(-1, 1, 1).repeat(1, 3, 2))
        if len(ours) > 3:
            b = ours.sample_batch(3, generator=torch.Generator(device=DEV).manual_seed(1))
            assert b["id"].shape == (3,) and len(set(b["id"].tolist())) == 3 and set(b["id"].tolist()) <= set(case["memory"])
            assert torch.equal(b["x"][:, 0], b["id"].float()) and torch.equal(b["flag"], b["id"] % 2 == 0)
        assert ours.sample_batch(10 ** 6)["id"].shape[0] == len(case["memory"])      # len <= size -> everything


def test_dual_buffer_on_the_gpu_equals_recorded_reference_counts():
    from cm3_amd.replay import DeviceDualReplayBuffer
    for case in _fixture()["dual"]:
        n1, n2 = case["n_bad"], case["n_good"]
        buf = DeviceDualReplayBuffer(size=1000, device=DEV)
        cols = _cols(0, n1 + n2)
        perm = torch.randperm(n1 + n2, generator=torch.Generator(device=DEV).manual_seed(n1 * 131 + n2), device=DEV)
        cols = {k: v[perm].contiguous() for k, v in cols.items()}            # bad and good transitions interleaved
        buf.add(cols, cols["id"] < n1)
        assert (len(buf.mem1), len(buf.mem2)) == (n1, n2)
        if n1:
            assert torch.equal(buf.mem1.all()["id"], cols["id"][cols["id"] < n1])     # order of arrival kept inside each memory
        if n2:
            assert torch.equal(buf.mem2.all()["id"], cols["id"][cols["id"] >= n1])
        b = buf.sample_batch(case["size"], generator=torch.Generator(device=DEV).manual_seed(0))
        if b is None:
            assert case["taken_bad"] + case["taken_good"] == 0
            continue
        assert (int((b["id"] < n1).sum()), int((b["id"] >= n1).sum())) == (case["taken_bad"], case["taken_good"]), case
        assert len(set(b["id"].tolist())) == b["id"].numel()
        assert torch.equal(b["x"][:, 0], b["id"].float())


def test_dual_buffer_wraps_like_sequential_adds():
    """A batch with more bad transitions than the memory holds, on top of earlier contents: memory_1 ends up as after one
    add_1() per bad transition in order (replay_buffer_dual.py:26-31)."""
    from cm3_amd.replay import DeviceDualReplayBuffer
    size = 7
    buf = DeviceDualReplayBuffer(size=size, device=DEV)
    mem1, idx1, mem2, idx2 = [], 0, [], 0
    start = 0
    for n in (5, 4, 23, 3):
        cols = _cols(start, n)
        bad = (cols["id"] % 3) != 1
        buf.add(cols, bad)
        for i, bflag in zip(cols["id"].tolist(), bad.tolist()):
            if bflag:
                if idx1 >= len(mem1):
                    mem1.append(i)
                else:
                    mem1[idx1] = i
                idx1 = (idx1 + 1) % size
            else:
                if idx2 >= len(mem2):
                    mem2.append(i)
                else:
                    mem2[idx2] = i
                idx2 = (idx2 + 1) % size
        start += n
        assert buf.mem1.all()["id"].tolist() == mem1 and buf.mem2.all()["id"].tolist() == mem2, n
        assert (buf.mem1.idx, buf.mem2.idx) == (idx1, idx2)


def _rollout(E=512, N=4, T=66, cfg="particle_stage2_cross.json", **kw):
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    env = VecParticleEnv(load_cfg(cfg), N, 0.2, 33, E, device=DEV, dtype=torch.float32, auto_reset=True, seed=77)
    env.reset()
    ro = ParticleRollout(env, n_ticks=T, use_graph=True, **kw)
    ro.collect()
    return env, ro


@pytest.mark.parametrize("kw", [dict(), dict(sparse_goals=True, live_state=False), dict(live_state=True)])
@pytest.mark.parametrize("N,cfg", [(4, "particle_stage2_cross.json"), (8, "particle_merge8.json"), (1, "particle_stage1.json"), (3, "particle_merge8.json")])
def test_single_launch_transition_export_equals_the_torch_composition(N, cfg, kw):
    env, ro = _rollout(E=300, N=N, T=40, cfg=cfg, **kw)
    tt, ee = ro.valid_indices()
    a = ro.as_reference_batch(tt, ee, numpy=False)                 # cm3_transitions_gather_f32
    b = ro.as_reference_batch_torch(tt, ee, numpy=False)
    assert int(ro.done.sum()) > 0 and set(a) == set(b)
    for k in b:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
        assert torch.equal(a[k], b[k]), k
    pick = torch.randperm(tt.numel(), device=DEV)[:1000]
    a = ro.as_reference_batch(tt[pick], ee[pick], numpy=False)
    b = ro.as_reference_batch_torch(tt[pick], ee[pick], numpy=False)
    for k in b:
        assert torch.equal(a[k], b[k]), k
    assert a["v_global"].data_ptr() == a["v_local"].data_ptr()     # the reference stores the same array twice (train_onpolicy.py:338)
    a = ro.as_reference_batch(numpy=False)                         # no index arrays at all: every transition, time-major
    b = ro.as_reference_batch_torch(tt, ee, numpy=False)
    for k in b:
        assert torch.equal(a[k], b[k]), k
    ro.close()


def test_replay_buffers_filled_from_a_real_rollout():
    """The trainer's use (train_onpolicy.py:338-357): every transition of a collection goes into the buffer; with the dual buffer
    the transitions of an episode go to memory_1 iff scenario.collisions != 0 at the episode's end."""
    from cm3_amd.replay import DeviceDualReplayBuffer, DeviceReplayBuffer
    env, ro = _rollout(E=256, N=4, T=99)
    cols = ro.as_reference_batch(numpy=False)
    B = cols["reward"].shape[0]
    assert B == 99 * 256
    # per-transition flag: the flag of the episode the transition belongs to = episode_is_bad() at the tick that ends it, carried
    # backwards over the episode's ticks (an episode still running at the end of the collection: not bad yet)
    bad_end, done = ro.episode_is_bad(), ro.done.bool()                      # [T, E]
    flag = torch.zeros_like(done)
    carry = torch.zeros(done.shape[1], dtype=torch.bool, device=DEV)
    for t in range(done.shape[0] - 1, -1, -1):
        carry = torch.where(done[t], bad_end[t], carry)
        flag[t] = carry
    tt, ee = ro.valid_indices()
    is_bad = flag[tt, ee]
    assert 0 < int(is_bad.sum()) < B

    ring = DeviceReplayBuffer(size=B // 2 + 5, device=DEV)                   # wraps
    ring.add(cols)
    got = ring.all()
    keep = torch.arange(B - ring.maxsize, B, device=DEV)
    pos = (keep % ring.maxsize)
    for k, v in cols.items():
        assert torch.equal(got[k][pos], v[keep]), k
    s = ring.sample_batch(128, generator=torch.Generator(device=DEV).manual_seed(3))
    assert s["v_global"].shape == (128, 4, 4) and s["done"].dtype == torch.bool
    # a sampled transition is one of the rollout's: its next state continues its state under the stored action for non-terminal rows
    assert torch.isfinite(s["obs_others_next"]).all()

    dual = DeviceDualReplayBuffer(size=B, device=DEV)
    dual.add(cols, is_bad)
    n1 = int(is_bad.sum())
    assert (len(dual.mem1), len(dual.mem2)) == (n1, B - n1)
    for k, v in cols.items():
        assert torch.equal(dual.mem1.all()[k], v[is_bad]), k
        assert torch.equal(dual.mem2.all()[k], v[~is_bad]), k
    b = dual.sample_batch(128, generator=torch.Generator(device=DEV).manual_seed(4))
    assert b["reward"].shape[0] == 128
    ro.close()


def test_reading_goals_between_collections_keeps_the_captured_graph():
    """the `goals` getter clears the "slots incomplete" flag; that flag used to double as the key of the captured
    hipGraph, so a loop that read ro.goals re-captured every collection -- and a toggled mode could replay a stale graph."""
    env, ro = _rollout(E=256, N=4, T=33, sparse_goals=True, live_state=False)
    assert ro.n_captures == 1
    for _ in range(3):
        _ = ro.goals
        assert not ro._goals_sparse
        ro.collect(reset=False)
        assert ro._goals_sparse and ro.n_captures == 1                        # same graph, slots sparse again
    ref_env, ref = _rollout(E=256, N=4, T=33, sparse_goals=False, live_state=False)
    for _ in range(3):
        ref.collect(reset=False)
    _ = ro.goals
    ro.sparse_goals = False                                                   # toggled after a goals read: must re-capture, dense
    ro.collect(reset=False)
    ref.collect(reset=False)
    assert not ro._goals_sparse and ro.n_captures == 2
    assert torch.equal(ro.goals, ref.goals) and torch.equal(ro.state, ref.state) and torch.equal(env.goals, ref_env.goals)
    ro.close()
    ref.close()


def test_on_policy_minibatches_are_one_export_of_distinct_transitions():
    """train_onpolicy.py:359-377: 24 minibatches of 128 per collection phase -- drawn together, exported by one launch; every
    minibatch holds distinct transitions and exactly the columns of the transitions it names."""
    env, ro = _rollout(E=2048, N=4, T=66)
    g = torch.Generator(device=DEV).manual_seed(8)
    mbs = list(ro.on_policy_minibatches(epochs=24, batch_size=128, generator=g))
    pos = ro.last_sample_positions
    assert len(mbs) == 24 and pos.shape == (24, 128)
    E = env.E
    for m, mb in enumerate(mbs):
        assert len(set(pos[m].tolist())) == 128
        want = ro.as_reference_batch_torch(torch.div(pos[m], E, rounding_mode="floor"), pos[m] % E, numpy=False)
        for k in want:
            assert torch.equal(mb[k], want[k]), (m, k)
    assert len(set(pos.reshape(-1).tolist())) > 24 * 128 * 0.95          # independent draws: hardly any overlap between minibatches
    one = ro.sample_batch(128, generator=g, numpy=False)
    assert one["reward"].shape == (128,)
    ro.close()
    # episode-synchronous collection: only the transitions before each env's `done` are candidates
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    env2 = VecParticleEnv(load_cfg("particle_stage2_cross.json"), 4, 0.2, 12, 700, device=DEV, dtype=torch.float32, auto_reset=False, seed=5)
    ro2 = ParticleRollout(env2, use_graph=False).collect()
    mbs = list(ro2.on_policy_minibatches(epochs=3, batch_size=64, generator=g))
    tt, ee = ro2.valid_indices()
    pos = ro2.last_sample_positions
    for m, mb in enumerate(mbs):
        want = ro2.as_reference_batch_torch(tt[pos[m]], ee[pos[m]], numpy=False)
        for k in want:
            assert torch.equal(mb[k], want[k]), (m, k)
    ro2.close()


def test_off_policy_cadence_for_particle_and_checkers():
    """train_offpolicy.py:309-356: a persistent buffer filled chunk by chunk (steps_per_train = 10 ticks, alg/config.json:58), one
    batch of 128 sampled after each; the Checkers trainer of the reference's README takes this path."""
    import numpy as np
    from cm3_amd.checkers import VecCheckersEnv
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.replay import DeviceReplayBuffer, off_policy_batches
    from cm3_amd.rollout import CheckersRollout, ParticleRollout
    g = torch.Generator(device=DEV).manual_seed(11)
    env = VecParticleEnv(load_cfg("particle_stage2_antipodal.json"), 4, 0.2, 33, 64, device=DEV, dtype=torch.float32, auto_reset=True, seed=3)
    env.reset()
    ro = ParticleRollout(env, n_ticks=10, use_graph=True)
    buf = DeviceReplayBuffer(size=2000, device=DEV)              # 640 transitions per chunk: wraps in the fourth
    seen = []
    for k, batch in enumerate(off_policy_batches(ro, buf, 5, batch_size=128, generator=g, reset=False)):
        assert batch["v_global"].shape == (128, 4, 4) and batch["actions"].dtype == torch.int32
        assert len(buf) == min(640 * (k + 1), 2000)
        seen.append(ro.as_reference_batch(numpy=False))
    # ring contents = the newest 2000 transitions, in arrival order modulo the ring
    allc = {name: torch.cat([c[name] for c in seen]) for name in seen[0]}
    keep = torch.arange(5 * 640 - 2000, 5 * 640, device=DEV)
    for name, v in allc.items():
        assert torch.equal(buf.all()[name][keep % 2000], v[keep]), name
    ro.close()
    ck = load_cfg("checkers_stage2.json")
    cenv = VecCheckersEnv(ck["init"], 2, 33, 96, device=DEV, auto_reset=True, seed=4)
    cenv.reset(np.eye(2))
    cro = CheckersRollout(cenv, n_ticks=10, use_graph=True)
    cbuf = DeviceReplayBuffer(size=100000, device=DEV)
    n = 0
    for batch in off_policy_batches(cro, cbuf, 3, batch_size=128, generator=g, goals=np.eye(2)):
        n += 960
        assert len(cbuf) == n and batch["grid"].shape[0] == 128 and batch["done"].dtype == torch.bool
        assert set(batch) == set(CheckersRollout.ORDER)
    last = cro.as_reference_batch(numpy=False)
    for name, v in last.items():
        assert torch.equal(cbuf.all()[name][n - 960:n], v), name
    cro.close()


@pytest.mark.parametrize("kw", [dict(), dict(sparse_goals=True, live_state=False), dict(live_state=True)])
def test_add_rollout_is_export_plus_add_in_one_launch(kw):
    """DeviceReplayBuffer.add_rollout (cm3_transitions_gather_f32 writing ring rows) leaves the ring exactly as
    add(as_reference_batch()) does -- chunk after chunk, across the wrap-around."""
    from cm3_amd.replay import DeviceReplayBuffer
    env, ro = _rollout(E=200, N=4, T=20, **kw)
    a, b = DeviceReplayBuffer(size=9000, device=DEV), DeviceReplayBuffer(size=9000, device=DEV)
    for chunk in range(4):                                  # 4000 transitions per chunk: the third wraps
        if chunk:
            ro.collect(reset=False)
        a.add_rollout(ro)
        b.add(ro.as_reference_batch(numpy=False))
        assert (len(a), a.idx) == (len(b), b.idx)
        for name in b.cols:
            assert torch.equal(a.all()[name], b.all()[name]), (chunk, name)
    assert a.cols["v_global"].data_ptr() == a.cols["v_local"].data_ptr()
    s1 = a.sample_batch(64, generator=torch.Generator(device=DEV).manual_seed(2))
    assert s1["v_local"].shape == (64, 4, 4) and torch.equal(s1["v_local"], s1["v_global"])
    ro.close()


def test_row_kernels_against_torch_indexing_on_random_layouts():
    """cm3_rows_scatter / cm3_rows_gather / cm3_rows_tile (csrc/batch.hip) on random column sets -- row sizes that select the 16-, 8-,
    4- and 1-byte copy units, more than 16 columns (two launches), negative (skipped) rows, ring wrap-around, and random
    (divisor, modulus, multiplier) tilings of every kind -- against plain torch indexing."""
    import ctypes
    from cm3_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(17)
    stream = torch.cuda.current_stream(DEV).cuda_stream
    dtypes = [torch.float32, torch.float64, torch.int32, torch.int64, torch.uint8, torch.bool, torch.int16]
    for trial in range(6):
        n_cols = [3, 16, 19, 1, 7, 33][trial]
        n_src, n_dst = int(torch.randint(50, 3000, (1,), generator=g)), int(torch.randint(50, 3000, (1,), generator=g))
        srcs, dsts = [], []
        for c in range(n_cols):
            dt = dtypes[int(torch.randint(0, len(dtypes), (1,), generator=g))]
            shape = tuple(int(x) for x in torch.randint(1, 6, (int(torch.randint(0, 3, (1,), generator=g)),), generator=g))
            src = (torch.rand((n_src,) + shape, generator=g) * 100).to(dt).to(DEV)
            srcs.append(src)
            dsts.append(torch.zeros((n_dst,) + shape, dtype=dt, device=DEV))
        # gather
        idx = torch.randint(0, n_src, (n_dst,), generator=g).to(DEV)
        _lib.rows_gather(list(zip(dsts, srcs)), n_dst, idx, stream)
        for d, s in zip(dsts, srcs):
            assert torch.equal(d, s[idx]), (trial, d.dtype, tuple(d.shape))
        # scatter with skipped rows (distinct destinations)
        perm = torch.randperm(n_dst, generator=g)[:min(n_src, n_dst)]
        dst_row = torch.full((n_src,), -1, dtype=torch.int64)
        dst_row[:perm.numel()] = perm
        dst_row = dst_row[torch.randperm(n_src, generator=g)].to(DEV)
        want = [d.clone() for d in dsts]
        keep = dst_row >= 0
        for w, s in zip(want, srcs):
            w[dst_row[keep]] = s[keep]
        _lib.rows_scatter(list(zip(dsts, srcs)), n_src, stream, dst_row=dst_row)
        for d, w in zip(dsts, want):
            assert torch.equal(d, w), (trial, d.dtype)
        # ring positions
        n_add = min(n_src, n_dst)
        start = int(torch.randint(0, n_dst, (1,), generator=g))
        pos = (start + torch.arange(n_add, device=DEV)) % n_dst
        want = [d.clone() for d in dsts]
        for w, s in zip(want, srcs):
            w[pos] = s[:n_add]
        _lib.rows_scatter([(d, s[:n_add].contiguous()) for d, s in zip(dsts, srcs)], n_add, stream, ring_start=start, ring_size=n_dst)
        for d, w in zip(dsts, want):
            assert torch.equal(d, w), (trial, "ring", d.dtype)
    # tiling kinds
    from cm3_amd.batch import _Tiler
    for trial in range(8):
        N = int(torch.randint(2, 9, (1,), generator=g))
        B = int(torch.randint(1, 40, (1,), generator=g))
        A = 5
        x = torch.randn(B, N, 4, generator=g).to(DEV)
        acts = torch.randint(0, A, (B, N), generator=g).to(torch.int32).to(DEV)
        done = (torch.rand(B, generator=g) < 0.3).to(DEV)
        t = _Tiler(torch.device(DEV))
        R = B * N
        rep_n = t.add(x, R * N, 4, x.dtype, _lib.TILE_COPY, terms=((N * N, 0, N), (1, N, 1)))
        rep_m_a = t.add(x, R * N * A, 4, torch.float64, _lib.TILE_F32_TO_F64, terms=((A * N, 0, 1), (1, 0, 0)))
        oh = t.add(acts, R, A, torch.int64, _lib.TILE_ONEHOT_I64)
        oth = t.add(x, R * (N - 1), 4, torch.float64, _lib.TILE_F32_TO_F64, others=(N, N * (N - 1), N - 1), shape=(R, (N - 1) * 4))
        oth_oh = t.add(acts, R * (N - 1), A, torch.float64, _lib.TILE_ONEHOT_F64, others=(N, N * (N - 1), N - 1), shape=(R, N - 1, A))
        eye = t.add(None, R * A, A, torch.float64, _lib.TILE_EYE_F64)
        nd = t.add(done.view(torch.uint8), R, 1, torch.int64, _lib.TILE_NOT_I64, terms=((N, 0, 1), (1, 0, 0)), shape=(R,))
        t.run()
        rows = x.reshape(R, 4)
        assert torch.equal(rep_n, rows.reshape(B, N, 4).repeat_interleave(N, dim=0).reshape(-1, 4))
        assert torch.equal(rep_m_a, rows.repeat_interleave(N, dim=0).repeat_interleave(A, dim=0).to(torch.float64))
        assert torch.equal(oh, torch.nn.functional.one_hot(acts.long(), A).reshape(R, A))
        idx = torch.tensor([[j for j in range(N) if j != n] for n in range(N)], device=DEV)
        assert torch.equal(oth, x[:, idx].reshape(R, (N - 1) * 4).to(torch.float64))
        assert torch.equal(oth_oh, torch.nn.functional.one_hot(acts.long(), A)[:, idx].reshape(R, N - 1, A).to(torch.float64))
        assert torch.equal(eye, torch.eye(A, dtype=torch.float64, device=DEV).repeat(R, 1))
        assert torch.equal(nd, 1 - done.repeat_interleave(N).to(torch.int64))
    # argument checks
    rc = _lib.RowCols()
    assert _lib.lib().cm3_rows_gather(ctypes.byref(rc), 4, None, stream) != 0


def _env(E, n=4, cfg="particle_stage2_cross.json", **kw):
    from cm3_amd.particle import VecParticleEnv
    return VecParticleEnv(load_cfg(cfg), n, 0.2, 33, E, device=DEV, dtype=torch.float32, auto_reset=True, seed=11, **kw)


def test_sample_batch_of_a_small_buffer_is_a_copy_and_export_checks_its_columns():
    """sample_batch() with len <= size used to hand out VIEWS of the ring, which the next add() overwrites in place (the
    reference's np.array(self.memory) is a copy); add_rollout advanced the ring before an export that could raise; export_into
    checked neither dtype nor trailing shape of caller-supplied columns."""
    from cm3_amd import Cm3Error
    from cm3_amd.replay import DeviceReplayBuffer
    from cm3_amd.rollout import ParticleRollout
    dev = "cuda:0"
    buf = DeviceReplayBuffer(size=64, device=dev)
    a = {"x": torch.arange(10, dtype=torch.float32, device=dev).reshape(10, 1), "y": torch.arange(10, dtype=torch.int32, device=dev)}
    buf.add(a)
    got = buf.sample_batch(32)
    keep = {k: v.clone() for k, v in got.items()}
    buf.add({"x": torch.full((60, 1), -1.0, device=dev), "y": torch.full((60,), -1, dtype=torch.int32, device=dev)})   # wraps: overwrites rows 0..5
    assert all(torch.equal(got[k], keep[k]) for k in got)
    # export_into: a wrong layout is refused before any launch, and the ring does not move
    env = _env(64)
    env.reset()
    ro = ParticleRollout(env, n_ticks=5, use_graph=False).collect()
    rb = DeviceReplayBuffer(size=5 * 64 * 2, device=dev)
    rb.add_rollout(ro)
    idx, ln = rb.idx, rb.len
    rb.cols["actions"] = rb.cols["actions"].to(torch.int64)             # a ring with a wrong column dtype
    with pytest.raises(Cm3Error):
        rb.add_rollout(ro)
    assert (rb.idx, rb.len) == (idx, ln)
    ro.close()


def test_off_policy_cadence_with_the_dual_buffer():
    """off_policy_batches with a DeviceDualReplayBuffer fell through to buffer.add(cols) without the flag and raised
    TypeError.  Now every transition carries the flag of its episode (scenario.collisions != 0 at the tick that ends it,
    train_onpolicy.py:356): checked against a host recomputation, and the two memories hold exactly the bad / good transitions."""
    import numpy as np
    from cm3_amd.replay import DeviceDualReplayBuffer, _transition_flags, off_policy_batches
    from cm3_amd.rollout import ParticleRollout
    env = _env(128)
    env.reset()
    ro = ParticleRollout(env, n_ticks=40, use_graph=False)
    dual = DeviceDualReplayBuffer(size=40 * 128 * 4, device="cuda:0")
    batches = list(off_policy_batches(ro, dual, 2, batch_size=64, generator=torch.Generator(device="cuda:0").manual_seed(0)))
    assert len(batches) == 2 and all(b["reward"].shape[0] == 64 for b in batches)
    flag = _transition_flags(ro).reshape(40, 128).cpu().numpy()
    coll, done = ro.collisions.cpu().numpy(), ro.done.bool().cpu().numpy()
    want = np.zeros((40, 128), bool)
    for e in range(128):
        cur = coll[39, e] != 0
        for t in range(39, -1, -1):
            if done[t, e]:
                cur = coll[t, e] != 0
            want[t, e] = cur
    assert np.array_equal(flag, want) and want.any() and (~want).any()
    assert len(dual.mem1) + len(dual.mem2) == 2 * 40 * 128
    ro.close()
