"""CPU: the multi-GPU plumbing with world_size 2 on the gloo backend -- shard ranges, shard-invariant RNG
keying, and the moments all-gather behind advantage normalisation (the path's only collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cm3_amd.shard import (gather_moments, global_moments, local_moments, normalize_advantages, returns_to_go,
                            shard_range)
from oracle import philox


def test_shard_ranges_partition_the_envs():
    for n, w in [(32768, 8), (65536, 8), (4096, 1), (10, 4), (7, 8)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (b0, c0), (b1, _) in zip(spans, spans[1:]):
            assert b0 + c0 == b1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_rng_keyed_by_global_env_id_is_shard_invariant():
    """What rank r draws for its block equals the corresponding slice of the single-rank draw."""
    cfg = dict(agents_x=[-0.9, -0.9], agents_y=[0.2, -0.2], landmarks_x=[0.9, 0.9], landmarks_y=[-0.2, 0.2],
               initial_std=0.05)
    n, w = 1000, 8
    full_pos, full_lm, _ = philox.expected_reset(12341, np.arange(n), 3, cfg, 2, 0.2)
    full_act = philox.expected_actions(12341, np.arange(n), 3, 17, 2)
    for r in range(w):
        base, cnt = shard_range(n, r, w)
        ids = base + np.arange(cnt)
        pos, lm, _ = philox.expected_reset(12341, ids, 3, cfg, 2, 0.2)
        assert np.array_equal(pos, full_pos[base:base + cnt]) and np.array_equal(lm, full_lm[base:base + cnt])
        assert np.array_equal(philox.expected_actions(12341, ids, 3, 17, 2), full_act[base:base + cnt])


def test_returns_to_go():
    r = torch.tensor([[1.0, 1.0], [2.0, 2.0], [3.0, 3.0]])
    d = torch.tensor([[0, 0], [0, 1], [1, 0]], dtype=torch.uint8)
    g = returns_to_go(r, d, gamma=0.5)
    assert torch.allclose(g[:, 0], torch.tensor([1 + 0.5 * (2 + 0.5 * 3), 2 + 0.5 * 3, 3.0]))
    assert torch.allclose(g[:, 1], torch.tensor([1 + 0.5 * 2, 2.0, 3.0]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        adv_all = torch.randn(33, n_global, 4, generator=g, dtype=torch.float64) * 3 + 1.5
        valid_all = torch.rand(33, n_global, generator=g) > 0.1
        base, cnt = shard_range(n_global, rank, world)
        adv, valid = adv_all[:, base:base + cnt], valid_all[:, base:base + cnt]
        mean, std, count = global_moments(adv, valid)
        norm = normalize_advantages(adv, valid)
        # single-process reference over the whole batch
        sel = adv_all[valid_all.unsqueeze(-1).expand_as(adv_all)]
        ok = (abs(float(mean) - float(sel.mean())) < 1e-12 and abs(float(std) - float(sel.std(unbiased=False))) < 1e-12
              and int(count) == sel.numel())
        want = (adv - sel.mean()) / (sel.std(unbiased=False) + 1e-8)
        want = torch.where(valid.unsqueeze(-1).expand_as(want), want, torch.zeros_like(want))
        ok = ok and bool(torch.allclose(norm, want, atol=1e-12))
        # gather_moments is THE collective of the product path (normalized_returns hands its result to cm3_normalize_*
        # as `parts`, n_parts): rank-ordered triples, identical on every rank
        mine = local_moments(adv, valid)
        parts, n_parts = gather_moments(mine)
        ok = ok and n_parts == world and parts.shape == (3 * world,) and bool(torch.equal(parts[3 * rank:3 * rank + 3], mine))
        for r in range(world):
            b, c = shard_range(n_global, r, world)
            ok = ok and bool(torch.equal(parts[3 * r:3 * r + 3], local_moments(adv_all[:, b:b + c], valid_all[:, b:b + c])))
        # every rank must hold bit-identical statistics
        both = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, torch.stack([mean, std]))
        ok = ok and all(torch.equal(both[0], b) for b in both)
        out[rank] = 1 if ok else 0
    finally:
        dist.destroy_process_group()


def test_moments_all_gather_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), 37, out), nprocs=world, join=True)
    assert dict(out) == {0: 1, 1: 1}


def _segments_worker(rank, world, port, n_global, K, out):
    """The collection phase's collective at `world` ranks: K rollouts' moment triples travel in ONE all-gather of 24 K bytes per
    rank and come back as [world][K][3] in rank order -- the layout cm3_normalize_segments_* sums over."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1)
        T = 33
        adv_all = torch.randn(K, T, n_global, 4, generator=g, dtype=torch.float64) * 2 - 0.5
        spans = [shard_range(n_global, r, world) for r in range(world)]
        base, cnt = spans[rank]
        mine = torch.stack([local_moments(adv_all[k, :, base:base + cnt]) for k in range(K)])          # [K, 3]
        parts, n_parts = gather_moments(mine)
        ok = n_parts == world and parts.shape == (world * K * 3,)
        parts = parts.view(world, K, 3)
        for r, (b, c) in enumerate(spans):
            for k in range(K):
                ok = ok and bool(torch.equal(parts[r, k], local_moments(adv_all[k, :, b:b + c])))
        tot = parts[0].clone()
        for r in range(1, world):                     # rank order, as the device kernel adds them
            tot = tot + parts[r]
        mean = tot[:, 0] / tot[:, 2]
        std = (tot[:, 1] / tot[:, 2] - mean * mean).clamp(min=0).sqrt()
        for k in range(K):
            ok = ok and abs(float(mean[k]) - float(adv_all[k].mean())) < 1e-12
            ok = ok and abs(float(std[k]) - float(adv_all[k].std(unbiased=False))) < 1e-11
            ok = ok and int(tot[k, 2]) == adv_all[k].numel()
        every = [torch.zeros(K, 2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, torch.stack([mean, std], dim=1))
        ok = ok and all(torch.equal(every[0], e) for e in every)          # bit-identical statistics on every rank
        out[rank] = 1 if ok else 0
    finally:
        dist.destroy_process_group()


def test_segment_moments_all_gather_world8_gloo():
    """the C4 phase's collective at the node's real rank count (8), K = 10 rollouts per phase, uneven shards."""
    world, K = 8, 10
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_segments_worker, args=(world, _free_port(), 37, K, out), nprocs=world, join=True)
    assert dict(out) == {r: 1 for r in range(world)}


def test_single_process_path_needs_no_process_group():
    x = torch.arange(12.0).view(3, 4)
    mean, std, n = global_moments(x)
    assert float(mean) == 5.5 and int(n) == 12
    z = normalize_advantages(x)
    assert abs(float(z.mean())) < 1e-6
