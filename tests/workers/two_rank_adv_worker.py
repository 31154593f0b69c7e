"""Worker of tests/test_gpu_multirank.py: one rank of a world-size-2 run of the PRODUCT advantage-normalisation path
(cm3_returns_moments_* -> gather_moments -> cm3_normalize_* with n_parts = 2) on a box with ONE GPU: both ranks use
cuda:0, the process group is gloo (RCCL refuses two ranks on one device), so the 24-byte all-gather is staged through the
host by cm3_amd.shard.gather_moments -- every HIP kernel of the path runs exactly as under RCCL.

    python -m torch.distributed.run --nproc-per-node 2 ... two_rank_adv_worker.py OUT_DIR E_PER_RANK
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import cm3_amd
    from cm3_amd.particle import VecParticleEnv
    from cm3_amd.rollout import ParticleRollout
    from cm3_amd.shard import normalized_returns, shard_range
    out_dir, e_total = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    base, count = shard_range(e_total, rank, world)
    cfg = cm3_amd.load_config("particle_stage2_cross")
    env = VecParticleEnv(cfg, 4, 0.2, 33, count, device="cuda:0", auto_reset=True, env_id_base=base, seed=12341)
    env.reset()
    if len(sys.argv) > 3:
        # the collector's own multi-rank path: K rollouts of 11 ticks in one hipGraph (graph ends at the K moment triples), ONE
        # all-gather of K x 3 float64 per rank, cm3_normalize_segments_* over [world][K][3]
        K = int(sys.argv[3])
        ro = ParticleRollout(env, n_ticks=11 * K, use_graph=True)
        for rep in range(2):          # the second call replays the captured graph
            out, (mean, std, cnt) = ro.collect_normalized(gamma=0.99, segments=K)
        torch.cuda.synchronize()
        torch.save({"reward_n": ro.reward_n.cpu(), "done": ro.done.cpu(), "norm": out.cpu(), "mean": mean.cpu(), "std": std.cpu(),
                    "count": cnt.cpu(), "moments": ro._norm.moments.cpu(), "base": base, "n": count},
                   os.path.join(out_dir, "rank%d.pt" % rank))
        dist.barrier()
        dist.destroy_process_group()
        return
    ro = ParticleRollout(env, n_ticks=33, use_graph=True).collect(reset=False)
    out, (mean, std, cnt) = normalized_returns(ro.reward_n, ro.done, None, gamma=0.99)
    torch.cuda.synchronize()
    torch.save({"reward_n": ro.reward_n.cpu(), "done": ro.done.cpu(), "norm": out.cpu(), "mean": float(mean),
                "std": float(std), "count": float(cnt), "base": base, "n": count},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
