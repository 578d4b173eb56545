"""Multi-GPU sharding helpers: stereo pairs are independent units, so the
path shards with NO data-path collective.  One process per GPU; torch.distributed (backend
"nccl" = RCCL on the GPU box, "gloo" in the CPU tests) is used only for the barrier and for the
max-over-ranks wall clock that bench.py reports."""
import os


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def pairs_for_rank(rank, world, pairs_per_rank):
    """Global stereo-pair indices owned by `rank`: pair k -> rank k mod N (round robin)."""
    return [i * world + rank for i in range(pairs_per_rank)]


def owner_of_pair(k, world):
    return k % world


def sequence_seed_for_rank(base_seed, rank):
    """Tracking workload: the temporal tail is a strict chain per sequence, so each rank tracks
    its own sequence (replicas only)."""
    return base_seed + rank


def max_over_ranks(seconds, dist=None, device="cpu"):
    """Wall time of the slowest rank (what throughput is computed from)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank, world, seconds):
    return units_per_rank * world / seconds
