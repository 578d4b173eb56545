"""N > 1 path on CPU: two gloo processes exercise the sharding rule and the max-over-ranks clock
that bench.py uses under torchrun (no GPU, no data-path collective)."""
import importlib
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    import svo_loader
    svo_loader.load()
    shard = importlib.import_module("stereo_semantic_vo_amd.shard")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.pairs_for_rank(rank, world, 8)
    dist.barrier()
    slow = shard.max_over_ranks(1.0 + rank, dist)          # rank 1 is the slow one
    # every rank learns the whole assignment only for the test (the product never gathers)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, slow, gathered))


def test_two_rank_sharding_and_clock():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, s0, g0), (r1, m1, s1, g1) = res
    assert m0 == [0, 2, 4, 6, 8, 10, 12, 14] and m1 == [1, 3, 5, 7, 9, 11, 13, 15]
    assert s0 == s1 == 2.0                                   # the slowest rank's clock everywhere
    assert sorted(g0[0] + g0[1]) == list(range(16))          # disjoint, complete cover of the pairs


def test_sharding_rule(pkg):
    shard = importlib.import_module("stereo_semantic_vo_amd.shard")
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            own = shard.pairs_for_rank(r, world, 5)
            assert all(shard.owner_of_pair(k, world) == r for k in own)
            seen += own
        assert sorted(seen) == list(range(5 * world))
    assert shard.whole_job_rate(128 * 20, 8, 2.0) == 128 * 20 * 8 / 2.0
    assert shard.sequence_seed_for_rank(100, 3) == 103


def test_bench_refuses_a_launcher_with_another_rank_count():
    """bench.py --gpus 4 under a launcher that started 2 ranks must exit non-zero BEFORE touching a GPU instead of printing
    a line that misstates n_gpus (runs without a GPU: the check comes first)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--frames", "8"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)
    assert not any(l.startswith("{") for l in out.stdout.splitlines())
