"""N > 1 path on the CPU: two gloo ranks shard sequences with no data-path collective (world_size 2)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from uuo_mocap_amd.parallel import fit_sharded, shard_indices


def test_shard_indices_partition():
    for n in (0, 1, 7, 8, 9):
        for world in (1, 2, 4, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def fit_fn(sid):  # stands in for one multimodal_video_mocap call: per-sequence, own parameters
            torch.manual_seed(sid)
            return {"seq": sid, "rank": rank, "betas": torch.randn(10).tolist()}

        results, elapsed = fit_sharded(list(range(5)), fit_fn)
        if rank == 0:
            out.put((results, elapsed))
        else:
            assert results is None and elapsed > 0
    finally:
        dist.destroy_process_group()


def test_two_rank_sequence_sharding_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results, elapsed = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(results) == [0, 1, 2, 3, 4]
    assert [results[i]["rank"] for i in range(5)] == [0, 1, 0, 1, 0]
    for sid in range(5):  # independent of which rank fitted it
        torch.manual_seed(sid)
        assert results[sid]["betas"] == torch.randn(10).tolist()
    assert elapsed > 0
