"""N > 1 path on the CPU: two gloo ranks shard sequences with no data-path collective (world_size 2)."""
import os
import socket

import numpy as np
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


def _hyp_rank(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from uuo_mocap_amd import parallel

        with parallel.shard_hypotheses() as shard:
            assert parallel.hypothesis_shard() is shard and shard.world == world
            mine = shard.mine(4)
            local = {i: {"angle": i, "by": rank, "payload": np.full(3, float(i))} for i in mine}
            everything = shard.exchange(local, 4)
        assert parallel.hypothesis_shard() is None
        torch.save({"mine": mine, "all": everything}, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_hypothesis_shard_exchange_two_gloo_ranks(tmp_path):
    """rank r owns hypotheses r, r + world, ...; after one all_gather_object every rank holds all of them
    in hypothesis order."""
    import torch
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000) + 1
    mp.spawn(_hyp_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(2)]
    assert res[0]["mine"] == [0, 2] and res[1]["mine"] == [1, 3]
    for r in range(2):
        assert [h["angle"] for h in res[r]["all"]] == [0, 1, 2, 3]
        assert [h["by"] for h in res[r]["all"]] == [0, 1, 0, 1]


def _frames_rank(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from uuo_mocap_amd import parallel

        with parallel.shard_frames() as fs:
            F = 7
            lo, hi = fs.block(F)
            full = torch.arange(F * 2 * 3, dtype=torch.float32).reshape(F, 2, 3) * 0.25 + 1e-3
            got = fs.gather_frames(full[lo:hi].clone(), F)
            flat = fs.gather_frames(full[lo:hi, 0, 0].clone(), F)
            torch.save({"bounds": fs.bounds(F), "block": (lo, hi), "equal": bool(torch.equal(got, full)),
                        "flat_equal": bool(torch.equal(flat, full[:, 0, 0])), "inside": parallel.frame_shard() is fs},
                       os.path.join(out_dir, "rank%d.pt" % rank))
        assert parallel.frame_shard() is None
    finally:
        dist.destroy_process_group()


def test_frame_shard_blocks_and_exchange_two_gloo_ranks(tmp_path):
    """contiguous frame blocks (sizes differ by at most one), and the exchange that gives every rank the full
    per-frame tensors again -- bit for bit."""
    from uuo_mocap_amd.dist_lbfgs import LocalReducer
    from uuo_mocap_amd.parallel import FrameShard

    class _R:  # a reducer's (rank, world) is all `bounds` needs
        def __init__(self, rank, world):
            self.rank, self.world = rank, world

    for F in (1, 7, 8, 300):
        for world in (1, 2, 3, 8):
            if F < world:
                continue
            e = FrameShard(_R(0, world)).bounds(F)
            sizes = [e[r + 1] - e[r] for r in range(world)]
            assert e[0] == 0 and e[-1] == F and max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    one = FrameShard(LocalReducer())
    t = torch.randn(5, 3)
    assert torch.equal(one.gather_frames(t, 5), t) and one.block(5) == (0, 5)

    port = 29500 + (os.getpid() % 2000) + 2
    mp.spawn(_frames_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(2)]
    assert res[0]["bounds"] == [0, 4, 7] and res[0]["block"] == (0, 4) and res[1]["block"] == (4, 7)
    for r in range(2):
        assert res[r]["equal"] and res[r]["flat_equal"] and res[r]["inside"]
