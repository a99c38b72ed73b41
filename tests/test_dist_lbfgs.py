"""The sharded L-BFGS driver (uuo_mocap_amd/dist_lbfgs.py: shared-beta extension) on the CPU:
(1) with one rank it is torch.optim.LBFGS(strong_wolfe) evaluation for evaluation; (2) with two gloo ranks, each holding
its own block of the parameter vector plus a replicated shared tail, it reproduces the one-process solve of the joint
problem and leaves the shared parameters bit-identical on both ranks."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from uuo_mocap_amd.dist_lbfgs import DistReducer, LocalReducer, ShardedLBFGS  # noqa: E402

N_LOCAL, N_SHARED = 40, 6


def _block(rank):
    """rank r's term: a well-scaled quadratic in its own block x_r, coupled to the shared tail s through
    0.5 * c_r * sum_j (x_r[j] - s[j])^2 over the first N_SHARED entries, plus a smooth non-quadratic term."""
    i = torch.arange(N_LOCAL, dtype=torch.float32)
    a = 1.0 + 3.0 * i / (N_LOCAL - 1) + 0.5 * rank
    b = 0.2 * torch.sin(0.37 * i + rank)
    c = 0.7 + 0.2 * rank

    def f(x_r, s):
        q = 0.5 * (a * (x_r - b) ** 2).sum()
        cpl = 0.5 * c * ((x_r[:N_SHARED] - s) ** 2).sum()
        return q + cpl + 0.05 * torch.log1p(s ** 2).sum() / (1 + rank)

    return f


def _torch_reference(world, max_iter, lr):
    fs = [_block(r) for r in range(world)]
    x = torch.zeros(world * N_LOCAL + N_SHARED, requires_grad=True)
    opt = torch.optim.LBFGS([x], max_iter=max_iter, tolerance_grad=1e-7, tolerance_change=1e-9, lr=lr,
                            line_search_fn="strong_wolfe")
    losses = []

    def closure():
        opt.zero_grad()
        s = x[world * N_LOCAL:]
        loss = sum(fs[r](x[r * N_LOCAL:(r + 1) * N_LOCAL], s) for r in range(world))
        loss.backward()
        losses.append(float(loss))
        return loss

    opt.step(closure)
    return x.detach().clone(), losses, int(opt.state[opt._params[0]]["n_iter"])


def _sharded_solve(rank, world, reducer, max_iter, lr):
    f = _block(rank)
    x = torch.zeros(N_LOCAL + N_SHARED)
    losses = []

    def evaluate(xl):
        xl = xl.detach().clone().requires_grad_(True)
        loss = f(xl[:N_LOCAL], xl[N_LOCAL:])
        loss.backward()
        losses.append(float(loss))
        return float(loss), xl.grad

    st = ShardedLBFGS(x, N_SHARED, evaluate, reducer=reducer, lr=lr, max_iter=max_iter).solve()
    return x, st, losses


@pytest.mark.parametrize("lr", [1.0, 0.3])
def test_one_rank_is_torch_lbfgs(lr):
    x_ref, losses_ref, n_iter_ref = _torch_reference(1, 40, lr)
    x, st, losses = _sharded_solve(0, 1, LocalReducer(), 40, lr)
    # evaluation for evaluation until the loss stops changing in fp32; in that plateau torch's fp32 directional
    # derivatives and the driver's fp64 ones may disagree on the sign of a ~1e-9 quantity, so the very last line search
    # can take a few evaluations more or fewer
    assert st["n_iter"] == n_iter_ref and st["n_eval"] == len(losses)
    assert abs(st["n_eval"] - len(losses_ref)) <= 6, (st["n_eval"], len(losses_ref))
    head = min(len(losses), len(losses_ref))
    np.testing.assert_allclose(losses[:head], losses_ref[:head], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(x.numpy(), x_ref.numpy(), atol=2e-5)


def _rank_main(rank, world, port, max_iter, lr, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, st, losses = _sharded_solve(rank, world, DistReducer(), max_iter, lr)
        torch.save({"x": x, "st": st, "n_local_evals": len(losses)}, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lr", [1.0])
def test_two_gloo_ranks_reproduce_the_joint_solve(tmp_path, lr):
    import torch.multiprocessing as mp

    world, max_iter = 2, 40
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(world, port, max_iter, lr, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    x_ref, losses_ref, n_iter_ref = _torch_reference(world, max_iter, lr)
    # lock-step: same decisions on every rank
    assert res[0]["st"] == res[1]["st"]
    assert res[0]["n_local_evals"] == res[1]["n_local_evals"] == res[0]["st"]["n_eval"]
    # the replicated shared parameters never diverge
    assert torch.equal(res[0]["x"][N_LOCAL:], res[1]["x"][N_LOCAL:])
    # and the sharded solve is the joint solve
    # (two partial sums added in fp64 are not the one-process fp32 dot product: at the fp32 plateau of the loss the last
    # iteration or two may differ)
    assert abs(res[0]["st"]["n_iter"] - n_iter_ref) <= 2 and abs(res[0]["st"]["n_eval"] - len(losses_ref)) <= 6
    np.testing.assert_allclose(res[0]["st"]["final_loss"], min(losses_ref), rtol=1e-4, atol=1e-7)
    joint = torch.cat([res[0]["x"][:N_LOCAL], res[1]["x"][:N_LOCAL], res[0]["x"][N_LOCAL:]])
    np.testing.assert_allclose(joint.numpy(), x_ref.numpy(), atol=5e-5)


def _gather_main(rank, world, port, out_dir):
    import ctypes

    import torch.distributed as dist

    from uuo_mocap_amd._lib import GATHER_FN

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        red = DistReducer()
        got = {}

        # exactly the adapter engine._StageProblem.solve_shared hands to uuo_lbfgs_solve_shared, called the way the C driver
        # calls it: raw double pointers, several message lengths, repeatedly (the buffers are cached per length)
        def gather(user, mine, n, out):
            red.gather_array(np.ctypeslib.as_array(mine, shape=(n,)), np.ctypeslib.as_array(out, shape=(world, n)))
            return 0

        fn = GATHER_FN(gather)
        for rep in range(3):
            for n in (10, 16, 627):
                mine = (np.arange(n, dtype=np.float64) + 1000.0 * rank + 0.25 * rep)
                out = np.full((world, n), np.nan)
                rc = fn(None, mine.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n,
                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
                assert rc == 0
                got[(rep, n)] = out
        torch.save(got, os.path.join(out_dir, "gather%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_gather_hook_of_the_device_solver_over_gloo(tmp_path):
    """The uuo_gather_fn adapter (rank-ordered all_gather of float64 vectors through torch.distributed) with two gloo
    ranks: every rank receives every rank's vector, in rank order, for each of the message lengths the device solver sends
    (10 betas, 16 evaluation statistics, 627 Gram-row entries).  The C driver itself needs a GPU: tests/test_gpu_multirank.py."""
    import torch.multiprocessing as mp

    world = 2
    port = 29500 + ((os.getpid() + 7) % 2000)
    mp.spawn(_gather_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "gather%d.pt" % r), weights_only=False) for r in range(world)]
    for key, table in res[0].items():
        rep, n = key
        assert np.array_equal(table, res[1][key])
        for r in range(world):
            assert np.array_equal(table[r], np.arange(n, dtype=np.float64) + 1000.0 * r + 0.25 * rep)


def _mailbox_main(rank, world, port, out_dir):
    import ctypes

    import torch.distributed as dist

    from uuo_mocap_amd import parallel
    from uuo_mocap_amd.dist_lbfgs import ShmReducer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    assert parallel.ensure_process_group(None) == (rank, world)   # no pre-initialised group: the helper brings up gloo
    try:
        red = parallel._default_reducer(None, None, "auto")       # one host -> the shared-memory mailbox
        assert isinstance(red, ShmReducer) and (red.rank, red.world) == (rank, world)
        assert parallel._default_reducer(None, None, "auto") is red
        fn, user = red.native()                                     # what engine.solve_shared hands to the C driver
        got = {}
        for rep in range(50):
            for n in (11, 17, 628):                                 # betas, evaluation statistics, Gram rows (+ status word)
                mine = (np.arange(n, dtype=np.float64) + 1000.0 * rank + 0.25 * rep)
                out = np.full((world, n), np.nan)
                rc = fn(user, mine.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n,
                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
                assert rc == 0
                got[(rep, n)] = out
        # a long vector (frame blocks after a solve) is cut into mailbox-sized messages; lanes are independent tables
        big = np.arange(5000, dtype=np.float64) * (rank + 1)
        out = np.empty((world, 5000))
        red.gather_array(big, out)
        got["big"] = out
        lanes = red.fork(3)
        got["lane"] = lanes[2].gather([float(rank), 7.0])
        # the sharded Python driver over the mailbox: same joint solve as over gloo
        x, st, _ = _sharded_solve(rank, world, red, 40, 1.0)
        got["st"], got["x"] = st, x
        torch.save(got, os.path.join(out_dir, "mb%d.pt" % rank))
        dist.barrier()
        red.close()
    finally:
        dist.destroy_process_group()


def test_shared_memory_mailbox_two_ranks(tmp_path):
    """The node-local transport of the shared solves (csrc/mailbox.hip through dist_lbfgs.ShmReducer; no GPU involved): two
    processes, the C function pointer called the way the device driver calls it, every message length of a solve, 150
    gathers in a row (the two-slot rows must never be overwritten before both ranks have read them), a long vector, a
    forked lane, and the Python checker's joint solve over it."""
    import torch.multiprocessing as mp

    world = 2
    port = 29500 + ((os.getpid() + 13) % 2000)
    mp.spawn(_mailbox_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "mb%d.pt" % r), weights_only=False) for r in range(world)]
    for key, table in res[0].items():
        if not isinstance(key, tuple):
            continue
        rep, n = key
        assert np.array_equal(table, res[1][key])
        for r in range(world):
            assert np.array_equal(table[r], np.arange(n, dtype=np.float64) + 1000.0 * r + 0.25 * rep)
    for r in range(world):
        assert np.array_equal(res[r]["big"][1], np.arange(5000, dtype=np.float64) * 2)
        assert res[r]["lane"].tolist() == [[0.0, 7.0], [1.0, 7.0]]
    assert res[0]["st"] == res[1]["st"]
    assert torch.equal(res[0]["x"][N_LOCAL:], res[1]["x"][N_LOCAL:])
    x_ref, losses_ref, n_iter_ref = _torch_reference(world, 40, 1.0)
    joint = torch.cat([res[0]["x"][:N_LOCAL], res[1]["x"][:N_LOCAL], res[0]["x"][N_LOCAL:]])
    np.testing.assert_allclose(joint.numpy(), x_ref.numpy(), atol=5e-5)
