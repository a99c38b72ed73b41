"""Two ranks on the GPU box (gloo rendezvous, both on cuda:0: RCCL refuses two ranks on one device; the collectives of
this path are < 1 KB host-side exchanges, so the backend does not change what is tested): the three ways the fit spreads
over ranks with REAL fits -- sequences per rank, yaw hypotheses per rank, shared betas across ranks."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F, M, ITERS = 20, 12, 25


def _setup():
    from uuo_mocap_amd.body_model import synthetic_smpl
    from uuo_mocap_amd.config import packaged_config
    from uuo_mocap_amd.smpl import SmplInference

    dev = torch.device("cuda:0")
    tables = synthetic_smpl(0)
    cfg = packaged_config("video_mocap")
    for k in ("part", "chamfer", "marker"):
        cfg["stages"][k]["num_iters"] = ITERS
    return dev, tables, cfg, SmplInference(dev, tables=tables)


def _fit(smpl, seq, cfg, dev):
    import contextlib
    import io

    from uuo_mocap_amd.multimodal import last_run_stats, multimodal_video_mocap

    with contextlib.redirect_stdout(io.StringIO()):
        out = multimodal_video_mocap(seq.img_smpl, copy.deepcopy(seq.markers), dev, cfg, offset=0, print_options=[],
                                     save_stages=False, smpl_inference=smpl)
    res = {k: np.asarray(out[k]) for k in ("trans", "pose_body", "betas", "root_orient")}
    res["yaw_scores"] = np.asarray(last_run_stats()["yaw_scores"])
    res["n_chamfer"] = len(last_run_stats()["chamfer"])
    for key in ("chamfer", "marker", "marker_final"):
        res[key + "_first"] = [float(s_["first_loss"]) for s_ in last_run_stats().get(key, [])]
        res[key + "_final"] = [float(s_["final_loss"]) for s_ in last_run_stats().get(key, [])]
        res[key + "_driver"] = [str(s_.get("driver", "")) for s_ in last_run_stats().get(key, [])]
    return res


def _rank_main(rank, world, port, out_dir):
    import torch.distributed as dist

    import faulthandler

    faulthandler.enable()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from uuo_mocap_amd import parallel
        from uuo_mocap_amd.synthetic import make_sequence

        dev, tables, cfg, smpl = _setup()
        seqs = [make_sequence(tables, seed=40 + i, num_frames=F, num_markers=M) for i in range(4)]
        out = {}
        # (1) sequences per rank: real fits, gathered on rank 0
        merged, elapsed = parallel.fit_sharded(list(range(4)), lambda sid: _fit(smpl, seqs[sid], cfg, dev), device=dev)
        out["sharded"] = merged
        out["sharded_elapsed"] = elapsed
        # (2) yaw hypotheses per rank (8e.2): ONE sequence, every rank ends with the full, identical result
        with parallel.shard_hypotheses():
            out["hyp"] = _fit(smpl, seqs[0], cfg, dev)
        # (3) shared betas (8e.3 / 8e.4, extension): rank r fits sequence r of the same subject, one shape vector
        # (every rank keeps its OWN HMR shape estimate: the solves have to bring the replicas of the shared vector together)
        same_subject = [make_sequence(tables, seed=70 + r, num_frames=F, num_markers=M) for r in range(world)]
        with parallel.shared_betas(device=dev) as red:
            out["shared"] = _fit(smpl, same_subject[rank], cfg, dev)
            out["shared_world"] = red.world
            # (4) one chamfer-stage problem per rank as a joint solve: the device driver (product) and the Python checker
            from uuo_mocap_amd.engine import ChamferProblem

            sq = same_subject[rank]
            markers = torch.from_numpy(sq.markers.get_points()).float().to(dev)
            o_betas = (sq.img_smpl.betas.sum(0, keepdim=True) / sq.img_smpl.img_mask.sum()).to(dev)
            prob = ChamferProblem(smpl, markers, sq.img_smpl.pose_body.to(dev), o_betas, sq.img_smpl.root_orient.to(dev), cfg)
            x0 = prob.pack(torch.median(markers, dim=1)[0], torch.zeros(F, 1, 1, device=dev), o_betas,
                           sq.img_smpl.pose_body.to(dev))
            xa, xb = x0.clone(), x0.clone()
            losses = []
            sa = prob.solve_shared(xa, red, max_iter=15, lr=0.1, callback=lambda i, l: losses.append(l))
            xb[4 * F:4 * F + 10] = xa.new_tensor(red.gather(x0[4 * F:4 * F + 10].double().cpu().tolist())[0])  # rank 0's start
            sb = prob.solve_shared_reference(xb, red, max_iter=15, lr=0.1)
            out["joint"] = {"device": sa, "checker": sb, "losses": losses, "betas_device": xa[4 * F:4 * F + 10].cpu().numpy(),
                            "betas_checker": xb[4 * F:4 * F + 10].cpu().numpy(),
                            "x_diff": float((xa - xb).abs().max()), "start_betas": x0[4 * F:4 * F + 10].cpu().numpy()}
        # (5) frame blocks per rank (8e.3): ONE sequence, every chamfer / marker solve is a joint problem over the ranks' blocks
        with parallel.shard_frames(device=dev) as fs:
            out["frames"] = _fit(smpl, seqs[0], cfg, dev)
            out["frames_block"] = fs.block(F)
        # (6) the same two collective modes with one process group (lane) per yaw hypothesis: the hypotheses run concurrently
        # on their threads, every lane ordered by its own thread -- the same solves, so the same results bit for bit
        with parallel.shard_frames(device=dev, lanes=4):
            out["frames_lanes"] = _fit(smpl, seqs[0], cfg, dev)
        with parallel.shared_betas(device=dev, lanes=4) as red4:
            out["shared_lanes"] = _fit(smpl, same_subject[rank], cfg, dev)
            out["shared_transport"] = type(red4).__name__
        # (6b) the transports carry the same rank-ordered tables: the node-local mailbox (the default on one host) and
        # torch.distributed's all_gather on the group itself give the same solves bit for bit
        with parallel.shared_betas(device=dev, lanes=4, transport="gloo") as redg:
            out["shared_lanes_gloo"] = _fit(smpl, same_subject[rank], cfg, dev)
            out["shared_transport_gloo"] = type(redg).__name__
        # (7) the batch runner with all ranks on every sequence (--rank_mode frames): rank 0 writes, both take part
        from uuo_mocap_amd import runner
        from uuo_mocap_amd.config import CONFIG_DIR

        root = os.path.join(out_dir, "data")
        d = os.path.join(root, "moyo_val", "mocap", "subj")
        if rank == 0:
            os.makedirs(d, exist_ok=True)
            runner.write_sequence_npz(os.path.join(d, "seq0.npz"), seqs[1].markers.get_points(), 30.0,
                                      seqs[1].img_smpl.pose_body, seqs[1].img_smpl.root_orient, seqs[1].img_smpl.betas)
            with open(os.path.join(out_dir, "cfg.yaml"), "w") as fh:
                fh.write("parent: %s\nname: unit\nstages:\n  part:\n    num_iters: 6\n  chamfer:\n    num_iters: 6\n"
                         "  marker:\n    num_iters: 6\n" % os.path.join(CONFIG_DIR, "video_mocap.yaml"))
        dist.barrier()
        args = runner.build_parser().parse_args(["--config", os.path.join(out_dir, "cfg.yaml"), "--dataset", "moyo_val",
                                                 "--input_dir", root, "--gpu", "0", "--rank_mode", "frames",
                                                 "--print_options"])
        import contextlib
        import io

        with contextlib.redirect_stdout(io.StringIO()):
            out["runner_written"] = runner.run(args)
        dist.barrier()
        out["runner_file"] = os.path.isfile(os.path.join(root, "moyo_val", "results", "unit", "subj", "seq0_stageii.npz"))
        torch.save(out, os.path.join(out_dir, "rank%d.pt" % rank))
    except BaseException:
        import traceback

        with open(os.path.join(out_dir, "rank%d.err" % rank), "w") as fh:
            traceback.print_exc(file=fh)
        raise
    finally:
        dist.destroy_process_group()


def test_two_ranks_sequences_hypotheses_and_shared_betas(tmp_path):
    import torch.multiprocessing as mp

    world = 2
    port = 29500 + (os.getpid() % 2000)
    try:
        mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    except Exception:
        for r in range(world):  # every rank's own traceback, not only the first one mp.spawn saw
            fn = os.path.join(str(tmp_path), "rank%d.err" % r)
            if os.path.isfile(fn):
                print("---- rank %d ----\n%s" % (r, open(fn).read()))
        raise
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]

    # the single-process answers
    from uuo_mocap_amd.synthetic import make_sequence

    dev, tables, cfg, smpl = _setup()
    seqs = [make_sequence(tables, seed=40 + i, num_frames=F, num_markers=M) for i in range(4)]
    alone = [_fit(smpl, sq, cfg, dev) for sq in seqs]

    # (1) every sequence was fitted by exactly one rank, rank 0 holds them all, each is the fit it is alone
    assert res[1]["sharded"] is None and sorted(res[0]["sharded"].keys()) == [0, 1, 2, 3]
    for sid in range(4):
        for k in ("trans", "pose_body", "betas", "root_orient"):
            assert np.array_equal(res[0]["sharded"][sid][k], alone[sid][k]), (sid, k)
    assert res[0]["sharded_elapsed"] == res[1]["sharded_elapsed"]  # max over ranks, identical on both

    # (2) hypotheses per rank: both ranks hold the whole result and it is the one-process fit, bit for bit
    for r in range(world):
        assert res[r]["hyp"]["n_chamfer"] == 4
        np.testing.assert_array_equal(res[r]["hyp"]["yaw_scores"], alone[0]["yaw_scores"])
        for k in ("trans", "pose_body", "betas", "root_orient"):
            assert np.array_equal(res[r]["hyp"][k], alone[0][k]), (r, k)

    # (3) shared betas: one shape vector, bit-identical on both ranks, and sensible fits
    assert res[0]["shared_world"] == 2
    assert np.array_equal(res[0]["shared"]["betas"][0], res[1]["shared"]["betas"][0])
    assert not np.array_equal(res[0]["shared"]["trans"], res[1]["shared"]["trans"])
    for r in range(world):
        assert np.isfinite(res[r]["shared"]["pose_body"]).all()
    np.testing.assert_array_equal(res[0]["shared"]["yaw_scores"], res[1]["shared"]["yaw_scores"])  # rank-summed: one winner
    assert res[0]["shared"]["yaw_scores"].min() < 1e-2   # the best hypothesis hugs the markers (m^2, summed over 2 ranks)

    # (4) the joint chamfer solve on the device driver: the ranks started from different shape estimates and end with ONE,
    # they took the same decisions (same counts, same joint losses evaluation by evaluation), and the solve is the checker's
    j0, j1 = res[0]["joint"], res[1]["joint"]
    assert not np.array_equal(j0["start_betas"], j1["start_betas"])
    assert np.array_equal(j0["betas_device"], j1["betas_device"])
    assert j0["device"]["n_iter"] == j1["device"]["n_iter"] and j0["device"]["n_eval"] == j1["device"]["n_eval"]
    assert j0["losses"] == j1["losses"] and len(j0["losses"]) == j0["device"]["n_eval"]
    assert j0["device"]["driver"].startswith("device-lbfgs(shared betas, world=2")
    for j in (j0, j1):
        d_, c_ = j["device"], j["checker"]
        print("joint solve: device", d_, "checker", c_, "max |x_device - x_checker| %.2e" % j["x_diff"])
        assert d_["first_loss"] == pytest.approx(c_["first_loss"], rel=1e-6)
        assert d_["final_loss"] == pytest.approx(c_["final_loss"], rel=2e-2)
        assert abs(d_["n_iter"] - c_["n_iter"]) <= 1 and abs(d_["n_eval"] - c_["n_eval"]) <= 3
        assert np.abs(j["betas_device"] - j["betas_checker"]).max() < 2e-2
    assert j0["device"]["final_loss"] < 0.7 * j0["device"]["first_loss"]

    # (5) frame blocks per rank: both ranks hold the whole result, identical; every solve ran as a joint problem whose FIRST
    # evaluation is the one-process objective (global normalisers; only the summation order differs); after 25 iterations the
    # fit is where the one-process fit is, up to the drift of two fp32 trajectories
    assert res[0]["frames_block"] == (0, F // 2) and res[1]["frames_block"] == (F // 2, F)
    for k in ("trans", "pose_body", "betas", "root_orient", "yaw_scores"):
        assert np.array_equal(res[0]["frames"][k], res[1]["frames"][k]), k
    fr = res[0]["frames"]
    assert all("frame blocks, world=2" in d for d in fr["chamfer_driver"] + fr["marker_driver"] + fr["marker_final_driver"])
    np.testing.assert_allclose(fr["chamfer_first"], alone[0]["chamfer_first"], rtol=2e-5)
    # (the marker stages start from the chamfer stages' results, which have drifted apart by then: compared loosely)
    np.testing.assert_allclose(fr["chamfer_final"], alone[0]["chamfer_final"], rtol=0.15)
    np.testing.assert_allclose(fr["marker_final_final"], alone[0]["marker_final_final"], rtol=0.5)
    print("OBS frame blocks: chamfer first", fr["chamfer_first"], "vs", alone[0]["chamfer_first"], "final", fr["chamfer_final"],
          "vs", alone[0]["chamfer_final"], "| final marker", fr["marker_final_final"], "vs", alone[0]["marker_final_final"],
          "| median |dtrans| %.3g" % np.median(np.abs(fr["trans"] - alone[0]["trans"])))
    assert np.median(np.abs(fr["trans"] - alone[0]["trans"])) < 2e-2


    # (7) the runner: one output file, written by rank 0, both ranks went through the fit
    assert res[0]["runner_written"] == 1 and res[1]["runner_written"] == 0
    assert res[0]["runner_file"] and res[1]["runner_file"]

    # (6) lanes: concurrent hypotheses give the serial results
    for r in range(world):
        for k in ("trans", "pose_body", "betas", "root_orient", "yaw_scores"):
            assert np.array_equal(res[r]["frames_lanes"][k], res[r]["frames"][k]), ("frames", r, k)
            assert np.array_equal(res[r]["shared_lanes"][k], res[r]["shared"][k]), ("shared", r, k)
            assert np.array_equal(res[r]["shared_lanes_gloo"][k], res[r]["shared"][k]), ("shared over gloo", r, k)
    assert res[0]["shared_transport"] == "ShmReducer" and res[0]["shared_transport_gloo"] == "DistReducer"


@pytest.mark.parametrize("stage", ["chamfer", "marker"])
def test_shared_betas_with_one_rank_is_the_device_driver_bit_for_bit(stage):
    """uuo_lbfgs_solve_shared with world = 1 goes through the whole exchange machinery (gather of the evaluation's
    statistics, Gram rows through the host, summed shape gradient written back) and must still BE uuo_lbfgs_solve: same
    iterates bit for bit, same counts -- the reductions over one rank change nothing.  The Python checker
    (dist_lbfgs.ShardedLBFGS) on the same problem follows closely (fp64 dot products in both, different summation orders)."""
    from uuo_mocap_amd.dist_lbfgs import LocalReducer
    from uuo_mocap_amd.engine import ChamferProblem, MarkerProblem
    from uuo_mocap_amd.synthetic import make_sequence

    dev, tables, cfg, smpl = _setup()
    seq = make_sequence(tables, seed=81, num_frames=F, num_markers=M)
    markers = torch.from_numpy(seq.markers.get_points()).float().to(dev)
    o_pose = seq.img_smpl.pose_body.to(dev)
    o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
    root = seq.img_smpl.root_orient.to(dev)
    if stage == "chamfer":
        prob = ChamferProblem(smpl, markers, o_pose, o_betas, root, cfg)
        x0 = prob.pack(torch.median(markers, dim=1)[0], torch.zeros(F, 1, 1, device=dev), o_betas, o_pose)
        lr = 0.1
    else:
        prob = MarkerProblem(smpl, markers, o_pose, o_betas, torch.from_numpy(seq.gt["marker_vids"]).to(dev), cfg)
        x0 = prob.pack(o_pose, o_betas, root, torch.median(markers, dim=1)[0])
        lr = 1.0
    xa, xb, xc = x0.clone(), x0.clone(), x0.clone()
    la, lb = [], []
    sa = prob.solve(xa, max_iter=40, lr=lr, callback=lambda i, l: la.append(l))
    sb = prob.solve_shared(xb, LocalReducer(), max_iter=40, lr=lr, callback=lambda i, l: lb.append(l))
    assert (sa["n_iter"], sa["n_eval"], sa["stop_reason"]) == (sb["n_iter"], sb["n_eval"], sb["stop_reason"]), (sa, sb)
    assert la == lb and sa["n_iter"] >= 30
    assert torch.equal(xa, xb)
    sc = prob.solve_shared_reference(xc, LocalReducer(), max_iter=40, lr=lr)
    assert abs(sa["n_iter"] - sc["n_iter"]) <= 2, (sa, sc)
    assert sa["first_loss"] == pytest.approx(sc["first_loss"], rel=1e-6)
    assert sc["final_loss"] == pytest.approx(sa["final_loss"], rel=0.1), (sa, sc)  # 40 iterations: the trajectories have parted


def _nccl_one_rank_main(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dev = torch.device("cuda:0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)   # "nccl" IS RCCL on ROCm
    try:
        from uuo_mocap_amd import parallel
        from uuo_mocap_amd.dist_lbfgs import DistReducer
        from uuo_mocap_amd.engine import ChamferProblem
        from uuo_mocap_amd.synthetic import make_sequence

        dev, tables, cfg, smpl = _setup()
        out = {"backend": dist.get_backend()}
        seq = make_sequence(tables, seed=81, num_frames=F, num_markers=M)
        markers = torch.from_numpy(seq.markers.get_points()).float().to(dev)
        o_pose = seq.img_smpl.pose_body.to(dev)
        o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
        prob = ChamferProblem(smpl, markers, o_pose, o_betas, seq.img_smpl.root_orient.to(dev), cfg)
        x0 = prob.pack(torch.median(markers, dim=1)[0], torch.zeros(F, 1, 1, device=dev), o_betas, o_pose)
        red = DistReducer(None, dev)
        out["reducer_device"] = str(red.device)
        xa, xb = x0.clone(), x0.clone()
        la, lb = [], []
        sa = prob.solve(xa, max_iter=40, lr=0.1, callback=lambda i, l: la.append(l))
        sb = prob.solve_shared(xb, red, max_iter=40, lr=0.1, callback=lambda i, l: lb.append(l))
        out["solve"] = (sa, sb, la == lb, bool(torch.equal(xa, xb)))
        # whole fits: the two collective modes with lanes = 0, i.e. every gather on the RCCL communicator of the default group
        out["alone"] = _fit(smpl, seq, cfg, dev)
        with parallel.shared_betas(device=dev, lanes=0, transport="rccl") as r1:
            out["shared"] = _fit(smpl, seq, cfg, dev)
            out["shared_reducer"] = (type(r1).__name__, str(r1.device))
        with parallel.shard_frames(device=dev, lanes=0, transport="rccl", joint_with_one_rank=True):
            out["frames"] = _fit(smpl, seq, cfg, dev)
        out["gathers"] = r1.stats()["gathers"]
        torch.save(out, os.path.join(out_dir, "nccl.pt"))
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_group_runs_the_device_branch_of_the_exchange(tmp_path):
    """SURVEY 8e / north_star: the shared-beta exchange over RCCL.  One GPU box cannot hold two RCCL ranks, but a
    world-size-1 "nccl" group CAN be created on it: every gather of a shared solve then goes through
    all_gather_into_tensor on device buffers of the RCCL communicator (DistReducer's device branch: the code a multi-GPU node
    runs), and with one rank the result must BE uuo_lbfgs_solve, bit for bit -- as a single joint solve and as whole fits in
    both collective modes (shared betas, frame blocks) with the hypotheses serial on the default group (lanes = 0)."""
    import torch.multiprocessing as mp

    port = 29500 + ((os.getpid() + 31) % 2000)
    mp.spawn(_nccl_one_rank_main, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    out = torch.load(os.path.join(str(tmp_path), "nccl.pt"), weights_only=False)
    assert out["backend"] == "nccl" and out["reducer_device"].startswith("cuda")
    sa, sb, same_losses, same_x = out["solve"]
    assert (sa["n_iter"], sa["n_eval"], sa["stop_reason"]) == (sb["n_iter"], sb["n_eval"], sb["stop_reason"]), (sa, sb)
    assert same_losses and same_x and sa["n_iter"] >= 30
    assert out["shared_reducer"][0] == "DistReducer" and out["shared_reducer"][1].startswith("cuda")
    for mode in ("shared", "frames"):
        for k in ("trans", "pose_body", "betas", "root_orient", "yaw_scores"):
            assert np.array_equal(out[mode][k], out["alone"][k]), (mode, k)
        assert all("world=1" in d for d in out[mode]["chamfer_driver"] + out[mode]["marker_driver"]), out[mode]["chamfer_driver"]
    assert out["gathers"] > 500   # every closure evaluation and iteration of both fits went through the communicator


def _runner_cli_main(rank, world, port, root, cfg_path):
    # what `torchrun -m uuo_mocap_amd.runner` gives a rank: the launcher's environment and NOTHING initialised
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world), UUO_SHARE_GPU="1")
    import contextlib
    import io

    import torch.distributed as dist

    from uuo_mocap_amd import runner

    assert not dist.is_initialized()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runner.main(["--config", cfg_path, "--dataset", "moyo_val", "--input_dir", root, "--rank_mode", "frames",
                     "--print_options"])
    assert dist.is_initialized() and dist.get_world_size() == world
    with open(os.path.join(root, "cli%d.txt" % rank), "w") as fh:
        fh.write(buf.getvalue().strip().splitlines()[-1])
    dist.destroy_process_group()


def test_runner_cli_brings_up_its_own_process_group(tmp_path):
    """`--rank_mode frames` from the command line (no process group initialised by the caller) must create the
    group itself -- it used to fall back to every rank fitting the whole sequence on its own and discarding the result."""
    import torch.multiprocessing as mp

    from uuo_mocap_amd import runner
    from uuo_mocap_amd.body_model import synthetic_smpl
    from uuo_mocap_amd.config import CONFIG_DIR
    from uuo_mocap_amd.synthetic import make_sequence

    root = str(tmp_path)
    d = os.path.join(root, "moyo_val", "mocap", "subj")
    os.makedirs(d, exist_ok=True)
    seq = make_sequence(synthetic_smpl(0), seed=41, num_frames=F, num_markers=M)
    runner.write_sequence_npz(os.path.join(d, "seq0.npz"), seq.markers.get_points(), 30.0, seq.img_smpl.pose_body,
                              seq.img_smpl.root_orient, seq.img_smpl.betas)
    cfg_path = os.path.join(root, "cfg.yaml")
    with open(cfg_path, "w") as fh:
        fh.write("parent: %s\nname: unit\nstages:\n  part:\n    num_iters: 6\n  chamfer:\n    num_iters: 6\n"
                 "  marker:\n    num_iters: 6\n" % os.path.join(CONFIG_DIR, "video_mocap.yaml"))
    port = 29500 + ((os.getpid() + 57) % 2000)
    mp.spawn(_runner_cli_main, args=(2, port, root, cfg_path), nprocs=2, join=True)
    assert os.path.isfile(os.path.join(root, "moyo_val", "results", "unit", "subj", "seq0_stageii.npz"))
    assert open(os.path.join(root, "cli0.txt")).read() == "wrote 1 sequence(s)"
    assert open(os.path.join(root, "cli1.txt")).read() == "wrote 0 sequence(s)"


def _failing_rank_main(rank, world, port, out_dir):
    import time

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), UUO_SHARE_GPU="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from uuo_mocap_amd import parallel
        from uuo_mocap_amd.engine import ChamferProblem
        from uuo_mocap_amd.synthetic import make_sequence

        dev, tables, cfg, smpl = _setup()
        res = {}
        for transport in ("shm", "gloo"):
            red = parallel._default_reducer(None, dev, transport)
            sq = make_sequence(tables, seed=70 + rank, num_frames=F, num_markers=M)
            markers = torch.from_numpy(sq.markers.get_points()).float().to(dev)
            o_betas = (sq.img_smpl.betas.sum(0, keepdim=True) / sq.img_smpl.img_mask.sum()).to(dev)
            prob = ChamferProblem(smpl, markers, sq.img_smpl.pose_body.to(dev), o_betas, sq.img_smpl.root_orient.to(dev), cfg)
            x = prob.pack(torch.median(markers, dim=1)[0], torch.zeros(F, 1, 1, device=dev), o_betas,
                          sq.img_smpl.pose_body.to(dev))
            if rank == 1:
                prob.problem.M = 0   # this rank's problem does not validate: its solve fails before the first exchange
            t0 = time.perf_counter()
            try:
                prob.solve_shared(x, red, max_iter=10, lr=0.1)
                res[transport] = ("returned", time.perf_counter() - t0, "")
            except RuntimeError as exc:
                res[transport] = ("raised", time.perf_counter() - t0, str(exc))
            dist.barrier()
        torch.save(res, os.path.join(out_dir, "fail%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_takes_its_peers_out_of_a_shared_solve(tmp_path):
    """a rank that leaves a shared solve (here: its problem does not validate) used to leave the other ranks
    blocked in their next gather until the transport's time-out.  Every gathered message carries a status word now and a
    rank that fails still takes part in the exchange its peers are waiting in: both ranks raise within seconds, over the
    shared-memory mailbox and over gloo, and the healthy rank's message names the one that failed."""
    import torch.multiprocessing as mp

    port = 29500 + ((os.getpid() + 77) % 2000)
    mp.spawn(_failing_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "fail%d.pt" % r), weights_only=False) for r in range(2)]
    for transport in ("shm", "gloo"):
        for r in range(2):
            kind, seconds, msg = res[r][transport]
            assert kind == "raised" and seconds < 30.0, (transport, r, res[r][transport])
        assert "rank 1" in res[0][transport][2], res[0][transport]
