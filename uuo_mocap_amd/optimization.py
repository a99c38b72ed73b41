"""Stage solvers with the reference's signatures and in-place semantics
(reference src/video_mocap/optimization.py:21-32,147-163,288-304,402-417,645-724), each one call into the
device-resident L-BFGS of libuuo_hip.so instead of a Python closure driven by torch.optim.LBFGS."""
from __future__ import annotations

import threading
from collections.abc import Callable
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .device_lbfgs import DeviceLBFGS
from .engine import MARKER_DISTANCE, ChamferProblem, MarkerProblem
from .losses import (MarkerLoss, chamfer_distance, soft_weighted_chamfer_distance,  # noqa: F401  (re-exported)
                     weighted_chamfer_distance)
from .smpl import SmplInference
from .transforms import compute_root_orient_y, compute_root_orient_z, normalize_rot  # noqa: F401

#: per-solve statistics of the most recent calls (n_iter, n_eval, losses, device ms) -- bench.py and tests read it
LAST_STATS: Dict[str, Dict] = {}
_tls_stats = threading.local()  # same, per calling thread (concurrent yaw hypotheses)


def last_stats(kind: str) -> Dict:
    return getattr(_tls_stats, kind)


def _printer(tag: str, verbose: bool):
    if not verbose:
        return None
    return lambda i, loss: print(tag, i, float(loss))


def optim_root(*args, **kwargs):
    """reference optimization.py:21-144.  Disabled in every shipped config (stages.root.num_iters: 0) and not
    runnable as written there (undefined o_betas :112, missing config key 'lr' :51) -- not reproduced."""
    raise NotImplementedError("optim_root is disabled in every shipped configuration of the reference")


def optim_chamfer(
    markers: torch.Tensor,  # [F, M, 3]
    pose_body: torch.Tensor,  # [F, J-1, 3, 3]
    o_pose_body: torch.Tensor,  # [F, J-1, 3, 3]
    betas: torch.Tensor,  # [1, 10]
    o_betas: torch.Tensor,  # [1, 10]
    root_orient: torch.Tensor,  # [F, 1, 3, 3]
    trans: torch.Tensor,  # [F, 3]
    img_mask: torch.Tensor,  # [F]
    marker_labels: torch.Tensor,  # [F, M]
    smpl_inference: SmplInference,
    config: Dict,
    initial_angle: float = 0,
    repeat: int = 0,
    verbose: bool = False,
    iter_fn: Callable = None,
):
    """Chamfer (pose fitting) stage: L-BFGS over [trans, z_angle, betas, pose_body], lr 0.1.  Mutates
    trans / betas / pose_body in place and applies the optimised yaw to root_orient in place."""
    st = config["stages"]["chamfer"]
    fused_losses = _CHAMFER_FUSED_LOSSES
    if float(st["losses"].get("soft_chamfer", 0.0)) != 0.0 and markers.is_cuda and \
            bool((config.get("execution") or {}).get("chamfer_soft_fused", True)):
        # EXTENSION: the soft-assignment data term has a fused closure (dense backward on the matrix pipe, csrc/dense_bwd.hip);
        # execution.chamfer_soft_fused: False keeps the operator-composed closure, its checker
        fused_losses = _CHAMFER_FUSED_LOSSES | {"soft_chamfer"}
    if (set(st["losses"]) - fused_losses) or not st["yaw_lock"]:
        return _optim_chamfer_general(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, marker_labels,
                                      smpl_inference, config, initial_angle, repeat, verbose, iter_fn)
    from .parallel import frame_shard

    fs = frame_shard()
    if fs is not None and fs.active:
        if "soft_chamfer" in fused_losses:
            raise NotImplementedError("stages.chamfer.losses.soft_chamfer (extension) is not built for frame-block sharding "
                                      "(parallel.shard_frames); set execution.chamfer_soft_fused: False or use another mode")
        return _optim_chamfer_frame_sharded(fs, markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans,
                                            smpl_inference, config, iter_fn)
    prob = ChamferProblem(smpl_inference, markers, o_pose_body, o_betas, root_orient, config)
    z_angle = torch.zeros((root_orient.shape[0], root_orient.shape[1], 1), device=root_orient.device)
    x = prob.pack(trans, z_angle, betas, pose_body)
    point_cb = None
    if iter_fn is not None:
        root_np = normalize_rot(root_orient.detach()).cpu().numpy()  # the closure reports the un-yawed input (:271)

        def point_cb(i, loss, x_eval):  # what closure_stage_chamfer hands to iter_fn after every evaluation (:263-272)
            e_trans, _, e_betas, e_pose = prob.unpack(x_eval)
            iter_fn(stage="chamfer_" + str(repeat), iteration=i, initial_angle=np.array([initial_angle]),
                    pose_body=normalize_rot(e_pose).numpy(), betas=e_betas.numpy().copy(), trans=e_trans.numpy().copy(),
                    root_orient=root_np)

    stats = _solve(prob, x, config, "chamfer", 0.1, "Chamfer", verbose, point_cb)
    new_trans, new_z, new_betas, new_pose = prob.unpack(x)
    with torch.no_grad():
        trans.copy_(new_trans)
        betas.copy_(new_betas)
        pose_body.copy_(new_pose)
        root_orient.requires_grad_(False)
        root_orient[:] = compute_root_orient_z(new_z) @ root_orient
    root_orient.requires_grad_(True)
    LAST_STATS["chamfer"] = stats
    _tls_stats.chamfer = stats
    return None


def _frame_sharded_solve(fs, prob, x, config, stage: str, lr: float, frames_here: int, frames_all: int, data_share: float):
    """One stage problem on this rank's frame block as a member of the joint problem over `fs`'s ranks.  The
    weights of the block carry the GLOBAL normalisers: the data term's share of the global denominator (`data_share`), the
    pose prior's F_r / F, the shape prior once in total."""
    opt = config["optimizer"]
    if str(opt.get("type", "lbfgs")).lower() != "lbfgs":
        raise NotImplementedError("frame sharding: L-BFGS driver only")
    prob.problem.w_data = float(prob.problem.w_data) * float(data_share)
    prob.problem.w_pose = float(prob.problem.w_pose) * float(frames_here) / float(frames_all)
    prob.problem.w_betas = float(prob.problem.w_betas) / float(fs.world)
    stats = prob.solve_shared(x, fs.reducer, max_iter=config["stages"][stage]["num_iters"], lr=lr,
                              tolerance_grad=opt["tolerance_grad"], tolerance_change=opt["tolerance_change"])
    stats["driver"] = "device-lbfgs(frame blocks, world=%d)" % fs.world
    return stats


def _optim_chamfer_frame_sharded(fs, markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, smpl_inference,
                                 config, iter_fn):
    """optim_chamfer's fused solve spread over ranks by frame blocks (parallel.shard_frames); same in-place contract, every
    rank ends with the full result."""
    if iter_fn is not None:
        raise NotImplementedError("frame sharding: no per-evaluation iter_fn")
    F = int(markers.shape[0])
    lo, hi = fs.block(F)
    prob = ChamferProblem(smpl_inference, markers[lo:hi], o_pose_body[lo:hi], o_betas, root_orient[lo:hi], config)
    mask = get_marker_mask(markers)                      # full_chamfer is normalised by the number of present markers
    share = float(mask[lo:hi].sum().item()) / float(mask.sum().item())
    z_angle = torch.zeros((hi - lo, root_orient.shape[1], 1), device=root_orient.device)
    x = prob.pack(trans[lo:hi], z_angle, betas, pose_body[lo:hi])
    stats = _frame_sharded_solve(fs, prob, x, config, "chamfer", 0.1, hi - lo, F, share)
    new_trans, new_z, new_betas, new_pose = prob.unpack(x)
    with torch.no_grad():
        trans.copy_(fs.gather_frames(new_trans, F))
        betas.copy_(new_betas)
        pose_body.copy_(fs.gather_frames(new_pose, F))
        root_orient.requires_grad_(False)
        root_orient[:] = compute_root_orient_z(fs.gather_frames(new_z, F)) @ root_orient
    root_orient.requires_grad_(True)
    LAST_STATS["chamfer"] = stats
    _tls_stats.chamfer = stats
    return None


def _optim_markers_frame_sharded(fs, markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, assign,
                                 smpl_inference, config, iter_fn):
    """optim_markers' fused solve (one-hot placement) spread over ranks by frame blocks; same in-place contract."""
    if iter_fn is not None:
        raise NotImplementedError("frame sharding: no per-evaluation iter_fn")
    F = int(markers.shape[0])
    lo, hi = fs.block(F)
    prob = MarkerProblem(smpl_inference, markers[lo:hi], o_pose_body[lo:hi], o_betas, assign, config)
    x = prob.pack(pose_body[lo:hi], betas, root_orient[lo:hi], trans[lo:hi])
    # the marker term is a mean over all F x M entries: the block's share of the denominator is F_r / F
    stats = _frame_sharded_solve(fs, prob, x, config, "marker", 1.0, hi - lo, F, float(hi - lo) / float(F))
    new_pose, new_betas, new_root, new_trans = prob.unpack(x)
    with torch.no_grad():
        pose_body.copy_(fs.gather_frames(new_pose, F))
        betas.copy_(new_betas)
        root_orient.copy_(fs.gather_frames(new_root, F))
        trans.copy_(fs.gather_frames(new_trans, F))
    LAST_STATS["marker"] = stats
    _tls_stats.marker = stats
    return None


def lockstep_supported(config: Dict, stage: str) -> bool:
    """True when `stage` ("chamfer" / "marker") of this configuration runs on the fused device closure with the L-BFGS
    driver, i.e. when independent solves of it can be stepped together (engine.solve_batch)."""
    st = config["stages"][stage]
    if str(config["optimizer"].get("type", "lbfgs")).lower() != "lbfgs":
        return False
    if stage == "chamfer":
        return not (set(st["losses"]) - _CHAMFER_FUSED_LOSSES) and bool(st["yaw_lock"])
    return not (set(st["losses"]) - {"marker", "reg_pose_body", "reg_betas"}) and not st.get("use_sdf")


def optim_chamfer_lockstep(markers, hyps, o_pose_body, o_betas, smpl_inference, config):
    """`optim_chamfer` for several independent hypotheses at once (the yaw hypotheses of multimodal_video_mocap,
    reference multimodal.py:462-497): `hyps` is a list of dicts with the leaves `pose_body`, `betas`, `root_orient`,
    `trans` of each.  One lock-step batch (engine.solve_batch) instead of one solve per host thread; each hypothesis ends
    exactly where `optim_chamfer` would take it alone (bit-identical), with the same in-place updates.  Returns the
    solver statistics per hypothesis."""
    from .engine import solve_batch

    probs, xs = [], []
    for h in hyps:
        prob = ChamferProblem(smpl_inference, markers, o_pose_body, o_betas, h["root_orient"], config)
        z_angle = torch.zeros((h["root_orient"].shape[0], h["root_orient"].shape[1], 1), device=h["root_orient"].device)
        probs.append(prob)
        xs.append(prob.pack(h["trans"], z_angle, h["betas"], h["pose_body"]))
    opt = config["optimizer"]
    stats = solve_batch(probs, xs, max_iter=config["stages"]["chamfer"]["num_iters"], lr=0.1,
                        tolerance_grad=opt["tolerance_grad"], tolerance_change=opt["tolerance_change"])
    for h, prob, x in zip(hyps, probs, xs):
        new_trans, new_z, new_betas, new_pose = prob.unpack(x)
        with torch.no_grad():
            h["trans"].copy_(new_trans)
            h["betas"].copy_(new_betas)
            h["pose_body"].copy_(new_pose)
            h["root_orient"].requires_grad_(False)
            h["root_orient"][:] = compute_root_orient_z(new_z) @ h["root_orient"]
        h["root_orient"].requires_grad_(True)
    return stats


def optim_markers_lockstep(markers, hyps, o_pose_bodies, o_betas, one_hots, smpl_inference, config):
    """`optim_markers` for several independent hypotheses at once (reference multimodal.py:549-565), one-hot placements
    only; same contract as optim_chamfer_lockstep."""
    from .engine import solve_batch

    probs, xs = [], []
    for h, o_pose, one_hot in zip(hyps, o_pose_bodies, one_hots):
        prob = MarkerProblem(smpl_inference, markers, o_pose, o_betas, torch.argmax(one_hot, dim=-1), config)
        probs.append(prob)
        xs.append(prob.pack(h["pose_body"], h["betas"], h["root_orient"], h["trans"]))
    opt = config["optimizer"]
    stats = solve_batch(probs, xs, max_iter=config["stages"]["marker"]["num_iters"], lr=1.0,
                        tolerance_grad=opt["tolerance_grad"], tolerance_change=opt["tolerance_change"])
    for h, prob, x in zip(hyps, probs, xs):
        new_pose, new_betas, new_root, new_trans = prob.unpack(x)
        with torch.no_grad():
            h["pose_body"].copy_(new_pose)
            h["betas"].copy_(new_betas)
            h["root_orient"].copy_(new_root)
            h["trans"].copy_(new_trans)
    return stats


def placement_corners(placement: torch.Tensor):
    """A placement matrix [M, V] with at most three non-zeros per row (compute_nearest_points with use_barycentric) as
    (corner vertex ids [M, 3] int32 in ascending order, their weights [M, 3]); rows with fewer non-zeros are padded with
    zero-weight corners."""
    _, i3 = torch.topk(placement.abs(), 3, dim=1)
    i3, _ = torch.sort(i3, dim=1)  # ascending vertex order: a fixed summation order of the virtual marker
    return i3.to(torch.int32), torch.gather(placement, 1, i3).to(torch.float32)


def is_one_hot_placement(one_hot: torch.Tensor) -> bool:
    rows_nz = (one_hot != 0).sum(dim=1)
    return bool(((rows_nz == 1) & (one_hot.sum(dim=1) == 1.0)).all())


def _solve(prob, x, config, stage: str, lr: float, verbose_tag: str, verbose: bool, point_cb):
    """L-BFGS (the reference's only driver) unless the config carries the EXTENSION key `optimizer.type: adam`
    (BASELINE's north star names Adam; the reference has no such option): then `optimizer.adam_steps` (default: the
    stage's num_iters) Adam steps of `optimizer.adam_lr` (default: the stage's L-BFGS lr / 100) on the same fused closure."""
    opt = config["optimizer"]
    kind = str(opt.get("type", "lbfgs")).lower()
    from .parallel import shared_betas_reducer

    shared = shared_betas_reducer()
    if shared is not None:  # EXTENSION: one joint problem over the ranks, shared betas (parallel.shared_betas)
        if kind != "lbfgs" or point_cb is not None:
            raise NotImplementedError("shared betas: L-BFGS driver without per-evaluation callbacks only")
        return prob.solve_shared(x, shared, max_iter=config["stages"][stage]["num_iters"], lr=lr,
                                 tolerance_grad=opt["tolerance_grad"], tolerance_change=opt["tolerance_change"])
    if kind == "lbfgs":
        return prob.solve(x, max_iter=config["stages"][stage]["num_iters"], lr=lr, tolerance_grad=opt["tolerance_grad"],
                          tolerance_change=opt["tolerance_change"], callback=_printer(verbose_tag, verbose),
                          point_callback=point_cb,
                          history_size=int(opt.get("history_size", 100)))  # torch.optim.LBFGS's default; the reference never sets it
    if kind != "adam":
        raise ValueError("optimizer.type must be 'lbfgs' or 'adam' (got %r)" % kind)
    if point_cb is not None:
        raise NotImplementedError("iter_fn is reported by the L-BFGS driver only")
    return prob.solve_adam(x, num_steps=int(opt.get("adam_steps", config["stages"][stage]["num_iters"])),
                           lr=float(opt.get("adam_lr", lr / 100.0)), callback=_printer(verbose_tag, verbose))


#: chamfer-stage loss terms the device solver fuses (the only ones the shipped configs enable)
_CHAMFER_FUSED_LOSSES = {"full_chamfer", "reg_pose_body", "reg_betas"}


def _optim_chamfer_general(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, marker_labels,
                           smpl_inference, config, initial_angle, repeat, verbose, iter_fn):
    """Chamfer stage with the reference's optional terms (`part_chamfer`, `trans_vel`, `ground`; plus the labelled
    extension `soft_chamfer`, a soft-assignment data term the reference does not have) and / or
    `yaw_lock: False` (reference optimization.py:164-285; none is in a shipped config): the closure is composed from the
    differentiable HIP operators and driven by torch.optim.LBFGS with the reference's parameter list
    [trans, z_angle, betas, pose_body], lr 0.1.  `root_orient_vel` stops in a debugger in the reference (:241) and is
    refused.  Same in-place semantics as the fused path."""
    st = config["stages"]["chamfer"]
    w = st["losses"]
    unknown = set(w) - _CHAMFER_FUSED_LOSSES - {"part_chamfer", "trans_vel", "ground", "soft_chamfer"}
    if unknown:
        raise NotImplementedError("chamfer-stage losses that cannot run in the reference: %s" % sorted(unknown))
    device = root_orient.device
    num_frames = pose_body.shape[0]
    root_fixed = root_orient.detach().clone()
    if st["yaw_lock"]:
        z_angle = torch.zeros((root_orient.shape[0], root_orient.shape[1], 1), device=device).requires_grad_(True)
    else:  # a free rotation that REPLACES the root orientation inside the closure and multiplies it afterwards (:167-172,196,282)
        z_angle = torch.eye(3, device=device).expand(root_orient.shape[0], root_orient.shape[1], 3, 3).clone().requires_grad_(True)
    p_trans, p_betas, p_pose = (t.detach().clone().requires_grad_(True) for t in (trans, betas, pose_body))
    params = [p_trans, z_angle, p_betas, p_pose]
    optimizer = DeviceLBFGS(params, max_iter=st["num_iters"], tolerance_grad=config["optimizer"]["tolerance_grad"],
                                  tolerance_change=config["optimizer"]["tolerance_change"], lr=0.1,
                                  line_search_fn="strong_wolfe")
    mask = get_marker_mask(markers)
    lbs_weights = smpl_inference.get_lbs_weights()
    n_eval = [0]
    trace = []

    def z_root():
        return compute_root_orient_z(z_angle) @ root_fixed if st["yaw_lock"] else normalize_rot(z_angle)

    def closure():
        optimizer.zero_grad()
        out = smpl_inference(poses=normalize_rot(p_pose), betas=torch.repeat_interleave(p_betas, dim=0, repeats=num_frames),
                             root_orient=normalize_rot(z_root()), trans=p_trans)
        loss = 0
        if "part_chamfer" in w:
            loss = loss + chamfer_distance_by_part(markers, out["vertices"], marker_labels, lbs_weights,
                                                   single_directional=st["single_directional"]) * w["part_chamfer"]
        if "full_chamfer" in w:
            loss = loss + weighted_chamfer_distance(x=markers, y=out["vertices"], x_weights=mask,
                                                    single_directional=st["single_directional"])[0] * w["full_chamfer"]
        if "soft_chamfer" in w:  # EXTENSION (not in the reference): soft-assignment data term, temperature stages.chamfer.soft_tau
            loss = loss + soft_weighted_chamfer_distance(markers, out["vertices"], mask,
                                                         float(st.get("soft_tau", 1e-3)))[0] * w["soft_chamfer"]
        if "reg_pose_body" in w:
            loss = loss + F.mse_loss(p_pose, o_pose_body) * w["reg_pose_body"]
        if "trans_vel" in w:
            markers_mean = torch.mean(markers, dim=1)
            loss = loss + F.mse_loss(p_trans[1:] - p_trans[:-1], markers_mean[1:] - markers_mean[:-1]) * w["trans_vel"]
        if "ground" in w:
            loss = loss + torch.mean(F.relu(-out["joints"][..., 2])) * w["ground"]
        if "reg_betas" in w:
            loss = loss + F.mse_loss(p_betas, o_betas) * w["reg_betas"]
        loss.backward()
        if verbose:
            print("Chamfer", n_eval[0], float(loss))
        if iter_fn is not None:
            iter_fn(stage="chamfer_" + str(repeat), iteration=n_eval[0], initial_angle=np.array([initial_angle]),
                    pose_body=normalize_rot(p_pose).detach().cpu().numpy(), betas=p_betas.detach().cpu().numpy(),
                    trans=p_trans.detach().cpu().numpy(), root_orient=normalize_rot(root_fixed).cpu().numpy())
        n_eval[0] += 1
        trace.append(loss.detach())
        return loss

    optimizer.step(closure)
    with torch.no_grad():
        trans.copy_(p_trans)
        betas.copy_(p_betas)
        pose_body.copy_(p_pose)
        root_orient.requires_grad_(False)
        root_orient[:] = (compute_root_orient_z(z_angle) if st["yaw_lock"] else normalize_rot(z_angle)) @ root_fixed
    root_orient.requires_grad_(True)
    stats = {"n_eval": n_eval[0], "n_iter": int(optimizer.state[params[0]].get("n_iter", 0)), "device_ms": 0.0,
             "driver": optimizer.stats.get("driver", "device-lbfgs(host closure)"), "loss_first": float(trace[0]), "loss_final": float(min(trace))}
    LAST_STATS["chamfer"] = stats
    _tls_stats.chamfer = stats
    return None


def optim_markers(
    markers: torch.Tensor,
    pose_body: torch.Tensor,
    o_pose_body: torch.Tensor,
    betas: torch.Tensor,
    o_betas: torch.Tensor,
    root_orient: torch.Tensor,
    trans: torch.Tensor,
    barycentric_coords_one_hot: torch.Tensor,
    img_mask: torch.Tensor,  # [F]
    smpl_inference: SmplInference,
    config: Dict,
    initial_angle: float = 0,
    repeat: int = 0,
    verbose: bool = False,
    iter_fn: Callable = None,
):
    """Marker (inverse kinematics) stage: L-BFGS over [pose_body, betas, root_orient, trans], lr 1.0, with the
    fixed marker -> vertex placement given as a one-hot [M, V] matrix.  Mutates the four leaves in place."""
    one_hot = barycentric_coords_one_hot
    if one_hot.dim() != 2 or one_hot.shape[1] != smpl_inference.device_model.V:
        raise ValueError("barycentric_coords_one_hot must be [M, %d]" % smpl_inference.device_model.V)
    rows_nz = (one_hot != 0).sum(dim=1)
    from .parallel import frame_shard

    fs = frame_shard()
    sharded = fs is not None and fs.active
    bary = None
    if not bool(((rows_nz == 1) & (one_hot.sum(dim=1) == 1.0)).all()):
        # barycentric placement (compute_locations.use_barycentric): up to three weighted vertices per marker.  Fused closure
        # (k_bary_fwd + k_bwd_items) when it is that and nothing else; execution.marker_bary_fused: False, more than three
        # non-zeros in a row, or frame-block sharding keep the closure composed from the operators
        fused = (one_hot.is_cuda and int(rows_nz.max()) <= 3 and not sharded and
                 bool((config.get("execution") or {}).get("marker_bary_fused", True)))
        if not fused:
            return _optim_markers_general(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, one_hot,
                                          smpl_inference, config, verbose, iter_fn, initial_angle, repeat)
        bary = placement_corners(one_hot)
    assign = bary[0] if bary is not None else torch.argmax(one_hot, dim=-1)
    if sharded:
        return _optim_markers_frame_sharded(fs, markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, assign,
                                            smpl_inference, config, iter_fn)
    prob = MarkerProblem(smpl_inference, markers, o_pose_body, o_betas, assign, config, bary=None if bary is None else bary[1])
    x = prob.pack(pose_body, betas, root_orient, trans)
    point_cb = None
    if iter_fn is not None:
        def point_cb(i, loss, x_eval):  # closure_stage_marker_pose's iter_fn call (:382-391)
            e_pose, e_betas, e_root, e_trans = prob.unpack(x_eval)
            iter_fn(stage="marker_" + str(repeat), iteration=i, initial_angle=np.array([initial_angle]),
                    pose_body=normalize_rot(e_pose).numpy(), betas=e_betas.numpy().copy(), trans=e_trans.numpy().copy(),
                    root_orient=normalize_rot(e_root).numpy())

    stats = _solve(prob, x, config, "marker", 1.0, "Marker", verbose, point_cb)
    new_pose, new_betas, new_root, new_trans = prob.unpack(x)
    with torch.no_grad():
        pose_body.copy_(new_pose)
        betas.copy_(new_betas)
        root_orient.copy_(new_root)
        trans.copy_(new_trans)
    LAST_STATS["marker"] = stats
    _tls_stats.marker = stats
    return None


def _optim_markers_general(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, coords, smpl_inference,
                           config, verbose, iter_fn=None, initial_angle=0, repeat=0):
    """Marker stage for a general placement matrix [M, V] (reference optimization.py:288-399 as written: virtual
    markers = coords @ vertices).  The fused device solver covers the one-hot placements of the shipped configs; this
    path composes the same closure from the differentiable HIP operators (SmplInference forward / uuo_smpl_backward)
    and drives it with torch.optim.LBFGS like the reference.  Mutates the four leaves in place."""
    st = config["stages"]["marker"]
    unsupported = set(st["losses"]) - {"marker", "reg_pose_body", "reg_betas"}
    if unsupported:
        raise NotImplementedError("marker-stage losses outside the shipped configs: %s" % sorted(unsupported))
    if st.get("use_sdf"):
        raise NotImplementedError("stages.marker.use_sdf is off in every shipped config")
    num_frames = pose_body.shape[0]
    leaves = [pose_body, betas, root_orient, trans]
    params = [p.detach().clone().requires_grad_(True) for p in leaves]
    p_pose, p_betas, p_root, p_trans = params
    optimizer = DeviceLBFGS(params, max_iter=st["num_iters"], tolerance_grad=config["optimizer"]["tolerance_grad"],
                                  tolerance_change=config["optimizer"]["tolerance_change"], lr=1.0,
                                  line_search_fn="strong_wolfe")
    weights = get_marker_mask(markers)
    coords = coords.to(torch.float32)
    n_eval = [0]
    trace = []

    def closure():
        optimizer.zero_grad()
        out = smpl_inference(poses=normalize_rot(p_pose), betas=torch.repeat_interleave(p_betas, dim=0, repeats=num_frames),
                             root_orient=normalize_rot(p_root), trans=p_trans)
        virtual = torch.einsum("mv,fvc->fmc", coords, out["vertices"])
        loss = 0
        if "marker" in st["losses"]:
            loss = loss + torch.mean(MarkerLoss(markers=markers, virtual_markers=virtual, marker_weights=weights,
                                                marker_distance=MARKER_DISTANCE)) * st["losses"]["marker"]
        if "reg_pose_body" in st["losses"]:
            loss = loss + F.mse_loss(p_pose, o_pose_body) * st["losses"]["reg_pose_body"]
        if "reg_betas" in st["losses"]:
            loss = loss + F.mse_loss(p_betas, o_betas) * st["losses"]["reg_betas"]
        loss.backward()
        if verbose:
            print("Marker", n_eval[0], float(loss))
        if iter_fn is not None:
            iter_fn(stage="marker_" + str(repeat), iteration=n_eval[0], initial_angle=np.array([initial_angle]),
                    pose_body=normalize_rot(p_pose).detach().cpu().numpy(), betas=p_betas.detach().cpu().numpy(),
                    trans=p_trans.detach().cpu().numpy(), root_orient=normalize_rot(p_root).detach().cpu().numpy())
        n_eval[0] += 1
        trace.append(loss.detach())
        return loss

    optimizer.step(closure)
    with torch.no_grad():
        for leaf, p in zip(leaves, params):
            leaf.copy_(p)
    stats = {"n_eval": n_eval[0], "n_iter": int(optimizer.state[params[0]].get("n_iter", 0)), "device_ms": 0.0,
             "driver": optimizer.stats.get("driver", "device-lbfgs(host closure)"), "loss_first": float(trace[0]), "loss_final": float(min(trace))}
    LAST_STATS["marker"] = stats
    _tls_stats.marker = stats
    return None


def compute_nearest_points(
    markers: torch.Tensor,
    pose_body: torch.Tensor,
    betas: torch.Tensor,
    root_orient: torch.Tensor,
    trans: torch.Tensor,
    smpl_inference: SmplInference,
    marker_labels: np.array,
    granularity: str,
    img_mask: torch.Tensor,
    device: torch.device,
    config: Dict,
    o_pose_body: torch.Tensor = None,
    window_size: int = 1,
    use_velocity: bool = True,
):
    """Marker placement [M, 6890] (reference optimization.py:402-642).

    `compute_locations.use_mean` (every shipped config): one-hot of argmin_v mean_f |v_fv - x_fm| over the frames with
    img_mask == 1 (one fused kernel).  `use_barycentric`: the closest point on the body surface (HIP brute force over
    the 13 776 faces) expressed in the winning face's three corners, chosen per `granularity` ("full" | "marker" |
    "part") exactly as the reference's window loop chooses it.  The third mode (neither flag) is not runnable in the
    reference: its three scatter_ calls write (1, 0, 0) to the SAME entry, so it returns an all-zero matrix
    (:549-561)."""
    cl = config["stages"]["compute_locations"]
    if not cl["use_barycentric"] and not cl["use_mean"]:
        raise NotImplementedError("compute_locations without use_mean / use_barycentric returns an all-zero placement "
                                  "in the reference (scatter_ overwrite, optimization.py:549-561): not reproduced")
    if granularity not in ("full", "marker", "part"):
        raise ValueError("granularity must be 'full', 'marker' or 'part'")
    if cl["use_mean"] and window_size != 1:
        raise NotImplementedError("use_mean with window_size != 1 (the reference's only callers pass 1; its distance "
                                  "table is then indexed inconsistently, optimization.py:457-484,597)")
    with torch.no_grad():
        verts = smpl_inference(
            poses=normalize_rot(pose_body.detach()),
            betas=torch.repeat_interleave(torch.mean(betas.detach(), dim=0, keepdim=True), dim=0,
                                          repeats=betas.shape[0]),
            root_orient=normalize_rot(root_orient.detach()),
            trans=trans.detach(),
        )["vertices"]
        if cl["use_mean"] and not cl["use_barycentric"]:
            # the window loop only fills the distance matrix in this mode; the result is the argmin below (:595-603)
            valid = _valid_frames(img_mask, markers.shape[0])
            idx = smpl_inference.device_model.assign_mean_argmin(verts, markers, valid)
            one_hot = torch.zeros((markers.shape[1], verts.shape[1]), dtype=torch.float32, device=verts.device)
            one_hot.scatter_(1, idx.long()[:, None], 1.0)
            return one_hot.to(device)
        coords = _barycentric_placement(markers, verts, smpl_inference, marker_labels, granularity, img_mask,
                                        pose_body.shape[1], window_size, use_velocity and o_pose_body is not None)
        if cl["use_mean"]:
            # both flags: the mean-distance argmin overwrites the window loop's result (:595-603)
            idx = smpl_inference.device_model.assign_mean_argmin(verts, markers, _valid_frames(img_mask, markers.shape[0]))
            coords = torch.zeros_like(coords)
            coords.scatter_(1, idx.long()[:, None], 1.0)
    return coords.to(device)


def _valid_frames(img_mask: torch.Tensor, num_frames: int) -> torch.Tensor:
    """[num_frames] bool.  The reference indexes its per-frame tables with np.where(img_mask == 1) (:466,597): the mask
    is in VIDEO frames, the tables in mocap frames, so after frame-rate resampling only the leading len(img_mask)
    frames can be valid, and a set index beyond the sequence is an IndexError there."""
    idx = torch.where(img_mask == 1)[0]
    if idx.numel() and int(idx.max()) >= num_frames:
        raise IndexError("img_mask marks frame %d but the sequence has %d frames" % (int(idx.max()), num_frames))
    valid = torch.zeros(num_frames, dtype=torch.bool, device=img_mask.device)
    valid[idx] = True
    return valid


#: diagnostics of the last barycentric placement (selected frame per marker, distances, velocity factors)
LAST_PLACEMENT: Dict = {}


def _barycentric_placement(markers, verts, smpl_inference, marker_labels, granularity, img_mask, num_joints,
                           window_size, use_velocity):
    """The reference's window loop (optimization.py:464-591) for `use_barycentric`.  One frame per window is examined
    (the loop's stride equals the window size, :455,468), each window owns its running minimum (:449), so every
    examined frame passes the `< min_distance[window]` test against +inf and the rows it selects overwrite the
    previous ones: the result is a per-marker "last examined frame that selects the marker" rule.  The closest points
    of ALL examined frames are computed in one launch; the selection itself is O(F M) host logic."""
    F, M = markers.shape[0], markers.shape[1]
    V = verts.shape[1]
    dm = smpl_inference.device_model
    faces = torch.from_numpy(np.asarray(smpl_inference.smpl.faces).astype(np.int64)).to(verts.device)
    valid_frames = set(torch.where(img_mask == 1)[0].tolist())
    # window w examines frame w * window_size and files it under index w; the validity test is on the window index
    # (:472-474), as in the reference
    frames = [(w, f) for w, f in enumerate(range(0, F, window_size)) if w in valid_frames]
    final_vid = torch.zeros((M, 3), dtype=torch.long, device=verts.device)
    final_w = torch.zeros((M, 3), dtype=torch.float32, device=verts.device)
    chosen = np.full((M,), -1, dtype=np.int64)
    info = {"frames": [f for _, f in frames], "chosen_frame": chosen}
    if frames:
        fsel = torch.tensor([f for _, f in frames], device=verts.device)
        dist, face, closest, bary = dm.mesh_closest_points(verts[fsel], faces, markers[fsel])
        dist_np = dist.double().cpu().numpy()              # [K, M]
        vids = faces[face.long()]                          # [K, M, 3]
        vel = np.ones((len(frames), M))
        if use_velocity:
            # vel_factor[w, m] of the placement found in examined frame k (:563-580): velocity of the placed point
            # (between trajectory frames w-1 and w) dotted with the marker's velocity; row w of the [F, M] table
            widx = torch.tensor([w for w, _ in frames], device=verts.device)
            prev = torch.clamp(widx - 1, min=0)
            gi = vids.reshape(len(frames), -1)                                             # [K, 3M]
            p_now = torch.gather(verts[widx].double(), 1, gi[..., None].expand(-1, -1, 3)).view(len(frames), M, 3, 3)
            p_prev = torch.gather(verts[prev].double(), 1, gi[..., None].expand(-1, -1, 3)).view(len(frames), M, 3, 3)
            wts = bary.double()[..., None]
            pv = ((p_now * wts).sum(2) - (p_prev * wts).sum(2)) * (widx > 0).double()[:, None, None]
            mv = (markers[widx].double() - markers[prev].double()) * (widx > 0).double()[:, None, None]
            vel = (pv * mv).sum(-1).cpu().numpy()
        info.update(distance=dist_np, vel_factor=vel)
        labels = np.asarray(marker_labels)
        for k, (w, f) in enumerate(frames):  # ascending, so later frames overwrite earlier ones
            if granularity == "full":
                if np.mean(dist_np[k]) * np.mean(vel[k]) < np.inf:
                    chosen[:] = k
            elif granularity == "marker":
                chosen[dist_np[k] < np.inf] = k
            else:  # "part": markers whose label this frame is one of the first `num_joints` joints (:583-589)
                lab = labels[f]
                for j in range(num_joints):
                    sel = lab == j
                    if sel.any() and np.median(dist_np[k][sel]) < np.inf:
                        if j >= M:  # the reference files the part's point under column j of an [windows, M] table (:578)
                            raise IndexError("granularity 'part': populated joint id %d >= number of markers %d" % (j, M))
                        chosen[sel] = k
        has = torch.from_numpy(chosen >= 0).to(verts.device)
        kk = torch.from_numpy(np.maximum(chosen, 0)).to(verts.device)
        ar = torch.arange(M, device=verts.device)
        final_vid = vids[kk, ar]
        final_w = bary[kk, ar] * has[:, None].float()
    coords = torch.zeros((M, V), dtype=torch.float32, device=verts.device)
    # three scatter_ calls in corner order (:533-535): assignment, so a repeated corner keeps the last weight
    for c in range(3):
        coords.scatter_(1, final_vid[:, [c]], final_w[:, [c]])
    LAST_PLACEMENT.clear()
    LAST_PLACEMENT.update(info)
    return coords


def compute_marker_labels_from_coords(smpl_inference: SmplInference, barycentric_coords_one_hot: torch.Tensor,
                                      num_frames: int):
    vertex_ids = torch.argmax(smpl_inference.get_lbs_weights(), dim=-1)  # [V]
    coords_ids = torch.argmax(barycentric_coords_one_hot, dim=-1)
    labels = vertex_ids[coords_ids]
    return torch.repeat_interleave(labels.unsqueeze(0), repeats=num_frames, dim=0)


def chamfer_distance_by_part(markers, vertices, marker_labels, vertex_weights, single_directional: bool = False):
    """reference optimization.py:682-700 (used only by loss keys no shipped config enables)."""
    vertex_mask = torch.argmax(vertex_weights, dim=-1)
    labels_mode = torch.mode(marker_labels, dim=0)[0]
    loss = 0
    for i in torch.unique(labels_mode).tolist():
        part = chamfer_distance(vertices[:, vertex_mask == i], markers[:, labels_mode == i],
                                single_directional=single_directional)[0]
        loss = loss + (part - MARKER_DISTANCE) ** 2
    return loss


def get_marker_mask(markers: torch.Tensor) -> torch.Tensor:
    """[F, M] bool: marker present (missing markers are exact zeros)."""
    return torch.sum(torch.abs(markers), axis=-1) != 0.0


def weighted_mse_loss(input: torch.Tensor, target: torch.Tensor, weights: torch.Tensor):
    return torch.mean(F.mse_loss(input, target, reduction="none") * weights)
