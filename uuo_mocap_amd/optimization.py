"""Stage solvers with the reference's signatures and in-place semantics
(reference src/video_mocap/optimization.py:21-32,147-163,288-304,402-417,645-724), each one call into the
device-resident L-BFGS of libuuo_hip.so instead of a Python closure driven by torch.optim.LBFGS."""
from __future__ import annotations

import threading
from collections.abc import Callable
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .engine import MARKER_DISTANCE, ChamferProblem, MarkerProblem
from .losses import MarkerLoss, chamfer_distance, weighted_chamfer_distance  # noqa: F401  (re-exported like the reference)
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


def _reject_iter_fn(iter_fn):
    if iter_fn is not None:
        raise NotImplementedError(
            "iter_fn / save_iterations is a visualisation hook that copies every iterate to the host "
            "(reference multimodal.py:102-142); the device-resident solver does not expose iterates")


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
    _reject_iter_fn(iter_fn)
    prob = ChamferProblem(smpl_inference, markers, o_pose_body, o_betas, root_orient, config)
    z_angle = torch.zeros((root_orient.shape[0], root_orient.shape[1], 1), device=root_orient.device)
    x = prob.pack(trans, z_angle, betas, pose_body)
    stats = prob.solve(
        x, max_iter=config["stages"]["chamfer"]["num_iters"], lr=0.1,
        tolerance_grad=config["optimizer"]["tolerance_grad"], tolerance_change=config["optimizer"]["tolerance_change"],
        callback=_printer("Chamfer", verbose))
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
    _reject_iter_fn(iter_fn)
    one_hot = barycentric_coords_one_hot
    if one_hot.dim() != 2 or one_hot.shape[1] != smpl_inference.device_model.V:
        raise ValueError("barycentric_coords_one_hot must be [M, %d]" % smpl_inference.device_model.V)
    if not bool(((one_hot != 0).sum(dim=1) == 1).all()):
        raise NotImplementedError("only one-hot vertex placements (compute_locations.use_mean) are supported")
    assign = torch.argmax(one_hot, dim=-1)
    prob = MarkerProblem(smpl_inference, markers, o_pose_body, o_betas, assign, config)
    x = prob.pack(pose_body, betas, root_orient, trans)
    stats = prob.solve(
        x, max_iter=config["stages"]["marker"]["num_iters"], lr=1.0,
        tolerance_grad=config["optimizer"]["tolerance_grad"], tolerance_change=config["optimizer"]["tolerance_change"],
        callback=_printer("Marker", verbose))
    new_pose, new_betas, new_root, new_trans = prob.unpack(x)
    with torch.no_grad():
        pose_body.copy_(new_pose)
        betas.copy_(new_betas)
        root_orient.copy_(new_root)
        trans.copy_(new_trans)
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
    """Marker placement: one-hot [M, 6890] of argmin_v mean_f |v_fv - x_fm| over the frames with img_mask == 1."""
    cl = config["stages"]["compute_locations"]
    if cl["use_barycentric"] or not cl["use_mean"] or granularity != "full" or window_size != 1:
        raise NotImplementedError("only the shipped placement (use_mean, granularity 'full', window 1) is built")
    with torch.no_grad():
        verts = smpl_inference(
            poses=normalize_rot(pose_body.detach()),
            betas=torch.repeat_interleave(torch.mean(betas.detach(), dim=0, keepdim=True), dim=0,
                                          repeats=betas.shape[0]),
            root_orient=normalize_rot(root_orient.detach()),
            trans=trans.detach(),
        )["vertices"]
        valid = (img_mask == 1)
        idx = smpl_inference.device_model.assign_mean_argmin(verts, markers, valid)
        one_hot = torch.zeros((markers.shape[1], verts.shape[1]), dtype=torch.float32, device=verts.device)
        one_hot.scatter_(1, idx.long()[:, None], 1.0)
    return one_hot.to(device)


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
