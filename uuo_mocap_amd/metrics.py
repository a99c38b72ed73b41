"""Evaluation metrics with the reference's names and definitions (reference src/video_mocap/evaluation/metrics.py:27-190):
marker-to-surface distance (m2s), MPJPE / MPJVE and their Procrustes-aligned variants, V2V.  All of them run on the
tensors' device; m2s is the one with real work -- the closest point on the body surface for every marker and frame --
and uses the same HIP kernel as the barycentric marker placement (`uuo_mesh_closest_points`) where the reference
loops over frames calling igl.signed_distance."""
from __future__ import annotations

from typing import List

import torch


def compute_marker_to_surface_distance(vertices: torch.Tensor, faces: torch.Tensor, markers: torch.Tensor) -> torch.Tensor:
    """mean_{f,m} |distance(markers[f,m], mesh(vertices[f], faces))| (metrics.py:27-45).  vertices [F,V,3], markers
    [F,M,3]; faces [NF,3] or the reference's per-frame [F,NF,3] (the first frame's list is used: one topology)."""
    from .engine import mesh_closest_points

    if faces.dim() == 3:
        if faces.shape[0] > 1 and not bool((faces == faces[:1]).all()):
            raise ValueError("per-frame face lists must share one topology")
        faces = faces[0]
    dist = mesh_closest_points(vertices, faces, markers)[0]
    return torch.mean(dist.double()).cpu()


def compute_MPJPE(pred_joints: torch.Tensor, gt_joints: torch.Tensor) -> torch.Tensor:
    return torch.mean(torch.norm(pred_joints - gt_joints, dim=-1))


def compute_MPJPE_joints(pred_joints: torch.Tensor, gt_joints: torch.Tensor, joints_ids: List[int]) -> torch.Tensor:
    return torch.mean(torch.norm(pred_joints[:, joints_ids] - gt_joints[:, joints_ids], dim=-1))


def _velocity_error(pred: torch.Tensor, gt: torch.Tensor, freq: float, joints_ids=None) -> torch.Tensor:
    pred_vel = (pred[1:] - pred[:-1]) * freq
    gt_vel = (gt[1:] - gt[:-1]) * freq
    if joints_ids is not None:
        pred_vel, gt_vel = pred_vel[:, joints_ids], gt_vel[:, joints_ids]
    return torch.mean(torch.norm(pred_vel - gt_vel, dim=-1))


def compute_MPJVE(pred_joints: torch.Tensor, gt_joints: torch.Tensor, freq: float) -> torch.Tensor:
    return _velocity_error(pred_joints, gt_joints, freq)


def compute_MPJVE_joints(pred_joints: torch.Tensor, gt_joints: torch.Tensor, freq: float,
                         joints_ids: List[int]) -> torch.Tensor:
    return _velocity_error(pred_joints, gt_joints, freq, joints_ids)


def compute_PA_MPJPE(pred_joints: torch.Tensor, gt_joints: torch.Tensor) -> torch.Tensor:
    return compute_MPJPE(compute_similarity_transform(pred_joints, gt_joints), gt_joints)


def compute_PA_MPJPE_joints(pred_joints: torch.Tensor, gt_joints: torch.Tensor, joints_ids: List[int]) -> torch.Tensor:
    return compute_MPJPE_joints(compute_similarity_transform(pred_joints, gt_joints), gt_joints, joints_ids)


def compute_PA_MPJVE(pred_joints: torch.Tensor, gt_joints: torch.Tensor, freq: float) -> torch.Tensor:
    return _velocity_error(compute_similarity_transform(pred_joints, gt_joints), gt_joints, freq)


def compute_PA_MPJVE_joints(pred_joints: torch.Tensor, gt_joints: torch.Tensor, freq: float,
                            joints_ids: List[int]) -> torch.Tensor:
    return _velocity_error(compute_similarity_transform(pred_joints, gt_joints), gt_joints, freq, joints_ids)


def compute_V2V(pred_vertices: torch.Tensor, gt_vertices: torch.Tensor) -> torch.Tensor:
    return torch.mean(torch.norm(pred_vertices - gt_vertices, dim=-1))


def compute_similarity_transform(S1: torch.Tensor, S2: torch.Tensor) -> torch.Tensor:
    """Per-frame orthogonal Procrustes: S1 [B,N,3] mapped by the similarity (s R, t) that brings it closest to S2
    (metrics.py:141-190, the HMR2.0 formulation: R = V Z U^T from the SVD of the 3x3 cross-covariance, Z fixing
    det R = +1, s = tr(R K) / var(S1))."""
    X1 = S1.permute(0, 2, 1)
    X2 = S2.permute(0, 2, 1)
    mu1 = X1.mean(dim=2, keepdim=True)
    mu2 = X2.mean(dim=2, keepdim=True)
    X1c, X2c = X1 - mu1, X2 - mu2
    var1 = (X1c ** 2).sum(dim=(1, 2))
    K = torch.matmul(X1c, X2c.permute(0, 2, 1))
    U, _, Vh = torch.linalg.svd(K)
    V = Vh.permute(0, 2, 1)
    Z = torch.eye(3, device=S1.device, dtype=S1.dtype).unsqueeze(0).repeat(S1.shape[0], 1, 1)
    Z[:, -1, -1] *= torch.sign(torch.linalg.det(torch.matmul(U, Vh)))
    R = torch.matmul(torch.matmul(V, Z), U.permute(0, 2, 1))
    trace = torch.matmul(R, K).diagonal(offset=0, dim1=-1, dim2=-2).sum(dim=-1)
    scale = (trace / var1)[:, None, None]
    t = mu2 - scale * torch.matmul(R, mu1)
    return (scale * torch.matmul(R, X1) + t).permute(0, 2, 1)
