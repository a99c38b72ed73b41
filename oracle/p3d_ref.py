"""CPU restatement of the pytorch3d pieces the reference calls.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restated from the published pytorch3d 0.7.x algorithms (package not vendored; reference
install.sh:5-7), anchored on the reference call sites:

* ``knn_points`` K=1 / ``chamfer_distance`` -- losses/chamfer_distance.py:15-20,
  markers/markers_utils.py:471-475,575-579, optimization.py:697
* ``rotation_6d_to_matrix`` / ``matrix_to_rotation_6d`` -- optimization.py:66-74,197,200,336,338
* ``axis_angle_to_matrix`` -- optimization.py:662-679
"""
from __future__ import annotations

import ctypes
import os
from collections import namedtuple
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

_KNN = namedtuple("KNN", "dists idx knn")

_HERE = os.path.dirname(os.path.abspath(__file__))
_knn_lib = None


def _load_knn_c():
    """ctypes handle to oracle/_build/knn_cpu.so (built by __graft_entry__.build()); None if absent."""
    global _knn_lib
    if _knn_lib is None:
        path = os.path.join(_HERE, "_build", "knn_cpu.so")
        if os.path.isfile(path):
            lib = ctypes.CDLL(path)
            lib.knn1_cpu.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
            lib.knn1_cpu.restype = None
            _knn_lib = lib
        else:
            _knn_lib = False
    return _knn_lib or None


def knn1_loop(p1: np.ndarray, p2: np.ndarray):
    """pytorch3d's CPU K=1 loop, executed by the C restatement (oracle/knn_cpu.c). p1 [N,P1,D], p2 [N,P2,D]."""
    lib = _load_knn_c()
    if lib is None:
        raise RuntimeError("oracle/_build/knn_cpu.so missing: run __graft_entry__.build()")
    p1 = np.ascontiguousarray(p1, dtype=np.float32)
    p2 = np.ascontiguousarray(p2, dtype=np.float32)
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    dists = np.empty((N, P1), dtype=np.float32)
    idx = np.empty((N, P1), dtype=np.int64)
    lib.knn1_cpu(p1.ctypes.data, p2.ctypes.data, N, P1, P2, D, dists.ctypes.data, idx.ctypes.data)
    return dists, idx


def _knn1_forward(p1: torch.Tensor, p2: torch.Tensor, chunk_elems: int = 1 << 24):
    """Vectorised K=1 search with the C loop's exact fp32 op order ((dx*dx + dy*dy) + dz*dz),
    first index on ties (torch.argmin returns the first minimal index)."""
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    dists = torch.empty((N, P1), dtype=p1.dtype)
    idx = torch.empty((N, P1), dtype=torch.int64)
    rows = max(1, chunk_elems // max(1, P1 * P2))
    for s in range(0, N, rows):
        e = min(N, s + rows)
        diff = p1[s:e, :, None, :] - p2[s:e, None, :, :]
        sq = diff * diff
        d = sq[..., 0]
        for k in range(1, D):
            d = d + sq[..., k]
        dmin, imin = torch.min(d, dim=-1)
        # torch.min(dim) does not document the tie rule; argmin does ("first")
        imin = torch.argmin(d, dim=-1)
        dists[s:e] = torch.gather(d, -1, imin[..., None])[..., 0]
        idx[s:e] = imin
    return dists, idx


class _Knn1(torch.autograd.Function):
    """K=1 nearest neighbour, squared L2.  Backward as pytorch3d knn.cpp: g = 2*grad*(p1-p2[idx]);
    +g -> p1, -g accumulated into p2[idx]."""

    @staticmethod
    def forward(ctx, p1, p2):
        with torch.no_grad():
            dists, idx = _knn1_forward(p1.detach(), p2.detach())
        ctx.save_for_backward(p1, p2, idx)
        ctx.mark_non_differentiable(idx)
        return dists, idx

    @staticmethod
    def backward(ctx, grad_dists, _grad_idx):
        p1, p2, idx = ctx.saved_tensors
        nn = torch.gather(p2, 1, idx[..., None].expand(-1, -1, p2.shape[2]))
        g = 2.0 * grad_dists[..., None] * (p1 - nn)
        grad_p2 = torch.zeros_like(p2)
        grad_p2.scatter_add_(1, idx[..., None].expand(-1, -1, p2.shape[2]), -g)
        return g, grad_p2


def knn_points(p1: torch.Tensor, p2: torch.Tensor, lengths1=None, lengths2=None, norm: int = 2, K: int = 1,
               return_nn: bool = False):
    if K != 1 or norm != 2:
        raise NotImplementedError("oracle restates K=1, squared-L2 only (all the reference uses)")
    if lengths1 is not None and not bool((lengths1 == p1.shape[1]).all()):
        raise NotImplementedError("heterogeneous lengths unused by the reference")
    if lengths2 is not None and not bool((lengths2 == p2.shape[1]).all()):
        raise NotImplementedError("heterogeneous lengths unused by the reference")
    dists, idx = _Knn1.apply(p1, p2)
    return _KNN(dists=dists[..., None], idx=idx[..., None], knn=None)


def _chamfer_single_direction(x, y, weights, batch_reduction, point_reduction):
    N, P1, _ = x.shape
    if weights is not None:
        if weights.size(0) != N:
            raise ValueError("weights must be of shape (N,).")
        if not (weights >= 0).all():
            raise ValueError("weights cannot be negative.")
        if weights.sum() == 0.0:
            weights = weights.view(N, 1)
            if batch_reduction in ["mean", "sum"]:
                return (x.sum((1, 2)) * weights).sum() * 0.0
            return (x.sum((1, 2)) * weights) * 0.0
    x_nn = knn_points(x, y, K=1)
    cham_x = x_nn.dists[..., 0]  # (N, P1)
    if weights is not None:
        cham_x = cham_x * weights.view(N, 1)
    if point_reduction is not None:
        cham_x = cham_x.sum(1)  # (N,)
        if point_reduction == "mean":
            cham_x = cham_x / float(max(P1, 1))
        if batch_reduction is not None:
            cham_x = cham_x.sum()
            if batch_reduction == "mean":
                div = weights.sum() if weights is not None else max(N, 1)
                cham_x = cham_x / div
    return cham_x


def chamfer_distance(x, y, x_lengths=None, y_lengths=None, x_normals=None, y_normals=None, weights=None,
                     batch_reduction: Optional[str] = "mean", point_reduction: Optional[str] = "mean",
                     norm: int = 2, single_directional: bool = False, abs_cosine: bool = True):
    """pytorch3d.loss.chamfer_distance (point clouds as padded tensors, no normals)."""
    if x_normals is not None or y_normals is not None:
        raise NotImplementedError
    if x.ndim != 3 or y.ndim != 3:
        raise ValueError("Expected points to be of shape (N, P, D)")
    cham_x = _chamfer_single_direction(x, y, weights, batch_reduction, point_reduction)
    if single_directional:
        return cham_x, None
    cham_y = _chamfer_single_direction(y, x, weights, batch_reduction, point_reduction)
    return cham_x + cham_y, None


# ----------------------------------------------------------------------------------------------
# rotation transforms
# ----------------------------------------------------------------------------------------------

def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def matrix_to_rotation_6d(matrix: torch.Tensor) -> torch.Tensor:
    batch_dim = matrix.size()[:-2]
    return matrix[..., :2, :].clone().reshape(batch_dim + (6,))


def axis_angle_to_quaternion(axis_angle: torch.Tensor) -> torch.Tensor:
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half_angles = angles * 0.5
    eps = 1e-6
    small_angles = angles.abs() < eps
    sin_half_angles_over_angles = torch.empty_like(angles)
    sin_half_angles_over_angles[~small_angles] = torch.sin(half_angles[~small_angles]) / angles[~small_angles]
    sin_half_angles_over_angles[small_angles] = 0.5 - (angles[small_angles] * angles[small_angles]) / 48
    return torch.cat([torch.cos(half_angles), axis_angle * sin_half_angles_over_angles], dim=-1)


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def axis_angle_to_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    return quaternion_to_matrix(axis_angle_to_quaternion(axis_angle))


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    ret = torch.zeros_like(x)
    positive_mask = x > 0
    ret[positive_mask] = torch.sqrt(x[positive_mask])
    return ret


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.matrix_to_quaternion (used off-path by multimodal.py:161-165 when Hz differ)."""
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    out = quat_candidates[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(batch_dim + (4,))
    return out


def so3_relative_angle(R1: torch.Tensor, R2: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """Angle of R1 R2^T (dead code on the shipped configs: optimization.py:204-211 results are unused)."""
    R12 = torch.bmm(R1, R2.permute(0, 2, 1))
    rot_trace = R12[:, 0, 0] + R12[:, 1, 1] + R12[:, 2, 2]
    phi_cos = ((rot_trace - 1.0) * 0.5).clamp(-1.0, 1.0)
    return torch.acos(phi_cos)
