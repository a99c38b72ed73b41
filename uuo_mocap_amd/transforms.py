"""Rotation helpers used around (not inside) the fitted path: output normalisation and the in-place root update
after a solve (reference multimodal.py:34-35, optimization.py:280-285,662-679).  Same definitions as the
pytorch3d functions the reference imports; the per-closure versions of these run inside the HIP kernels."""
from __future__ import annotations

import torch


def matrix_to_rotation_6d(matrix: torch.Tensor) -> torch.Tensor:
    return matrix[..., :2, :].clone().reshape(matrix.shape[:-2] + (6,))


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = a1 / a1.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    u2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = u2 / u2.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def normalize_rot(rot: torch.Tensor) -> torch.Tensor:
    return rotation_6d_to_matrix(matrix_to_rotation_6d(rot))


def axis_angle_to_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    """quaternion route with the small-angle series, as pytorch3d.transforms.axis_angle_to_matrix."""
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    ratio = torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half) / safe)
    quat = torch.cat([torch.cos(half), axis_angle * ratio], dim=-1)
    r, i, j, k = torch.unbind(quat, -1)
    two_s = 2.0 / (quat * quat).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quat.shape[:-1] + (3, 3))


def _axis_rotation(angle: torch.Tensor, axis: int) -> torch.Tensor:
    vec = torch.zeros(list(angle.shape[:-1]) + [3], device=angle.device, dtype=angle.dtype)
    vec[..., [axis]] = angle
    return axis_angle_to_matrix(vec)


def compute_root_orient_z(angle: torch.Tensor) -> torch.Tensor:
    """[..., J, 1] yaw angles -> [..., J, 3, 3] (reference optimization.py:672-679)."""
    return _axis_rotation(angle, 2)


def compute_root_orient_y(angle: torch.Tensor) -> torch.Tensor:
    """reference optimization.py:662-669."""
    return _axis_rotation(angle, 1)
