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


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    ret = torch.zeros_like(x)
    positive = x > 0
    ret[positive] = torch.sqrt(x[positive])
    return ret


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """Rotation matrices [..., 3, 3] -> quaternions (r, i, j, k): the four |q_x| candidates, the best-conditioned
    one picked per matrix (pytorch3d 0.7.x semantics; no sign standardisation)."""
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    one_hot = torch.nn.functional.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    return quat_candidates[one_hot, :].reshape(batch_dim + (4,))


def quaternion_to_axis_angle(quaternions: torch.Tensor) -> torch.Tensor:
    norms = torch.norm(quaternions[..., 1:], p=2, dim=-1, keepdim=True)
    half_angles = torch.atan2(norms, quaternions[..., :1])
    angles = 2 * half_angles
    small = angles.abs() < 1e-6
    ratio = torch.empty_like(angles)
    ratio[~small] = torch.sin(half_angles[~small]) / angles[~small]
    ratio[small] = 0.5 - (angles[small] * angles[small]) / 48
    return quaternions[..., 1:] / ratio


def matrix_to_axis_angle(matrix: torch.Tensor) -> torch.Tensor:
    """as pytorch3d.transforms.matrix_to_axis_angle (used by the batch runner for the `poses` output, reference
    test/test.py:120)."""
    return quaternion_to_axis_angle(matrix_to_quaternion(matrix))
