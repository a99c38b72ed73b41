"""Video -> mocap frame-rate resampling of the HMR estimate (reference src/video_mocap/multimodal.py:145-182): output
frame i sits at video time i * f_video / f_mocap; translations and foot contacts are interpolated linearly, rotations by
spherical interpolation of their quaternions (the reference calls roma.utils.unitquat_slerp with shortest_arc; the
interpolation is independent of the quaternion component order, so pytorch3d's w-first quaternions pass through it
unchanged).  All frames are produced by a handful of batched tensor expressions on the device."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .transforms import matrix_to_quaternion


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.quaternion_to_matrix (real part first; the input need not be normalised)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def unitquat_slerp(q0: torch.Tensor, q1: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """Shortest-arc spherical interpolation, q0 / q1 [..., 4], t broadcastable to [...]: sin((1-t)w) q0 + sin(t w) q1
    normalised, with linear weights when the quaternions are closer than cos w > 1 - 1e-3 (roma's rule)."""
    cos_omega = (q0 * q1).sum(-1)
    q1 = torch.where((cos_omega < 0)[..., None], -q1, q1)
    cos_omega = cos_omega.abs()
    near = cos_omega > (1.0 - 1e-3)
    omega = torch.acos(torch.clamp(cos_omega, max=1.0))
    a = torch.where(near, 1 - t, torch.sin((1 - t) * omega))
    b = torch.where(near, t, torch.sin(t * omega))
    q = a[..., None] * q0 + b[..., None] * q1
    return q / q.norm(dim=-1, keepdim=True)


def resample_plan(num_frames: int, src_freq: float, dst_freq: float):
    """(frame [N] long, alpha [N] float64, interpolate [N] bool) exactly as the reference's loop computes them in
    Python floats (multimodal.py:151-155)."""
    n_new = round(num_frames * (dst_freq / src_freq))
    step = src_freq / dst_freq
    frame, alpha = [], []
    for i in range(n_new):
        fr = int(i * step)
        frame.append(fr)
        alpha.append(i * step - fr)
    frame = torch.tensor(frame, dtype=torch.long)
    alpha = torch.tensor(alpha, dtype=torch.float64)
    return frame, alpha, frame + 1 < num_frames


def resample_hmr(trans: torch.Tensor, root_orient: torch.Tensor, pose_body: torch.Tensor,
                 foot_contacts: Optional[torch.Tensor], src_freq: float, dst_freq: float
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """trans [F,3], root_orient [F,1,3,3], pose_body [F,23,3,3], foot_contacts [F,2] at `src_freq` (video) ->
    the same at `dst_freq` (mocap), round(F * dst/src) frames."""
    device = trans.device
    F = trans.shape[0]
    frame, alpha, interp = resample_plan(F, src_freq, dst_freq)
    frame, interp = frame.to(device), interp.to(device)
    nxt = torch.clamp(frame + 1, max=F - 1)
    a32 = alpha.to(device=device, dtype=trans.dtype)   # the reference multiplies fp32 tensors by the Python float
    a = torch.where(interp, a32, torch.zeros_like(a32))

    def lerp(x):
        w = a.reshape((-1,) + (1,) * (x.dim() - 1))
        return torch.where(interp.reshape(w.shape), x[nxt] * w + x[frame] * (1.0 - w), x[frame])

    def slerp(r):
        q = unitquat_slerp(matrix_to_quaternion(r[frame]), matrix_to_quaternion(r[nxt]),
                           a.reshape((-1,) + (1,) * (r.dim() - 3)))
        keep = interp.reshape((-1,) + (1,) * (r.dim() - 1))
        return torch.where(keep, quaternion_to_matrix(q), r[frame])

    out_fc = lerp(foot_contacts) if foot_contacts is not None else None
    return lerp(trans), slerp(root_orient), slerp(pose_body), out_fc
