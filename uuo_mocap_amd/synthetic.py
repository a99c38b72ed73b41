"""Deterministic synthetic mocap sequences (SURVEY.md 8d recipe) and the two input duck-types the
orchestrator consumes (reference img_smpl/img_smpl.py:26-31,100-132 ``ImgSmpl`` fields and
markers/markers.py:35-54 ``Markers`` accessors).

Data tooling, not the fitted path: ground-truth vertices come from a float64 numpy evaluation of the
SMPL equations so the generated inputs do not depend on either the HIP kernels or the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from .body_model import NUM_JOINTS, SmplTables, hash_normal, hash_uniform


def _rodrigues(aa: np.ndarray) -> np.ndarray:
    """axis-angle [...,3] -> rotation matrices [...,3,3] (float64)."""
    theta = np.linalg.norm(aa, axis=-1, keepdims=True)
    k = aa / np.maximum(theta, 1e-12)
    kx, ky, kz = k[..., 0], k[..., 1], k[..., 2]
    zero = np.zeros_like(kx)
    K = np.stack([zero, -kz, ky, kz, zero, -kx, -ky, kx, zero], axis=-1).reshape(aa.shape[:-1] + (3, 3))
    s = np.sin(theta)[..., None]
    c = np.cos(theta)[..., None]
    eye = np.broadcast_to(np.eye(3), K.shape)
    return eye + s * K + (1.0 - c) * (K @ K)


def lbs_f64(tables: SmplTables, rot: np.ndarray, betas: np.ndarray, trans: np.ndarray):
    """SMPL forward in float64 numpy. rot [F,24,3,3], betas [F,10] or [1,10], trans [F,3].
    Returns verts [F,V,3], joints [F,24,3], per-vertex blended rotations [F,V,3,3]."""
    F = rot.shape[0]
    vt = tables.v_template.astype(np.float64)
    S = tables.shapedirs.astype(np.float64)
    P = tables.posedirs.astype(np.float64)
    Jr = tables.J_regressor.astype(np.float64)
    W = tables.lbs_weights.astype(np.float64)
    parents = tables.parents
    betas = np.broadcast_to(betas.astype(np.float64), (F, 10))
    v_shaped = vt[None] + np.einsum("bl,mkl->bmk", betas, S)
    J = np.einsum("bik,ji->bjk", v_shaped, Jr)
    pf = (rot[:, 1:] - np.eye(3)).reshape(F, -1)
    v_posed = v_shaped + (pf @ P).reshape(F, -1, 3)
    G_R = np.zeros((F, NUM_JOINTS, 3, 3))
    G_t = np.zeros((F, NUM_JOINTS, 3))
    G_R[:, 0] = rot[:, 0]
    G_t[:, 0] = J[:, 0]
    for j in range(1, NUM_JOINTS):
        p = parents[j]
        G_R[:, j] = G_R[:, p] @ rot[:, j]
        G_t[:, j] = np.einsum("fab,fb->fa", G_R[:, p], J[:, j] - J[:, p]) + G_t[:, p]
    A_t = G_t - np.einsum("fjab,fjb->fja", G_R, J)
    T_R = np.einsum("vj,fjab->fvab", W, G_R)
    T_t = np.einsum("vj,fja->fva", W, A_t)
    verts = np.einsum("fvab,fvb->fva", T_R, v_posed) + T_t + trans[:, None, :]
    return verts, G_t + trans[:, None, :], T_R


def farthest_point_vertices(points: np.ndarray, count: int, start: int = 0) -> np.ndarray:
    chosen = [int(start)]
    d = np.linalg.norm(points - points[start], axis=1)
    for _ in range(count - 1):
        nxt = int(np.argmax(d))
        chosen.append(nxt)
        d = np.minimum(d, np.linalg.norm(points - points[nxt], axis=1))
    return np.array(chosen, dtype=np.int64)


class SyntheticMarkers:
    """``Markers`` duck-type (reference markers/markers.py:35-54)."""

    def __init__(self, points: np.ndarray, freq: float = 30.0):
        self._points = points
        self._freq = freq

    def get_points(self):
        return self._points

    def set_points(self, points):
        self._points = points

    def get_frequency(self):
        return self._freq


@dataclass
class SyntheticImgSmpl:
    """``ImgSmpl`` duck-type: the fields multimodal_video_mocap reads (reference multimodal.py:88-100)."""

    trans: torch.Tensor
    root_orient: torch.Tensor
    hmr_root_orient: torch.Tensor
    pose_body: torch.Tensor
    betas: torch.Tensor
    foot_contacts: torch.Tensor
    camera_bbox: torch.Tensor
    center: torch.Tensor
    scale: torch.Tensor
    size: torch.Tensor
    img_mask: torch.Tensor
    freq: float = 30.0


@dataclass
class SyntheticSequence:
    img_smpl: SyntheticImgSmpl
    markers: SyntheticMarkers
    gt: dict


def make_sequence(tables: SmplTables, seed: int = 0, num_frames: int = 300, num_markers: int = 50,
                  limb_only: bool = False, yaw_offset_deg: float = 100.0, dropout: float = 0.02,
                  hmr_pose_noise: float = 0.1, hmr_beta_noise: float = 0.5, subject_seed: int = None) -> SyntheticSequence:
    """One synthetic sequence: smooth GT motion, unlabeled-but-tracked markers 9.5 mm off the
    surface with 1 mm noise and block dropout, and an HMR stand-in (noisy pose/shape, wrong yaw).  `subject_seed` fixes the
    ground-truth shape independently of `seed`: sequences of ONE subject (the shared-betas extension fits them together)."""
    F, M = num_frames, num_markers
    s = 7919 * (seed + 1)
    t = np.arange(F, dtype=np.float64) / max(F, 1)

    # --- ground-truth motion
    amp = 0.6 * hash_uniform(s + 1, NUM_JOINTS, 3)
    amp[0] = 0.0
    amp[[10, 11, 22, 23]] *= 0.3  # feet / hands move less
    freq = 0.5 + 2.5 * hash_uniform(s + 2, NUM_JOINTS, 3)
    phase = 2.0 * np.pi * hash_uniform(s + 3, NUM_JOINTS, 3)
    aa = amp[None] * np.sin(2.0 * np.pi * freq[None] * t[:, None, None] + phase[None])  # [F,24,3]
    rot = _rodrigues(aa)
    yaw = 1.0 * np.sin(2.0 * np.pi * 0.5 * t + 2.0 * np.pi * hash_uniform(s + 4))
    up = _rodrigues(np.array([np.pi / 2.0, 0.0, 0.0]))  # SMPL y-up -> mocap z-up
    wobble = _rodrigues(0.15 * np.sin(2.0 * np.pi * t[:, None] * np.array([1.0, 1.5, 0.7])[None]
                                      + 2.0 * np.pi * hash_uniform(s + 5, 3)[None]))
    Rz = _rodrigues(np.stack([np.zeros(F), np.zeros(F), yaw], axis=-1))
    rot[:, 0] = Rz @ up[None] @ wobble
    steps = hash_normal(s + 6, F, 3) * 0.02
    walk = np.cumsum(steps, axis=0)
    k = np.exp(-0.5 * (np.arange(-15, 16) / 5.0) ** 2)
    k /= k.sum()
    walk = np.stack([np.convolve(np.pad(walk[:, a], 15, mode="edge"), k, mode="valid") for a in range(3)], axis=1)
    walk = np.clip(walk, -1.0, 1.0)
    trans = walk + np.array([0.0, 0.0, 0.95])[None]
    beta_gt = np.clip(hash_normal((s if subject_seed is None else 7919 * (int(subject_seed) + 1)) + 7, 10), -2.0, 2.0)[None]

    verts, joints, T_R = lbs_f64(tables, rot, beta_gt, trans)

    # --- markers: farthest-point vertex ids, 9.5 mm outward, 1 mm noise
    vt = tables.v_template.astype(np.float64)
    owner = np.argmax(tables.lbs_weights, axis=1)
    if limb_only:
        cand = np.where(np.isin(owner, [16, 18, 20, 22]))[0]  # left arm
    else:
        cand = np.arange(vt.shape[0])
    pick = cand[farthest_point_vertices(vt[cand], M, start=int(hash_uniform(s + 8) * len(cand)))]
    J0 = tables.J_regressor.astype(np.float64) @ vt
    ends = J0.copy()
    for j in range(NUM_JOINTS):
        kids = np.where(tables.parents == j)[0]
        ends[j] = J0[kids].mean(axis=0) if len(kids) else J0[j] + (J0[j] - J0[tables.parents[j]])
    a = J0[owner[pick]]
    b = ends[owner[pick]]
    ab = b - a
    tt = np.clip(np.sum((vt[pick] - a) * ab, axis=1) / np.maximum(np.sum(ab * ab, axis=1), 1e-12), 0.0, 1.0)
    out_dir = vt[pick] - (a + tt[:, None] * ab)
    out_dir /= np.maximum(np.linalg.norm(out_dir, axis=1, keepdims=True), 1e-9)
    out_world = np.einsum("fmab,mb->fma", T_R[:, pick], out_dir)
    markers = verts[:, pick] + 0.0095 * out_world + 0.001 * hash_normal(s + 9, F, M, 3)
    perm = np.argsort(hash_uniform(s + 10, M), kind="stable")
    markers = markers[:, perm]
    # block dropout: (marker, 10-frame block) zeroed (style of reference markers/markers_noise.py:39-66)
    nblocks = (F + 9) // 10
    drop = hash_uniform(s + 11, nblocks, M) < dropout
    drop_f = np.repeat(drop, 10, axis=0)[:F]
    markers = np.where(drop_f[..., None], 0.0, markers)

    # --- HMR stand-in
    noise_aa = hmr_pose_noise * hash_normal(s + 12, F, NUM_JOINTS, 3)
    hmr_rot = rot @ _rodrigues(noise_aa)
    yaw_off = _rodrigues(np.array([0.0, 0.0, np.deg2rad(yaw_offset_deg)]))
    hmr_root = yaw_off[None] @ hmr_rot[:, 0]
    hmr_betas = beta_gt + hmr_beta_noise * hash_normal(s + 13, F, 10)
    hmr_trans = trans + 0.05 * hash_normal(s + 14, F, 3)

    f32 = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    img = SyntheticImgSmpl(
        trans=f32(hmr_trans),
        root_orient=f32(hmr_root[:, None]),
        hmr_root_orient=f32(hmr_root[:, None]),
        pose_body=f32(hmr_rot[:, 1:]),
        betas=f32(hmr_betas),
        foot_contacts=torch.zeros(F, 2),
        camera_bbox=torch.zeros(F, 3),
        center=torch.zeros(F, 2),
        scale=torch.zeros(F, 1),
        size=torch.zeros(F, 2),
        img_mask=torch.ones(F, dtype=torch.bool),
        freq=30.0,
    )
    gt = {
        "rot": rot.astype(np.float32), "betas": beta_gt.astype(np.float32), "trans": trans.astype(np.float32),
        "verts": verts.astype(np.float32), "joints": joints.astype(np.float32),
        "marker_vids": pick[perm],
    }
    return SyntheticSequence(img_smpl=img, markers=SyntheticMarkers(markers.astype(np.float32), 30.0), gt=gt)


def synthetic_hmr_camera(num_frames: int, seed: int = 5):
    """A plausible HMR 2.0 weak-perspective camera for the reprojection stage: `camera_bbox` (scale, shift x, shift y)
    [F,3], bounding-box `center` [F,2] in pixels, image `size` (H, W) [F,2] and bounding-box `scale` [F,1]."""
    from .body_model import hash_uniform

    u = hash_uniform(seed, num_frames, 6)
    pred_cam = np.stack([0.85 + 0.2 * u[:, 0], 0.1 * (u[:, 1] - 0.5), 0.1 * (u[:, 2] - 0.5)], axis=1)
    center = np.stack([300.0 + 40 * u[:, 3], 220.0 + 30 * u[:, 4]], axis=1)
    size = np.tile(np.array([[480.0, 640.0]]), (num_frames, 1))
    scale = 0.35 + 0.1 * u[:, [5]]
    return [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in (pred_cam, center, size, scale)]
