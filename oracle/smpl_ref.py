"""CPU restatement of smplx's SMPL forward (linear blend skinning).  TEST INFRASTRUCTURE (oracle/__init__.py).

Restates the published ``smplx.lbs.lbs`` / ``batch_rigid_transform`` / ``blend_shapes`` /
``vertices2joints`` and ``smplx.SMPL.forward`` (package is an empty submodule in the reference:
.gitmodules:4-6), anchored on the reference call sites utils/smpl.py:22-27 (``smplx.create(...,
model_type="smpl", gender="neutral", batch_size=1)``) and utils/smpl.py:39-45 (``pose2rot=False``).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def blend_shapes(betas: torch.Tensor, shape_disps: torch.Tensor) -> torch.Tensor:
    return torch.einsum("bl,mkl->bmk", [betas, shape_disps])


def vertices2joints(J_regressor: torch.Tensor, vertices: torch.Tensor) -> torch.Tensor:
    return torch.einsum("bik,ji->bjk", [vertices, J_regressor])


def transform_mat(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    return torch.cat([F.pad(R, [0, 0, 0, 1]), F.pad(t, [0, 0, 0, 1], value=1)], dim=2)


def batch_rigid_transform(rot_mats, joints, parents):
    joints = torch.unsqueeze(joints, dim=-1)
    rel_joints = joints.clone()
    rel_joints[:, 1:] -= joints[:, parents[1:]]
    transforms_mat = transform_mat(rot_mats.reshape(-1, 3, 3), rel_joints.reshape(-1, 3, 1)).reshape(
        -1, joints.shape[1], 4, 4)
    transform_chain = [transforms_mat[:, 0]]
    for i in range(1, parents.shape[0]):
        curr_res = torch.matmul(transform_chain[parents[i]], transforms_mat[:, i])
        transform_chain.append(curr_res)
    transforms = torch.stack(transform_chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_homogen = F.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - F.pad(torch.matmul(transforms, joints_homogen), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, pose2rot: bool = False):
    """smplx.lbs.lbs with pose2rot=False (pose holds rotation matrices [B,24,3,3])."""
    if pose2rot:
        raise NotImplementedError("the reference always passes pose2rot=False (utils/smpl.py:44)")
    batch_size = max(betas.shape[0], pose.shape[0])
    dtype = betas.dtype
    v_shaped = v_template + blend_shapes(betas, shapedirs)
    J = vertices2joints(J_regressor, v_shaped)
    ident = torch.eye(3, dtype=dtype)
    pose_feature = pose[:, 1:].view(batch_size, -1, 3, 3) - ident
    rot_mats = pose.view(batch_size, -1, 3, 3)
    pose_offsets = torch.matmul(pose_feature.view(batch_size, -1), posedirs).view(batch_size, -1, 3)
    v_posed = pose_offsets + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    W = lbs_weights.unsqueeze(dim=0).expand([batch_size, -1, -1])
    num_joints = J_regressor.shape[0]
    T = torch.matmul(W, A.view(batch_size, num_joints, 16)).view(batch_size, -1, 4, 4)
    homogen_coord = torch.ones([batch_size, v_posed.shape[1], 1], dtype=dtype)
    v_posed_homo = torch.cat([v_posed, homogen_coord], dim=2)
    v_homo = torch.matmul(T, torch.unsqueeze(v_posed_homo, dim=-1))
    verts = v_homo[:, :, :3, 0]
    return verts, J_transformed


class SMPLRef(nn.Module):
    """What ``smplx.create(path, model_type="smpl", gender="neutral", batch_size=1)`` returns, as far as
    the reference touches it: ``forward(...)`` -> ``.vertices`` / ``.joints`` (45), ``.faces``,
    ``.parents``, ``.lbs_weights``."""

    def __init__(self, tables):
        super().__init__()
        self.faces = np.asarray(tables.faces)
        self.register_buffer("v_template", torch.from_numpy(tables.v_template).float())
        self.register_buffer("shapedirs", torch.from_numpy(tables.shapedirs).float())
        self.register_buffer("posedirs", torch.from_numpy(tables.posedirs).float())
        self.register_buffer("J_regressor", torch.from_numpy(tables.J_regressor).float())
        parents = torch.from_numpy(np.asarray(tables.parents)).long().clone()
        parents[0] = -1
        self.register_buffer("parents", parents)
        self.register_buffer("lbs_weights", torch.from_numpy(tables.lbs_weights).float())
        self.register_buffer("extra_joints_idxs", torch.from_numpy(np.asarray(tables.extra_joint_vids)).long())

    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, pose2rot: bool = False,
                **kwargs):
        batch_size = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
        if betas.shape[0] != batch_size:
            betas = betas.expand(int(batch_size / betas.shape[0]), -1)
        full_pose = torch.cat([global_orient.reshape(-1, 1, 3, 3), body_pose.reshape(-1, 23, 3, 3)], dim=1)
        vertices, joints = lbs(betas, full_pose, self.v_template, self.shapedirs, self.posedirs,
                               self.J_regressor, self.parents, self.lbs_weights, pose2rot=False)
        extra = torch.index_select(vertices, 1, self.extra_joints_idxs)
        joints = torch.cat([joints, extra], dim=1)
        if transl is not None:
            joints = joints + transl.unsqueeze(dim=1)
            vertices = vertices + transl.unsqueeze(dim=1)
        return SimpleNamespace(vertices=vertices, joints=joints, betas=betas, body_pose=body_pose,
                               global_orient=global_orient, full_pose=full_pose)


class SmplInferenceRef(nn.Module):
    """Restates reference utils/smpl.py:9-53 (SmplInference) over SMPLRef."""

    def __init__(self, tables, device=torch.device("cpu"), gender="neutral"):
        super().__init__()
        self.device = device
        self.gender = gender
        self.smpl = SMPLRef(tables).to(device)

    def forward(self, poses, betas, root_orient, trans):
        if betas.shape[1] != 10:
            raise ValueError("Betas array must have 10 beta values")
        out = self.smpl(body_pose=poses, betas=betas, global_orient=root_orient, transl=trans, pose2rot=False)
        return {"joints": out.joints, "vertices": out.vertices}

    def get_lbs_weights(self):
        return self.smpl.lbs_weights


def batch_rodrigues(rot_vecs: torch.Tensor, epsilon: float = 1e-8) -> torch.Tensor:
    """smplx.lbs.batch_rodrigues (published algorithm)."""
    batch_size = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((batch_size, 1), dtype=rot_vecs.dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view((batch_size, 3, 3))
    ident = torch.eye(3, dtype=rot_vecs.dtype).unsqueeze(dim=0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


class SmplInferenceGenderRef(nn.Module):
    """Restates reference utils/smpl.py:56-131 (SmplInferenceGender) over two SMPLRef models."""

    def __init__(self, tables_male, tables_female):
        super().__init__()
        self.smpls = {"male": SMPLRef(tables_male), "female": SMPLRef(tables_female)}

    def forward(self, poses, betas, root_orient, trans, gender_one_hot, pose2rot: bool = True,
                compute_part_labels: bool = False):
        if betas.shape[1] != 10:
            raise ValueError("Betas array must have 10 beta values")
        if len(gender_one_hot.shape) != 2:
            raise ValueError("Gender one-hot vector must have 2 dimensions")
        batch_size, num_frames, _ = trans.shape
        poses_rs = torch.reshape(poses, (-1, poses.shape[-1]))
        root_rs = torch.reshape(root_orient, (-1, root_orient.shape[-1]))
        trans_rs = torch.reshape(trans, (-1, trans.shape[-1]))
        betas_rs = torch.repeat_interleave(torch.unsqueeze(betas, dim=0), dim=0, repeats=num_frames)
        betas_rs = torch.reshape(betas_rs, (-1, betas.shape[-1]))
        g = torch.repeat_interleave(torch.unsqueeze(gender_one_hot, dim=1), dim=1, repeats=num_frames)
        g = torch.reshape(g, (-1, gender_one_hot.shape[-1], 1))
        out = {}
        for gender in ["male", "female"]:
            if pose2rot:  # what smplx.SMPL.forward does with axis-angle input before lbs
                body = batch_rodrigues(poses_rs.reshape(-1, 3)).reshape(-1, 23, 3, 3)
                glob = batch_rodrigues(root_rs.reshape(-1, 3)).reshape(-1, 1, 3, 3)
            else:
                body = poses.reshape(-1, 23, 3, 3)
                glob = root_orient.reshape(-1, 1, 3, 3)
            out[gender] = self.smpls[gender](body_pose=body, betas=betas_rs, global_orient=glob, transl=trans_rs)
        joints = out["male"].joints[:, :24] * g[:, [0], :] + out["female"].joints[:, :24] * g[:, [1], :]
        vertices = out["male"].vertices * g[:, [0], :] + out["female"].vertices * g[:, [1], :]
        output = {"joints": torch.reshape(joints, (batch_size, num_frames, 24, 3)),
                  "vertices": torch.reshape(vertices, (batch_size, num_frames, 6890, 3))}
        if compute_part_labels:
            lab = self.smpls["male"].lbs_weights * g[[0], [0], :] + self.smpls["female"].lbs_weights * g[[0], [1], :]
            output["vertex_part_labels"] = torch.repeat_interleave(torch.unsqueeze(lab, 0), repeats=batch_size, dim=0)
        return output
