"""``SmplInference`` with the reference's interface (reference src/video_mocap/utils/smpl.py:9-53), computing on
the MI355X through libuuo_hip.so."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .body_model import SmplTables, load_model
from .engine import DeviceModel


class _SmplForward(torch.autograd.Function):
    """Forward and backward on the GPU.  Gradients of the fit flow through the fused stage closures
    (engine.*Problem), never through this operator; a caller who
    differentiates the operator itself (own closure, own loss) gets the same sparse-gather backward kernel run over
    all vertices (uuo_smpl_backward)."""

    @staticmethod
    def forward(ctx, model: DeviceModel, poses, betas, root_orient, trans):
        verts, joints = model.smpl_forward(poses, betas, root_orient, trans, want_joints=True)
        ctx.model = model
        ctx.has_trans = trans is not None
        ctx.save_for_backward(poses, betas, root_orient, trans if trans is not None else poses.new_zeros(0))
        return joints, verts

    @staticmethod
    def backward(ctx, d_joints, d_verts):
        poses, betas, root_orient, trans = ctx.saved_tensors
        if d_joints is None and d_verts is None:
            return None, None, None, None, None
        g_poses, g_betas, g_root, g_trans = ctx.model.smpl_backward(
            poses, betas, root_orient, trans if ctx.has_trans else None, d_verts, d_joints)
        return (None, g_poses.to(poses.dtype), g_betas.to(betas.dtype), g_root.to(root_orient.dtype),
                g_trans.to(trans.dtype) if ctx.has_trans else None)


class SmplInference(nn.Module):
    def __init__(self, device=torch.device("cpu"), gender: str = "neutral", tables: Optional[SmplTables] = None):
        super().__init__()
        self.body_model_path = "./body_models/"
        self.device = torch.device(device)
        self.gender = gender
        if self.device.type != "cuda":
            raise RuntimeError("uuo_mocap_amd.SmplInference runs on the GPU only (device=%s); the reference's CPU "
                               "path is not reimplemented here" % (device,))
        self.tables = tables if tables is not None else load_model(self.body_model_path, gender)
        self.device_model = DeviceModel(self.tables, self.device)
        parents = torch.from_numpy(np.asarray(self.tables.parents)).long().clone()
        parents[0] = -1
        # the attributes callers reach through `.smpl` (multimodal.py:348, optimization.py:92,431,490)
        self.smpl = SimpleNamespace(
            faces=np.asarray(self.tables.faces),
            parents=parents.to(self.device),
            lbs_weights=torch.from_numpy(self.tables.lbs_weights).float().to(self.device),
        )

    def forward(self, poses: torch.Tensor, betas: torch.Tensor, root_orient: torch.Tensor,
                trans: torch.Tensor) -> Dict:
        if betas.shape[1] != 10:
            raise ValueError("Betas array must have 10 beta values")
        needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                     for t in (poses, betas, root_orient, trans))
        if needs_grad:
            joints, verts = _SmplForward.apply(self.device_model, poses, betas, root_orient, trans)
        else:
            verts, joints = self.device_model.smpl_forward(poses, betas, root_orient, trans)
        return {"joints": joints, "vertices": verts}

    def get_lbs_weights(self):
        return self.smpl.lbs_weights


def batch_rodrigues(rot_vecs: torch.Tensor) -> torch.Tensor:
    """smplx.lbs.batch_rodrigues (what smplx applies to axis-angle inputs when pose2rot=True): [N,3] -> [N,3,3] with
    angle = |r + 1e-8|, R = I + sin(angle) K + (1 - cos(angle)) K^2."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros_like(rx)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(-1, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device).unsqueeze(dim=0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


class SmplInferenceGender(nn.Module):
    """reference src/video_mocap/utils/smpl.py:56-131: a male and a female SMPL evaluated on the same parameters and
    blended by `gender_one_hot` [N,2] (soft genders allowed: the blend is linear in the two outputs).  Used by the
    reference's dataset / synthetic-marker tooling, not by the fit itself; BASELINE configs[4] names
    "mixed male/female SMPL".  Both models live on the GPU (two uuo_model_t table sets); every forward is the HIP path
    of SmplInference, differentiable the same way.  `tables` = (male, female) SmplTables; by default the licensed
    ./body_models/smpl/SMPL_{MALE,FEMALE}.pkl, else two synthetic SMPL-shaped models."""

    def __init__(self, device=torch.device("cpu"), tables=None):
        super().__init__()
        self.body_model_path = "./body_models/"
        self.device = torch.device(device)
        if tables is None:
            tables = (load_model(self.body_model_path, "male"), load_model(self.body_model_path, "female"))
        self.smpls = {"male": SmplInference(self.device, "male", tables=tables[0]),
                      "female": SmplInference(self.device, "female", tables=tables[1])}

    def forward(self, poses, betas, root_orient, trans, gender_one_hot, pose2rot: bool = True,
                compute_part_labels: bool = False) -> Dict:
        if betas.shape[1] != 10:
            raise ValueError("Betas array must have 10 beta values")
        if len(gender_one_hot.shape) != 2:
            raise ValueError("Gender one-hot vector must have 2 dimensions")
        batch_size, num_frames, _ = trans.shape
        if pose2rot:  # axis-angle [N,F,69] / [N,F,3]
            pose_m = batch_rodrigues(poses.reshape(-1, 3)).reshape(-1, 23, 3, 3)
            root_m = batch_rodrigues(root_orient.reshape(-1, 3)).reshape(-1, 1, 3, 3)
        else:         # rotation matrices [N,F,23,3,3] / [N,F,3,3]
            pose_m = poses.reshape(-1, 23, 3, 3)
            root_m = root_orient.reshape(-1, 1, 3, 3)
        trans_rs = trans.reshape(-1, 3)
        # the reference repeats betas FRAME-major ([F,N,10] flattened) although poses are sequence-major ([N,F,...]):
        # rows agree only for N = 1; reproduced as written (smpl.py:100-101)
        betas_rs = torch.repeat_interleave(torch.unsqueeze(betas, dim=0), dim=0, repeats=num_frames).reshape(-1, betas.shape[-1])
        g = torch.repeat_interleave(torch.unsqueeze(gender_one_hot, dim=1), dim=1, repeats=num_frames).reshape(-1, 2, 1)
        out = {k: m(pose_m, betas_rs, root_m, trans_rs) for k, m in self.smpls.items()}
        joints = out["male"]["joints"][:, :24] * g[:, [0], :] + out["female"]["joints"][:, :24] * g[:, [1], :]
        vertices = out["male"]["vertices"] * g[:, [0], :] + out["female"]["vertices"] * g[:, [1], :]
        output = {"joints": joints.reshape(batch_size, num_frames, 24, 3),
                  "vertices": vertices.reshape(batch_size, num_frames, -1, 3)}
        if compute_part_labels:
            labels = self.smpls["male"].smpl.lbs_weights * g[[0], [0], :] + self.smpls["female"].smpl.lbs_weights * g[[0], [1], :]
            output["vertex_part_labels"] = torch.repeat_interleave(torch.unsqueeze(labels, 0), repeats=batch_size, dim=0)
        return output
