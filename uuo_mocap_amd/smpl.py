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
