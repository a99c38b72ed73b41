"""Golden fixture of the evaluation metrics.  TEST INFRASTRUCTURE; runs ONLY in the build container.

Executes the reference's own `video_mocap.evaluation.metrics` functions (pure torch, except the marker-to-surface
distance whose `igl.signed_distance` is routed to oracle/mesh_ref.py by the import shim) on seeded random inputs and
stores inputs and outputs as tests/golden/metrics.npz (data only).

    python -m oracle.make_golden_metrics
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import mesh_ref  # noqa: E402
from oracle.shim.install import _mod, install  # noqa: E402
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402


def main():
    torch.manual_seed(0)
    tables = install(synthetic_smpl(0))
    for name in ("pytorch3d.structures", "pytorch3d.loss.point_mesh_distance", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            _mod(name)
    sys.modules["pytorch3d.loss.point_mesh_distance"]._DEFAULT_MIN_TRIANGLE_AREA = 5e-3
    sys.modules["igl"].signed_distance = mesh_ref.signed_distance
    import video_mocap.evaluation.metrics as ref

    g = torch.Generator().manual_seed(5)
    F, J, V, NF, M = 6, 24, 200, 350, 9
    gt = torch.randn(F, J, 3, generator=g)
    q, _ = torch.linalg.qr(torch.randn(F, 3, 3, generator=g))
    pred = (gt @ q) * (1.0 + 0.1 * torch.randn(F, 1, 1, generator=g)) + 0.05 * torch.randn(F, J, 3, generator=g) + \
        torch.randn(F, 1, 3, generator=g)
    pv, gv = torch.randn(F, V, 3, generator=g), torch.randn(F, V, 3, generator=g)
    faces = torch.randint(0, V, (NF, 3), generator=g)
    markers = torch.randn(F, M, 3, generator=g) * 1.2
    ids = [1, 4, 7, 10, 20]
    freq = 30.0
    out = {
        "m2s": ref.compute_marker_to_surface_distance(gv, faces[None].repeat(F, 1, 1), markers),
        "mpjpe": ref.compute_MPJPE(pred, gt), "mpjpe_joints": ref.compute_MPJPE_joints(pred, gt, ids),
        "mpjve": ref.compute_MPJVE(pred, gt, freq), "mpjve_joints": ref.compute_MPJVE_joints(pred, gt, freq, ids),
        "pa_mpjpe": ref.compute_PA_MPJPE(pred, gt), "pa_mpjpe_joints": ref.compute_PA_MPJPE_joints(pred, gt, ids),
        "pa_mpjve": ref.compute_PA_MPJVE(pred, gt, freq),
        "pa_mpjve_joints": ref.compute_PA_MPJVE_joints(pred, gt, freq, ids), "v2v": ref.compute_V2V(pv, gv),
        "aligned": ref.compute_similarity_transform(pred, gt),
    }
    for k, v in out.items():
        print(k, v if v.numel() == 1 else tuple(v.shape))
    np.savez_compressed(os.path.join(GOLDEN, "metrics.npz"), pred=pred.numpy(), gt=gt.numpy(), pred_verts=pv.numpy(),
                        gt_verts=gv.numpy(), faces=faces.numpy(), markers=markers.numpy(), joint_ids=np.array(ids),
                        freq=freq, **{"out_" + k: v.numpy() for k, v in out.items()})


if __name__ == "__main__":
    main()
