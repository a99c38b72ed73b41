"""Fixture for the HMR-track ingestion.  TEST INFRASTRUCTURE; build container only.

    python -m oracle.make_golden_ingest

Feeds the reference's OWN ``ImgSmpl`` class (/root/reference/src/video_mocap/img_smpl/img_smpl.py:12-132, with its own
``get_foot_contacts``) a synthetic 4D-Humans style per-frame dictionary with missing detections at the start, in the middle
(two gaps of different lengths) and at the end, and stores the dictionary's arrays plus every attribute the class produces
in ``tests/golden/ingest_img_smpl.npz``.  The rotation conversions / slerp the class imports (pytorch3d, roma) are the
oracle's restatements (oracle/shim/install.py; ``roma.utils.unitquat_slerp`` -> oracle/stages_ref.py).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import p3d_ref, stages_ref  # noqa: E402
from oracle.shim.install import install  # noqa: E402

F_ = 16
MISSING = [0, 1, 5, 9, 10, 11, 15]


def make_data(seed=3):
    g = torch.Generator().manual_seed(seed)
    rot = p3d_ref.rotation_6d_to_matrix(torch.randn(F_, 24, 6, generator=g)).numpy().astype(np.float32)
    betas = torch.randn(F_, 10, generator=g).numpy().astype(np.float32)
    j3d = torch.randn(F_, 45, 3, generator=g).numpy().astype(np.float32)
    j2d = (torch.rand(F_, 45, 2, generator=g) * 200 + 100).numpy().astype(np.float32)
    j2d[6:9, 19:24] = j2d[5:6, 19:24]  # still toes over a few frames -> foot contacts
    cam = torch.rand(F_, 3, generator=g).numpy().astype(np.float32)
    center = (torch.rand(F_, 2, generator=g) * 300).numpy().astype(np.float32)
    scale = (torch.rand(F_, generator=g) + 0.3).numpy().astype(np.float32)
    size = np.tile(np.array([[480.0, 640.0]], np.float32), (F_, 1))
    data = {}
    for f in range(F_):
        key = "frame_%05d.jpg" % f
        if f in MISSING:
            data[key] = {"tracked_ids": [], "smpl": [], "3d_joints": [], "camera_bbox": [], "center": [], "scale": [],
                         "size": [], "2d_joints": []}
        else:
            data[key] = {"tracked_ids": [1], "smpl": [{"global_orient": rot[f, :1], "body_pose": rot[f, 1:],
                                                        "betas": betas[f]}],
                         "3d_joints": [j3d[f]], "camera_bbox": [cam[f]], "center": [center[f]], "scale": [scale[f]],
                         "size": [size[f]], "2d_joints": [j2d[f].reshape(-1)]}
    arrays = {"rot": rot, "betas": betas, "j3d": j3d, "j2d": j2d, "cam": cam, "center": center, "scale": scale,
              "size": size, "missing": np.array(MISSING)}
    return data, arrays


def main():
    install()
    sys.modules["roma.utils"].unitquat_slerp = stages_ref.unitquat_slerp
    sys.modules["roma"].utils = sys.modules["roma.utils"]
    from video_mocap.img_smpl.img_smpl import ImgSmpl as RefImgSmpl

    data, arrays = make_data()
    ref = RefImgSmpl(data, 30.0)
    out = {"out_" + k: getattr(ref, k).numpy() for k in ("trans", "root_orient", "hmr_root_orient", "pose_body", "betas",
                                                          "camera_bbox", "center", "scale", "size", "foot_contacts")}
    out["out_img_mask"] = ref.img_mask.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ingest_img_smpl.npz"), freq=30.0, **arrays, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
