"""Golden fixture of the reprojection closure AT THE BASELINE SIZE (300 frames x 50 markers).  TEST INFRASTRUCTURE; runs ONLY
in the build container (about two minutes of CPU).

Executes the reference's own `video_mocap.utils.hmr_utils.optim_reprojection` (over the oracle's restated third-party
primitives, oracle/shim/install.py) for both yaw hypotheses of the small fixture's set-up, 20 L-BFGS iterations each, and
stores what its torch.optim.LBFGS saw: the first parameter vector, the first gradient and the loss of every closure
evaluation -> tests/golden/reprojection_stage_300x50.npz.  The inputs are the deterministic synthetic sequence
`make_sequence(synthetic_smpl(0), seed=0, 300, 50)` with `synthetic_hmr_camera(300)`; a checksum of the marker array guards
the generator.

    python -m oracle.make_golden_reprojection_full
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle.make_golden import RecordingLBFGS  # noqa: E402
from oracle.shim.install import install  # noqa: E402
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence, synthetic_hmr_camera  # noqa: E402


def main():
    torch.manual_seed(0)
    tables = install(synthetic_smpl(0))
    import video_mocap.utils.hmr_utils as ref_hmr
    from video_mocap.utils.smpl import SmplInference as RefSmplInference

    torch.optim.LBFGS = RecordingLBFGS
    F, M, seed, iters = 300, 50, 0, 20
    seq = make_sequence(tables, seed=seed, num_frames=F, num_markers=M)
    smpl = RefSmplInference(torch.device("cpu"))
    cfg = packaged_config("video_mocap")
    cfg["stages"]["reprojection_part"]["num_iters"] = iters
    markers = torch.from_numpy(seq.markers.get_points()).float()
    img = seq.img_smpl
    betas = (img.betas.sum(0, keepdim=True) / img.img_mask.sum()).clone()
    trans = torch.median(markers, dim=1)[0].clone()
    pred_cam, center, size, scale = synthetic_hmr_camera(F)
    out = {}
    for name, angle in (("a0", 0.0), ("a1", float(np.pi / 2))):
        RecordingLBFGS.records.clear()
        t0 = time.time()
        r = ref_hmr.optim_reprojection(
            markers=markers, pose_body=img.pose_body.clone(), betas=betas.clone().requires_grad_(True),
            hmr_betas=img.betas.clone(), root_orient=img.hmr_root_orient.clone(), trans=trans.clone().requires_grad_(True),
            pred_cam=pred_cam, cam_center=center, cam_size=size, cam_scale=scale, angle=torch.tensor(angle),
            img_mask=img.img_mask, smpl_inference=smpl, num_iters=iters, config=cfg, verbose=False, iter_fn=None)
        rec = RecordingLBFGS.records[-1]
        out.update({name + "_losses": np.array(rec["losses"], np.float64), name + "_first_grad": rec["first_grad"],
                    name + "_first_params": rec["first_params"],
                    name + "_angles": np.array([r["input_angle"], r["output_angle"]]),
                    name + "_metrics": np.array([r["metrics"]["chamfer"], r["metrics"]["reproject"]])})
        print(name, "evals", len(rec["losses"]), "loss", rec["losses"][0], "->", rec["losses"][-1], r["metrics"],
              "%.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(GOLDEN, "reprojection_stage_300x50.npz"), seed=seed, F=F, M=M, num_iters=iters,
                        markers_checksum=np.float64(np.abs(markers.double().numpy()).sum()), **out)


if __name__ == "__main__":
    main()
