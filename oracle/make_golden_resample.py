"""Golden fixture of the video -> mocap frame-rate resampling.  TEST INFRASTRUCTURE; runs ONLY in the build container.

Executes the reference's own `video_mocap.multimodal.multimodal_video_mocap` (hmr_full.yaml budgets, over the oracle's
restated third-party primitives, oracle/shim/install.py; `roma.utils.unitquat_slerp` -> oracle/stages_ref.py) on a
synthetic sequence whose HMR track runs at 15 Hz under 30 Hz markers, and stores inputs and outputs as
tests/golden/e2e_resample.npz (data only).  The output `pose_body` is the normalised RESAMPLED HMR pose, so the fixture
pins the reference's resampling loop (frame / alpha arithmetic, which fields are interpolated, the tail rule).

    python -m oracle.make_golden_resample
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import stages_ref  # noqa: E402
from oracle.make_golden import RecordingLBFGS, seq_inputs, small_config  # noqa: E402
from oracle.shim.install import install  # noqa: E402
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.synthetic import SyntheticImgSmpl, make_sequence  # noqa: E402


def half_rate(img: SyntheticImgSmpl) -> SyntheticImgSmpl:
    """Every second frame of the HMR track, declared at half the frame rate."""
    f = {k: getattr(img, k) for k in ("trans", "root_orient", "hmr_root_orient", "pose_body", "betas", "foot_contacts",
                                      "camera_bbox", "center", "scale", "size", "img_mask")}
    return SyntheticImgSmpl(**{k: v[::2].clone() for k, v in f.items()}, freq=img.freq / 2)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    tables = install(synthetic_smpl(0))
    sys.modules["roma.utils"].unitquat_slerp = stages_ref.unitquat_slerp
    sys.modules["roma"].utils = sys.modules["roma.utils"]
    import video_mocap.multimodal as ref_mm

    torch.optim.LBFGS = RecordingLBFGS
    cfg = small_config("hmr_full")
    seq = make_sequence(tables, seed=3, num_frames=9, num_markers=12)
    img = half_rate(seq.img_smpl)  # 5 video frames at 15 Hz -> round(5 * 2) = 10 frames, cut to the 9 marker frames
    RecordingLBFGS.records = []
    out = ref_mm.multimodal_video_mocap(img, copy.deepcopy(seq.markers), torch.device("cpu"), cfg, offset=0,
                                        print_options=[], save_stages=True)
    recs = RecordingLBFGS.records
    print("frames", out["trans"].shape[0], "solves", len(recs), "evals", [len(r["losses"]) for r in recs])
    inp = seq_inputs(seq)
    np.savez_compressed(
        os.path.join(GOLDEN, "e2e_resample.npz"), markers=inp["markers"], video_freq=img.freq, mocap_freq=30.0,
        hmr_trans=img.trans.numpy(), hmr_root_orient=img.root_orient.numpy(), hmr_pose_body=img.pose_body.numpy(),
        hmr_betas=img.betas.numpy(), img_mask=img.img_mask.numpy(), part_iters=cfg["stages"]["part"]["num_iters"],
        n_evals=np.array([len(r["losses"]) for r in recs]), first_losses=np.array([r["losses"][0] for r in recs]),
        final_losses=np.array([r["losses"][-1] for r in recs]),
        out_trans=out["trans"].numpy(), out_root_orient=out["root_orient"].numpy(),
        out_pose_body=out["pose_body"].numpy(), out_betas=out["betas"].numpy())


if __name__ == "__main__":
    main()
