"""Golden fixture of the barycentric marker placement and of the marker stage on a barycentric placement.
TEST INFRASTRUCTURE; runs ONLY in the build container.

Executes the reference's own `video_mocap.optimization.compute_nearest_points` with
`compute_locations.use_barycentric` (granularity "full" with the velocity factor, "marker" and "part", one case with
masked frames) and its own `optim_markers` on the resulting [M, 6890] matrix, over the oracle's restated third-party
primitives (oracle/shim/install.py; `igl.signed_distance`, `trimesh.Trimesh`, `trimesh.triangles.points_to_barycentric`
-> oracle/mesh_ref.py).  What this pins is the reference's window / granularity / scatter logic and its marker closure
on a three-corner placement; the mesh primitives themselves stay "parity unpinned" (see oracle/mesh_ref.py).  Stores
inputs and outputs as tests/golden/placement_barycentric.npz (data only).

    python -m oracle.make_golden_placement
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
from oracle.make_golden import RecordingLBFGS, small_config  # noqa: E402
from oracle.shim.install import install  # noqa: E402
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402


def sparse(mat: torch.Tensor):
    nz = torch.nonzero(mat)
    return nz.numpy().astype(np.int32), mat[nz[:, 0], nz[:, 1]].numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    tables = install(synthetic_smpl(0))
    sys.modules["igl"].signed_distance = mesh_ref.signed_distance
    sys.modules["trimesh"].Trimesh = mesh_ref.TrimeshRef
    sys.modules["trimesh"].triangles = sys.modules["trimesh.triangles"]
    sys.modules["trimesh.triangles"].points_to_barycentric = mesh_ref.points_to_barycentric
    import video_mocap.optimization as ref_opt
    from video_mocap.utils.smpl import SmplInference as RefSmplInference

    torch.optim.LBFGS = RecordingLBFGS
    smpl = RefSmplInference(torch.device("cpu"))
    g = np.load(os.path.join(GOLDEN, "chamfer_stage.npz"))  # the converged chamfer stage of the F=8, M=12 fixture
    t = lambda k: torch.from_numpy(np.asarray(g[k])).float()
    markers, o_pose, o_betas = t("markers"), t("hmr_pose_body"), t("o_betas")
    pose, betas, root, trans = t("out_pose_body"), t("out_betas"), t("out_root_orient"), t("out_trans")
    F, M = markers.shape[0], markers.shape[1]
    cfg = small_config()
    cfg["stages"]["compute_locations"].update(use_barycentric=True, use_mean=False)
    # per-frame labels that change over time; all < M: the reference's "part" branch indexes a [windows, M] table by the
    # joint id (optimization.py:578) and raises IndexError for a populated joint id >= M
    labels = (np.arange(M)[None, :] + (np.arange(F)[:, None] // 3)) % 10
    mask_some = torch.ones(F)
    mask_some[[2, 7]] = 0  # the last frame is masked: the last EXAMINED frame is 6
    out = {}
    for tag, gran, vel, mask in (("full", "full", True, torch.ones(F)), ("marker", "marker", False, mask_some),
                                 ("part", "part", False, mask_some)):
        mat = ref_opt.compute_nearest_points(
            markers=markers, pose_body=pose, betas=betas, root_orient=root, trans=trans, smpl_inference=smpl,
            marker_labels=labels, granularity=gran, img_mask=mask, device=torch.device("cpu"), config=cfg,
            o_pose_body=o_pose, window_size=1, use_velocity=vel)
        out[tag + "_nz"], out[tag + "_val"] = sparse(mat)
        out[tag + "_mask"] = mask.numpy()
        print(tag, "rows with weights", int((mat != 0).any(1).sum()), "nnz", int((mat != 0).sum()))
        if tag == "full":
            full = mat
    # marker stage on the barycentric placement
    RecordingLBFGS.records = []
    p_pose, p_root = pose.clone().requires_grad_(True), root.clone().requires_grad_(True)
    p_betas, p_trans = betas.clone().requires_grad_(True), trans.clone().requires_grad_(True)
    ref_opt.optim_markers(markers=markers, pose_body=p_pose, o_pose_body=o_pose, betas=p_betas, o_betas=o_betas,
                          root_orient=p_root, trans=p_trans, barycentric_coords_one_hot=full, img_mask=torch.ones(F),
                          smpl_inference=smpl, config=cfg)
    rec = RecordingLBFGS.records[-1]
    print("marker stage evals", len(rec["losses"]), rec["losses"][0], "->", rec["losses"][-1])
    np.savez_compressed(
        os.path.join(GOLDEN, "placement_barycentric.npz"), markers=markers.numpy(), o_pose_body=o_pose.numpy(),
        o_betas=o_betas.numpy(), in_pose_body=pose.numpy(), in_betas=betas.numpy(), in_root_orient=root.numpy(),
        in_trans=trans.numpy(), labels=labels, num_iters=cfg["stages"]["marker"]["num_iters"],
        losses=np.array(rec["losses"]), first_grad=rec["first_grad"], first_params=rec["first_params"],
        out_pose_body=p_pose.detach().numpy(), out_betas=p_betas.detach().numpy(),
        out_root_orient=p_root.detach().numpy(), out_trans=p_trans.detach().numpy(), **out)


if __name__ == "__main__":
    main()
