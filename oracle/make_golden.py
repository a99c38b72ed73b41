"""Golden-fixture generator.  TEST INFRASTRUCTURE; runs ONLY in the build container.

Executes the reference's *own* Python modules (/root/reference/src/video_mocap/{optimization,
markers/markers_utils,multimodal,losses/*,utils/*}.py) over the oracle's restated third-party
primitives (oracle/shim/install.py) on tiny synthetic inputs and stores inputs + expected outputs
as ``tests/golden/*.npz``.  The fixtures are data only (arrays), never reference text.

    python -m oracle.make_golden            # rewrites tests/golden/

Tests then check (a) oracle/stages_ref.py reproduces the reference's orchestration (CPU, -m "not gpu")
and (b) the HIP path agrees with both (-m gpu).
"""
from __future__ import annotations

import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle.shim.install import install  # noqa: E402
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence  # noqa: E402


class RecordingLBFGS(torch.optim.LBFGS):
    """torch.optim.LBFGS that records every closure evaluation (loss, flat grad of the first one)."""

    records = []

    def step(self, closure):
        rec = {"losses": [], "first_grad": None, "first_params": None}
        RecordingLBFGS.records.append(rec)
        params = self.param_groups[0]["params"]

        def wrapped():
            if rec["first_params"] is None:
                rec["first_params"] = torch.cat([p.detach().reshape(-1) for p in params]).numpy().copy()
            loss = closure()
            rec["losses"].append(float(loss))
            if rec["first_grad"] is None:
                rec["first_grad"] = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                               for p in params]).numpy().copy()
            return loss

        return super().step(wrapped)


def small_config(name="video_mocap", part=10000, chamfer=10000, marker=10000):
    """The shipped iteration budgets (10000): at the fixture sizes every solve converges in < 500 closure
    evaluations, so the recorded outputs are converged quantities, not mid-flight iterates."""
    cfg = packaged_config(name)
    if cfg["stages"]["part"]["num_iters"] > 0:
        cfg["stages"]["part"]["num_iters"] = part
    if cfg["stages"]["chamfer"]["num_iters"] > 0:
        cfg["stages"]["chamfer"]["num_iters"] = chamfer
    if cfg["stages"]["marker"]["num_iters"] > 0:
        cfg["stages"]["marker"]["num_iters"] = marker
    return cfg


def seq_inputs(seq):
    img = seq.img_smpl
    return {
        "markers": seq.markers.get_points().copy(),
        "hmr_trans": img.trans.numpy(), "hmr_root_orient": img.root_orient.numpy(),
        "hmr_pose_body": img.pose_body.numpy(), "hmr_betas": img.betas.numpy(),
        "img_mask": img.img_mask.numpy(),
    }


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(1)  # fixed reduction order for the recorded trajectories
    tables = install(synthetic_smpl(0))
    import video_mocap.losses.chamfer_distance as ref_cd
    import video_mocap.losses.losses as ref_losses
    import video_mocap.markers.markers_utils as ref_mu
    import video_mocap.multimodal as ref_mm
    import video_mocap.optimization as ref_opt
    import video_mocap.utils.points as ref_points
    from video_mocap.utils.settings import MARKER_DISTANCE
    from video_mocap.utils.smpl import SmplInference as RefSmplInference
    from video_mocap.utils.smpl_utils import get_sub_hierachies, remove_approximately_redundant_hierarchies

    real_lbfgs = torch.optim.LBFGS
    torch.optim.LBFGS = RecordingLBFGS
    meta = {"model_checksum": str(tables.checksum()), "torch": torch.__version__, "numpy": np.__version__}

    # ------------------------------------------------------------------ KATs
    x = torch.arange(0, 16 * 7 * 3).float().reshape(16, 7, 3)
    y = torch.arange(0, 16 * 19 * 3).float().reshape(16, 19, 3)
    w = torch.ones(16, 7)
    w[:, ::2] = 0
    ka_loss, _ = ref_cd.weighted_chamfer_distance(x, y, w)
    m = torch.tensor([[[0, 0, 0], [1, 2, 2]], [[0, 3, 4], [0, 0, 0]]]).float()
    vm = torch.tensor([[[1, 0, 0], [1, 2, 2.0095]], [[0, 0, 0], [5, 5, 5]]]).float()
    kb = ref_losses.MarkerLoss(m, vm, ref_opt.get_marker_mask(m), MARKER_DISTANCE)
    pts = np.array([[0., 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [.3, .3, .3], [1, 0, 0], [1, 0, 1], [1, 1, 0],
                    [1, 1, 1]])
    kc = ref_points.geometric_median(pts)
    kd = ref_points.closest_point(np.array([[.1, .1, .1], [.9, .9, .2], [.5, .5, .5]]), pts)
    sub_counts = {}
    for k in (1, 3, 5, 12, 24):
        s = get_sub_hierachies(torch.from_numpy(tables.parents), k)
        sub_counts[k] = (len(s), len(remove_approximately_redundant_hierarchies(s, 0.9)))
    sub5 = get_sub_hierachies(torch.from_numpy(tables.parents), 5)
    np.savez_compressed(
        os.path.join(GOLDEN, "kats.npz"),
        ka_loss=np.float32(ka_loss.item()), kb=kb.numpy(), kb_mean=np.float32(kb.mean().item()),
        kc=kc, kd_idx=kd["vertex_indices"], kd_dist=kd["distances"],
        sub_counts=np.array([[k, a, b] for k, (a, b) in sub_counts.items()]),
        sub5=np.array(sub5), sub5_pruned=np.array(remove_approximately_redundant_hierarchies(sub5, 0.9)),
    )

    # ------------------------------------------------------------------ SMPL forward + chamfer operator
    device = torch.device("cpu")
    ref_smpl = RefSmplInference(device)
    F_, M_ = 8, 12
    seq = make_sequence(tables, seed=1, num_frames=F_, num_markers=M_)
    inp = seq_inputs(seq)
    markers = torch.from_numpy(inp["markers"]).float()
    o_pose = seq.img_smpl.pose_body.clone()
    o_root = seq.img_smpl.root_orient.clone()
    o_betas = (torch.sum(seq.img_smpl.betas, dim=0, keepdim=True) / torch.sum(seq.img_smpl.img_mask)).clone()
    trans0 = torch.median(markers, dim=1)[0].clone()
    with torch.no_grad():
        fwd = ref_smpl(poses=o_pose, betas=seq.img_smpl.betas, root_orient=o_root, trans=trans0)
        cd_loss, _ = ref_cd.weighted_chamfer_distance(markers, fwd["vertices"], ref_opt.get_marker_mask(markers))
    np.savez_compressed(
        os.path.join(GOLDEN, "smpl_forward.npz"), **inp, trans0=trans0.numpy(),
        vertices=fwd["vertices"].numpy()[:, ::13].copy(), vertex_stride=13,
        joints=fwd["joints"].numpy(), chamfer_loss=np.float32(cd_loss.item()),
    )

    # ------------------------------------------------------------------ chamfer stage
    cfg = small_config()
    RecordingLBFGS.records = []
    pose = o_pose.clone().requires_grad_(True)
    betas = o_betas.clone().requires_grad_(True)
    root = o_root.clone().requires_grad_(True)
    trans = trans0.clone().requires_grad_(True)
    ref_opt.optim_chamfer(markers, pose_body=pose, o_pose_body=o_pose, betas=betas, o_betas=o_betas,
                          root_orient=root, trans=trans, img_mask=seq.img_smpl.img_mask,
                          marker_labels=torch.zeros(F_, M_).long(), smpl_inference=ref_smpl, config=cfg)
    rec = RecordingLBFGS.records[-1]
    np.savez_compressed(
        os.path.join(GOLDEN, "chamfer_stage.npz"), **inp, o_betas=o_betas.numpy(), trans0=trans0.numpy(),
        num_iters=cfg["stages"]["chamfer"]["num_iters"], losses=np.array(rec["losses"]),
        first_grad=rec["first_grad"], first_params=rec["first_params"],
        out_trans=trans.detach().numpy(), out_betas=betas.detach().numpy(), out_pose_body=pose.detach().numpy(),
        out_root_orient=root.detach().numpy(),
    )

    # ------------------------------------------------------------------ placement + marker stage
    one_hot = ref_opt.compute_nearest_points(
        markers=markers, pose_body=pose, betas=betas, root_orient=root, trans=trans, smpl_inference=ref_smpl,
        marker_labels=np.zeros((F_, M_)), granularity="full", img_mask=seq.img_smpl.img_mask, device=device,
        config=cfg, o_pose_body=o_pose, window_size=1, use_velocity=False)
    place_idx = torch.argmax(one_hot, dim=-1).numpy()
    RecordingLBFGS.records = []
    root_m = root.clone().detach().requires_grad_(True)
    pose_m = pose.clone().detach().requires_grad_(True)
    in_pose, in_root = pose_m.detach().numpy().copy(), root_m.detach().numpy().copy()
    in_betas, in_trans = betas.detach().numpy().copy(), trans.detach().numpy().copy()
    ref_opt.optim_markers(markers=markers, pose_body=pose_m, o_pose_body=o_pose, betas=betas, o_betas=o_betas,
                          root_orient=root_m, trans=trans, barycentric_coords_one_hot=one_hot,
                          img_mask=seq.img_smpl.img_mask, smpl_inference=ref_smpl, config=cfg)
    rec = RecordingLBFGS.records[-1]
    np.savez_compressed(
        os.path.join(GOLDEN, "marker_stage.npz"), markers=inp["markers"], o_pose_body=o_pose.numpy(),
        o_betas=o_betas.numpy(), in_pose_body=in_pose, in_root_orient=in_root, in_betas=in_betas,
        in_trans=in_trans, img_mask=inp["img_mask"], place_idx=place_idx,
        num_iters=cfg["stages"]["marker"]["num_iters"], losses=np.array(rec["losses"]),
        first_grad=rec["first_grad"], first_params=rec["first_params"],
        out_trans=trans.detach().numpy(), out_betas=betas.detach().numpy(), out_pose_body=pose_m.detach().numpy(),
        out_root_orient=root_m.detach().numpy(),
    )

    # ------------------------------------------------------------------ part stage (full skeleton and sub-tree search)
    for tag, cfg_name, limb in (("full", "hmr_full", False), ("tree", "hmr_part", True)):
        cfg_p = small_config(cfg_name)
        seq_p = make_sequence(tables, seed=2, num_frames=F_, num_markers=8 if limb else M_, limb_only=limb)
        inp_p = seq_inputs(seq_p)
        markers_p = torch.from_numpy(inp_p["markers"]).float()
        groups = ref_mu.segment_rigid(markers_p.numpy())
        if limb:  # force a small, fixed cluster structure so the sub-tree search stays tiny
            groups = [[0, 1, 2], [3, 4, 5], [6, 7]]
        seg = torch.zeros(markers_p.shape[:2])
        for gi, g in enumerate(groups):
            seg[:, g] = gi
        seg = seg.long()
        ob = (torch.sum(seq_p.img_smpl.betas, dim=0, keepdim=True) / torch.sum(seq_p.img_smpl.img_mask)).clone()
        RecordingLBFGS.records = []
        out = ref_mu.find_best_part_fits(
            markers=markers_p, pose_body=seq_p.img_smpl.pose_body.clone(), betas=ob,
            root_orient=seq_p.img_smpl.root_orient.clone(), marker_labels=seg, smpl_inference=ref_smpl,
            hierarchy=ref_smpl.smpl.parents, joints_2d_gt=None, focal_length=None, reproject_mask=None,
            cam_trans=None, camera_center=None, config=cfg_p, foot_contacts=torch.zeros(F_, 2))
        recs = RecordingLBFGS.records
        np.savez_compressed(
            os.path.join(GOLDEN, "part_stage_%s.npz" % tag), **inp_p, o_betas=ob.numpy(), seg=seg.numpy(),
            num_iters=cfg_p["stages"]["part"]["num_iters"], n_subtrees=len(recs),
            first_losses=np.array([r["losses"][0] for r in recs]), n_evals=np.array([len(r["losses"]) for r in recs]),
            final_losses=np.array([r["losses"][-1] for r in recs]), first_grad0=recs[0]["first_grad"],
            out_betas=out["betas"].detach().numpy(), out_marker_labels=out["marker_labels"].numpy(),
            out_marker_weights=out["marker_weights"].numpy(), out_root_orient=out["root_orient"].detach().numpy(),
            out_trans=out["trans"].detach().numpy(), out_aabb=out["aabb_volume_ratio"].numpy(),
            out_chain=out["chain"],
        )

    # ------------------------------------------------------------------ end to end
    for tag, cfg_name in (("default", "video_mocap"), ("hmr_full", "hmr_full")):
        cfg_e = small_config(cfg_name)
        seq_e = make_sequence(tables, seed=3, num_frames=F_, num_markers=M_)
        inp_e = seq_inputs(seq_e)
        RecordingLBFGS.records = []
        out = ref_mm.multimodal_video_mocap(seq_e.img_smpl, copy.deepcopy(seq_e.markers), device, cfg_e,
                                            offset=0, print_options=[], save_stages=True)
        recs = RecordingLBFGS.records
        np.savez_compressed(
            os.path.join(GOLDEN, "e2e_%s.npz" % tag), **inp_e,
            part_iters=cfg_e["stages"]["part"]["num_iters"], chamfer_iters=cfg_e["stages"]["chamfer"]["num_iters"],
            marker_iters=cfg_e["stages"]["marker"]["num_iters"],
            n_solves=len(recs), n_evals=np.array([len(r["losses"]) for r in recs]),
            first_losses=np.array([r["losses"][0] for r in recs]),
            final_losses=np.array([r["losses"][-1] for r in recs]),
            out_trans=out["trans"].numpy(), out_root_orient=out["root_orient"].numpy(),
            out_pose_body=out["pose_body"].numpy(), out_betas=out["betas"].numpy(),
            out_markers_labels=np.asarray(out["markers_labels"]), out_chain=out["chain"],
            stage_keys=np.array(sorted(out["stages"].keys())),
        )

    # ------------------------------------------------------------------ config contract
    cwd = os.getcwd()
    os.chdir("/root/reference")
    try:
        from video_mocap.utils.config import load_config as ref_load_config
        cfgs = {n: ref_load_config("config/%s.yaml" % n) for n in ("video_mocap", "hmr_full", "hmr_part",
                                                                    "mht_rotation")}
    finally:
        os.chdir(cwd)
    with open(os.path.join(GOLDEN, "configs.json"), "w") as fh:
        json.dump(cfgs, fh, indent=1, sort_keys=True)
    with open(os.path.join(GOLDEN, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    torch.optim.LBFGS = real_lbfgs
    for fn in sorted(os.listdir(GOLDEN)):
        print("%-24s %8d B" % (fn, os.path.getsize(os.path.join(GOLDEN, fn))))


if __name__ == "__main__":
    main()
