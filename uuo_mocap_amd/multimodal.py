"""``multimodal_video_mocap`` with the reference's signature, stage order and output dictionary
(reference src/video_mocap/multimodal.py:38-726): HMR prior + unlabeled markers -> SMPL parameters.

Stages (reference :217-677): rigid marker clustering -> part fit (yaw/translation/shape per candidate body
part) -> for each of `num_root_orient_angles` yaw hypotheses { chamfer L-BFGS -> marker placement -> marker
L-BFGS } -> best hypothesis by chamfer distance -> final placement + marker L-BFGS.  The reprojection and
root stages are disabled in every shipped config and are not built."""
from __future__ import annotations

import contextlib
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict

import numpy as np
import torch

from . import markers_utils, optimization
from .engine import set_workspace_group, set_workspace_slot, worker_pool, worker_streams, workspace_group
from .markers_utils import find_best_part_fits, get_aabb, get_aabb_volume, segment_rigid
from .optimization import (compute_marker_labels_from_coords, compute_nearest_points, get_marker_mask,
                           optim_chamfer, optim_markers, weighted_chamfer_distance)
from .losses import knn_points_k1
from .smpl import SmplInference
from .transforms import compute_root_orient_z, normalize_rot

#: per-stage solver statistics of the most recent call (n_iter / n_eval / device ms per solve)
LAST_RUN_STATS: Dict = {}
_tls_stats = threading.local()


def last_run_stats() -> Dict:
    """Solver statistics of the calling thread's last multimodal_video_mocap call (LAST_RUN_STATS is the
    process-wide copy, ambiguous when sequences are fitted concurrently)."""
    return getattr(_tls_stats, "last", {})


def pad(sequence, offset):
    """Repeats the first (offset > 0) or last (offset < 0) frame |offset| times (reference multimodal.py:713-726)."""
    if offset == 0:
        return sequence
    edge = sequence[[0]] if offset > 0 else sequence[[-1]]
    padding = torch.repeat_interleave(edge, repeats=abs(offset), dim=0)
    return torch.cat((sequence, padding), dim=0) if offset < 0 else torch.cat((padding, sequence), dim=0)


def _np(t):
    return t.clone().detach().cpu().numpy()


def _np_dict(**tensors):
    """Host copies of several float tensors of one device with ONE device-to-host copy (packed into a flat buffer first)
    instead of a blocking copy each."""
    keys = list(tensors)
    ts = [tensors[k].detach() for k in keys]
    if not ts[0].is_cuda or any(t.dtype != ts[0].dtype for t in ts):
        return {k: t.clone().cpu().numpy() for k, t in zip(keys, ts)}
    flat = torch.cat([t.reshape(-1) for t in ts]).cpu().numpy()
    out, o = {}, 0
    for k, t in zip(keys, ts):
        n = t.numel()
        out[k] = flat[o:o + n].reshape(tuple(t.shape)).copy()
        o += n
    return out


def multimodal_video_mocap(
    img_smpl,
    mocap_markers,
    device: torch.device,
    config: Dict,
    offset: int = None,
    print_options=[],
    save_stages: bool = False,
    save_iterations: bool = False,
    visualize_fits: bool = False,
    smpl_inference: SmplInference = None,
    execution: Dict = None,
) -> Dict:
    """See the reference docstring (multimodal.py:49-84) for the meaning of the inputs and output keys.
    `smpl_inference` (extension) lets callers reuse one model/workspace across sequences.  `execution` (extension; how the
    independent solves are scheduled on the device, never what they compute -- markers_utils.EXECUTION_DEFAULTS):
    {"hypothesis_lockstep": False, "hypothesis_threads": 4, "subtree_lockstep": True, "subtree_batch": 256,
    "subtree_threads": 4}; an `execution` section of `config` sets the same keys, the argument wins."""
    exe = markers_utils.merge_execution(config, execution)
    if visualize_fits:
        raise NotImplementedError("visualize_fits renders with pyrender, not built")
    for key in ("reprojection_full", "root"):
        # both are disabled in every shipped config and cannot run in the reference as written (reprojection_full calls
        # optim_reprojection without its img_mask argument, multimodal.py:396-413; optim_root reads an undefined
        # o_betas and a missing 'lr' key, optimization.py:51,112)
        if config["stages"][key]["num_iters"] > 0:
            raise NotImplementedError("stage '%s' is disabled in every shipped config and is not built" % key)
    device = torch.device(device)
    if smpl_inference is None:
        smpl_inference = SmplInference(device)
    stats: Dict = {"part": [], "chamfer": [], "marker": [], "marker_final": [], "timeline": []}
    t_start = time.perf_counter()

    def mark(label: str):  # host-side stage boundaries of this call, seconds since entry (bench.py reports them)
        stats["timeline"].append((label, time.perf_counter() - t_start))

    verbose = "loss" in print_options

    o_trans = img_smpl.trans.clone().detach().to(device)
    o_root_orient = img_smpl.root_orient.clone().detach().to(device)
    o_pose_body = img_smpl.pose_body.clone().detach().to(device)
    o_betas = torch.sum(img_smpl.betas, dim=0, keepdim=True).clone().detach().to(device)
    o_betas = o_betas / torch.sum(img_smpl.img_mask)
    img_mask = img_smpl.img_mask.to(device)
    o_foot_contacts = getattr(img_smpl, "foot_contacts", None)
    if o_foot_contacts is not None:
        o_foot_contacts = o_foot_contacts.clone().detach().to(device)
    if mocap_markers.get_frequency() != img_smpl.freq:
        # bring the HMR track to the mocap frame rate (reference :145-182); img_mask and the camera stay in video
        # frames there, so the reprojection stage (which pairs them with the resampled track) cannot follow
        if config["find_best_part_fits"] and config["stages"]["reprojection_part"]["num_iters"] > 0:
            raise NotImplementedError("stages.reprojection_part with different mocap / video frame rates: the reference "
                                      "pairs the resampled HMR track with the un-resampled camera (multimodal.py:270-284)")
        from .resample import resample_hmr

        o_trans, o_root_orient, o_pose_body, o_foot_contacts = resample_hmr(
            o_trans, o_root_orient, o_pose_body, o_foot_contacts, float(img_smpl.freq),
            float(mocap_markers.get_frequency()))

    trans = o_trans.clone().detach().requires_grad_(True)
    root_orient = o_root_orient.clone().detach().requires_grad_(True)
    markers = torch.from_numpy(mocap_markers.get_points()).float().to(device)
    markers = torch.nan_to_num(markers, nan=0)

    min_frames = min(markers.shape[0], trans.shape[0])
    markers = markers[:min_frames]
    o_trans, o_root_orient, o_pose_body = o_trans[:min_frames], o_root_orient[:min_frames], o_pose_body[:min_frames]
    trans, root_orient = trans[:min_frames], root_orient[:min_frames]
    if o_foot_contacts is not None:
        o_foot_contacts = o_foot_contacts[:min_frames]

    if "progress" in print_options:
        print("Stage: computing temporal alignment...")
    if offset is None:
        offset = 0
    o_pose_body = pad(o_pose_body, offset).detach()
    o_betas = o_betas.detach()
    o_root_orient = pad(o_root_orient, offset).detach()
    o_trans = pad(o_trans, offset).detach()
    if o_foot_contacts is not None:
        o_foot_contacts = pad(o_foot_contacts, offset).detach()
    markers = pad(markers, -offset).detach().contiguous()
    num_frames = trans.shape[0]

    # ---- save_iterations: every closure evaluation of every stage is recorded through the stages' iter_fn hook, in the
    # reference's nesting iterations[stage][initial_angle | part][iteration][parameter] (multimodal.py:102-142)
    iter_output = None
    save_iter_fn = None
    if save_iterations:
        iter_output = {"input": {"markers": mocap_markers.get_points()}}
        iter_lock = threading.Lock()
        recorded = ("betas", "pose_body", "trans", "root_orient", "markers", "pred_angle", "pred_2d_joints",
                    "gt_2d_joints", "part_joints")

        def save_iter_fn(stage, iteration, **kwargs):
            with iter_lock:  # the yaw hypotheses report from their own threads
                node = iter_output.setdefault(stage, {})
                if "initial_angle" in kwargs:
                    node = node.setdefault(np.asarray(kwargs["initial_angle"]).item(), {})
                elif "part" in kwargs:
                    node = node.setdefault(kwargs["part"], {})
                node[iteration] = {k: kwargs[k] for k in recorded if k in kwargs}

    mark("inputs")
    # ---- marker segmentation
    print("Stage: computing marker segmentation...")
    with torch.no_grad():
        if config["stages"]["part"]["mode"] != "cluster":
            raise NotImplementedError("stages.part.mode 'network' is not built")
        segmented_markers = torch.zeros((markers.shape[:2]))
        for group_index, group in enumerate(segment_rigid(markers)):
            segmented_markers[:, group] = group_index
        segmented_markers = segmented_markers.long().to(device)
        mean_out = smpl_inference(poses=o_pose_body, betas=o_betas * 0, root_orient=o_root_orient, trans=o_trans * 0)
        aabb_volume_ratio = torch.median(get_aabb_volume(get_aabb(markers)) /
                                         get_aabb_volume(get_aabb(mean_out["vertices"])))

    mark("segmentation")
    filter_output = None
    smpl_part = None
    camera = {"joints_2d_gt": None, "focal_length": None, "reproject_mask": None, "cam_trans": None,
              "camera_center": None}
    if config["find_best_part_fits"]:
        rp = config["stages"]["reprojection_part"]
        if rp["num_iters"] > 0:
            # Stage [reprojection_part] (reference multimodal.py:248-331; off in every shipped config): yaw hypotheses
            # of the camera-consistent placement, the best one by reprojection (or chamfer) error replaces the HMR
            # root orientation / translation / shape that the part search starts from.
            from .reprojection import optim_reprojection

            trans = torch.median(markers, dim=1)[0].requires_grad_(True)
            betas = o_betas.clone().requires_grad_(True)
            angles = torch.arange(0, 2 * np.pi, (2 * np.pi) / rp["num_angles"])
            hyps = [optim_reprojection(
                markers=markers, pose_body=o_pose_body, betas=betas, hmr_betas=img_smpl.betas.clone().detach().to(device),
                root_orient=img_smpl.hmr_root_orient.clone().detach().to(device), trans=trans,
                pred_cam=img_smpl.camera_bbox.clone().detach().to(device),
                cam_center=img_smpl.center.clone().detach().to(device), cam_size=img_smpl.size.clone().detach().to(device),
                cam_scale=img_smpl.scale.clone().detach().to(device), angle=angles[k], img_mask=img_mask,
                smpl_inference=smpl_inference, config=config, num_iters=rp["num_iters"], verbose=verbose,
                iter_fn=save_iter_fn)
                for k in range(rp["num_angles"])]
            key = {"reprojection": "reproject", "chamfer": "chamfer"}[rp["criterion"]]
            best = int(np.argmin([h["metrics"][key] for h in hyps]))
            stats["reprojection_part"] = [dict(h["metrics"], input_angle=h["input_angle"], output_angle=h["output_angle"])
                                          for h in hyps]
            if save_iterations:
                iter_output["reprojection_output"] = {"metrics": [h["metrics"] for h in hyps],
                                                      "input_angle": [h["input_angle"] for h in hyps],
                                                      "output_angle": [h["output_angle"] for h in hyps]}
            o_betas = torch.mean(hyps[best]["betas"][0], dim=0, keepdim=True).clone().detach()
            o_root_orient = hyps[best]["root_orient"][0].clone().detach()
            o_trans = hyps[best]["trans"][0].clone().detach()
            # camera of the winning hypothesis for the part stage's optional 'reproject' loss (reference :325-335)
            camera = {"joints_2d_gt": hyps[best]["joints_2d_gt"][0].clone().detach(),
                      "focal_length": hyps[best]["focal_length"].clone().detach(),
                      "reproject_mask": hyps[best]["reproject_mask"].clone().detach(),
                      "cam_trans": hyps[best]["cam_trans"][0].clone().detach(),
                      "camera_center": hyps[best]["camera_center"].clone().detach()}
        filter_output = find_best_part_fits(
            markers=markers, pose_body=o_pose_body, betas=o_betas, root_orient=o_root_orient,
            marker_labels=segmented_markers, smpl_inference=smpl_inference, hierarchy=smpl_inference.smpl.parents,
            config=config, foot_contacts=o_foot_contacts, iter_fn=save_iter_fn, execution=exe, **camera)
        stats["part"] = list(markers_utils.LAST_STATS.get("part", []))
        segmented_markers = filter_output["marker_labels"].detach().clone()
        root_orient = filter_output["root_orient"].detach().clone()
        trans = filter_output["trans"].detach().clone()
        betas = filter_output["betas"].detach().clone()
        smpl_part = {"trans": _np(trans), "root_orient": _np(normalize_rot(root_orient)), "betas": _np(betas[0]),
                     "pose_body": _np(normalize_rot(o_pose_body))}
    marker_labels = segmented_markers.detach().cpu().numpy()
    mark("part")

    if not config["find_best_part_fits"] or aabb_volume_ratio > 0.4:
        trans = torch.median(markers, dim=1)[0].requires_grad_(True)
        root_orient = o_root_orient.clone().requires_grad_(True)
        betas = o_betas.clone().requires_grad_(True)

    if "progress" in print_options:
        print("Stage [root]: optimizing root...")
    pose_body = o_pose_body.clone().requires_grad_(True)
    root_orient = root_orient.detach()

    smpl_chamfer_rotations, smpl_marker_rotations = {}, {}
    run_chamfer = config["stages"]["chamfer"]["num_iters"] > 0
    run_marker = config["stages"]["marker"]["num_iters"] > 0
    root_orient_angles = torch.arange(0, 2 * np.pi, (2 * np.pi) / config["num_root_orient_angles"]).tolist()
    recompute_labels = bool(config["recompute_marker_labels"]) and run_marker

    def labels_from_placement(coords):
        """config.recompute_marker_labels (reference :529-539,632-642): the markers' part labels become the dominant
        joint of the vertex they were placed on, optionally smoothed over the rigid clusters."""
        labels = compute_marker_labels_from_coords(smpl_inference, coords, num_frames).detach().cpu().numpy()
        if config["stages"]["segment"]["rigid_filter"]:
            labels = markers_utils.filter_rigid(markers, labels)
        return labels

    group = workspace_group()  # worker threads do not inherit thread-locals

    def yaw_score(r) -> float:
        """Masked chamfer distance of a hypothesis' marker-stage result (reference :576-599)."""
        angle_betas = torch.repeat_interleave(torch.from_numpy(r["betas"]).to(device)[None], dim=0,
                                              repeats=r["pose_body"].shape[0])
        with torch.no_grad():
            vertices = smpl_inference(
                poses=torch.from_numpy(r["pose_body"]).to(device), betas=angle_betas,
                root_orient=torch.from_numpy(r["root_orient"]).to(device),
                trans=torch.from_numpy(r["trans"]).to(device))["vertices"]
            score = weighted_chamfer_distance(x=markers, y=vertices, x_weights=get_marker_mask(markers),
                                              single_directional=True)[0]
        return float(score)

    def yaw_scores_batched(records):
        """The scores of ALL hypotheses from one SMPL forward over their stacked frames and one K=1 search (the reference
        runs one forward + one chamfer call per hypothesis, :576-599; four of each were 4 ms of an `hmr_full` fit).  Each
        hypothesis' frames go through exactly the arithmetic of `yaw_score` (the kernels treat frames independently, the
        masked mean is taken on a copy of the hypothesis' own [F, M] block): bit-identical scores."""
        H, Fh = len(records), records[0]["pose_body"].shape[0]
        cat = lambda key: torch.from_numpy(np.concatenate([r[key] for r in records], axis=0)).to(device)  # noqa: E731
        betas_all = torch.from_numpy(np.concatenate([np.repeat(r["betas"][None], Fh, axis=0) for r in records], axis=0)).to(device)
        with torch.no_grad():
            vertices = smpl_inference(poses=cat("pose_body"), betas=betas_all, root_orient=cat("root_orient"),
                                      trans=cat("trans"))["vertices"]
            d, _ = knn_points_k1(markers.repeat(H, 1, 1), vertices)
            w = get_marker_mask(markers)
            wf = w.to(d.dtype)
            wsum = w.sum()
            if wsum == 0.0:
                return [0.0] * H
            per = torch.stack([(d[k * Fh:(k + 1) * Fh].clone() * wf).sum() / wsum for k in range(H)])
        return [float(v) for v in per.cpu()]

    def final_stage(r, labels):
        """Final placement + marker L-BFGS from a hypothesis' marker-stage result (reference :601-677)."""
        root_f = torch.from_numpy(r["root_orient"]).to(device).requires_grad_(True)
        trans_f = torch.from_numpy(r["trans"]).to(device).requires_grad_(True)
        pose_f = torch.from_numpy(r["pose_body"]).to(device).requires_grad_(True)
        betas_f = torch.from_numpy(r["betas"][None]).to(device).requires_grad_(True)
        final_np, final_stats = None, []
        for stage_i in range(config["stage_repeats"]):
            pose_stage = torch.clone(pose_f).detach().requires_grad_(False)
            if "progress" in print_options:
                print("Stage: computing marker placement... [{}/{}]".format(stage_i + 1, config["stage_repeats"]))
            if run_marker:
                one_hot = compute_nearest_points(
                    markers=markers, pose_body=pose_f, betas=betas_f, root_orient=root_f, trans=trans_f,
                    smpl_inference=smpl_inference, marker_labels=labels,
                    granularity=config["stages"]["segment"]["granularity"], img_mask=img_mask, device=device,
                    config=config, o_pose_body=pose_stage, window_size=1,
                    use_velocity=config["stages"]["compute_locations"]["use_velocity"])
                if recompute_labels:
                    labels = labels_from_placement(one_hot)
                if "progress" in print_options:
                    print("Stage [marker]: optimizing SMPL parameters... [{}/{}]".format(stage_i + 1,
                                                                                          config["stage_repeats"]))
                root_f = root_f.clone().detach().requires_grad_(True)
                pose_f = pose_f.clone().detach().requires_grad_(True)
                optim_markers(markers=markers, pose_body=pose_f, o_pose_body=pose_stage, betas=betas_f,
                              o_betas=o_betas, root_orient=root_f, trans=trans_f, barycentric_coords_one_hot=one_hot,
                              img_mask=img_mask, smpl_inference=smpl_inference, config=config, initial_angle=0,
                              repeat=1, verbose=verbose, iter_fn=save_iter_fn)
                final_stats.append(optimization.last_stats("marker"))
            root_f = normalize_rot(root_f).clone().detach().requires_grad_(True)
            pose_f = normalize_rot(pose_f).clone().detach().requires_grad_(True)
            final_np = _np_dict(trans=trans_f, root_orient=root_f, betas=betas_f[0], pose_body=pose_f)
        return {"trans": trans_f, "root_orient": root_f, "pose_body": pose_f, "betas": betas_f, "np": final_np,
                "stats": final_stats, "labels": labels}

    def fit_hypothesis(index: int, root_orient_angle: float, stream, marker_labels=marker_labels):
        """One yaw hypothesis (reference multimodal.py:463-574): chamfer L-BFGS -> placement -> marker L-BFGS.
        Hypotheses are independent, so each runs on its own host thread, HIP stream and solver workspace."""
        set_workspace_group(group)
        set_workspace_slot(index)
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        local = {}
        with ctx:
            angle_t = torch.tensor([[[root_orient_angle]]]).float().to(device)
            z_root = compute_root_orient_z(torch.repeat_interleave(angle_t, repeats=root_orient.shape[0], dim=0)) @ \
                root_orient.clone().detach()
            z_root = z_root.clone().detach().requires_grad_(True)
            trans_angle = trans.clone().detach().requires_grad_(True)
            pose_angle = pose_body.clone().detach().requires_grad_(True)
            betas_angle = betas.clone().detach().requires_grad_(True)
            if "progress" in print_options:
                print("Stage [pose]: optimizing poses and shapes...")
            if run_chamfer:
                optim_chamfer(markers, pose_body=pose_angle, o_pose_body=o_pose_body, betas=betas_angle,
                              o_betas=o_betas, root_orient=z_root, trans=trans_angle,
                              marker_labels=torch.from_numpy(np.asarray(marker_labels)).to(device),
                              img_mask=img_mask, smpl_inference=smpl_inference, initial_angle=root_orient_angle,
                              repeat=0, config=config, verbose=verbose, iter_fn=save_iter_fn)
                local["chamfer_stats"] = optimization.last_stats("chamfer")
            local["chamfer"] = _np_dict(trans=trans_angle, root_orient=normalize_rot(z_root), betas=betas_angle[0],
                                        pose_body=normalize_rot(pose_angle))
            if "progress" in print_options:
                print("Stage: computing marker placement... [{}/{}]".format(1, config["stage_repeats"]))
            if run_marker:
                one_hot = compute_nearest_points(
                    markers=markers, pose_body=pose_angle, betas=betas_angle, root_orient=z_root, trans=trans_angle,
                    smpl_inference=smpl_inference, marker_labels=marker_labels,
                    granularity=config["stages"]["segment"]["granularity"], img_mask=img_mask, device=device,
                    config=config, o_pose_body=o_pose_body, window_size=1,
                    use_velocity=config["stages"]["compute_locations"]["use_velocity"])
                if recompute_labels:
                    local["marker_labels"] = labels_from_placement(one_hot)
                if "progress" in print_options:
                    print("Stage [marker]: optimizing SMPL parameters... [{}/{}]".format(1, config["stage_repeats"]))
                z_root = z_root.clone().detach().requires_grad_(True)
                pose_angle = pose_angle.clone().detach().requires_grad_(True)
                optim_markers(markers=markers, pose_body=pose_angle, o_pose_body=o_pose_body, betas=betas_angle,
                              o_betas=o_betas, root_orient=z_root, trans=trans_angle,
                              barycentric_coords_one_hot=one_hot, img_mask=img_mask, smpl_inference=smpl_inference,
                              config=config, initial_angle=root_orient_angle, repeat=0, verbose=verbose,
                              iter_fn=save_iter_fn)
                local["marker_stats"] = optimization.last_stats("marker")
            if not run_chamfer and not run_marker:
                # nothing was optimised: the marker-stage record is the chamfer-stage record (both hold the normalised
                # rotations of the yawed part-stage result)
                local["marker"] = {k: v.copy() for k, v in local["chamfer"].items()}
            else:
                z_root = normalize_rot(z_root).clone().detach().requires_grad_(True)
                pose_angle = normalize_rot(pose_angle).clone().detach().requires_grad_(True)
                local["marker"] = _np_dict(trans=trans_angle, root_orient=z_root, betas=betas_angle[0],
                                           pose_body=pose_angle)
            if stream is not None:
                stream.synchronize()
        return local

    def hypotheses_without_stages():
        """hmr_full.yaml / hmr_part.yaml: neither the chamfer nor the marker stage runs, so a hypothesis is the part-stage
        result with its root turned by the hypothesis' yaw and both rotations normalised (what fit_hypothesis leaves in
        that case) -- all of them from ONE batched expression and one read-back instead of four of each.  Element-wise the
        same operations as fit_hypothesis (tests/test_gpu_parity.py compares the records bit for bit)."""
        H = len(root_orient_angles)
        angles = torch.tensor(root_orient_angles, dtype=torch.float32).reshape(H, 1, 1, 1).to(device)
        rz = compute_root_orient_z(angles.expand(H, root_orient.shape[0], 1, 1))           # [H, F, 1, 3, 3]
        z_all = normalize_rot(rz @ root_orient.detach()[None])                                # [H, F, 1, 3, 3]
        host = _np_dict(trans=trans.detach(), root_orient=z_all, betas=betas.detach()[0], pose_body=normalize_rot(pose_body.detach()))
        out = []
        for h in range(H):
            rec = {"trans": host["trans"].copy(), "root_orient": host["root_orient"][h].copy(), "betas": host["betas"].copy(),
                   "pose_body": host["pose_body"].copy()}
            out.append({"chamfer": rec, "marker": {k: v.copy() for k, v in rec.items()}})
        return out

    def fit_hypotheses_lockstep():
        """All yaw hypotheses stage by stage: their chamfer solves are one lock-step batch (one launch per kernel and
        round for all of them instead of one kernel stream per host thread), then the placements, then the marker solves as
        another batch.  Every hypothesis ends exactly where fit_hypothesis takes it (same arithmetic, bit-identical)."""
        hyps = []
        for angle in root_orient_angles:
            angle_t = torch.tensor([[[angle]]]).float().to(device)
            z_root = compute_root_orient_z(torch.repeat_interleave(angle_t, repeats=root_orient.shape[0], dim=0)) @ \
                root_orient.clone().detach()
            hyps.append({"root_orient": z_root.clone().detach().requires_grad_(True),
                         "trans": trans.clone().detach().requires_grad_(True),
                         "pose_body": pose_body.clone().detach().requires_grad_(True),
                         "betas": betas.clone().detach().requires_grad_(True)})
        locals_ = [dict() for _ in hyps]
        if run_chamfer:
            for local, stt in zip(locals_, optimization.optim_chamfer_lockstep(markers, hyps, o_pose_body, o_betas,
                                                                               smpl_inference, config)):
                local["chamfer_stats"] = stt
        for local, h in zip(locals_, hyps):
            local["chamfer"] = _np_dict(trans=h["trans"], root_orient=normalize_rot(h["root_orient"]), betas=h["betas"][0],
                                        pose_body=normalize_rot(h["pose_body"]))
        if run_marker:
            one_hots = [compute_nearest_points(
                markers=markers, pose_body=h["pose_body"], betas=h["betas"], root_orient=h["root_orient"], trans=h["trans"],
                smpl_inference=smpl_inference, marker_labels=marker_labels,
                granularity=config["stages"]["segment"]["granularity"], img_mask=img_mask, device=device, config=config,
                o_pose_body=o_pose_body, window_size=1,
                use_velocity=config["stages"]["compute_locations"]["use_velocity"]) for h in hyps]
            for h in hyps:
                h["root_orient"] = h["root_orient"].clone().detach().requires_grad_(True)
                h["pose_body"] = h["pose_body"].clone().detach().requires_grad_(True)
            if all(optimization.is_one_hot_placement(oh) for oh in one_hots):
                all_stats = optimization.optim_markers_lockstep(markers, hyps, [o_pose_body] * len(hyps), o_betas, one_hots,
                                                                smpl_inference, config)
            else:
                all_stats = []
                for h, oh, angle in zip(hyps, one_hots, root_orient_angles):
                    optim_markers(markers=markers, pose_body=h["pose_body"], o_pose_body=o_pose_body, betas=h["betas"],
                                  o_betas=o_betas, root_orient=h["root_orient"], trans=h["trans"],
                                  barycentric_coords_one_hot=oh, img_mask=img_mask, smpl_inference=smpl_inference,
                                  config=config, initial_angle=angle, repeat=0, verbose=verbose)
                    all_stats.append(optimization.last_stats("marker"))
            for local, stt in zip(locals_, all_stats):
                local["marker_stats"] = stt
        for local, h in zip(locals_, hyps):
            if not run_chamfer and not run_marker:
                local["marker"] = {k: v.copy() for k, v in local["chamfer"].items()}
            else:
                local["marker"] = _np_dict(trans=h["trans"], root_orient=normalize_rot(h["root_orient"]),
                                           betas=h["betas"][0], pose_body=normalize_rot(h["pose_body"]))
        return locals_

    # lock-step batches need the fused device closures, no per-evaluation callbacks and the labels fixed during the loop
    lockstep = (device.type == "cuda" and len(root_orient_angles) > 1 and (run_chamfer or run_marker)
                and save_iter_fn is None and not verbose and not recompute_labels
                and (not run_chamfer or optimization.lockstep_supported(config, "chamfer"))
                and (not run_marker or optimization.lockstep_supported(config, "marker"))
                and bool(exe["hypothesis_lockstep"]))
    n_threads = min(len(root_orient_angles), max(1, int(exe["hypothesis_threads"])))
    if not run_chamfer and not run_marker:
        n_threads = 1  # nothing to solve per hypothesis (hmr_full.yaml): worker threads would only add their start-up
    if recompute_labels and config["stages"]["segment"]["granularity"] == "part":
        # the reference's hypotheses run one after the other and each placement reads the labels the previous one
        # recomputed (only the "part" granularity looks at them): keep that order
        n_threads = 1
    from .parallel import collective_lanes, frame_shard, hypothesis_shard, shared_betas_reducer

    hyp_shard = hypothesis_shard()
    lanes = None
    if (frame_shard() is not None and frame_shard().active) or shared_betas_reducer() is not None:
        # frame blocks across ranks / shared betas (extension): every solve is a collective, so all ranks must
        # issue them in one order -- one hypothesis after the other, or every hypothesis on a lane (process group) of its own
        if hyp_shard is not None or (frame_shard() is not None and shared_betas_reducer() is not None):
            raise NotImplementedError("frame sharding, hypothesis sharding and shared betas are different uses of the ranks")
        lockstep = False
        lanes = collective_lanes(len(root_orient_angles))
        if lanes is None or n_threads < len(root_orient_angles) or device.type != "cuda":
            lanes, n_threads = None, 1
    if lanes is not None:
        _fit_plain = fit_hypothesis

        def fit_hypothesis(index, root_orient_angle, stream, marker_labels=marker_labels):  # noqa: F811
            with lanes[index]():
                return _fit_plain(index, root_orient_angle, stream, marker_labels)
    if hyp_shard is not None and hyp_shard.world > 1:
        # this rank fits hypotheses rank, rank + world, ...; one all_gather_object brings every rank all results
        if recompute_labels:
            raise NotImplementedError("recompute_marker_labels chains the hypotheses (reference :529-539); not shardable")
        mine = hyp_shard.mine(len(root_orient_angles))
        local = {}
        if len(mine) > 1 and device.type == "cuda":
            main_stream = torch.cuda.current_stream(device)
            streams = worker_streams(device, len(mine), "hypothesis")
            for st_ in streams:
                st_.wait_stream(main_stream)
            pool = worker_pool(len(mine), "hypothesis")
            futures = [pool.submit(fit_hypothesis, k, root_orient_angles[i], streams[k]) for k, i in enumerate(mine)]
            for i, f in zip(mine, futures):
                local[i] = f.result()
            for st_ in streams:
                main_stream.wait_stream(st_)
        else:
            for i in mine:
                local[i] = fit_hypothesis(0, root_orient_angles[i], None, marker_labels)
        results = hyp_shard.exchange(local, len(root_orient_angles))
    elif not run_chamfer and not run_marker and device.type == "cuda" and bool(exe.get("batch_trivial_hypotheses", True)):
        results = hypotheses_without_stages()
    elif lockstep:
        results = fit_hypotheses_lockstep()
    elif n_threads > 1 and device.type == "cuda":
        main_stream = torch.cuda.current_stream(device)
        streams = worker_streams(device, len(root_orient_angles), "hypothesis")
        for st_ in streams:
            st_.wait_stream(main_stream)
        pool = worker_pool(n_threads, "hypothesis")  # persistent threads: no per-fit thread / BLAS-handle churn
        futures = [pool.submit(fit_hypothesis, i, a, streams[i]) for i, a in enumerate(root_orient_angles)]
        results = [f.result() for f in futures]
        for st_ in streams:
            main_stream.wait_stream(st_)
    else:
        results = []
        for a in root_orient_angles:
            results.append(fit_hypothesis(0, a, None, marker_labels))
            marker_labels = results[-1].get("marker_labels", marker_labels)
    if recompute_labels and results:
        marker_labels = results[-1].get("marker_labels", marker_labels)  # the last hypothesis' labels survive the loop
    set_workspace_slot(0)
    mark("hypotheses")
    for root_orient_angle, local in zip(root_orient_angles, results):
        smpl_chamfer_rotations[root_orient_angle] = local["chamfer"]
        smpl_marker_rotations[root_orient_angle] = local["marker"]
        if "chamfer_stats" in local:
            stats["chamfer"].append(local["chamfer_stats"])
        if "marker_stats" in local:
            stats["marker"].append(local["marker_stats"])

    # ---- best yaw hypothesis by masked chamfer distance (first minimum wins)
    best_angle_chamfer, best_angle, best_index = np.inf, None, 0
    records = [smpl_marker_rotations[a] for a in root_orient_angles]
    batched = device.type == "cuda" and len(records) > 1 and len({r["pose_body"].shape[0] for r in records}) == 1
    yaw_scores = yaw_scores_batched(records) if batched else [yaw_score(r) for r in records]
    shared_red = shared_betas_reducer()
    if shared_red is not None and shared_red.world > 1:
        # shared betas (extension): hypothesis k was ONE joint solve over all ranks, so the ranks must agree on the winner --
        # it is picked from the scores summed over the ranks (rank order, fp64: the same number everywhere); a rank that
        # picked by its own score could start the final stage from the betas of a different joint solve
        table = np.zeros((shared_red.world, len(yaw_scores)))
        shared_red.gather_array(np.asarray(yaw_scores, dtype=np.float64), table)
        stats["yaw_scores_local"] = yaw_scores
        yaw_scores = [float(v) for v in table.sum(axis=0)]
    for k, (root_orient_angle, score) in enumerate(zip(root_orient_angles, yaw_scores)):
        if score < best_angle_chamfer:
            best_angle_chamfer, best_angle, best_index = score, root_orient_angle, k
    stats["yaw_scores"] = yaw_scores
    stats["best_angle"] = best_angle
    mark("selection")

    smpl_chamfer = smpl_chamfer_rotations[best_angle]
    smpl_marker = smpl_marker_rotations[best_angle]
    print("Final marker optimization")
    # (Running this stage speculatively inside the hypothesis threads -- for the best-scoring hypothesis finished so far
    # -- was tried: the winner tends to be among the last to finish, so it saved nothing and cost extra solves.)
    fin = final_stage(smpl_marker, marker_labels)
    root_orient, trans, pose_body, betas = fin["root_orient"], fin["trans"], fin["pose_body"], fin["betas"]
    smpl_marker_final = fin["np"]
    stats["marker_final"].extend(fin["stats"])
    marker_labels = fin["labels"]

    mark("final_marker")
    output = {
        "trans": trans.detach().cpu(),
        "root_orient": normalize_rot(root_orient).detach().cpu(),
        "pose_body": normalize_rot(pose_body).detach().cpu(),
        "betas": torch.repeat_interleave(torch.mean(betas, dim=0, keepdim=True), dim=0,
                                         repeats=pose_body.shape[0]).detach().cpu(),
        "mocap_frame_rate": mocap_markers.get_frequency(),
    }
    mocap_markers.set_points(markers.detach().cpu().numpy())
    output["mocap_markers"] = mocap_markers
    output["markers_labels"] = marker_labels
    if save_stages:
        output["stages"] = {}
        if config["find_best_part_fits"]:
            output["stages"]["part"] = smpl_part
        if run_chamfer:
            output["stages"]["chamfer"] = smpl_chamfer
        if run_marker:
            output["stages"]["marker"] = smpl_marker
        if config["stage_repeats"] > 0:
            output["stages"]["marker_final"] = smpl_marker_final
    if filter_output is not None:
        output["chain"] = filter_output["chain"]
    if save_iterations:
        output["iterations"] = iter_output
    mark("outputs")
    LAST_RUN_STATS.clear()
    LAST_RUN_STATS.update(stats)
    _tls_stats.last = stats
    return output
