"""CPU restatement of the reference's stage solvers.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows, for the branches the shipped configs enable:

* ``optim_chamfer``            reference optimization.py:147-285
* ``optim_markers``            reference optimization.py:288-399  (+ losses/losses.py:43-51)
* ``compute_nearest_points``   reference optimization.py:402-642  (use_mean, granularity "full")
* ``find_best_part_fits``      reference markers/markers_utils.py:274-638 (mode "cluster", no reprojection)
* ``segment_rigid``            reference markers/markers_utils.py:244-271
* sub-hierarchy enumeration    reference utils/smpl_utils.py:106-188
* ``multimodal_video_mocap``   reference multimodal.py:38-726 (equal frame rates, offset 0)

Same materialisations as the reference (repeat_interleave of the vertex cloud per marker, dense one-hot
gather) and the real ``torch.optim.LBFGS`` -- this is also what bench.py times as ``cpu_baseline``
(kind "port").  Pinned against the reference's own modules by tests/golden (oracle/make_golden.py).
"""
from __future__ import annotations

import itertools
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .p3d_ref import (axis_angle_to_matrix, chamfer_distance, matrix_to_rotation_6d, rotation_6d_to_matrix)

MARKER_DISTANCE = 0.0095  # reference utils/settings.py:1


# ----------------------------------------------------------------------------------------------
# small helpers (optimization.py:662-724, multimodal.py:34-35, losses/*)
# ----------------------------------------------------------------------------------------------

def normalize_rot(rot: torch.Tensor) -> torch.Tensor:
    return rotation_6d_to_matrix(matrix_to_rotation_6d(rot))


def compute_root_orient_z(angle: torch.Tensor) -> torch.Tensor:
    z = torch.zeros(list(angle.shape[:-1]) + [3], device=angle.device)
    z[..., [2]] = angle
    return axis_angle_to_matrix(z)


def get_marker_mask(markers: torch.Tensor) -> torch.Tensor:
    return torch.sum(torch.abs(markers), axis=-1) != 0.0


def weighted_chamfer_distance(x, y, x_weights, single_directional: bool = False):
    """losses/chamfer_distance.py:5-21 -- one cloud per marker; the flag argument is ignored there too."""
    x_flat = torch.reshape(x, (-1, 1, x.shape[2]))
    y_flat = torch.repeat_interleave(y, repeats=x.shape[1], dim=0)
    w_flat = torch.flatten(x_weights)
    return chamfer_distance(x_flat, y_flat, weights=w_flat, single_directional=True)


def marker_loss(markers, virtual_markers, marker_weights, marker_distance):
    """losses/losses.py:43-51."""
    out = (torch.norm(markers - virtual_markers, dim=-1) - marker_distance) ** 2
    return out * marker_weights


def get_aabb(points: torch.Tensor) -> torch.Tensor:
    aabb = torch.zeros((points.shape[0], 3, 2), dtype=points.dtype, device=points.device)
    for a in range(3):
        aabb[:, a, 0] = torch.min(points[..., a], dim=1)[0]
        aabb[:, a, 1] = torch.max(points[..., a], dim=1)[0]
    return aabb


def get_aabb_volume(aabb: torch.Tensor) -> torch.Tensor:
    d = aabb[:, :, 1] - aabb[:, :, 0]
    return d[:, 0] * d[:, 1] * d[:, 2]


def _lbfgs(params, num_iters, config, lr):
    return torch.optim.LBFGS(
        params,
        max_iter=num_iters,
        tolerance_grad=config["optimizer"]["tolerance_grad"],
        tolerance_change=config["optimizer"]["tolerance_change"],
        lr=lr,
        line_search_fn="strong_wolfe",
    )


def _smpl_repeat_betas(smpl_inference, poses, betas, root_orient, trans):
    return smpl_inference(
        poses=poses,
        betas=torch.repeat_interleave(betas, dim=0, repeats=poses.shape[0]),
        root_orient=root_orient,
        trans=trans,
    )


# ----------------------------------------------------------------------------------------------
# chamfer stage
# ----------------------------------------------------------------------------------------------

def chamfer_distance_by_part(markers, vertices, marker_labels, vertex_weights, single_directional: bool = False):
    """optimization.py:682-700."""
    vertex_mask = torch.argmax(vertex_weights, dim=-1)
    labels_mode = torch.mode(marker_labels, dim=0)[0]
    loss = 0
    for i in torch.unique(labels_mode).tolist():
        part = chamfer_distance(vertices[:, vertex_mask == i], markers[:, labels_mode == i],
                                single_directional=single_directional)[0]
        loss = loss + (part - MARKER_DISTANCE) ** 2
    return loss


def chamfer_stage_loss(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, z_angle,
                       smpl_inference, config, marker_labels=None):
    """One forward of closure_stage_chamfer (optimization.py:187-256): the shipped terms and the optional
    part_chamfer (:227-235), trans_vel (:245-249), ground (:250-253) ones, yaw_lock True or False (:190-193)."""
    st = config["stages"]["chamfer"]
    w = st["losses"]
    z_root = compute_root_orient_z(z_angle) @ root_orient if st["yaw_lock"] else normalize_rot(z_angle)
    out = _smpl_repeat_betas(smpl_inference, normalize_rot(pose_body), betas, normalize_rot(z_root), trans)
    loss = 0
    unsupported = set(w) - {"full_chamfer", "reg_pose_body", "reg_betas", "part_chamfer", "trans_vel", "ground"}
    if unsupported:
        raise NotImplementedError("chamfer-stage losses that cannot run in the reference: %s" % sorted(unsupported))
    if "part_chamfer" in w:
        loss = loss + chamfer_distance_by_part(markers, out["vertices"], marker_labels, smpl_inference.smpl.lbs_weights,
                                               single_directional=st["single_directional"]) * w["part_chamfer"]
    if "full_chamfer" in w:
        c = weighted_chamfer_distance(markers, out["vertices"], get_marker_mask(markers),
                                      single_directional=st["single_directional"])[0]
        loss = loss + c * w["full_chamfer"]
    if "reg_pose_body" in w:
        loss = loss + F.mse_loss(pose_body, o_pose_body) * w["reg_pose_body"]
    if "trans_vel" in w:
        markers_mean = torch.mean(markers, dim=1)
        loss = loss + F.mse_loss(trans[1:] - trans[:-1], markers_mean[1:] - markers_mean[:-1]) * w["trans_vel"]
    if "ground" in w:
        loss = loss + torch.mean(F.relu(-out["joints"][..., 2])) * w["ground"]
    if "reg_betas" in w:
        loss = loss + F.mse_loss(betas, o_betas) * w["reg_betas"]
    return loss, out


def optim_chamfer(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, smpl_inference, config,
                  trace: Optional[list] = None, marker_labels=None):
    """optimization.py:147-285. Mutates trans/betas/pose_body (L-BFGS) and root_orient (in place)."""
    if config["stages"]["chamfer"]["yaw_lock"]:
        z_angle = torch.zeros((root_orient.shape[0], root_orient.shape[1], 1), device=root_orient.device)
    else:
        z_angle = torch.zeros((root_orient.shape[0], root_orient.shape[1], 3, 3), device=root_orient.device)
        z_angle[..., 0, 0] = z_angle[..., 1, 1] = z_angle[..., 2, 2] = 1.0
    z_angle.requires_grad_(True)
    params = [trans, z_angle, betas, pose_body]
    opt = _lbfgs(params, config["stages"]["chamfer"]["num_iters"], config, lr=0.1)
    root_orient.requires_grad_(False)

    def closure():
        opt.zero_grad()
        loss, _ = chamfer_stage_loss(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans,
                                     z_angle, smpl_inference, config, marker_labels=marker_labels)
        loss.backward()
        if trace is not None:
            trace.append(float(loss))
        return loss

    opt.step(closure)
    with torch.no_grad():
        if config["stages"]["chamfer"]["yaw_lock"]:
            root_orient[:] = compute_root_orient_z(z_angle) @ root_orient
        else:
            root_orient[:] = normalize_rot(z_angle) @ root_orient
    root_orient.requires_grad_(True)
    return z_angle.detach()


# ----------------------------------------------------------------------------------------------
# marker stage
# ----------------------------------------------------------------------------------------------

def virtual_markers_dense(vertices: torch.Tensor, one_hot: torch.Tensor) -> torch.Tensor:
    """optimization.py:345-351 -- the dense [F,M,V,3] product the reference materialises."""
    num_markers = one_hot.shape[0]
    v = torch.repeat_interleave(torch.unsqueeze(vertices, dim=1), repeats=num_markers, dim=1)
    bc = torch.unsqueeze(torch.unsqueeze(one_hot, dim=0), dim=-1)
    bc = torch.repeat_interleave(bc, repeats=v.shape[0], dim=0)
    bc = torch.repeat_interleave(bc, repeats=3, dim=-1)
    return torch.sum(v * bc, dim=2)


def marker_stage_loss(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, one_hot,
                      smpl_inference, config):
    st = config["stages"]["marker"]
    if st.get("use_sdf"):
        raise NotImplementedError("use_sdf is off in every shipped config")
    unsupported = set(st["losses"]) - {"marker", "reg_pose_body", "reg_betas"}
    if unsupported:
        raise NotImplementedError("marker-stage losses outside the shipped configs: %s" % sorted(unsupported))
    out = _smpl_repeat_betas(smpl_inference, normalize_rot(pose_body), betas, normalize_rot(root_orient), trans)
    vm = virtual_markers_dense(out["vertices"], one_hot)
    loss = 0
    if "marker" in st["losses"]:
        ml = marker_loss(markers, vm, get_marker_mask(markers), MARKER_DISTANCE)
        loss = loss + torch.mean(ml) * st["losses"]["marker"]
    if "reg_pose_body" in st["losses"]:
        loss = loss + F.mse_loss(pose_body, o_pose_body) * st["losses"]["reg_pose_body"]
    if "reg_betas" in st["losses"]:
        loss = loss + F.mse_loss(betas, o_betas) * st["losses"]["reg_betas"]
    return loss, out


def optim_markers(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans, one_hot,
                  smpl_inference, config, trace: Optional[list] = None):
    """optimization.py:288-399."""
    params = [pose_body, betas, root_orient, trans]
    opt = _lbfgs(params, config["stages"]["marker"]["num_iters"], config, lr=1.0)

    def closure():
        opt.zero_grad()
        loss, _ = marker_stage_loss(markers, pose_body, o_pose_body, betas, o_betas, root_orient, trans,
                                    one_hot, smpl_inference, config)
        loss.backward()
        if trace is not None:
            trace.append(float(loss))
        return loss

    opt.step(closure)


# ----------------------------------------------------------------------------------------------
# marker placement
# ----------------------------------------------------------------------------------------------

def compute_nearest_points(markers, pose_body, betas, root_orient, trans, smpl_inference, img_mask, config,
                           return_indices: bool = False, marker_labels=None, granularity: str = "full",
                           window_size: int = 1, use_velocity: bool = False):
    """optimization.py:402-642: compute_locations.use_mean (granularity "full") or use_barycentric (any granularity)."""
    cl = config["stages"]["compute_locations"]
    if cl["use_barycentric"]:
        return _barycentric_nearest_points(markers, pose_body, betas, root_orient, trans, smpl_inference, img_mask,
                                           config, marker_labels, granularity, window_size, use_velocity)
    if not cl["use_mean"]:
        raise NotImplementedError("the closest_point mode returns an all-zero matrix in the reference (:549-561)")
    num_frames, num_markers = markers.shape[0], markers.shape[1]
    with torch.no_grad():
        out = smpl_inference(
            poses=normalize_rot(pose_body),
            betas=torch.repeat_interleave(torch.mean(betas, dim=0, keepdim=True), dim=0, repeats=betas.shape[0]),
            root_orient=normalize_rot(root_orient),
            trans=trans,
        )
    vertices = out["vertices"].detach().cpu().numpy()
    num_verts = vertices.shape[1]
    dist = np.zeros((num_frames, num_markers, num_verts), dtype=np.float32)
    valid = torch.where(img_mask == 1)[0].tolist()
    markers_np = markers.detach().cpu().numpy()
    for f in range(num_frames):
        if f not in valid:
            continue
        vf = np.repeat(vertices[f][None, :, :], axis=0, repeats=num_markers)
        mf = np.repeat(markers_np[f][:, None, :], axis=1, repeats=num_verts)
        dist[f] = np.linalg.norm(vf - mf, axis=-1)
    mask_np = img_mask.detach().cpu().numpy()
    reduced = np.mean(dist[np.where(mask_np == 1)], axis=0)
    vidx = np.argmin(reduced, axis=-1)
    one_hot = torch.zeros((num_markers, num_verts)).float().to(markers.device)
    for m in range(num_markers):
        one_hot[m, vidx[m]] = 1.0
    if return_indices:
        return one_hot, vidx
    return one_hot


def _barycentric_nearest_points(markers, pose_body, betas, root_orient, trans, smpl_inference, img_mask, config,
                                marker_labels, granularity, window_size, use_velocity):
    """The use_barycentric branch of compute_nearest_points, loop for loop (optimization.py:444-603), over
    oracle/mesh_ref.py instead of igl / trimesh."""
    from . import mesh_ref

    num_frames, num_markers, num_joints = markers.shape[0], markers.shape[1], pose_body.shape[1]
    with torch.no_grad():
        out = smpl_inference(
            poses=normalize_rot(pose_body),
            betas=torch.repeat_interleave(torch.mean(betas, dim=0, keepdim=True), dim=0, repeats=betas.shape[0]),
            root_orient=normalize_rot(root_orient), trans=trans)
    vertices = out["vertices"].detach().cpu().numpy()
    o_vertices = vertices  # the reference recomputes it from the same inputs (:434-442)
    faces = np.asarray(smpl_inference.smpl.faces)
    torch_faces = torch.from_numpy(faces.astype(np.int64))
    num_windows = int(np.ceil(num_frames / window_size))
    size = {"part": num_joints, "marker": num_markers, "full": 1}[granularity]
    min_distance = np.full((num_windows, size), np.inf)
    final = torch.zeros((num_markers, vertices.shape[1]))
    stride = window_size
    n_str = int(np.ceil(num_frames / stride))
    distance = np.zeros((n_str, num_markers))
    face_indices = np.zeros((n_str, num_markers), dtype=np.int32)
    points_3d = np.zeros((n_str, num_markers, 3))
    valid_frames = torch.where(img_mask == 1)[0].tolist()
    markers_np = markers.detach().cpu().numpy()
    window_index = 0
    for w_start in range(0, num_frames, window_size):
        for f_index in range(w_start, min(w_start + window_size, num_frames), stride):
            fs = f_index // stride
            if fs not in valid_frames:
                continue
            distance[fs], face_indices[fs], points_3d[fs] = mesh_ref.signed_distance(
                markers_np[f_index], vertices[f_index], faces.astype(np.int32))
            distance[fs] = np.abs(distance[fs])
            tri = vertices[f_index][faces.astype(np.int64)][face_indices[fs]]
            bc_np = mesh_ref.points_to_barycentric(tri, points_3d[fs])
            bc = torch.from_numpy(bc_np).float()
            i0, i1, i2 = (torch_faces[face_indices[fs]][:, k] for k in range(3))
            one_hot = torch.zeros((num_markers, vertices.shape[1]))
            one_hot.scatter_(1, i0.unsqueeze(1), bc[:, [0]])
            one_hot.scatter_(1, i1.unsqueeze(1), bc[:, [1]])
            one_hot.scatter_(1, i2.unsqueeze(1), bc[:, [2]])
            vel_factor = np.ones((num_frames, num_markers))
            if use_velocity:
                pts = sum(o_vertices[:, i.numpy()] * np.repeat(np.reshape(bc_np[..., k], (1, -1, 1)), num_frames, axis=0)
                          for k, i in enumerate((i0, i1, i2)))
                pv = pts[1:] - pts[:-1]
                pv = np.concatenate((pv[[0]] * 0, pv), axis=0)
                mv = markers_np[1:] - markers_np[:-1]
                mv = np.concatenate((mv[[0]] * 0, mv), axis=0)
                vel_factor = np.sum(mv * pv, axis=-1)
            if granularity == "part":
                for m in range(num_joints):
                    sel = marker_labels[f_index] == m
                    part_distance = (sel.astype(np.float32) * distance[fs])[sel]
                    if part_distance.size > 0 and np.median(part_distance) < min_distance[window_index, m]:
                        final[sel] = one_hot[sel]
                        min_distance[window_index, m] = np.median(part_distance)
            elif granularity == "marker":
                for m in range(num_markers):
                    if distance[fs, m] < min_distance[window_index, m]:
                        final[m] = one_hot[m]
                        min_distance[window_index, m] = distance[fs, m]
            else:
                if np.mean(distance[fs]) * np.mean(vel_factor[fs]) < min_distance[window_index]:
                    final = one_hot
                    min_distance[window_index] = np.mean(distance[fs])
        window_index += 1
    if cl_use_mean(config):
        raise NotImplementedError("use_mean together with use_barycentric is not restated")
    return final


def cl_use_mean(config) -> bool:
    return bool(config["stages"]["compute_locations"]["use_mean"])


# ----------------------------------------------------------------------------------------------
# part stage
# ----------------------------------------------------------------------------------------------

def segment_rigid(points: np.ndarray) -> List[List[int]]:
    """markers/markers_utils.py:244-271: std of pairwise distance over time -> average-linkage clusters."""
    from sklearn.cluster import AgglomerativeClustering

    num_markers = points.shape[1]
    mat = np.zeros((num_markers, num_markers))
    for i in range(num_markers):
        for j in range(num_markers):
            mat[i, j] = np.std(np.linalg.norm(points[:, i] - points[:, j], axis=-1))
    labels = AgglomerativeClustering(n_clusters=None, distance_threshold=0.005, metric="precomputed",
                                     linkage="average").fit(mat).labels_
    return [np.where(labels == v)[0].tolist() for v in np.unique(labels).tolist()]


def get_sub_hierarchies(parents, num_bones: int) -> List[List[int]]:
    """utils/smpl_utils.py:106-164: connected sub-trees with exactly num_bones nodes, in the
    reference's enumeration order (children products, nodes visited from the last to the first)."""
    parents_np = parents if isinstance(parents, np.ndarray) else parents.detach().cpu().numpy()
    n = parents_np.shape[0]
    num_bones = min(num_bones, n)
    children: Dict[int, List[int]] = {i: [] for i in range(n)}
    for i in range(1, n):
        children[int(parents_np[i])].append(i)
    table: Dict[int, List[List[int]]] = {}
    for node in list(children.keys())[::-1]:
        table[node] = [[]]
        for combo in itertools.product(*[table[c] for c in children[node]]):
            merged: List[int] = []
            for part in combo:
                merged = merged + part
            merged = sorted(merged)
            if [node] + merged not in table[node]:
                table[node].append([node] + merged)
    out = []
    for node, subtrees in table.items():
        for st in subtrees:
            if len(st) == num_bones:
                out.append(st)
    return out


def remove_approximately_redundant_hierarchies(subtrees, similarity_threshold: float = 0.9):
    """utils/smpl_utils.py:167-188."""
    kept = [subtrees[0]]
    for st in subtrees[1:]:
        limit = len(st) * similarity_threshold
        if all(len(set(st) & set(k)) <= limit for k in kept):
            kept.append(st)
    return kept


def part_stage_loss(markers_subset, pose_body, betas, o_betas, root_orient, trans, z_angle, vertex_indices,
                    smpl_inference, config, camera=None, foot_contacts=None, markers_subset_mean=None):
    """closure_fit_subtree (markers_utils.py:454-533): chamfer + reg_betas (the shipped terms) and the optional
    reproject (:477-514), foot_contact (:520-524), foot_velocity (:526-532), velocity (:534-538), ground (:540-544)
    terms, added in the reference's order."""
    st = config["stages"]["part"]
    w = st["losses"]
    unsupported = set(w) - {"chamfer", "reg_betas", "reproject", "foot_contact", "foot_velocity", "velocity", "ground"}
    if unsupported:
        raise NotImplementedError("part-stage losses the reference does not define: %s" % sorted(unsupported))
    num_frames = pose_body.shape[0]
    z_root = compute_root_orient_z(torch.repeat_interleave(z_angle, repeats=num_frames, dim=0)) @ root_orient
    out = _smpl_repeat_betas(smpl_inference, pose_body, betas, z_root, trans)
    verts_sub = out["vertices"][:, vertex_indices]
    loss = chamfer_distance(markers_subset, verts_sub, single_directional=True)[0] * w["chamfer"]
    if "reproject" in w:
        correction = torch.tensor([[[[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]]]])
        correction = torch.repeat_interleave(correction, repeats=num_frames, dim=0)
        hmr_cam_trans = torch.repeat_interleave(mocap_to_hmr(camera["cam_trans"]), dim=0, repeats=num_frames)
        hmr_root_orient = torch.linalg.inv(correction) @ root_orient
        camera_offset = mocap_to_hmr(trans) - hmr_cam_trans
        inv_translation = (compute_root_orient_y(z_angle)[:, 0] @ camera_offset[..., None])[..., 0] + hmr_cam_trans
        joints = _smpl_repeat_betas(smpl_inference, pose_body, betas, hmr_root_orient, inv_translation)["joints"]
        kp = perspective_projection(joints, hmr_cam_trans,
                                    torch.repeat_interleave(camera["focal_length"], dim=0, repeats=num_frames),
                                    camera["camera_center"],
                                    torch.eye(3).unsqueeze(0).expand(num_frames, -1, -1)).reshape(num_frames, 45, 2) + 0.5
        loss = loss + torch.mean((kp - camera["joints_2d_gt"]) ** 2 * camera["reproject_mask"][:, None, None]) * w["reproject"]
    if "reg_betas" in w:
        loss = loss + F.mse_loss(betas, o_betas) * w["reg_betas"]
    feet = [10, 11]  # left_foot, right_foot (utils/smpl_utils.py:11-36)
    if "foot_contact" in w and foot_contacts is not None:
        h = out["joints"][:, feet, 2]
        loss = loss + torch.mean(F.mse_loss(h, torch.ones_like(h) * 0.005, reduction="none") * foot_contacts) * w["foot_contact"]
    if "foot_velocity" in w and foot_contacts is not None:
        vel = out["joints"][1:, feet, :2] - out["joints"][:-1, feet, :2]
        speed = torch.norm(vel, dim=-1)
        loss = loss + torch.mean(F.mse_loss(speed, torch.zeros_like(speed), reduction="none") * w["foot_velocity"] *
                                 foot_contacts[1:]) * 1.0
    if "velocity" in w:
        loss = loss + F.mse_loss(trans[1:] - trans[:-1], markers_subset_mean[1:] - markers_subset_mean[:-1]) * w["velocity"]
    if "ground" in w:
        loss = loss + torch.mean(F.relu(-out["vertices"][..., 2])) * w["ground"]
    return loss, out, z_root


def find_best_part_fits(markers, pose_body, betas, root_orient, marker_labels, smpl_inference, hierarchy, config,
                        trace: Optional[dict] = None, joints_2d_gt=None, focal_length=None, reproject_mask=None,
                        camera_center=None, cam_trans=None, foot_contacts=None, subtree_limit: Optional[int] = None):
    """markers/markers_utils.py:274-638, mode "cluster".  `subtree_limit` (tests only) stops after that many
    candidates so a CPU test can pin the first trajectories without paying for all of them."""
    st = config["stages"]["part"]
    if st["mode"] != "cluster":
        raise NotImplementedError("part.mode 'network' needs checkpoints the reference does not ship")
    labels_mode = torch.mode(marker_labels, axis=0)[0]
    groups = [x.tolist() for x in torch.unique(labels_mode, return_counts=True)]
    chain = groups[0]
    final_labels = torch.zeros_like(marker_labels)
    final_weights = torch.zeros_like(marker_labels, dtype=torch.float)
    num_frames = markers.shape[0]
    o_betas = betas

    indices = torch.cat([torch.where(labels_mode == j)[0] for j in chain], dim=0)
    markers_subset = markers[:, indices]
    if st.get("use_full_skeleton"):
        subtrees = [np.arange(0, hierarchy.shape[0]).tolist()]
    else:
        subtrees = get_sub_hierarchies(hierarchy, len(chain))
        if "similarity_threshold" in st:
            subtrees = remove_approximately_redundant_hierarchies(subtrees, similarity_threshold=0.9)

    weights = smpl_inference.get_lbs_weights()
    vertex_labels = torch.argmax(weights, dim=-1)
    best = {"distance": np.inf}
    subtree_losses = []
    for subtree in subtrees[:subtree_limit]:
        z_angle = torch.zeros((1, 1, 1), device=markers.device).requires_grad_(True)
        trans = torch.median(markers, dim=1)[0].clone().requires_grad_(True)
        betas_s = o_betas.clone().requires_grad_(True)
        camera = None
        if "reproject" in st["losses"]:  # the camera translation joins the parameter list but gets no gradient (:422-424)
            camera = {"joints_2d_gt": joints_2d_gt, "focal_length": focal_length, "reproject_mask": reproject_mask,
                      "camera_center": camera_center, "cam_trans": cam_trans[[0]].clone()}
        opt = _lbfgs([z_angle, trans, betas_s] + ([camera["cam_trans"]] if camera else []), st["num_iters"], config, lr=1.0)
        extra = dict(camera=camera, foot_contacts=foot_contacts, markers_subset_mean=torch.mean(markers_subset, dim=1))
        vertex_indices = torch.cat([(vertex_labels == j).nonzero(as_tuple=True)[0] for j in subtree], dim=0)
        evals = []

        def closure():
            opt.zero_grad()
            loss, _, _ = part_stage_loss(markers_subset, pose_body, betas_s, o_betas, root_orient, trans, z_angle,
                                         vertex_indices, smpl_inference, config, **extra)
            loss.backward()
            evals.append(float(loss))
            return loss

        opt.step(closure)
        with torch.no_grad():
            _, out, z_root = part_stage_loss(markers_subset, pose_body, betas_s, o_betas, root_orient, trans,
                                             z_angle, vertex_indices, smpl_inference, config, **extra)
            verts_sub = out["vertices"][:, vertex_indices]
            distance = chamfer_distance(markers_subset, verts_sub, single_directional=False)[0].item()
        subtree_losses.append([subtree, distance])
        if trace is not None:
            trace.setdefault("evals", []).append(evals)
            trace.setdefault("distance", []).append(distance)
        if distance < best["distance"]:
            best = {
                "distance": distance, "betas": betas_s.clone(), "markers_subset": markers_subset.clone(),
                "root_orient": z_root.clone(), "trans": trans.clone(),
                "aabb": get_aabb_volume(get_aabb(markers_subset)) / get_aabb_volume(get_aabb(markers)),
            }
            with torch.no_grad():
                for i in range(indices.shape[0]):
                    m_i = indices[i]
                    d = torch.norm(out["vertices"] - markers_subset[:, [i]], dim=-1)
                    final_labels[:, m_i] = vertex_labels[torch.argmin(torch.mean(d, dim=0), dim=-1)]

    if len(subtree_losses) > 1:
        subtree_losses = sorted(subtree_losses, key=lambda x: x[1])
        for i in range(indices.shape[0]):
            final_weights[:, indices[i]] = subtree_losses[1][1] / subtree_losses[0][1]
            if indices.shape[0] == 1:
                final_weights *= 0
    final_weights = final_weights / torch.max(final_weights)
    return {
        "betas": best["betas"].clone(), "marker_labels": final_labels.clone(),
        "markers_subset": best["markers_subset"].clone(), "marker_weights": final_weights.clone(),
        "root_orient": best["root_orient"].clone(), "trans": best["trans"].clone(),
        "aabb_volume_ratio": best["aabb"].clone(),
        "chain": np.array(list(subtree_losses[0][0]), dtype=np.int32),
    }


# ----------------------------------------------------------------------------------------------
# frame-rate resampling (multimodal.py:145-182)
# ----------------------------------------------------------------------------------------------

def unitquat_slerp(q0: torch.Tensor, q1: torch.Tensor, steps: torch.Tensor, shortest_arc: bool = True) -> torch.Tensor:
    """roma.utils.unitquat_slerp (roma is not installed here: restated from the package's published source; the
    result is also checked against scipy's Slerp in tests/test_oracle_stages.py).  q0, q1 [..., 4], steps [S] ->
    [S, ..., 4].  sin((1-t)w) q0 + sin(t w) q1, linear weights where cos w > 1 - 1e-3, normalised."""
    batch = q0.shape[:-1]
    q0, q1 = q0.reshape(-1, 4), q1.reshape(-1, 4)
    cos_omega = torch.sum(q0 * q1, dim=-1)
    if shortest_arc:
        q1 = q1.clone()
        q1[cos_omega < 0, :] *= -1
        cos_omega = torch.abs(cos_omega)
    nearby = cos_omega > (1.0 - 1e-3)
    omega = torch.acos(cos_omega)
    alpha = torch.sin((1 - steps.unsqueeze(-1)) * omega)
    beta = torch.sin(steps.unsqueeze(-1) * omega)
    alpha[..., nearby] = (1 - steps.unsqueeze(-1)).expand_as(alpha)[..., nearby]
    beta[..., nearby] = steps.unsqueeze(-1).expand_as(beta)[..., nearby]
    q = alpha.unsqueeze(-1) * q0 + beta.unsqueeze(-1) * q1
    q = q / torch.norm(q, dim=-1, keepdim=True)
    return q.reshape(steps.shape + batch + (4,))


def resample_hmr(o_trans, o_root_orient, o_pose_body, img_freq, mocap_freq):
    """The resampling loop of multimodal.py:145-182, frame by frame (foot contacts are carried by the product only)."""
    from .p3d_ref import matrix_to_quaternion, quaternion_to_matrix

    tl, rl, pl = [], [], []
    new_num_frames = round(o_trans.shape[0] * (mocap_freq / img_freq))
    for i in range(new_num_frames):
        frame = int(i * (img_freq / mocap_freq))
        alpha = i * (img_freq / mocap_freq) - frame
        inv_alpha = 1.0 - alpha
        if frame + 1 < o_trans.shape[0]:
            tl.append((o_trans[frame + 1] * alpha) + (o_trans[frame] * inv_alpha))
            steps = torch.tensor([alpha]).to(o_trans.device)
            rl.append(quaternion_to_matrix(unitquat_slerp(matrix_to_quaternion(o_root_orient[frame]),
                                                          matrix_to_quaternion(o_root_orient[frame + 1]), steps))[0])
            pl.append(quaternion_to_matrix(unitquat_slerp(matrix_to_quaternion(o_pose_body[frame]),
                                                          matrix_to_quaternion(o_pose_body[frame + 1]), steps))[0])
        else:
            tl.append(o_trans[frame])
            rl.append(o_root_orient[frame])
            pl.append(o_pose_body[frame])
    return torch.stack(tl, dim=0), torch.stack(rl, dim=0), torch.stack(pl, dim=0)


# ----------------------------------------------------------------------------------------------
# orchestrator
# ----------------------------------------------------------------------------------------------

def multimodal_video_mocap(img_smpl, mocap_markers, smpl_inference, config, device=torch.device("cpu"),
                           stats: Optional[dict] = None) -> Dict:
    """multimodal.py:38-710 for offset 0, reprojection and root stages off."""
    for key in ("reprojection_part", "reprojection_full", "root"):
        if config["stages"][key]["num_iters"] > 0:
            raise NotImplementedError("stage %s is disabled in every shipped config" % key)
    stats = stats if stats is not None else {}
    o_trans = img_smpl.trans.clone().detach().to(device)
    o_root_orient = img_smpl.root_orient.clone().detach().to(device)
    o_pose_body = img_smpl.pose_body.clone().detach().to(device)
    o_betas = torch.sum(img_smpl.betas, dim=0, keepdim=True).clone().detach().to(device)
    o_betas = o_betas / torch.sum(img_smpl.img_mask)
    img_mask = img_smpl.img_mask.to(device)
    if mocap_markers.get_frequency() != img_smpl.freq:
        o_trans, o_root_orient, o_pose_body = resample_hmr(o_trans, o_root_orient, o_pose_body, img_smpl.freq,
                                                           mocap_markers.get_frequency())

    trans = o_trans.clone().detach().requires_grad_(True)
    root_orient = o_root_orient.clone().detach().requires_grad_(True)
    markers = torch.from_numpy(mocap_markers.get_points()).float().to(device)
    markers = torch.nan_to_num(markers, nan=0)
    n = min(markers.shape[0], trans.shape[0])
    markers, o_trans, o_root_orient, o_pose_body = markers[:n], o_trans[:n], o_root_orient[:n], o_pose_body[:n]
    trans, root_orient = trans[:n], root_orient[:n]
    num_frames = n

    with torch.no_grad():
        groups = segment_rigid(markers.detach().cpu().numpy())
        seg = torch.zeros((markers.shape[:2]))
        for gi, g in enumerate(groups):
            seg[:, g] = gi
        seg = seg.long().to(device)
    mean_out = smpl_inference(poses=o_pose_body, betas=o_betas * 0, root_orient=o_root_orient, trans=o_trans * 0)
    aabb_ratio = torch.median(get_aabb_volume(get_aabb(markers)) / get_aabb_volume(get_aabb(mean_out["vertices"])))

    filter_output = None
    if config["find_best_part_fits"]:
        trans = torch.median(markers, dim=1)[0].requires_grad_(True)
        root_orient = o_root_orient.clone().requires_grad_(True)
        betas = o_betas.clone().requires_grad_(True)
        tr = {}
        filter_output = find_best_part_fits(markers, o_pose_body, o_betas, o_root_orient, seg, smpl_inference,
                                            smpl_inference.smpl.parents, config, trace=tr)
        stats["part"] = tr
        seg = filter_output["marker_labels"].detach().clone()
        root_orient = filter_output["root_orient"].detach().clone()
        trans = filter_output["trans"].detach().clone()
        betas = filter_output["betas"].detach().clone()
    marker_labels = seg.detach().cpu().numpy()
    if not config["find_best_part_fits"] or aabb_ratio > 0.4:
        trans = torch.median(markers, dim=1)[0].requires_grad_(True)
        root_orient = o_root_orient.clone().requires_grad_(True)
        betas = o_betas.clone().requires_grad_(True)

    pose_body = o_pose_body.clone().requires_grad_(True)
    root_orient = root_orient.detach()
    chamfer_rot, marker_rot = {}, {}
    angles = torch.arange(0, 2 * np.pi, (2 * np.pi) / config["num_root_orient_angles"]).tolist()
    run_chamfer = config["stages"]["chamfer"]["num_iters"] > 0
    run_marker = config["stages"]["marker"]["num_iters"] > 0
    for angle in angles:
        a = torch.tensor([[[angle]]]).float().to(device)
        z_root = compute_root_orient_z(torch.repeat_interleave(a, repeats=root_orient.shape[0], dim=0)) @ \
            root_orient.clone().detach()
        z_root = z_root.clone().detach().requires_grad_(True)
        trans_a = trans.clone().detach().requires_grad_(True)
        pose_a = pose_body.clone().detach().requires_grad_(True)
        betas_a = betas.clone().detach().requires_grad_(True)
        if run_chamfer:
            tr = []
            optim_chamfer(markers, pose_a, o_pose_body, betas_a, o_betas, z_root, trans_a, smpl_inference, config,
                          trace=tr)
            stats.setdefault("chamfer", []).append(tr)
        chamfer_rot[angle] = {
            "trans": trans_a.clone().detach().cpu().numpy(),
            "root_orient": normalize_rot(z_root).clone().detach().cpu().numpy(),
            "betas": betas_a[0].clone().detach().cpu().numpy(),
            "pose_body": normalize_rot(pose_a).clone().detach().cpu().numpy(),
        }
        if run_marker:
            one_hot = compute_nearest_points(markers, pose_a, betas_a, z_root, trans_a, smpl_inference, img_mask,
                                             config)
            z_root = z_root.clone().detach().requires_grad_(True)
            pose_a = pose_a.clone().detach().requires_grad_(True)
            tr = []
            optim_markers(markers, pose_a, o_pose_body, betas_a, o_betas, z_root, trans_a, one_hot, smpl_inference,
                          config, trace=tr)
            stats.setdefault("marker", []).append(tr)
        z_root = normalize_rot(z_root).clone().detach().requires_grad_(True)
        pose_a = normalize_rot(pose_a).clone().detach().requires_grad_(True)
        marker_rot[angle] = {
            "trans": trans_a.clone().detach().cpu().numpy(),
            "root_orient": z_root.clone().detach().cpu().numpy(),
            "betas": betas_a[0].clone().detach().cpu().numpy(),
            "pose_body": pose_a.clone().detach().cpu().numpy(),
        }

    best_val, best_angle = np.inf, None
    yaw_scores = []
    for angle in angles:
        r = marker_rot[angle]
        ab = torch.repeat_interleave(torch.from_numpy(r["betas"]).to(device)[None], dim=0,
                                     repeats=r["pose_body"].shape[0])
        with torch.no_grad():
            verts = smpl_inference(poses=torch.from_numpy(r["pose_body"]).to(device), betas=ab,
                                   root_orient=torch.from_numpy(r["root_orient"]).to(device),
                                   trans=torch.from_numpy(r["trans"]).to(device))["vertices"]
            c = weighted_chamfer_distance(markers, verts, get_marker_mask(markers), single_directional=True)[0]
        yaw_scores.append(float(c))
        if c < best_val:
            best_val, best_angle = c, angle
    stats["yaw_scores"] = yaw_scores
    stats["best_angle"] = best_angle
    smpl_marker = marker_rot[best_angle]
    root_orient = torch.from_numpy(smpl_marker["root_orient"]).to(device).requires_grad_(True)
    trans = torch.from_numpy(smpl_marker["trans"]).to(device).requires_grad_(True)
    pose_body = torch.from_numpy(smpl_marker["pose_body"]).to(device).requires_grad_(True)
    betas = torch.from_numpy(smpl_marker["betas"][None]).to(device).requires_grad_(True)

    for _ in range(config["stage_repeats"]):
        pose_stage = torch.clone(pose_body).detach().requires_grad_(False)
        if run_marker:
            one_hot = compute_nearest_points(markers, pose_body, betas, root_orient, trans, smpl_inference, img_mask,
                                             config)
            root_orient = root_orient.clone().detach().requires_grad_(True)
            pose_body = pose_body.clone().detach().requires_grad_(True)
            tr = []
            optim_markers(markers, pose_body, pose_stage, betas, o_betas, root_orient, trans, one_hot,
                          smpl_inference, config, trace=tr)
            stats.setdefault("marker_final", []).append(tr)
        root_orient = normalize_rot(root_orient).clone().detach().requires_grad_(True)
        pose_body = normalize_rot(pose_body).clone().detach().requires_grad_(True)

    output = {
        "trans": trans.detach().cpu(),
        "root_orient": normalize_rot(root_orient).detach().cpu(),
        "pose_body": normalize_rot(pose_body).detach().cpu(),
        "betas": torch.repeat_interleave(torch.mean(betas, dim=0, keepdim=True), dim=0,
                                         repeats=pose_body.shape[0]).detach().cpu(),
        "mocap_frame_rate": mocap_markers.get_frequency(),
        "markers_labels": marker_labels,
        "stages": {"chamfer": chamfer_rot[best_angle], "marker": smpl_marker},
    }
    if filter_output is not None:
        output["chain"] = filter_output["chain"]
    return output


# ----------------------------------------------------------------------------------------------
# reprojection stage (utils/hmr_utils.py:14-124,127-167,170-425) -- disabled in every shipped config
# ----------------------------------------------------------------------------------------------

def perspective_projection(points, translation, focal_length, camera_center=None, rotation=None):
    """hmr_utils.py:14-52, with the explicit intrinsic matrix."""
    b = points.shape[0]
    if rotation is None:
        rotation = torch.eye(3, dtype=points.dtype).unsqueeze(0).expand(b, -1, -1)
    if camera_center is None:
        camera_center = torch.zeros(b, 2, dtype=points.dtype)
    K = torch.zeros([b, 3, 3], dtype=points.dtype)
    K[:, 0, 0] = focal_length[:, 0]
    K[:, 1, 1] = focal_length[:, 1]
    K[:, 2, 2] = 1.0
    K[:, :-1, -1] = camera_center
    points = torch.einsum("bij,bkj->bki", rotation, points) + translation.unsqueeze(1)
    projected = points / points[:, :, -1].unsqueeze(-1)
    projected = torch.einsum("bij,bkj->bki", K, projected)
    return projected[:, :, :-1]


def hmr_to_mocap(pos):
    return torch.cat((pos[..., [0]], pos[..., [2]], pos[..., [1]] * -1), dim=-1)


def mocap_to_hmr(pos):
    return torch.cat((pos[..., [0]], pos[..., [2]] * -1, pos[..., [1]]), dim=-1)


def compute_root_orient_y(angle: torch.Tensor) -> torch.Tensor:
    pad = torch.zeros_like(angle)
    return axis_angle_to_matrix(torch.cat((pad, angle, pad), dim=-1))


def get_3d_parameters(smpl_inference, betas, body_pose, global_orient, pred_cam, center, size, scale):
    """hmr_utils.py:57-124."""
    focal = 5000.0
    img = 256
    n = pred_cam.shape[0]
    new_size = torch.max(size, dim=-1, keepdim=True)[0]
    top, left = (new_size - size[:, [0]]) // 2, (new_size - size[:, [1]]) // 2
    ratio = 1.0 / torch.round(new_size) * img
    center = (center + torch.cat((left, top), dim=-1)) * ratio
    scale = scale * new_size * ratio
    focal_length = focal * torch.ones(n, 2)
    joints = smpl_inference(body_pose, betas, global_orient, torch.zeros((n, 3)))["joints"]
    tmp = torch.stack([pred_cam[:, 1], pred_cam[:, 2], 2 * focal_length[:, 0] / (pred_cam[:, 0] * scale[:, 0] + 1e-9)], dim=1)
    cam_t = torch.cat((tmp[:, :2] + (center - img / 2.0) * tmp[:, [2]] / focal_length, tmp[:, [2]]), dim=1)
    kp = perspective_projection(joints, cam_t, focal_length / img, torch.zeros(n, 2),
                                torch.eye(3).unsqueeze(0).expand(n, -1, -1))
    kp = (kp + 0.5) * img
    return {"camera_center": torch.zeros(n, 2), "focal_length": focal_length / img, "pred_cam_t": cam_t,
            "pred_joints": joints, "pred_keypoints_2d_smpl": kp / img}


def optim_reprojection(markers, pose_body, betas, hmr_betas, root_orient, trans, pred_cam, cam_center, cam_size,
                       cam_scale, angle, img_mask, smpl_inference, num_iters, config, trace: Optional[list] = None,
                       capture: Optional[dict] = None):
    """hmr_utils.py:170-425 (A = 1 hypothesis axis kept).  Derived outputs are those of the last closure evaluation,
    as the reference's nonlocal temporaries leave them.  `trace` collects the loss of every closure evaluation, `capture`
    the flat parameter vector and gradient of the FIRST one (the order of the params list: yaw, body_t, cam, betas)."""
    F_ = pose_body.shape[0]
    w = config["stages"]["reprojection_part"]["losses"]
    pose_body, root_orient, trans = pose_body.clone(), root_orient.clone(), trans.clone()
    betas = betas.clone().detach()
    correction = torch.tensor([[[[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]]]]).repeat_interleave(F_, dim=0)
    jo = get_3d_parameters(smpl_inference, hmr_betas, pose_body, root_orient, pred_cam.clone(), cam_center.clone(),
                           cam_size.clone(), cam_scale.clone())
    target = torch.nan_to_num(jo["pred_keypoints_2d_smpl"][None], 0)
    cam_translation = jo["pred_cam_t"]
    mask = torch.mean((cam_translation == cam_translation).float(), dim=-1).detach()
    cam_translation = torch.nan_to_num(cam_translation, 0)
    body_from_cam = cam_translation
    cam_translation = trans.clone().detach()
    offset = mocap_to_hmr(torch.median(markers.reshape(-1, 3), dim=0, keepdim=True)[0]) - \
        torch.median(body_from_cam, dim=0, keepdim=True)[0]
    body_t = (body_from_cam + offset)[None].clone()
    body_t.requires_grad_(True)
    cam_single = torch.mean(cam_translation - offset, dim=0, keepdim=True).clone()
    cam_single.requires_grad_(True)
    yaw = (torch.ones(1, 1, 1, 1) * angle).requires_grad_(True)
    focal = torch.mean(jo["focal_length"], dim=0, keepdim=True)
    opt = torch.optim.LBFGS([yaw, body_t, cam_single, betas], max_iter=num_iters,
                            tolerance_grad=config["optimizer"]["tolerance_grad"],
                            tolerance_change=config["optimizer"]["tolerance_change"], lr=1.0,
                            line_search_fn="strong_wolfe")
    betas_rep = torch.repeat_interleave(betas, repeats=F_, dim=0)
    last = {}

    def closure():
        opt.zero_grad()
        cam_tr = torch.repeat_interleave(cam_single[:, None], dim=1, repeats=F_)
        yaw_f = torch.repeat_interleave(yaw, repeats=F_, dim=1)
        y_root = compute_root_orient_y(yaw_f) @ root_orient
        off = body_t - cam_tr
        rot = compute_root_orient_y(-yaw_f)[:, 0]
        inv_t = (rot @ off[..., None])[..., 0] + cam_tr
        out = smpl_inference(pose_body, betas_rep, root_orient, inv_t.flatten(0, 1))
        kp = perspective_projection(out["joints"], cam_tr.flatten(0, 1), torch.repeat_interleave(focal, dim=0, repeats=F_),
                                    jo["camera_center"], torch.eye(3).unsqueeze(0).expand(F_, -1, -1)).reshape(1, F_, 45, 2) + 0.5
        loss = torch.mean((kp - target) ** 2 * mask[None, :, None, None]) * w["reprojection"]
        verts = smpl_inference(pose_body, betas_rep, (correction @ y_root).flatten(0, 1), hmr_to_mocap(body_t).flatten(0, 1))["vertices"]
        loss = loss + chamfer_distance(markers, verts, single_directional=True)[0] * w["chamfer"]
        loss.backward()
        if trace is not None:
            trace.append(float(loss))
        if capture is not None and "grad" not in capture:
            leaves = [yaw, body_t, cam_single, betas]
            capture["params"] = torch.cat([p.detach().reshape(-1) for p in leaves]).clone()
            capture["grad"] = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                         for p in leaves]).clone()
            capture["loss"] = float(loss)
        last.update(cam_tr=cam_tr.detach(), y_root=y_root.detach(), inv_t=inv_t.detach(), kp=kp.detach())
        return loss

    opt.step(closure)
    with torch.no_grad():
        world = smpl_inference(pose_body, betas_rep, (correction @ last["y_root"]).flatten(0, 1), last["inv_t"].flatten(0, 1))["vertices"]
        rep_err = torch.mean((last["kp"][0] - target[0]) ** 2 * mask[None, :, None, None]).item()
        ch_err = chamfer_distance(markers, world, single_directional=True)[0].item()
    return {"pose_body": pose_body[None].detach(), "betas": betas_rep[None].detach(),
            "root_orient": (correction @ last["y_root"]).detach(), "trans": hmr_to_mocap(body_t.detach()),
            "joints_2d": last["kp"], "joints_2d_gt": target, "cam_trans": hmr_to_mocap(last["cam_tr"]),
            "camera_center": jo["camera_center"].clone(), "focal_length": focal.clone(), "reproject_mask": mask.clone(),
            "input_angle": float(angle), "output_angle": yaw.item(), "metrics": {"chamfer": ch_err, "reproject": rep_err}}
