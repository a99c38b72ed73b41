"""Part stage with the reference's interface: rigid marker clustering, SMPL sub-hierarchy enumeration and
``find_best_part_fits`` (reference src/video_mocap/markers/markers_utils.py:244-271,274-638 and
utils/smpl_utils.py:106-188, utils/aabb.py:5-25).  Host logic stays Python; every L-BFGS solve and every
nearest-neighbour query runs on the GPU through libuuo_hip.so."""
from __future__ import annotations

import contextlib
import functools
import itertools
import queue
from collections.abc import Callable
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List

import numpy as np
import torch

from .body_model import SMPL_JOINT_NAMES
from .device_lbfgs import DeviceLBFGS
from .engine import _f32, PART_SOFT_MAX_MARKERS, PartProblem, set_workspace_group, set_workspace_slot, worker_pool, worker_streams, workspace_group
from .losses import chamfer_distance, soft_chamfer_distance
from .transforms import compute_root_orient_z

LAST_STATS: Dict[str, list] = {}

# how independent solves are scheduled on the device (never what they compute); see find_best_part_fits /
# multimodal_video_mocap.  These were environment variables in round 2: the product path reads no environment now.
EXECUTION_DEFAULTS = {"subtree_lockstep": True, "subtree_batch": 256, "subtree_threads": 4,
                      "hypothesis_lockstep": False, "hypothesis_threads": 4, "batch_trivial_hypotheses": True,
                      "part_soft_fused": True, "chamfer_soft_fused": True, "marker_bary_fused": True}


def merge_execution(config: Dict, execution: Dict = None) -> Dict:
    """EXECUTION_DEFAULTS, overridden by the config's `execution` section, overridden by the `execution` argument."""
    out = dict(EXECUTION_DEFAULTS)
    out.update(config.get("execution") or {})
    out.update(execution or {})
    unknown = set(out) - set(EXECUTION_DEFAULTS)
    if unknown:
        raise KeyError("unknown execution option(s): %s" % sorted(unknown))
    return out


def get_joint_name(joint_id: int) -> str:
    return SMPL_JOINT_NAMES[joint_id]


def get_joint_id(joint_name: str) -> int:
    return SMPL_JOINT_NAMES.index(joint_name)


def get_aabb(points: torch.Tensor) -> torch.Tensor:
    """[F, P, 3] -> [F, 3, 2] (min, max) per axis."""
    lo = torch.min(points, dim=1)[0]
    hi = torch.max(points, dim=1)[0]
    return torch.stack([lo, hi], dim=-1)


def get_aabb_volume(aabb: torch.Tensor) -> torch.Tensor:
    d = aabb[:, :, 1] - aabb[:, :, 0]
    return d[:, 0] * d[:, 1] * d[:, 2]


def rigid_distance_matrix(points, device=None) -> np.ndarray:
    """mat[i, j] = np.std(np.linalg.norm(points[:, i] - points[:, j], axis=-1)) for every marker pair, as the reference's
    double loop computes it (markers/markers_utils.py:254-259) -- on the GPU (uuo_rigid_distance_std: one thread per pair
    walking numpy's pairwise-summation tree, so the values are bit-equal to numpy's; the clustering below cuts them at
    5 mm).  `points` [F, M, 3]: a CUDA tensor, or a host array that is uploaded to `device` (default: the current CUDA
    device).  There is no host implementation: the reference's loop lives in oracle/stages_ref.py as the checker."""
    from . import _lib
    from .engine import _ptr, check, current_stream

    if torch.is_tensor(points) and points.is_cuda:
        pts = points.detach()
    else:
        if not torch.cuda.is_available():
            raise RuntimeError("rigid_distance_matrix runs on the GPU only (no CUDA/HIP device visible)")
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        pts = torch.as_tensor(np.asarray(points)).to(dev)
    if pts.dtype != torch.float32:
        # the reference hands float32 marker arrays (multimodal.py:187); numpy would compute a float64 input in float64
        raise TypeError("rigid_distance_matrix expects float32 points (got %s)" % pts.dtype)
    pts = pts.contiguous()
    F, M = int(pts.shape[0]), int(pts.shape[1])
    out = torch.empty((M, M), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        check(_lib.load().uuo_rigid_distance_std(current_stream(pts.device), F, M, _ptr(pts), _ptr(out)),
              "uuo_rigid_distance_std")
    return out.cpu().numpy().astype(np.float64)


def segment_rigid(points, device=None) -> List[List[int]]:
    """Clusters markers that keep their mutual distance over time (std of the pairwise distance, average-linkage
    agglomerative clustering cut at 5 mm).  points [F, M, 3] (host array as in the reference, or a CUDA tensor) -> list of
    marker-id lists.  The O(M^2 F) rigidity matrix is computed on the GPU; the clustering of the M x M matrix is host logic."""
    from sklearn.cluster import AgglomerativeClustering

    mat = rigid_distance_matrix(points, device=device)
    labels = AgglomerativeClustering(n_clusters=None, distance_threshold=0.005, metric="precomputed",
                                     linkage="average").fit(mat).labels_
    return [np.where(labels == v)[0].tolist() for v in np.unique(labels).tolist()]


def filter_rigid(points, labels: np.ndarray, device=None) -> np.ndarray:
    """Every rigid cluster of markers takes the median of its members' labels (reference markers_utils.py:220-241)."""
    output = np.array(labels)
    for group in segment_rigid(points, device=device):
        output[:, group] = np.median(labels[:, group])
    return output


def get_sub_hierachies(parents, num_bones: int) -> List[List[int]]:
    """All connected sub-trees of the kinematic tree with exactly `num_bones` joints, enumerated in the reference's
    order (it decides which candidate wins ties).  Name keeps the reference's spelling.  The enumeration depends on
    the tree and the size only, so it is memoised (the batch runner asks for the same one for every sequence)."""
    parents_np = parents if isinstance(parents, np.ndarray) else parents.detach().cpu().numpy()
    key = (tuple(int(v) for v in parents_np.tolist()), int(num_bones))
    return [list(st) for st in _sub_hierarchies_cached(key)]


@functools.lru_cache(maxsize=8)
def _rooted_subtrees(parents_t):
    """Every connected sub-tree of the kinematic tree, grouped by its root, in the reference's enumeration order.  It
    does not depend on the requested size, so it is built once per tree (15 ms for SMPL) and only filtered per call:
    the number of rigid marker clusters, hence the size, changes from sequence to sequence."""
    n = len(parents_t)
    kids: Dict[int, List[int]] = {i: [] for i in range(n)}
    for i in range(1, n):
        kids[int(parents_t[i])].append(i)
    rooted: Dict[int, List[List[int]]] = {}
    for node in reversed(range(n)):  # children carry larger ids, so they are finished first
        options = [[]]
        seen = {()}
        for pick in itertools.product(*[rooted[c] for c in kids[node]]):
            cand = tuple([node] + sorted(j for part in pick for j in part))
            if cand not in seen:
                seen.add(cand)
                options.append(list(cand))
        rooted[node] = options
    return tuple(tuple(tuple(st) for st in rooted[node]) for node in reversed(range(n)))


@functools.lru_cache(maxsize=64)
def _sub_hierarchies_cached(key):
    parents_t, num_bones = key
    num_bones = min(num_bones, len(parents_t))
    return tuple(st for per_root in _rooted_subtrees(parents_t) for st in per_root if len(st) == num_bones)


def remove_approximately_redundant_hierarchies(subtrees_list: List[List[int]], similarity_threshold: float = 0.9):
    kept = _retained_hierarchies_cached(tuple(tuple(st) for st in subtrees_list), float(similarity_threshold))
    print("Retained", str(len(kept)) + "/" + str(len(subtrees_list)), "elements")
    return [list(st) for st in kept]


@functools.lru_cache(maxsize=64)
def _retained_hierarchies_cached(subtrees, similarity_threshold):
    kept = [subtrees[0]]
    for st in subtrees[1:]:
        limit = len(st) * similarity_threshold
        if all(len(set(st) & set(k)) <= limit for k in kept):
            kept.append(st)
    return tuple(kept)


#: part-stage loss terms the device solver fuses (the only ones the shipped configs enable)
_PART_FUSED_LOSSES = {"chamfer", "reg_betas"}
#: further terms of the reference closure (markers_utils.py:477-533), evaluated by `part_extra_losses`
_PART_OPTIONAL_LOSSES = {"reproject", "foot_contact", "foot_velocity", "velocity", "ground"}
#: EXTENSION (BASELINE configs[2] names a "soft-assignment path"; the reference has none): the part stage's data
#: term with a soft minimum over the candidate's vertices, temperature stages.part.soft_tau (m^2); replaces or joins `chamfer`
_PART_EXTENSION_LOSSES = {"soft_chamfer"}


def part_extra_losses(losses: Dict, smpl_inference, smpl_output: Dict, pose_body, betas, root_orient, trans, z_angle,
                      markers_subset_mean, camera: Dict, foot_contacts):
    """The optional terms of the reference's part closure (markers_utils.py:477-544) as differentiable tensor
    expressions over the HIP operators; returns {name: weighted term} for the caller to add in the reference's order.  `camera` holds the best reprojection hypothesis' joints_2d_gt [F,45,2],
    focal_length [1,2], reproject_mask [F], camera_center [F,2] and the (fixed) camera translation [1,3]."""
    from .reprojection import apply_matrix_33_to_vector_3, convert_mocap_pos_to_hmr_pos, perspective_projection
    from .transforms import compute_root_orient_y

    num_frames = pose_body.shape[0]
    device = pose_body.device
    terms = {}
    if "reproject" in losses:
        correction = torch.tensor([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]], device=device).expand(num_frames, 1, 3, 3)
        hmr_cam_trans = torch.repeat_interleave(convert_mocap_pos_to_hmr_pos(camera["cam_trans"]), dim=0,
                                                repeats=num_frames)                       # [F, 3]
        hmr_root_orient = torch.linalg.inv(correction) @ root_orient
        camera_offset = convert_mocap_pos_to_hmr_pos(trans) - hmr_cam_trans
        # the yaw is applied to the body's offset from the camera, not to the body's orientation (:487-492)
        inv_translation = apply_matrix_33_to_vector_3(compute_root_orient_y(z_angle)[:, 0], camera_offset) + hmr_cam_trans
        joints = smpl_inference(poses=pose_body, betas=torch.repeat_interleave(betas, dim=0, repeats=num_frames),
                                root_orient=hmr_root_orient, trans=inv_translation)["joints"]
        kp = perspective_projection(
            points=joints, translation=hmr_cam_trans,
            focal_length=torch.repeat_interleave(camera["focal_length"], dim=0, repeats=num_frames),
            camera_center=camera["camera_center"],
            rotation=torch.eye(3, device=device).unsqueeze(0).expand(num_frames, -1, -1),
        ).reshape((num_frames, 45, 2)) + 0.5
        terms["reproject"] = torch.mean((kp - camera["joints_2d_gt"]) ** 2 * camera["reproject_mask"][:, None, None]) * \
            losses["reproject"]
    feet = [get_joint_id("left_foot"), get_joint_id("right_foot")]
    if "foot_contact" in losses and foot_contacts is not None:
        feet_height = smpl_output["joints"][:, feet, 2]
        terms["foot_contact"] = torch.mean((feet_height - 0.005) ** 2 * foot_contacts) * losses["foot_contact"]
    if "foot_velocity" in losses and foot_contacts is not None:
        vel_xy = smpl_output["joints"][1:, feet, :2] - smpl_output["joints"][:-1, feet, :2]
        speed = torch.norm(vel_xy, dim=-1)
        terms["foot_velocity"] = torch.mean(speed ** 2 * losses["foot_velocity"] * foot_contacts[1:]) * 1.0
    if "velocity" in losses:
        terms["velocity"] = torch.nn.functional.mse_loss(
            trans[1:] - trans[:-1], markers_subset_mean[1:] - markers_subset_mean[:-1]) * losses["velocity"]
    if "ground" in losses:
        terms["ground"] = torch.mean(torch.relu(-smpl_output["vertices"][..., 2])) * losses["ground"]
    return terms



def find_best_part_fits(
    markers: torch.Tensor,  # [F, M, 3]
    pose_body: torch.Tensor,  # [F, J-1, 3, 3]
    betas: torch.Tensor,  # [1, 10]
    root_orient: torch.Tensor,  # [F, 1, 3, 3]
    marker_labels: torch.Tensor,  # [F, M]
    smpl_inference,
    hierarchy: torch.Tensor,  # [J]
    joints_2d_gt: torch.Tensor,
    focal_length: torch.Tensor,
    reproject_mask: torch.Tensor,
    camera_center: torch.Tensor,
    cam_trans: torch.Tensor,
    config: Dict,
    foot_contacts: torch.Tensor = None,
    visualize_fn=None,
    iter_fn: Callable = None,
    execution: Dict = None,
):
    """Rigidly aligns the HMR body to the marker cloud for every candidate body part (yaw about z, translation,
    shape) and keeps the best by two-directional chamfer distance.  Returns the reference's dict.

    `execution` (not in the reference; how the independent candidate solves are scheduled, never what they compute):
    {"subtree_lockstep": True, "subtree_batch": 256, "subtree_threads": 4} -- one lock-step batch of up to
    `subtree_batch` candidates (default), or `subtree_threads` host threads with a stream each.  An `execution` section of
    `config` sets the same keys; the argument wins."""
    exe = merge_execution(config, execution)
    st = config["stages"]["part"]
    if st["mode"] != "cluster":
        raise NotImplementedError("stages.part.mode 'network' needs segmenter checkpoints the reference does not ship")
    unknown = set(st["losses"]) - _PART_FUSED_LOSSES - _PART_OPTIONAL_LOSSES - _PART_EXTENSION_LOSSES
    if unknown:
        raise NotImplementedError("part-stage losses the reference does not define: %s" % sorted(unknown))
    extra = {k for k in st["losses"] if k in (_PART_OPTIONAL_LOSSES | _PART_EXTENSION_LOSSES) and
             (k not in _PART_EXTENSION_LOSSES or float(st["losses"][k]) != 0.0)}
    # the soft-assignment term has a fused closure of its own (k_part_soft) when it stands alone beside the reference's fused
    # terms and the candidate markers fit its instantiations; execution["part_soft_fused"] = False keeps the operator-composed
    # closure (the fused one's checker)
    soft_fused = extra == {"soft_chamfer"} and markers.is_cuda and bool(exe["part_soft_fused"])
    if not any(float(st["losses"].get(k, 0.0)) != 0.0 for k in ("chamfer", "soft_chamfer")):
        raise ValueError("the part stage needs a data term: stages.part.losses.chamfer (reference) or soft_chamfer (extension)")
    if "reproject" in extra and any(v is None for v in (joints_2d_gt, focal_length, reproject_mask, camera_center,
                                                        cam_trans)):
        raise ValueError("the part-stage 'reproject' loss needs the camera of the reprojection_part stage "
                         "(stages.reprojection_part.num_iters > 0)")
    if visualize_fn is not None:
        raise NotImplementedError("visualize_fn is a rendering hook, not built")
    device = markers.device
    num_frames = markers.shape[0]
    # (one read-back of the M per-marker labels; the reference's torch.unique / one torch.where per cluster are a device
    # synchronisation each -- 50 of them for 50 single-marker clusters, 2 ms of an `hmr_full` fit)
    labels_mode_np = torch.mode(marker_labels, axis=0)[0].cpu().numpy()  # [M]
    chain = np.unique(labels_mode_np).tolist()
    print("Found sequence with length", str(len(chain)))
    final_marker_labels = torch.zeros_like(marker_labels)
    final_marker_weights = torch.zeros_like(marker_labels, dtype=torch.float)
    o_betas = betas

    indices = torch.from_numpy(np.concatenate([np.where(labels_mode_np == j)[0] for j in chain])).to(device)
    markers_subset = markers[:, indices].contiguous()
    if soft_fused and int(markers_subset.shape[1]) <= PART_SOFT_MAX_MARKERS:
        extra = set()
    if st.get("use_full_skeleton"):
        subtrees = [np.arange(0, hierarchy.shape[0]).tolist()]
    else:
        subtrees = get_sub_hierachies(hierarchy, len(chain))
        if "similarity_threshold" in st:
            subtrees = remove_approximately_redundant_hierarchies(subtrees, similarity_threshold=0.9)

    trans0 = torch.median(markers, dim=1)[0]
    valid = torch.ones(num_frames, dtype=torch.bool, device=device)

    # vertices owned by each joint (dominant skin weight), in vertex order: one pass instead of 24 masked `nonzero`
    # calls (each a device synchronisation) per candidate -- a constant of the body model, kept with it
    cache = getattr(smpl_inference, "_part_vertex_cache", None)
    if cache is None or cache[0] != (str(device), int(hierarchy.shape[0])):
        vertex_labels = torch.argmax(smpl_inference.get_lbs_weights(), dim=-1)
        order = torch.argsort(vertex_labels, stable=True)
        counts = torch.bincount(vertex_labels, minlength=hierarchy.shape[0]).tolist()
        cache = ((str(device), int(hierarchy.shape[0])), vertex_labels, torch.split(order, counts))
        try:
            smpl_inference._part_vertex_cache = cache
        except AttributeError:  # a caller's own stand-in without attribute storage: just recompute every time
            pass
    _, vertex_labels, joint_vertices = cache

    def part_vertex_indices(subtree):
        return torch.cat([joint_vertices[j] for j in subtree], dim=0)

    group = workspace_group()  # worker threads do not inherit thread-locals

    def fit_subtree(slot: int, subtree, stream):
        """One candidate body part (reference :416-597): L-BFGS over [z, trans, betas], then the ranking score and the
        per-marker labels.  Candidates are independent solves: each worker thread has its own stream and workspace."""
        set_workspace_group(group)
        set_workspace_slot(slot)
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        with ctx:
            vertex_indices = part_vertex_indices(subtree)
            prob = PartProblem(smpl_inference, markers_subset, pose_body, o_betas, root_orient, vertex_indices, config)
            x = prob.pack(torch.zeros((1, 1, 1), device=device), trans0, o_betas)
            point_cb = None
            if iter_fn is not None:
                part_name = ", ".join(get_joint_name(j) for j in subtree)
                pose_np, markers_np = pose_body.detach().cpu().numpy(), markers_subset.detach().cpu().numpy()

                def point_cb(i, loss, x_eval):  # closure_fit_subtree's iter_fn call (markers_utils.py:546-558)
                    e_z, e_trans, e_betas = prob.unpack(x_eval)
                    z_root_eval = compute_root_orient_z(torch.repeat_interleave(e_z, repeats=num_frames, dim=0)) @ \
                        root_orient.detach().cpu()
                    iter_fn(stage="part", iteration=i, pose_body=pose_np, betas=e_betas.numpy().copy(),
                            trans=e_trans.numpy().copy(), root_orient=z_root_eval.numpy(), markers=markers_np,
                            part=part_name, part_joints=np.array(subtree))

            stats = prob.solve(x, max_iter=st["num_iters"], lr=1.0,
                               tolerance_grad=config["optimizer"]["tolerance_grad"],
                               tolerance_change=config["optimizer"]["tolerance_change"], point_callback=point_cb)
            stats["n_subset"], stats["n_markers"] = int(vertex_indices.numel()), int(markers_subset.shape[1])
            z_angle, trans, betas_s = prob.unpack(x)
            with torch.no_grad():
                z_root = compute_root_orient_z(torch.repeat_interleave(z_angle, repeats=num_frames, dim=0)) @ root_orient
                verts = smpl_inference(poses=pose_body, betas=torch.repeat_interleave(betas_s, dim=0, repeats=num_frames),
                                       root_orient=z_root, trans=trans)["vertices"]
                verts_sub = verts[:, vertex_indices].contiguous()
                distance = chamfer_distance(markers_subset, verts_sub, single_directional=False)[0].item()
            out = {"stats": stats, "distance": distance, "betas": betas_s.clone(), "root_orient": z_root.clone(),
                   "trans": trans.clone()}
            if stream is not None:
                stream.synchronize()
        return out

    markers_subset_mean = torch.mean(markers_subset, dim=1)
    camera = None
    if "reproject" in extra:
        camera = {"joints_2d_gt": joints_2d_gt, "focal_length": focal_length, "reproject_mask": reproject_mask,
                  "camera_center": camera_center, "cam_trans": cam_trans[[0]].clone()}

    def fit_subtree_general(slot: int, subtree, stream):
        """Same candidate fit with any of the reference's optional loss terms enabled (none is in a shipped config):
        the closure is composed from the differentiable HIP operators and driven by torch.optim.LBFGS, exactly the
        reference's construction (:422-434,564)."""
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        with ctx:
            vertex_indices = part_vertex_indices(subtree)
            z_angle = torch.zeros((1, 1, 1), device=device).requires_grad_(True)
            trans = trans0.clone().requires_grad_(True)
            betas_s = o_betas.clone().requires_grad_(True)
            # with 'reproject' the camera translation is a (gradient-free, hence fixed) fourth parameter (:422-424)
            params = [z_angle, trans, betas_s] + ([camera["cam_trans"]] if camera is not None else [])
            optimizer = DeviceLBFGS(params, max_iter=st["num_iters"],
                                          tolerance_grad=config["optimizer"]["tolerance_grad"],
                                          tolerance_change=config["optimizer"]["tolerance_change"], lr=1.0,
                                          line_search_fn="strong_wolfe")
            n_eval = [0]

            def forward():
                z_root = compute_root_orient_z(torch.repeat_interleave(z_angle, repeats=num_frames, dim=0)) @ root_orient
                out = smpl_inference(poses=pose_body, betas=torch.repeat_interleave(betas_s, dim=0, repeats=num_frames),
                                     root_orient=z_root, trans=trans)
                return z_root, out

            def closure():
                optimizer.zero_grad()
                n_eval[0] += 1
                z_root_c, out = forward()
                verts_sub = out["vertices"][:, vertex_indices].contiguous()
                loss = 0
                if float(st["losses"].get("chamfer", 0.0)) != 0.0:  # (a child config switches the hard term off with weight 0)
                    loss = loss + chamfer_distance(markers_subset, verts_sub, single_directional=True)[0] * st["losses"]["chamfer"]
                if float(st["losses"].get("soft_chamfer", 0.0)) != 0.0:  # EXTENSION: soft assignment of every marker to the candidate's vertices
                    loss = loss + soft_chamfer_distance(markers_subset, verts_sub, float(st.get("soft_tau", 2.5e-4)))[0] * \
                        st["losses"]["soft_chamfer"]
                terms = part_extra_losses(st["losses"], smpl_inference, out, pose_body, betas_s, root_orient, trans,
                                          z_angle, markers_subset_mean, camera, foot_contacts)
                if "reg_betas" in st["losses"]:
                    terms["reg_betas"] = torch.nn.functional.mse_loss(betas_s, o_betas) * st["losses"]["reg_betas"]
                for name in ("reproject", "reg_betas", "foot_contact", "foot_velocity", "velocity", "ground"):
                    if name in terms:  # the reference's order of accumulation (:477-544)
                        loss = loss + terms[name]
                loss.backward()
                if iter_fn is not None:
                    iter_fn(stage="part", iteration=n_eval[0] - 1, pose_body=pose_body.detach().cpu().numpy(),
                            betas=betas_s.detach().cpu().numpy(), trans=trans.detach().cpu().numpy(),
                            root_orient=z_root_c.detach().cpu().numpy(), markers=markers_subset.detach().cpu().numpy(),
                            part=", ".join(get_joint_name(j) for j in subtree), part_joints=np.array(subtree))
                return loss

            optimizer.step(closure)
            with torch.no_grad():
                z_root, out = forward()
                verts = out["vertices"]
                distance = chamfer_distance(markers_subset, verts[:, vertex_indices].contiguous(),
                                            single_directional=False)[0].item()
            res = {"stats": {"n_eval": n_eval[0], "n_iter": int(optimizer.state[params[0]].get("n_iter", 0)),
                             "first_loss": optimizer.stats.get("first_loss"), "final_loss": optimizer.stats.get("final_loss"),
                             "device_ms": 0.0, "driver": optimizer.stats.get("driver", "device-lbfgs(host closure)")},
                   "distance": distance, "betas": betas_s.detach().clone(), "root_orient": z_root.clone(),
                   "trans": trans.detach().clone()}
            if stream is not None:
                stream.synchronize()
        return res

    def fit_subtrees_lockstep():
        """All candidates in ONE lock-step batch (engine.solve_batch / uuo_batch_solve): they are independent problems of
        one stage and size, so every round launches each kernel once for all of them instead of 202 x 8 launches from four
        host threads.  Same decisions and arithmetic per candidate as the one-by-one solve (bit-identical)."""
        from .engine import part_scores_batch, solve_batch

        vis = [part_vertex_indices(st_) for st_ in subtrees]
        # the candidates share their inputs: one contiguous fp32 copy each (uuo_batch_solve requires ONE body-pose pointer
        # for the shared pose-blend cache; a strided / fp64 caller tensor would otherwise be re-materialised per problem),
        # and no standalone (F, M) workspace per problem -- the batch owns its own
        pose_c, markers_c = _f32(pose_body, "pose_body"), _f32(markers_subset, "markers")
        root_c, betas_c = _f32(root_orient, "root_orient"), _f32(o_betas, "betas")
        probs = [PartProblem(smpl_inference, markers_c, pose_c, betas_c, root_c, vi, config, own_workspace=False)
                 for vi in vis]
        for p_ in probs[1:]:  # one body pose, one pose-blend cache for the whole batch
            p_.problem.pose_cache_id = probs[0].problem.pose_cache_id
        z0 = torch.zeros((1, 1, 1), device=device)
        x_all = probs[0].pack(z0, trans0, o_betas).repeat(len(probs), 1)   # same start for every candidate (:418-421)
        xs = [x_all[i] for i in range(len(probs))]
        out = []
        chunk = max(1, int(exe["subtree_batch"]))
        for c0 in range(0, len(probs), chunk):
            sl = slice(c0, c0 + chunk)
            stats_l = solve_batch(probs[sl], xs[sl], max_iter=st["num_iters"], lr=1.0,
                                  tolerance_grad=config["optimizer"]["tolerance_grad"],
                                  tolerance_change=config["optimizer"]["tolerance_change"])
            # ranking score (reference :566-579): two-directional chamfer at the solved parameters, all candidates of the
            # chunk in one batched forward + one score kernel (was one SMPL forward + two searches per candidate)
            dists = part_scores_batch(probs[sl], xs[sl])
            for k, (stt, dval) in enumerate(zip(stats_l, dists)):
                stt["n_subset"], stt["n_markers"] = int(vis[c0 + k].numel()), int(markers_subset.shape[1])
                out.append({"stats": stt, "distance": dval, "x": xs[c0 + k], "prob": probs[c0 + k]})
        return out

    if extra:
        fit_subtree = fit_subtree_general

    n_threads = min(len(subtrees), max(1, int(exe["subtree_threads"])))
    if extra:
        n_threads = 1  # autograd graphs of concurrent candidates would share the engine's forward scratch
    lockstep = (not extra and iter_fn is None and device.type == "cuda" and len(subtrees) > 1
                and bool(exe["subtree_lockstep"]))
    if lockstep:
        results = fit_subtrees_lockstep()
    elif n_threads > 1 and device.type == "cuda":
        main_stream = torch.cuda.current_stream(device)
        streams = worker_streams(device, n_threads, "subtree")
        for s_ in streams:
            s_.wait_stream(main_stream)
        free_slots = queue.Queue()
        for i in range(n_threads):
            free_slots.put(i)

        def worker(subtree):
            i = free_slots.get()
            try:
                return fit_subtree(i, subtree, streams[i])
            finally:
                free_slots.put(i)

        results = list(worker_pool(n_threads, "subtree").map(worker, subtrees))
        for s_ in streams:
            main_stream.wait_stream(s_)
        set_workspace_slot(0)
    else:
        results = [fit_subtree(0, subtree, None) for subtree in subtrees]

    best = None
    best_distance = np.inf
    subtree_losses = []
    LAST_STATS["part"] = []
    best_res = None
    for subtree, res in zip(subtrees, results):  # candidate order decides ties, exactly as the sequential reference
        LAST_STATS["part"].append(res["stats"])
        subtree_losses.append([subtree, res["distance"]])
        if res["distance"] < best_distance:
            best_distance = res["distance"]
            best_res = res
    if "betas" not in best_res:  # lock-step path: only the winner's parameters are unpacked
        z_angle, trans_w, betas_w = best_res["prob"].unpack(best_res["x"])
        best_res = dict(best_res, betas=betas_w.clone(), trans=trans_w.clone(),
                        root_orient=(compute_root_orient_z(torch.repeat_interleave(z_angle, repeats=num_frames, dim=0))
                                     @ root_orient).clone())
    best = {
        "betas": best_res["betas"], "markers_subset": markers_subset.clone(), "root_orient": best_res["root_orient"],
        "trans": best_res["trans"],
        "aabb": get_aabb_volume(get_aabb(markers_subset)) / get_aabb_volume(get_aabb(markers)),
    }

    # The reference relabels the markers every time a candidate improves on the best so far (:586-597); only the last
    # such relabelling survives, i.e. the winner's: label of marker i = dominant joint of argmin_v mean_f |v - x_i| over
    # ALL vertices of the winning candidate's body.  Computed once, for the winner.
    with torch.no_grad():
        verts = smpl_inference(poses=pose_body, betas=torch.repeat_interleave(best["betas"], dim=0, repeats=num_frames),
                               root_orient=best["root_orient"], trans=best["trans"])["vertices"]
        near = smpl_inference.device_model.assign_mean_argmin(verts, markers_subset, valid).long()
        final_marker_labels[:, indices] = vertex_labels[near][None, :].to(final_marker_labels.dtype)

    if len(subtree_losses) > 1:
        subtree_losses = sorted(subtree_losses, key=lambda e: e[1])
        final_marker_weights[:, indices] = subtree_losses[1][1] / subtree_losses[0][1]
        if indices.shape[0] == 1:
            final_marker_weights *= 0
    for k in range(min(len(subtree_losses), 3)):
        print(", ".join(get_joint_name(j) for j in subtree_losses[k][0]), "{:.6f}".format(subtree_losses[k][1]))
    print("--------------------")
    final_marker_weights = final_marker_weights / torch.max(final_marker_weights)

    return {
        "betas": best["betas"].clone(),
        "marker_labels": final_marker_labels.clone(),
        "markers_subset": best["markers_subset"].clone(),
        "marker_weights": final_marker_weights.clone(),
        "root_orient": best["root_orient"].clone(),
        "trans": best["trans"].clone(),
        "aabb_volume_ratio": best["aabb"].clone(),
        "chain": np.array(list(subtree_losses[0][0]), dtype=np.int32),
    }
