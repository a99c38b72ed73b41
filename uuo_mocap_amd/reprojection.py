"""Reprojection stage (reference src/video_mocap/utils/hmr_utils.py:14-124,136-161,170-425): one yaw hypothesis of the
camera-consistent placement -- the HMR body is rotated about the camera's vertical axis and translated so that its 45
joints reproject onto the HMR 2D key points while its surface stays on the markers.

Disabled in every shipped configuration (`stages.reprojection_part.num_iters: 0`).  Since round 3 the closure is ONE fused
evaluation of the library (`uuo_reprojection_solve`, csrc/reprojection.hip; `engine.ReprojectionProblem`): pose, shape and
HMR root orientation are constants of this solve, so one SMPL forward before it replaces the two per closure, and the
chamfer search runs against a constant cloud.  `driver="operators"` keeps the closure composed from the differentiable
operators of this package -- `SmplInference` (HIP forward, `uuo_smpl_backward`) and `chamfer_distance` (HIP nearest
neighbour) under `DeviceLBFGS` (3F + 14 parameters) -- as the cross-check of the fused one (tests/test_gpu_parity.py).
The camera algebra around the solve is a handful of element-wise tensor expressions."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from .device_lbfgs import DeviceLBFGS
from .engine import ReprojectionProblem
from .losses import chamfer_distance
from .smpl import SmplInference
from .transforms import compute_root_orient_y

HMR_FOCAL_LENGTH = 5000.0  # HMR 2.0 default (hmr_utils.py:67)
HMR_IMG_SIZE = 256


def perspective_projection(points: torch.Tensor, translation: torch.Tensor, focal_length: torch.Tensor,
                           camera_center: Optional[torch.Tensor] = None,
                           rotation: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B,N,3] points -> [B,N,2] pixels: rotate, translate, divide by depth, apply the intrinsics
    K = [[fx,0,cx],[0,fy,cy],[0,0,1]] (hmr_utils.py:14-52)."""
    if rotation is not None:
        points = torch.einsum("bij,bkj->bki", rotation, points)
    cam = points + translation.unsqueeze(1)
    cam = cam / cam[:, :, -1].unsqueeze(-1)
    uv = cam[:, :, :2] * focal_length.unsqueeze(1)
    if camera_center is not None:
        uv = uv + camera_center.unsqueeze(1)
    return uv


def convert_hmr_pos_to_mocap_pos(pos):
    """(x, y, z)_hmr -> (x, z, -y) (hmr_utils.py:127-134)."""
    return torch.cat((pos[..., [0]], pos[..., [2]], pos[..., [1]] * -1), dim=-1)


def convert_mocap_pos_to_hmr_pos(pos):
    """(x, y, z)_mocap -> (x, -z, y) (hmr_utils.py:137-144)."""
    return torch.cat((pos[..., [0]], pos[..., [2]] * -1, pos[..., [1]]), dim=-1)


def apply_matrix_33_to_vector_3(mat, vec):
    return (mat @ vec[..., None])[..., 0]


def get_3d_parameters(smpl_inference: SmplInference, pred_smpl_betas, pred_smpl_body_pose, pred_smpl_global_orient,
                      pred_cam, center, size, scale) -> Dict:
    """HMR 2.0 / PHALP weak-perspective camera -> full-perspective translation and the 2D key points it implies
    (hmr_utils.py:57-124)."""
    device, dtype = pred_cam.device, pred_cam.dtype
    n = pred_cam.shape[0]
    new_image_size = torch.max(size, dim=-1, keepdim=True)[0]
    top = (new_image_size - size[:, [0]]) // 2
    left = (new_image_size - size[:, [1]]) // 2
    ratio = 1.0 / torch.round(new_image_size) * HMR_IMG_SIZE
    center = (center + torch.cat((left, top), dim=-1).to(device)) * ratio
    scale = scale * new_image_size * ratio
    focal_length = HMR_FOCAL_LENGTH * torch.ones(n, 2, device=device, dtype=dtype)
    joints = smpl_inference(pred_smpl_body_pose, pred_smpl_betas, pred_smpl_global_orient,
                            torch.zeros((n, 3), device=device, dtype=dtype))["joints"]
    depth = 2 * focal_length[:, 0] / (pred_cam[:, 0] * scale[:, 0] + 1e-9)
    cam_xy = torch.stack([pred_cam[:, 1], pred_cam[:, 2]], dim=1)
    pred_cam_t = torch.cat((cam_xy + (center - HMR_IMG_SIZE / 2.0) * depth[:, None] / focal_length, depth[:, None]), dim=1)
    camera_center = torch.zeros(n, 2, device=device, dtype=dtype)
    rotation = torch.eye(3, device=device, dtype=dtype).unsqueeze(0).expand(n, -1, -1)
    kp = perspective_projection(joints, pred_cam_t, focal_length / HMR_IMG_SIZE, camera_center, rotation)
    kp = (kp + 0.5) * HMR_IMG_SIZE
    return {"camera_center": camera_center, "focal_length": focal_length / HMR_IMG_SIZE, "pred_cam_t": pred_cam_t,
            "pred_joints": joints, "pred_keypoints_2d_smpl": kp / HMR_IMG_SIZE, "rotation": rotation}


def _prepare(markers, pose_body, betas, hmr_betas, root_orient, trans, pred_cam, cam_center, cam_size, cam_scale, angle,
             smpl_inference: SmplInference, config: Dict) -> SimpleNamespace:
    """Everything optim_reprojection sets up before its solve (hmr_utils.py:196-279): the HMR camera, the target key
    points and frame mask, and the starting values of the optimised leaves."""
    device = markers.device
    F = pose_body.shape[0]
    w = config["stages"]["reprojection_part"]["losses"]
    pose_body = pose_body.clone()
    betas = betas.clone().detach()
    root_orient = root_orient.clone()
    trans = trans.clone()
    correction = torch.tensor([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]], device=device).expand(F, 1, 3, 3)  # HMR -> mocap axes

    cam = get_3d_parameters(smpl_inference, hmr_betas, pose_body, root_orient, pred_cam.clone(), cam_center.clone(),
                            cam_size.clone(), cam_scale.clone())
    kp_target = torch.nan_to_num(cam["pred_keypoints_2d_smpl"][None], 0)        # [1, F, 45, 2]
    cam_t = cam["pred_cam_t"]
    mask = torch.mean((cam_t == cam_t).float(), dim=-1).detach()                # [F]: frames with a valid HMR camera
    cam_t = torch.nan_to_num(cam_t, 0)

    # the body takes the camera-space translation (moved onto the marker cloud), the camera takes the old body one
    cam_translation = trans.detach().clone()
    offset = convert_mocap_pos_to_hmr_pos(torch.median(markers.reshape(-1, 3), dim=0, keepdim=True)[0]) - \
        torch.median(cam_t, dim=0, keepdim=True)[0]
    body_t = (cam_t + offset)[None].clone().requires_grad_(True)                           # [1, F, 3]
    cam_single = torch.mean(cam_translation - offset, dim=0, keepdim=True).clone().requires_grad_(True)  # [1, 3]
    yaw = (torch.ones(1, 1, 1, 1, device=device) * angle).to(device).requires_grad_(True)
    focal = torch.mean(cam["focal_length"], dim=0, keepdim=True)                           # [1, 2]
    return SimpleNamespace(device=device, F=F, w=w, pose_body=pose_body, betas=betas, root_orient=root_orient,
                           correction=correction, cam=cam, kp_target=kp_target, mask=mask, body_t=body_t,
                           cam_single=cam_single, yaw=yaw, focal=focal)


def _fused_problem(pr: SimpleNamespace, markers, smpl_inference: SmplInference):
    """(engine.ReprojectionProblem, x0) of a prepared hypothesis.  Pose, shape (detached: hmr_utils.py:218,292) and HMR root
    orientation never change in this solve, so ONE forward here stands for the two of every closure evaluation.
    x = [yaw | body_t | cam_single | betas]: the order of the reference's params list (:276-279)."""
    with torch.no_grad():
        fwd0 = smpl_inference(pr.pose_body, pr.betas.expand(pr.F, 10), pr.root_orient,
                              torch.zeros((pr.F, 3), device=pr.device))
        x0 = torch.cat([pr.yaw.detach().reshape(-1), pr.body_t.detach().reshape(-1), pr.cam_single.detach().reshape(-1),
                        pr.betas.reshape(-1)]).contiguous()
        problem = ReprojectionProblem(markers, fwd0["joints"], fwd0["vertices"], pr.kp_target[0], pr.mask,
                                      pr.focal[0].tolist(), pr.cam["camera_center"][0].tolist(), pr.w["reprojection"],
                                      pr.w["chamfer"])
    return problem, x0


def reprojection_problem(markers, pose_body, betas, hmr_betas, root_orient, trans, pred_cam, cam_center, cam_size,
                         cam_scale, angle, smpl_inference: SmplInference, config: Dict):
    """The fused closure of one yaw hypothesis and its starting point, without solving: (engine.ReprojectionProblem, x0)."""
    pr = _prepare(markers, pose_body, betas, hmr_betas, root_orient, trans, pred_cam, cam_center, cam_size, cam_scale, angle,
                  smpl_inference, config)
    return _fused_problem(pr, markers, smpl_inference)


def optim_reprojection(markers, pose_body, betas, hmr_betas, root_orient, trans, pred_cam, cam_center, cam_size,
                       cam_scale, angle, img_mask, smpl_inference: SmplInference, num_iters: int, config: Dict,
                       verbose: bool = False, iter_fn=None, driver: str = "fused") -> Dict:
    """One yaw hypothesis `angle` (0-d tensor) of the reprojection fit (hmr_utils.py:170-425).  Optimises the yaw
    about the camera's vertical axis [1], the per-frame body translation [F,3] (HMR axes), one camera translation
    [3] and the shape [10] on  mean((kp - kp_hmr)^2 * mask) * w_reprojection + chamfer(markers -> vertices) * w_chamfer.
    Returns the reference's dictionary (leading hypothesis axis of size 1 kept).  `driver`: "fused" (the library's fused
    closure) or "operators" (the closure composed from this package's differentiable operators; the cross-check)."""
    if driver not in ("fused", "operators"):
        raise ValueError("optim_reprojection: driver must be 'fused' or 'operators'")
    pr = _prepare(markers, pose_body, betas, hmr_betas, root_orient, trans, pred_cam, cam_center, cam_size, cam_scale,
                  angle, smpl_inference, config)
    device, F, w, pose_body, betas, root_orient, correction = pr.device, pr.F, pr.w, pr.pose_body, pr.betas, pr.root_orient, pr.correction
    cam, kp_target, mask, body_t, cam_single, yaw, focal = pr.cam, pr.kp_target, pr.mask, pr.body_t, pr.cam_single, pr.yaw, pr.focal
    eye = torch.eye(3, device=device).unsqueeze(0).expand(F, -1, -1)
    state = {}
    n_eval = [0]

    def forward_terms():
        cam_tr = cam_single[:, None].expand(1, F, 3)
        yaw_f = yaw.expand(1, F, 1, 1)
        y_root = compute_root_orient_y(yaw_f) @ root_orient                                     # [1, F, 1, 3, 3]
        # rotate the body about the camera instead of rotating the camera
        inv_t = apply_matrix_33_to_vector_3(compute_root_orient_y(-yaw_f)[:, 0], body_t - cam_tr) + cam_tr
        betas_f = betas.expand(F, 10)
        joints = smpl_inference(pose_body, betas_f, root_orient, inv_t[0])["joints"]
        kp = perspective_projection(joints, cam_tr[0], focal.expand(F, 2), cam["camera_center"], eye)[None] + 0.5
        state.update(cam_tr=cam_tr, y_root=y_root, inv_t=inv_t, kp=kp)
        return kp, y_root, betas_f

    def report(loss_value, kp, y_root):
        if verbose:
            print("Reprojection", float(loss_value))
        if iter_fn is not None:  # hmr_utils.py:350-362
            iter_fn(stage="reprojection", iteration=n_eval[0], pose_body=pose_body.detach().cpu().numpy(),
                    betas=betas.detach().cpu().numpy(), trans=convert_hmr_pos_to_mocap_pos(body_t)[0].detach().cpu().numpy(),
                    root_orient=(correction @ y_root)[0].detach().cpu().numpy(), pred_angle=yaw.item(), initial_angle=angle,
                    pred_2d_joints=kp[0].detach().cpu().numpy(), gt_2d_joints=kp_target[0].detach().cpu().numpy())
        n_eval[0] += 1

    if driver == "fused":
        problem, x = _fused_problem(pr, markers, smpl_inference)
        with torch.no_grad():
            sizes = (1, 3 * F, 3, betas.numel())

            def unpack(vec):
                a, b, c, _ = torch.split(vec, sizes)
                yaw.copy_(a.view_as(yaw)); body_t.copy_(b.view_as(body_t)); cam_single.copy_(c.view_as(cam_single))

            def on_point(i, loss_value, host_x):
                unpack(host_x.to(device))
                kp, y_root, _ = forward_terms()
                report(loss_value, kp, y_root)

            want_points = iter_fn is not None
            stats = problem.solve(x, num_iters, lr=1.0, tolerance_grad=config["optimizer"]["tolerance_grad"],
                                  tolerance_change=config["optimizer"]["tolerance_change"],
                                  callback=(lambda i, l: report(l, None, None)) if (verbose and not want_points) else None,
                                  point_callback=on_point if want_points else None)
            # the derived quantities of the LAST closure evaluation (below), then the accepted parameters
            unpack(stats["x_last"])
            forward_terms()
            state["kp"] = stats["kp_last"][None]
            unpack(x)
        state["solver"] = {k: v for k, v in stats.items() if k not in ("x_last", "kp_last")}
    else:
        # `betas` is in the parameter list but detached (hmr_utils.py:218,292): it receives no gradient and stays put
        optimizer = DeviceLBFGS([yaw, body_t, cam_single, betas], max_iter=num_iters,
                                tolerance_grad=config["optimizer"]["tolerance_grad"],
                                tolerance_change=config["optimizer"]["tolerance_change"], lr=1.0,
                                line_search_fn="strong_wolfe")

        def closure():
            optimizer.zero_grad()
            kp, y_root, betas_f = forward_terms()
            loss = torch.mean((kp - kp_target) ** 2 * mask[None, :, None, None]) * w["reprojection"]
            verts = smpl_inference(pose_body, betas_f, (correction @ y_root)[0],
                                   convert_hmr_pos_to_mocap_pos(body_t)[0])["vertices"]
            loss = loss + chamfer_distance(markers, verts, single_directional=True)[0] * w["chamfer"]
            loss.backward()
            report(loss, kp, y_root)
            return loss

        optimizer.step(closure)
        state["solver"] = dict(optimizer.stats)
    # Like the reference (its `nonlocal` temporaries), the derived quantities below are those of the LAST closure
    # evaluation, which is not necessarily the accepted point of the line search; the parameters themselves
    # (`trans`, `output_angle`) are the accepted ones.
    with torch.no_grad():
        kp, y_root = state["kp"].detach(), state["y_root"].detach()
        betas_f = betas.expand(F, 10)
        world = smpl_inference(pose_body, betas_f, (correction @ y_root)[0], state["inv_t"][0].detach())["vertices"]
        reproject_error = torch.mean((kp[0] - kp_target[0]) ** 2 * mask[None, :, None, None]).item()
        chamfer_error = chamfer_distance(markers, world, single_directional=True)[0].item()
    return {
        "pose_body": pose_body[None].clone().detach(), "betas": betas_f[None].clone().detach(),
        "root_orient": (correction @ y_root).clone().detach(),
        "trans": convert_hmr_pos_to_mocap_pos(body_t.detach()).clone(),
        "joints_2d": kp.clone().detach(), "joints_2d_gt": kp_target,
        "cam_trans": convert_hmr_pos_to_mocap_pos(state["cam_tr"].detach()).clone(),
        "camera_center": cam["camera_center"].clone(), "focal_length": focal.clone(), "reproject_mask": mask.clone(),
        "input_angle": float(angle), "output_angle": yaw.item(),
        "metrics": {"chamfer": chamfer_error, "reproject": reproject_error},
        "solver": state["solver"],
    }
