"""oracle/stages_ref.py must reproduce what the reference's own stage solvers produced (tests/golden/*.npz,
captured by oracle/make_golden.py from /root/reference's modules).  Same torch, same primitives, single
thread -> the trajectories agree to fp32 round-off; tolerances below leave room for a different host CPU."""
import copy

import numpy as np
import pytest
import torch

from oracle import stages_ref
from uuo_mocap_amd.config import packaged_config
from uuo_mocap_amd.synthetic import SyntheticImgSmpl, SyntheticMarkers


@pytest.fixture(autouse=True)
def _single_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def _t(x):
    return torch.from_numpy(np.asarray(x)).clone()


def _cfg(name, part, chamfer, marker):
    cfg = packaged_config(name)
    for k, v in (("part", part), ("chamfer", chamfer), ("marker", marker)):
        if cfg["stages"][k]["num_iters"] > 0:
            cfg["stages"][k]["num_iters"] = int(v)
    return cfg


def test_smpl_forward_and_chamfer(golden, oracle_smpl):
    g = golden("smpl_forward.npz")
    markers = _t(g["markers"])
    out = oracle_smpl(_t(g["hmr_pose_body"]), _t(g["hmr_betas"]), _t(g["hmr_root_orient"]), _t(g["trans0"]))
    np.testing.assert_allclose(out["vertices"].numpy()[:, ::int(g["vertex_stride"])], g["vertices"], atol=1e-6)
    np.testing.assert_allclose(out["joints"].numpy(), g["joints"], atol=1e-6)
    loss, _ = stages_ref.weighted_chamfer_distance(markers, out["vertices"], stages_ref.get_marker_mask(markers))
    np.testing.assert_allclose(loss.item(), float(g["chamfer_loss"]), rtol=1e-5)


def test_chamfer_stage_matches_reference(golden, oracle_smpl):
    g = golden("chamfer_stage.npz")
    cfg = _cfg("video_mocap", 12, g["num_iters"], 25)
    markers = _t(g["markers"])
    o_pose, o_betas = _t(g["hmr_pose_body"]), _t(g["o_betas"])
    pose = o_pose.clone().requires_grad_(True)
    betas = o_betas.clone().requires_grad_(True)
    root = _t(g["hmr_root_orient"]).requires_grad_(True)
    trans = _t(g["trans0"]).requires_grad_(True)
    # first closure: loss and flat gradient in the reference's packing [trans | z | betas | pose_body]
    z = torch.zeros(root.shape[0], 1, 1, requires_grad=True)
    loss, _ = stages_ref.chamfer_stage_loss(markers, pose, o_pose, betas, o_betas, root.detach(), trans, z,
                                            oracle_smpl, cfg)
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in (trans, z, betas, pose)]).numpy()
    np.testing.assert_allclose(loss.item(), g["losses"][0], rtol=1e-6)
    np.testing.assert_allclose(flat, g["first_grad"], rtol=1e-4, atol=1e-7)
    for p in (trans, betas, pose):
        p.grad = None
    trace = []
    stages_ref.optim_chamfer(markers, pose, o_pose, betas, o_betas, root, trans, oracle_smpl, cfg, trace=trace)
    assert len(trace) == len(g["losses"])
    np.testing.assert_allclose(trace, g["losses"], rtol=1e-4)
    np.testing.assert_allclose(trans.detach().numpy(), g["out_trans"], atol=1e-4)
    np.testing.assert_allclose(pose.detach().numpy(), g["out_pose_body"], atol=1e-4)
    np.testing.assert_allclose(betas.detach().numpy(), g["out_betas"], atol=1e-4)
    np.testing.assert_allclose(root.detach().numpy(), g["out_root_orient"], atol=1e-4)


def test_placement_and_marker_stage_match_reference(golden, oracle_smpl):
    g = golden("marker_stage.npz")
    cfg = _cfg("video_mocap", 12, 25, g["num_iters"])
    markers = _t(g["markers"])
    o_pose, o_betas = _t(g["o_pose_body"]), _t(g["o_betas"])
    pose = _t(g["in_pose_body"]).requires_grad_(True)
    root = _t(g["in_root_orient"]).requires_grad_(True)
    betas = _t(g["in_betas"]).requires_grad_(True)
    trans = _t(g["in_trans"]).requires_grad_(True)
    one_hot, idx = stages_ref.compute_nearest_points(markers, pose, betas, root, trans, oracle_smpl,
                                                     _t(g["img_mask"]), cfg, return_indices=True)
    np.testing.assert_array_equal(idx, g["place_idx"])
    loss, _ = stages_ref.marker_stage_loss(markers, pose, o_pose, betas, o_betas, root, trans, one_hot,
                                           oracle_smpl, cfg)
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in (pose, betas, root, trans)]).numpy()
    np.testing.assert_allclose(loss.item(), g["losses"][0], rtol=1e-6)
    np.testing.assert_allclose(flat, g["first_grad"], rtol=1e-4, atol=1e-7)
    for p in (pose, betas, root, trans):
        p.grad = None
    trace = []
    stages_ref.optim_markers(markers, pose, o_pose, betas, o_betas, root, trans, one_hot, oracle_smpl, cfg,
                             trace=trace)
    assert len(trace) == len(g["losses"])
    np.testing.assert_allclose(trace, g["losses"], rtol=1e-4)
    np.testing.assert_allclose(pose.detach().numpy(), g["out_pose_body"], atol=1e-4)
    np.testing.assert_allclose(trans.detach().numpy(), g["out_trans"], atol=1e-4)


@pytest.mark.parametrize("tag,cfg_name", [("full", "hmr_full"), ("tree", "hmr_part")])
def test_part_stage_matches_reference(golden, oracle_smpl, tag, cfg_name):
    g = golden("part_stage_%s.npz" % tag)
    cfg = _cfg(cfg_name, g["num_iters"], 25, 25)
    trace = {}
    out = stages_ref.find_best_part_fits(_t(g["markers"]), _t(g["hmr_pose_body"]), _t(g["o_betas"]),
                                         _t(g["hmr_root_orient"]), _t(g["seg"]), oracle_smpl,
                                         oracle_smpl.smpl.parents, cfg, trace=trace)
    assert len(trace["evals"]) == int(g["n_subtrees"])
    np.testing.assert_array_equal([len(e) for e in trace["evals"]], g["n_evals"])
    np.testing.assert_allclose([e[0] for e in trace["evals"]], g["first_losses"], rtol=1e-5)
    np.testing.assert_allclose([e[-1] for e in trace["evals"]], g["final_losses"], rtol=1e-3)
    np.testing.assert_array_equal(out["chain"], g["out_chain"])
    np.testing.assert_array_equal(out["marker_labels"].numpy(), g["out_marker_labels"])
    np.testing.assert_allclose(out["trans"].detach().numpy(), g["out_trans"], atol=1e-4)
    np.testing.assert_allclose(out["betas"].detach().numpy(), g["out_betas"], atol=1e-4)
    np.testing.assert_allclose(out["root_orient"].detach().numpy(), g["out_root_orient"], atol=1e-4)
    np.testing.assert_allclose(out["marker_weights"].numpy(), g["out_marker_weights"], rtol=1e-3, equal_nan=True)
    np.testing.assert_allclose(out["aabb_volume_ratio"].numpy(), g["out_aabb"], rtol=1e-5)


@pytest.mark.parametrize("tag,cfg_name", [("hmr_full", "hmr_full"), ("default", "video_mocap")])
def test_end_to_end_matches_reference(golden, oracle_smpl, tag, cfg_name):
    g = golden("e2e_%s.npz" % tag)
    cfg = _cfg(cfg_name, g["part_iters"], g["chamfer_iters"], g["marker_iters"])
    F = g["markers"].shape[0]
    img = SyntheticImgSmpl(
        trans=_t(g["hmr_trans"]), root_orient=_t(g["hmr_root_orient"]), hmr_root_orient=_t(g["hmr_root_orient"]),
        pose_body=_t(g["hmr_pose_body"]), betas=_t(g["hmr_betas"]), foot_contacts=torch.zeros(F, 2),
        camera_bbox=torch.zeros(F, 3), center=torch.zeros(F, 2), scale=torch.zeros(F, 1), size=torch.zeros(F, 2),
        img_mask=_t(g["img_mask"]), freq=30.0)
    stats = {}
    out = stages_ref.multimodal_video_mocap(img, SyntheticMarkers(g["markers"].copy(), 30.0), oracle_smpl, cfg,
                                            stats=stats)
    n_solves = len(stats["part"]["evals"]) + sum(len(stats.get(k, [])) for k in ("chamfer", "marker", "marker_final"))
    assert n_solves == int(g["n_solves"])
    np.testing.assert_array_equal(out["markers_labels"], g["out_markers_labels"])
    np.testing.assert_array_equal(out["chain"], g["out_chain"])
    np.testing.assert_allclose(out["trans"].numpy(), g["out_trans"], atol=2e-4)
    np.testing.assert_allclose(out["pose_body"].numpy(), g["out_pose_body"], atol=2e-4)
    np.testing.assert_allclose(out["root_orient"].numpy(), g["out_root_orient"], atol=2e-4)
    np.testing.assert_allclose(out["betas"].numpy(), g["out_betas"], atol=2e-4)


def test_reprojection_stage_matches_reference(oracle_smpl, golden):
    """The oracle's restatement of optim_reprojection against the fixture captured from the reference's own
    hmr_utils.optim_reprojection (two yaw hypotheses): first loss / gradient, the loss trajectory, final outputs."""
    g = golden("reprojection_stage.npz")
    cfg = packaged_config("video_mocap")
    cfg["stages"]["reprojection_part"]["num_iters"] = 200
    t = lambda k: torch.from_numpy(np.asarray(g[k])).float()
    for name, angle in (("a0", 0.0), ("a1", float(np.pi / 2))):
        trace, cap = [], {}
        out = stages_ref.optim_reprojection(
            markers=t("markers"), pose_body=t("hmr_pose_body"), betas=t("betas"), hmr_betas=t("hmr_betas"),
            root_orient=t("hmr_root_orient"), trans=t("trans"), pred_cam=t("pred_cam"), cam_center=t("center"),
            cam_size=t("size"), cam_scale=t("scale"), angle=torch.tensor(angle), img_mask=t("img_mask"),
            smpl_inference=oracle_smpl, num_iters=200, config=cfg, trace=trace, capture=cap)
        # the first closure evaluation: same point, same gradient as the reference's own torch.optim.LBFGS saw
        np.testing.assert_allclose(cap["params"].numpy(), g[name + "_first_params"], atol=1e-5)
        gref = g[name + "_first_grad"]
        assert np.linalg.norm(cap["grad"].numpy() - gref) / np.linalg.norm(gref) < 1e-4
        ref = g[name + "_losses"]
        n = min(len(trace), len(ref), 20)
        np.testing.assert_allclose(trace[:n], ref[:n], rtol=2e-4)
        assert abs(len(trace) - len(ref)) <= max(10, len(ref) // 5)
        np.testing.assert_allclose(out["joints_2d_gt"].numpy(), g[name + "_joints_2d_gt"], atol=1e-5)
        np.testing.assert_allclose(out["focal_length"].numpy(), g[name + "_focal_length"], rtol=1e-6)
        np.testing.assert_allclose(out["reproject_mask"].numpy(), g[name + "_reproject_mask"])
        assert out["input_angle"] == pytest.approx(float(g[name + "_angles"][0]))
        assert out["output_angle"] == pytest.approx(float(g[name + "_angles"][1]), abs=5e-3)
        np.testing.assert_allclose(out["trans"].numpy(), g[name + "_trans"], atol=5e-3)
        np.testing.assert_allclose(out["joints_2d"].numpy(), g[name + "_joints_2d"], atol=5e-3)
        assert trace[-1] == pytest.approx(float(ref[-1]), rel=0.05)


def test_reprojection_closure_at_baseline_size_matches_reference(oracle_smpl, tables, golden):
    """The oracle's reprojection closure at 300 x 50 against the reference's own first evaluation there
    (reprojection_stage_300x50.npz): same starting point, loss and gradient."""
    from uuo_mocap_amd.synthetic import make_sequence, synthetic_hmr_camera

    g = golden("reprojection_stage_300x50.npz")
    F, M = int(g["F"]), int(g["M"])
    seq = make_sequence(tables, seed=int(g["seed"]), num_frames=F, num_markers=M)
    markers = torch.from_numpy(seq.markers.get_points()).float()
    assert float(np.abs(markers.double().numpy()).sum()) == pytest.approx(float(g["markers_checksum"]), rel=1e-12)
    img = seq.img_smpl
    betas = (img.betas.sum(0, keepdim=True) / img.img_mask.sum()).clone()
    trans = torch.median(markers, dim=1)[0].clone()
    pred_cam, center, size, scale = synthetic_hmr_camera(F)
    cfg = packaged_config("video_mocap")
    cap = {}
    stages_ref.optim_reprojection(
        markers=markers, pose_body=img.pose_body.clone(), betas=betas, hmr_betas=img.betas.clone(),
        root_orient=img.hmr_root_orient.clone(), trans=trans, pred_cam=pred_cam, cam_center=center, cam_size=size,
        cam_scale=scale, angle=torch.tensor(0.0), img_mask=img.img_mask, smpl_inference=oracle_smpl, num_iters=1, config=cfg,
        capture=cap)
    np.testing.assert_allclose(cap["params"].numpy(), g["a0_first_params"], atol=1e-5)
    assert cap["loss"] == pytest.approx(float(g["a0_losses"][0]), rel=1e-5)
    gref = g["a0_first_grad"]
    assert np.linalg.norm(cap["grad"].numpy() - gref) / np.linalg.norm(gref) < 1e-4


def _part_losses_case(g):
    cfg = _cfg("hmr_part", g["num_iters"], 25, 25)
    cfg["stages"]["part"]["losses"] = {str(k): float(v) for k, v in zip(g["loss_names"], g["loss_weights"])}
    camera = {k[4:]: _t(g[k]) for k in ("cam_joints_2d_gt", "cam_focal_length", "cam_reproject_mask", "cam_cam_trans",
                                        "cam_camera_center")}
    return cfg, camera


def test_part_stage_optional_losses_match_reference(golden, oracle_smpl):
    """Part stage with every optional term of the reference closure enabled (reproject, foot_contact, foot_velocity,
    velocity, ground): the oracle against the fixture captured from the reference's own find_best_part_fits --
    first loss of all 26 candidates, first gradient, and the full L-BFGS trajectories of the first two."""
    g = golden("part_stage_losses.npz")
    cfg, camera = _part_losses_case(g)
    markers, pose, o_betas, root = _t(g["markers"]), _t(g["pose_body"]), _t(g["o_betas"]), _t(g["o_root_orient"])
    seg, fc = _t(g["seg"]), _t(g["foot_contacts"])
    subtrees = stages_ref.remove_approximately_redundant_hierarchies(
        stages_ref.get_sub_hierarchies(oracle_smpl.smpl.parents, 3), 0.9)
    assert len(subtrees) == int(g["n_subtrees"])
    vlabels = torch.argmax(oracle_smpl.get_lbs_weights(), dim=-1)
    labels_mode = torch.mode(seg, axis=0)[0]
    indices = torch.cat([torch.where(labels_mode == j)[0] for j in torch.unique(labels_mode).tolist()], dim=0)
    sub = markers[:, indices]
    cam = dict(camera, cam_trans=camera["cam_trans"][[0]].clone())
    first = []
    for k, subtree in enumerate(subtrees):
        vidx = torch.cat([(vlabels == j).nonzero(as_tuple=True)[0] for j in subtree], dim=0)
        leaves = [torch.zeros(1, 1, 1, requires_grad=True), torch.median(markers, dim=1)[0].clone().requires_grad_(True),
                  o_betas.clone().requires_grad_(True)]
        loss = stages_ref.part_stage_loss(sub, pose, leaves[2], o_betas, root, leaves[1], leaves[0], vidx, oracle_smpl,
                                          cfg, camera=cam, foot_contacts=fc, markers_subset_mean=sub.mean(1))[0]
        first.append(float(loss))
        if k == 0:
            loss.backward()
            grad = torch.cat([p.grad.reshape(-1) for p in leaves] + [torch.zeros(3)]).numpy()
            np.testing.assert_allclose(grad, g["first_grad0"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(first, g["first_losses"], rtol=1e-5)
    trace = {}
    stages_ref.find_best_part_fits(markers, pose, o_betas, root, seg, oracle_smpl, oracle_smpl.smpl.parents, cfg,
                                   trace=trace, foot_contacts=fc, subtree_limit=2, **camera)
    # the non-smooth terms (relu, norm) make the tail of a trajectory sensitive to the last bits of the loss: pin the
    # first 25 evaluations and the converged value, not the evaluation count
    for k in range(2):
        ref = g["losses%d" % k]
        np.testing.assert_allclose(trace["evals"][k][:25], ref[:25], rtol=2e-4)
        np.testing.assert_allclose(trace["evals"][k][-1], ref[-1], rtol=1e-4)


def _dense(g, tag, M, V=6890):
    mat = torch.zeros(M, V)
    nz = torch.from_numpy(g[tag + "_nz"].astype(np.int64))
    mat[nz[:, 0], nz[:, 1]] = torch.from_numpy(g[tag + "_val"]).float()
    return mat


def test_point_triangle_known_answers():
    """oracle/mesh_ref.py against hand-computed closest points in all seven regions of a triangle, and the
    reconstruction property of the barycentric coordinates."""
    from oracle import mesh_ref

    tri = np.array([[[0.0, 0, 0], [1, 0, 0], [0, 1, 0]]])
    cases = [((-1, -1, 0.5), (0, 0, 0)), ((2, -0.5, 0), (1, 0, 0)), ((-0.5, 2, 0), (0, 1, 0)),
             ((0.5, -1, 1), (0.5, 0, 0)), ((-1, 0.5, 1), (0, 0.5, 0)), ((1, 1, 3), (0.5, 0.5, 0)),
             ((0.25, 0.25, -2), (0.25, 0.25, 0))]
    for p, q in cases:
        got, d2 = mesh_ref.closest_on_triangles(np.array(p, float), tri)
        np.testing.assert_allclose(got[0], q, atol=1e-12)
        np.testing.assert_allclose(d2[0], np.sum((np.array(p) - np.array(q)) ** 2), atol=1e-12)
        bc = mesh_ref.points_to_barycentric(tri, got)
        np.testing.assert_allclose(bc.sum(), 1.0, atol=1e-12)
        np.testing.assert_allclose((bc[0][:, None] * tri[0]).sum(0), q, atol=1e-12)
    rng = np.random.default_rng(0)
    V = rng.normal(size=(40, 3))
    F = rng.integers(0, 40, size=(60, 3))
    P = rng.normal(size=(25, 3))
    S, I, C = mesh_ref.signed_distance(P, V, F)
    nearest_vertex = np.min(np.linalg.norm(P[:, None] - V[None, np.unique(F)], axis=-1), axis=1)
    assert np.all(S <= nearest_vertex + 1e-12) and np.all(S >= 0)
    np.testing.assert_allclose(np.linalg.norm(P - C, axis=-1), S, atol=1e-12)


def test_barycentric_placement_matches_reference(golden, oracle_smpl):
    """compute_nearest_points with use_barycentric: the oracle's restatement of the reference's window / granularity /
    scatter logic against the fixture captured from the reference's own function (over the same mesh primitives),
    and the marker stage on the three-corner placement."""
    g = golden("placement_barycentric.npz")
    cfg = _cfg("video_mocap", 25, 25, int(g["num_iters"]))
    cfg["stages"]["compute_locations"].update(use_barycentric=True, use_mean=False)
    markers, pose, betas = _t(g["markers"]), _t(g["in_pose_body"]), _t(g["in_betas"])
    root, trans = _t(g["in_root_orient"]), _t(g["in_trans"])
    M = markers.shape[1]
    for tag, vel in (("full", True), ("marker", False), ("part", False)):
        mat = stages_ref.compute_nearest_points(markers, pose, betas, root, trans, oracle_smpl, _t(g[tag + "_mask"]), cfg,
                                                marker_labels=g["labels"], granularity=tag, use_velocity=vel)
        np.testing.assert_allclose(mat.numpy(), _dense(g, tag, M).numpy(), atol=1e-6)
    o_pose, o_betas = _t(g["o_pose_body"]), _t(g["o_betas"])
    leaves = [x.clone().requires_grad_(True) for x in (pose, betas, root, trans)]
    trace = []
    stages_ref.optim_markers(markers, leaves[0], o_pose, leaves[1], o_betas, leaves[2], leaves[3], _dense(g, "full", M),
                             oracle_smpl, cfg, trace=trace)
    assert len(trace) == len(g["losses"])
    np.testing.assert_allclose(trace, g["losses"], rtol=1e-4)
    np.testing.assert_allclose(leaves[0].detach().numpy(), g["out_pose_body"], atol=1e-4)
    np.testing.assert_allclose(leaves[3].detach().numpy(), g["out_trans"], atol=1e-4)


def _resample_case(g):
    F_img = g["hmr_trans"].shape[0]
    img = SyntheticImgSmpl(
        trans=_t(g["hmr_trans"]), root_orient=_t(g["hmr_root_orient"]), hmr_root_orient=_t(g["hmr_root_orient"]),
        pose_body=_t(g["hmr_pose_body"]), betas=_t(g["hmr_betas"]), foot_contacts=torch.zeros(F_img, 2),
        camera_bbox=torch.zeros(F_img, 3), center=torch.zeros(F_img, 2), scale=torch.zeros(F_img, 1),
        size=torch.zeros(F_img, 2), img_mask=_t(g["img_mask"]), freq=float(g["video_freq"]))
    return img, SyntheticMarkers(g["markers"].copy(), float(g["mocap_freq"]))


def test_frame_rate_resampling_matches_reference(golden, oracle_smpl):
    """Video (15 Hz) -> mocap (30 Hz) resampling of the HMR track: the oracle's slerp against scipy's, and the
    oracle's orchestrator against the fixture captured from the reference's own multimodal_video_mocap."""
    from scipy.spatial.transform import Rotation, Slerp

    rot = Rotation.random(6, random_state=2)
    q = torch.from_numpy(rot.as_quat()).float()  # component order is irrelevant to slerp
    for t in (0.0, 0.25, 0.5, 0.9):
        got = stages_ref.unitquat_slerp(q[:-1], q[1:], torch.tensor([t]))[0].numpy()
        for k in range(5):
            ref = Slerp([0, 1], rot[k:k + 2])([t]).as_quat()[0]
            assert min(np.abs(got[k] - ref).max(), np.abs(got[k] + ref).max()) < 1e-6
    g = golden("e2e_resample.npz")
    cfg = _cfg("hmr_full", g["part_iters"], 25, 25)
    img, markers = _resample_case(g)
    stats = {}
    out = stages_ref.multimodal_video_mocap(img, markers, oracle_smpl, cfg, stats=stats)
    assert out["trans"].shape[0] == g["out_trans"].shape[0] == 9
    np.testing.assert_allclose(out["pose_body"].numpy(), g["out_pose_body"], atol=1e-6)
    np.testing.assert_allclose(out["trans"].numpy(), g["out_trans"], atol=2e-4)
    np.testing.assert_allclose(out["root_orient"].numpy(), g["out_root_orient"], atol=2e-4)
    np.testing.assert_allclose(out["betas"].numpy(), g["out_betas"], atol=2e-4)
    np.testing.assert_array_equal([len(e) for e in stats["part"]["evals"]], g["n_evals"])


def _chamfer_options_cfg(g, tag):
    cfg = _cfg("video_mocap", 25, int(g["num_iters"]), 25)
    if tag == "terms":
        cfg["stages"]["chamfer"]["losses"].update({str(k): float(v) for k, v in zip(g["extra_names"], g["extra_weights"])})
    else:
        cfg["stages"]["chamfer"]["yaw_lock"] = False
    return cfg


@pytest.mark.parametrize("tag", ["terms", "free"])
def test_chamfer_stage_options_match_reference(golden, oracle_smpl, tag):
    """Chamfer stage with part_chamfer + trans_vel + ground ("terms") and with yaw_lock False ("free"): the oracle
    against the fixture captured from the reference's own optim_chamfer (first gradient, the first 40 evaluations of
    the loss trajectory, the converged loss)."""
    g = golden("chamfer_stage_options.npz")
    cfg = _chamfer_options_cfg(g, tag)
    cfg["stages"]["chamfer"]["num_iters"] = 30  # enough for 40 evaluations; the full solves take minutes on the CPU
    leaves = [_t(g[k]).clone().requires_grad_(True) for k in ("o_pose_body", "o_betas", "o_root_orient", "trans0")]
    pose, betas, root, trans = leaves
    trace = []
    stages_ref.optim_chamfer(_t(g["markers"]), pose, _t(g["o_pose_body"]), betas, _t(g["o_betas"]), root, trans,
                             oracle_smpl, cfg, trace=trace, marker_labels=torch.from_numpy(g["labels"]))
    ref = g[tag + "_losses"]
    n = min(len(trace), 30)
    np.testing.assert_allclose(trace[:n], ref[:n], rtol=2e-4)


@pytest.mark.parametrize("case", ["e2e_config0", "e2e_mht_rotation", "e2e_hmr_part"])
def test_oracle_reproduces_the_references_losses_at_its_recorded_points(case, oracle_smpl):
    """The converged-point records of the end-to-end fixtures (oracle/make_golden_e2e.py: for every `LBFGS.step` of the
    reference's own run the parameters it ended on, the loss of the reference's closure there, its constants) against the
    oracle's closures on the CPU: the oracle evaluated AT the recorded final parameters gives the recorded loss, and
    evaluated at the start rebuilt from the records (the way the reference's orchestrator chains its solves) gives the
    recorded first loss.  This pins the oracle at the points where the solves END, not only where they begin, and checks
    the bookkeeping the GPU test (tests/test_gpu_converged.py) relies on."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from converged_records import _solves, _starts

    from uuo_mocap_amd.config import packaged_config

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", case + ".npz"), allow_pickle=False)
    cfg = packaged_config(str(d["yaml"]))
    solves = _solves(d, oracle_smpl)
    starts = _starts(solves)
    for s in (solves if len(solves) <= 12 else solves[::6]):
        at_final, _ = s.oracle(s.final, oracle_smpl, cfg)
        at_start, _ = s.oracle(starts[s.k], oracle_smpl, cfg)
        assert at_final == pytest.approx(s.loss_at_final, rel=1e-6), (case, s.k, s.stage)
        assert at_start == pytest.approx(s.first_loss, rel=1e-6), (case, s.k, s.stage)
