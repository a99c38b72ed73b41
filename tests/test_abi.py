"""CPU-side checks of the C-ABI library and the host mirror (no kernels are launched)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from uuo_mocap_amd import _lib


def test_library_loads_and_exports_every_header_symbol():
    lib = _lib.load()
    names = _lib.header_symbols()
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), name
    assert lib.uuo_abi_version() == 3
    assert set(_lib._SIGNATURES) == set(names), "ctypes signature table and include/uuo_hip.h disagree"


def test_product_library_exports_the_header_and_nothing_else():
    """The shipped library's dynamic symbol table is include/uuo_hip.h: no uuo_debug_* hook, no self-test entry, no C++
    launcher; and it reads no environment variable (the ablation knobs live in libuuo_hip_debug.so only)."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if " T " in line)
    assert exported == _lib.header_symbols(), set(exported) ^ set(_lib.header_symbols())
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], check=True, capture_output=True,
                               text=True).stdout
    assert "getenv" not in undefined
    dbg = _lib.load_debug()
    for name in _lib._DEBUG_SIGNATURES:
        assert hasattr(dbg, name), name
    for name in _lib.header_symbols():
        assert hasattr(dbg, name), name


def test_marker_loss_known_answer():
    """Product MarkerLoss (losses.py; reference losses/losses.py:43-51) against SURVEY.md K-B, on the CPU: it is plain
    torch arithmetic on the caller's tensors."""
    from uuo_mocap_amd.losses import MarkerLoss
    from uuo_mocap_amd.optimization import get_marker_mask

    m = torch.tensor([[[0, 0, 0], [1, 2, 2]], [[0, 3, 4], [0, 0, 0]]]).float()
    vm = torch.tensor([[[1, 0, 0], [1, 2, 2.0095]], [[0, 0, 0], [5, 5, 5]]]).float()
    out = MarkerLoss(m, vm, get_marker_mask(m), 0.0095)
    assert out.shape == (2, 2)
    assert out[0, 0].item() == 0.0 and out[1, 1].item() == 0.0
    assert out[1, 0].item() == 24.90509033203125
    assert out.mean().item() == 6.2262725830078125
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.npz"))
    np.testing.assert_array_equal(out.numpy(), g["kb"])


def test_problem_sizes_follow_the_reference_packing():
    lib = _lib.load()
    for stage, per_frame, const in ((_lib.UUO_STAGE_CHAMFER, 211, 10), (_lib.UUO_STAGE_MARKER, 219, 10),
                                    (_lib.UUO_STAGE_PART, 3, 11)):
        for F in (1, 30, 300):
            p = _lib.UuoProblem()
            p.stage, p.F, p.M = stage, F, 50
            assert lib.uuo_problem_num_params(ctypes.byref(p)) == per_frame * F + const
    p = _lib.UuoProblem()
    p.stage, p.F = 7, 10
    assert lib.uuo_problem_num_params(ctypes.byref(p)) < 0


def test_null_arguments_are_rejected_with_a_message():
    lib = _lib.load()
    rc = lib.uuo_model_create(None, None, None, None, None, None, None, 6890, ctypes.byref(ctypes.c_void_p()))
    assert rc != 0
    assert b"null" in lib.uuo_last_error()


def test_no_cpu_fallback():
    from uuo_mocap_amd.smpl import SmplInference

    with pytest.raises(RuntimeError, match="GPU only"):
        SmplInference(torch.device("cpu"))


def test_product_does_not_import_the_oracle():
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "uuo_mocap_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn


def test_pad_and_masks():
    from uuo_mocap_amd.multimodal import pad
    from uuo_mocap_amd.optimization import get_marker_mask

    x = torch.arange(12.0).reshape(4, 3)
    assert pad(x, 0) is x
    assert torch.equal(pad(x, 2)[:2], x[[0, 0]]) and pad(x, 2).shape[0] == 6
    assert torch.equal(pad(x, -1)[-1], x[-1]) and pad(x, -1).shape[0] == 5
    m = torch.tensor([[[0.0, 0, 0], [1, -1, 0]], [[0, 0, 1e-9], [0, 0, 0]]])
    assert get_marker_mask(m).tolist() == [[False, True], [True, False]]


def test_transforms_match_oracle():
    from oracle import p3d_ref
    from uuo_mocap_amd import transforms as T

    torch.manual_seed(0)
    m = torch.randn(7, 3, 3)
    torch.testing.assert_close(T.normalize_rot(m), p3d_ref.rotation_6d_to_matrix(p3d_ref.matrix_to_rotation_6d(m)))
    a = torch.tensor([[[0.0]], [[0.7]], [[-2.5]], [[3e-7]]])
    z = torch.zeros(4, 1, 3)
    z[..., 2:] = a
    torch.testing.assert_close(T.compute_root_orient_z(a), p3d_ref.axis_angle_to_matrix(z))


def test_subtrees_match_oracle(tables):
    """(segment_rigid's rigidity matrix is a GPU kernel since round 3: tests/test_gpu_parity.py checks it against numpy.)"""
    from oracle import stages_ref
    from uuo_mocap_amd import markers_utils as MU

    for k in (2, 5, 12, 24):
        assert MU.get_sub_hierachies(tables.parents, k) == stages_ref.get_sub_hierarchies(tables.parents, k)


def test_matrix_to_axis_angle_round_trip():
    """poses[F,72] of the runner's output: axis-angle through the quaternion route (pytorch3d semantics)."""
    from oracle import p3d_ref
    from uuo_mocap_amd import transforms as T

    g = torch.Generator().manual_seed(4)
    aa = torch.randn(200, 3, generator=g)
    aa = aa / aa.norm(dim=-1, keepdim=True) * (torch.rand(200, 1, generator=g) * 3.0)  # angles in [0, 3) rad
    aa[0] = 0.0
    aa[1] = torch.tensor([1e-8, 0.0, 0.0])
    R = p3d_ref.axis_angle_to_matrix(aa)
    back = T.matrix_to_axis_angle(R)
    # the quaternion is not sign-standardised (pytorch3d 0.7.4), so the vector may describe the same rotation with
    # an angle in (pi, 2 pi): compare the rotations
    torch.testing.assert_close(p3d_ref.axis_angle_to_matrix(back), R, atol=2e-6, rtol=0)
    torch.testing.assert_close(T.matrix_to_quaternion(R), p3d_ref.matrix_to_quaternion(R))


def test_runner_conventions(tmp_path, tables):
    """Counterpart of the reference's test/test.py: argument names, directory layout, skip-if-exists, output keys
    and shapes, per-stage files (the fit itself is replaced by a stub: no GPU here)."""
    from uuo_mocap_amd import runner
    from uuo_mocap_amd.synthetic import make_sequence

    root = tmp_path / "data"
    seq = make_sequence(tables, seed=2, num_frames=6, num_markers=5)
    for subject, name in (("s01", "walk"), ("s01", "run"), ("s02", "walk")):
        d = root / "cmu_kitchen_pilot_rb" / "mocap" / subject
        d.mkdir(parents=True, exist_ok=True)
        runner.write_sequence_npz(str(d / (name + ".npz")), seq.markers.get_points(), 30.0, seq.img_smpl.pose_body,
                                  seq.img_smpl.root_orient, seq.img_smpl.betas)
    (root / "cmu_kitchen_pilot_rb" / "mocap" / "s02" / "jump.c3d").write_bytes(b"")
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("name: unit\n")
    args = runner.build_parser().parse_args(["--config", str(cfg), "--dataset", "cmu_kitchen_pilot_rb", "--input_dir",
                                             str(root), "--subjects", "s01"])
    assert args.print_options == ["loss", "progress"] and args.num_files is None and args.sequences is None
    calls = []

    def fake_fit(img_smpl, markers):
        F = markers.get_points().shape[0]
        calls.append(F)
        eye = torch.eye(3).expand(F, 23, 3, 3).clone()
        out = {"betas": torch.zeros(F, 10), "trans": torch.zeros(F, 3), "root_orient": torch.eye(3).expand(F, 1, 3, 3),
               "pose_body": eye, "mocap_frame_rate": markers.get_frequency(), "mocap_markers": markers,
               "stages": {"chamfer": {"trans": np.zeros((F, 3), np.float32), "betas": np.zeros(10, np.float32),
                                      "root_orient": np.tile(np.eye(3, dtype=np.float32), (F, 1, 1, 1)),
                                      "pose_body": np.tile(np.eye(3, dtype=np.float32), (F, 23, 1, 1))}}}
        return out

    assert runner.run(args, fit_fn=fake_fit) == 2 and len(calls) == 2
    res = root / "cmu_kitchen_pilot_rb" / "results" / "unit" / "s01"
    out = np.load(res / "run_stageii.npz")
    assert set(out.files) == {"betas", "trans", "poses", "mocap_frame_rate", "mocap_markers", "gender"}
    assert out["betas"].shape == (10,) and out["trans"].shape == (6, 3) and out["poses"].shape == (6, 72)
    assert out["mocap_markers"].shape == (6, 5, 3) and str(out["gender"]) == "neutral"
    assert (res / "run_stageii.chamfer.npz").exists()
    assert runner.run(args, fit_fn=fake_fit) == 0 and len(calls) == 2  # skip-if-exists
    args2 = runner.build_parser().parse_args(["--config", str(cfg), "--dataset", "cmu_kitchen_pilot_rb", "--input_dir",
                                              str(root), "--subjects", "s02", "--sequences", "jump"])
    # a .c3d sequence whose 4D-Humans result is missing is skipped, as the reference does (test/test.py:91-93)
    assert runner.run(args2, fit_fn=fake_fit) == 0 and len(calls) == 2


def test_evaluation_metrics_match_reference():
    """uuo_mocap_amd.metrics against the fixture captured from the reference's own evaluation/metrics.py (the pure
    tensor metrics run anywhere; the marker-to-surface distance needs the GPU and is in tests/test_gpu_parity.py)."""
    import numpy as np
    import torch
    from uuo_mocap_amd import metrics as m

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))
    pred, gt = torch.from_numpy(g["pred"]), torch.from_numpy(g["gt"])
    ids, freq = g["joint_ids"].tolist(), float(g["freq"])
    got = {
        "mpjpe": m.compute_MPJPE(pred, gt), "mpjpe_joints": m.compute_MPJPE_joints(pred, gt, ids),
        "mpjve": m.compute_MPJVE(pred, gt, freq), "mpjve_joints": m.compute_MPJVE_joints(pred, gt, freq, ids),
        "pa_mpjpe": m.compute_PA_MPJPE(pred, gt), "pa_mpjpe_joints": m.compute_PA_MPJPE_joints(pred, gt, ids),
        "pa_mpjve": m.compute_PA_MPJVE(pred, gt, freq),
        "pa_mpjve_joints": m.compute_PA_MPJVE_joints(pred, gt, freq, ids),
        "v2v": m.compute_V2V(torch.from_numpy(g["pred_verts"]), torch.from_numpy(g["gt_verts"])),
        "aligned": m.compute_similarity_transform(pred, gt),
    }
    for k, v in got.items():
        np.testing.assert_allclose(v.numpy(), g["out_" + k], rtol=2e-5, atol=2e-6, err_msg=k)


def test_dropin_package_exposes_the_reference_module_paths():
    """uuo_mocap_amd/dropin on PYTHONPATH presents the reference's module paths -- import only, no GPU."""
    import importlib
    import sys

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uuo_mocap_amd", "dropin")
    saved = {k: v for k, v in sys.modules.items() if k == "video_mocap" or k.startswith("video_mocap.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, root)
    try:
        expect = {
            "video_mocap.utils.smpl": ["SmplInference"],
            "video_mocap.losses.chamfer_distance": ["weighted_chamfer_distance"],
            "video_mocap.losses.losses": ["MarkerLoss"],
            "video_mocap.optimization": ["optim_chamfer", "optim_markers", "compute_nearest_points",
                                         "compute_marker_labels_from_coords", "compute_root_orient_z",
                                         "chamfer_distance_by_part", "get_marker_mask", "weighted_mse_loss"],
            "video_mocap.markers.markers_utils": ["find_best_part_fits", "segment_rigid", "filter_rigid"],
            "video_mocap.multimodal": ["multimodal_video_mocap", "pad"],
            "video_mocap.utils.hmr_utils": ["optim_reprojection", "perspective_projection", "get_3d_parameters"],
            "video_mocap.evaluation.metrics": ["compute_marker_to_surface_distance", "compute_PA_MPJPE", "compute_V2V"],
        }
        for mod, names in expect.items():
            m = importlib.import_module(mod)
            for n in names:
                assert hasattr(m, n), (mod, n)
    finally:
        sys.path.remove(root)
        for k in [k for k in sys.modules if k == "video_mocap" or k.startswith("video_mocap.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def _staging(ops, region_cap=4096):
    """Runs a script against the batch's pinned-blob bookkeeping (csrc/uuo_common.h UuoStaging) through the debug library's
    host-only hook; returns per op (flag, offset, pending bits, used bytes of the op's region)."""
    dbg = _lib.load_debug()
    arr = np.ascontiguousarray(np.asarray(ops, dtype=np.int64).reshape(-1, 3))
    out = np.zeros((arr.shape[0], 4), dtype=np.int64)
    rc = dbg.uuo_debug_staging_script(arr.ctypes.data, arr.shape[0], region_cap, out.ctypes.data)
    assert rc == 0, dbg.uuo_last_error()
    return [tuple(int(v) for v in row) for row in out]


FLUSH, APPEND, REPORT, SYNCED = 0, 1, 2, 3


def test_batch_staging_state_machine():
    """Host logic behind the GPU memory fault of round 2 (gpurun_out/r2_t9.log: the score launch re-used the pinned
    argument blob while the previous flush's host-to-device copy was still pending), tested without a GPU: a flush that
    follows a flush with no report in between must synchronise first; a report clears that; the two regions (stepping
    groups / streams) are independent; structs appended behind a flush never overwrite it and never need a wait."""
    # flush -> flush without a wait: the second one has to synchronise; after a report it does not
    r = _staging([(FLUSH, 0, 1000), (FLUSH, 0, 500), (REPORT, 0, 0), (FLUSH, 0, 700)])
    assert [x[0] for x in r] == [0, 1, 0, 0]
    assert r[1][2] == 1 and r[2][2] == 0 and r[3][2] == 1 and r[3][3] == 700
    # an empty flush (a round in which nothing was recorded) neither waits nor changes anything
    r = _staging([(FLUSH, 0, 1000), (FLUSH, 0, 0), (FLUSH, 0, 10)])
    assert [x[0] for x in r] == [0, 0, 1] and r[1][3] == 1000
    # regions are independent: region 1's flushes neither see nor clear region 0's pending copy
    r = _staging([(FLUSH, 0, 100), (FLUSH, 1, 100), (REPORT, 1, 0), (FLUSH, 1, 100), (FLUSH, 0, 100)])
    assert [x[0] for x in r] == [0, 0, 0, 0, 1]
    assert r[1][2] == 3 and r[2][2] == 1
    # flush then scores: the score structs go behind the flush's (256-byte aligned), also behind an earlier append, need no
    # wait, and keep the region pending so that the NEXT flush waits for them too
    r = _staging([(FLUSH, 0, 1000), (APPEND, 0, 300), (APPEND, 0, 100), (FLUSH, 0, 50)])
    assert r[1][:2] == (1, 1024) and r[2][:2] == (1, 1536) and r[3][0] == 1
    # an append on a region whose flush has already reported still lands behind it (its device copy may be in use by kernels)
    r = _staging([(FLUSH, 0, 1000), (REPORT, 0, 0), (APPEND, 0, 300), (FLUSH, 0, 10)])
    assert r[2][:3] == (1, 1024, 1) and r[3][0] == 1
    # overflow is refused, state unchanged
    r = _staging([(FLUSH, 0, 4000), (APPEND, 0, 200)], region_cap=4096)
    assert r[1][0] == 0 and r[1][3] == 4000
    # error exit / end of a solve: everything was synchronised
    r = _staging([(FLUSH, 0, 100), (FLUSH, 1, 100), (SYNCED, 0, 0), (FLUSH, 0, 100), (FLUSH, 1, 100)])
    assert [x[0] for x in r] == [0, 0, 0, 0, 0]
    # ... and nothing is held any more: score calls that follow each other with nothing to flush in between must
    # not keep appending behind the previous call's structs until the region overflows
    r = _staging([(FLUSH, 0, 1000), (APPEND, 0, 300), (SYNCED, 0, 0), (FLUSH, 0, 0), (APPEND, 0, 300), (SYNCED, 0, 0),
                  (FLUSH, 0, 0), (APPEND, 0, 300)], region_cap=2048)
    assert r[2][3] == 0 and r[4][:2] == (1, 0) and r[7][:2] == (1, 0)


def test_model_with_dense_skin_weights_is_refused():
    """SMPL has at most four non-zero skinning weights per vertex and every kernel is built on that; a model that has more
    is refused at creation with a message instead of reaching a slow generic path (round 2 shipped one that spilled)."""
    lib = _lib.load()
    V = 32
    rng = np.random.default_rng(0)
    W = np.zeros((V, 24), np.float32)
    W[:, :5] = 0.2  # five non-zero weights
    arrs = [rng.standard_normal((V, 3)).astype(np.float32), np.zeros((V, 3, 10), np.float32),
            np.zeros((207, V * 3), np.float32), np.full((24, V), 1.0 / V, np.float32), W,
            np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int64),
            np.zeros(21, np.int64)]
    handle = ctypes.c_void_p()
    rc = lib.uuo_model_create(*[a.ctypes.data for a in arrs], V, ctypes.byref(handle))
    assert rc == -22 and b"non-zero skinning weights" in lib.uuo_last_error()


def test_compact_solver_packing_index_map():
    """The L-BFGS solver of the chamfer / marker stages runs on a compact packing that leaves out the third row of every
    optimised rotation (csrc/closure.hip stage_layout: its gradient is identically zero there, so the entry never moves).
    The map from solver coordinates to the reference's parameter packing, built on the host and evaluated inside the
    kernels that write trial points, must be exactly: every coordinate of the reference's packing except entries 6, 7, 8
    of each row-major 3x3 rotation, in order."""
    dbg = _lib.load_debug()
    for F in (1, 7, 300, 3000):
        # chamfer: [trans 3F | z F | betas 10 | pose 23 x 9 x F]; marker: [pose | betas 10 | root 9F | trans 3F]
        keep9 = np.arange(9) < 6
        cham = np.concatenate([np.ones(4 * F + 10, bool), np.tile(keep9, 23 * F)])
        mark = np.concatenate([np.tile(keep9, 23 * F), np.ones(10, bool), np.tile(keep9, F), np.ones(3 * F, bool)])
        for stage, keep, n_full in ((_lib.UUO_STAGE_CHAMFER, cham, 211 * F + 10), (_lib.UUO_STAGE_MARKER, mark, 219 * F + 10)):
            assert keep.size == n_full
            out = np.full(n_full, -1, np.int32)
            n = dbg.uuo_debug_index_map(stage, F, 1, out.ctypes.data)
            assert n == int(keep.sum()) == (142 * F + 10 if stage == _lib.UUO_STAGE_CHAMFER else 147 * F + 10)
            assert np.array_equal(out[:n], np.nonzero(keep)[0])
            n = dbg.uuo_debug_index_map(stage, F, 0, out.ctypes.data)
            assert n == n_full and np.array_equal(out, np.arange(n_full))
    out = np.full(3 * 5 + 11, -1, np.int32)
    assert dbg.uuo_debug_index_map(_lib.UUO_STAGE_PART, 5, 1, out.ctypes.data) == 26 and np.array_equal(out, np.arange(26))


def test_execution_options_merge():
    """defaults < config section < argument; the same key may appear at every level; unknown keys are refused."""
    from uuo_mocap_amd.markers_utils import EXECUTION_DEFAULTS, merge_execution

    assert merge_execution({}) == EXECUTION_DEFAULTS
    cfg = {"execution": {"hypothesis_lockstep": True, "subtree_batch": 64}}
    out = merge_execution(cfg, {"hypothesis_lockstep": False, "subtree_threads": 2})
    assert out["hypothesis_lockstep"] is False and out["subtree_batch"] == 64 and out["subtree_threads"] == 2
    assert merge_execution(cfg)["hypothesis_lockstep"] is True
    with pytest.raises(KeyError):
        merge_execution({"execution": {"no_such_knob": 1}})


def test_reprojection_entry_points_refuse_bad_arguments():
    """uuo_reprojection_* (the fused 2D-prior closure): argument checks that need no GPU."""
    import ctypes
    from uuo_mocap_amd import _lib

    lib = _lib.load()
    p = _lib.UuoReprojectionProblem()
    p.F, p.M, p.V, p.J = 8, 12, 6890, 45
    assert lib.uuo_reprojection_num_params(ctypes.byref(p)) == 3 * 8 + 14
    h = ctypes.c_void_p()
    assert lib.uuo_reprojection_create(ctypes.byref(p), ctypes.byref(h)) == -22  # null device pointers
    assert b"null device pointer" in lib.uuo_last_error()
    p.J = 65
    assert lib.uuo_reprojection_create(ctypes.byref(p), ctypes.byref(h)) == -22
    assert b"joints" in lib.uuo_last_error()
    assert lib.uuo_reprojection_create(None, ctypes.byref(h)) == -22
    assert lib.uuo_reprojection_destroy(None) == 0
    assert lib.uuo_reprojection_eval(None, None, None, None, None, None, None) == -22
    assert lib.uuo_reprojection_solve(None, None, None, None, None, None, None, None, None) == -22


def test_wait_policy_argument_checks():
    """uuo_set_wait_policy: host-only state, no GPU needed."""
    from uuo_mocap_amd import _lib

    lib = _lib.load()
    assert lib.uuo_set_wait_policy(250, 20000) == 0
    assert lib.uuo_set_wait_policy(0, -1) == -22 and b"sleep_ns" in lib.uuo_last_error()
    assert lib.uuo_set_wait_policy(0, 20000000) == -22
    assert lib.uuo_set_wait_policy(-1, 0) == 0  # back to spinning


def test_header_is_plain_c(tmp_path):
    """include/uuo_hip.h is the C ABI: it must compile on its own as C99 and as C++ (no torch / HIP types, nothing missing)."""
    import subprocess

    from uuo_mocap_amd import _lib

    for lang, std, cc in (("c", "-std=c99", "gcc"), ("c++", "-std=c++11", "g++")):
        subprocess.check_call([cc, "-x", lang, std, "-Wall", "-Werror", "-fsyntax-only", _lib.HEADER_PATH])


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The Python mirror binds the C ABI with hand-written ctypes structures: their sizes and the offsets of their last
    members must be what a C compiler makes of include/uuo_hip.h (uuo_problem_t grew two members in round 4)."""
    import ctypes
    import subprocess

    from uuo_mocap_amd import _lib
    from uuo_mocap_amd._lib import UuoLbfgsOptions, UuoLbfgsStats, UuoProblem, UuoReprojectionProblem

    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void) {\n'
                   '  printf("%%zu %%zu %%zu\\n", sizeof(uuo_problem_t), offsetof(uuo_problem_t, w_soft), offsetof(uuo_problem_t, soft_tau));\n'
                   '  printf("%%zu %%zu\\n", sizeof(uuo_lbfgs_options_t), offsetof(uuo_lbfgs_options_t, verbose));\n'
                   '  printf("%%zu %%zu\\n", sizeof(uuo_lbfgs_stats_t), offsetof(uuo_lbfgs_stats_t, device_ms));\n'
                   '  printf("%%zu %%zu\\n", sizeof(uuo_reprojection_problem_t), offsetof(uuo_reprojection_problem_t, w_chamfer));\n'
                   '  return 0;\n}\n' % _lib.HEADER_PATH)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])
    rows = [[int(x) for x in line.split()] for line in subprocess.check_output([str(exe)]).decode().splitlines()]
    assert rows[0] == [ctypes.sizeof(UuoProblem), UuoProblem.w_soft.offset, UuoProblem.soft_tau.offset]
    assert rows[1] == [ctypes.sizeof(UuoLbfgsOptions), UuoLbfgsOptions.verbose.offset]
    assert rows[2] == [ctypes.sizeof(UuoLbfgsStats), UuoLbfgsStats.device_ms.offset]
    assert rows[3] == [ctypes.sizeof(UuoReprojectionProblem), UuoReprojectionProblem.w_chamfer.offset]


def test_soft_assignment_options_are_validated_on_the_host(tables):
    """EXTENSION plumbing that needs no GPU: the packaged soft configurations resolve their parents, the part stage's fused
    soft closure is limited to 16 markers per frame by construction, and the new execution switches are known options."""
    from uuo_mocap_amd.config import packaged_config
    from uuo_mocap_amd.engine import PART_SOFT_MAX_MARKERS
    from uuo_mocap_amd.markers_utils import EXECUTION_DEFAULTS, merge_execution

    part = packaged_config("hmr_part_soft")["stages"]["part"]
    assert part["losses"]["soft_chamfer"] == 10.0 and part["losses"]["chamfer"] == 0.0 and part["soft_tau"] == 2.5e-4
    cham = packaged_config("video_mocap_soft")["stages"]["chamfer"]
    assert cham["losses"]["soft_chamfer"] == 10.0 and cham["losses"]["full_chamfer"] == 0.0 and cham["soft_tau"] == 1e-3
    assert cham["losses"]["reg_pose_body"] == 1.0 and cham["num_iters"] == 10000      # inherited from video_mocap.yaml
    assert PART_SOFT_MAX_MARKERS == 16
    assert EXECUTION_DEFAULTS["part_soft_fused"] is True and EXECUTION_DEFAULTS["chamfer_soft_fused"] is True
    assert merge_execution({"execution": {"chamfer_soft_fused": False}})["chamfer_soft_fused"] is False
