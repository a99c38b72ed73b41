"""End-to-end fixtures WITH converged-point records: the reference's OWN ``multimodal_video_mocap`` on the CPU.

TEST INFRASTRUCTURE; runs ONLY in the build container (needs /root/reference).

    python -m oracle.make_golden_e2e --case headline|config0|config1|hmr_part|mht_rotation [--threads N]

The reference's orchestrator (``/root/reference/src/video_mocap/multimodal.py:38-710``) is executed over the restated
third-party primitives (oracle/shim/install.py) with the shipped YAML of the case:

* ``headline``      -- ``config/video_mocap.yaml`` at the METRIC's own workload, 300 frames x 50 markers (BASELINE.json
                        ``metric``): part stage, 4 x (chamfer, placement, marker), selection, final placement + marker;
* ``config0``       -- the same YAML at 30 x 41 (BASELINE ``configs[0]``);
* ``config1``       -- ``config/hmr_full.yaml`` (``configs[1]``) at 300 x 50;
* ``hmr_part``      -- ``config/hmr_part.yaml`` (``configs[2]``), 60 x 10 on one limb, every candidate sub-hierarchy;
* ``mht_rotation``  -- ``config/mht_rotation.yaml`` (reference side of ``configs[4]``), 30 x 41.

Two fp32 L-BFGS runs of one problem separate after a few dozen evaluations ("trajectory chaos"), so
comparing final parameters needs wide bands.  What does NOT depend on the trajectory is the objective itself at a
given point.  For EVERY ``torch.optim.LBFGS.step`` of the run this generator therefore records, beside the counts and
the first / last evaluated losses of the earlier fixtures:

* the parameter tensors the solve ended on (``s<k>_p<j>``, in the reference's parameter-list order);
* the loss of ONE more evaluation of the reference's own closure at those final parameters (``loss_at_final``), and the
  nearest-vertex indices pytorch3d's K=1 search returned inside that evaluation (``s<k>_nn``);
* the constants of the closure as the calling frame held them when ``.step`` was entered (``torch.optim.LBFGS`` is
  subclassed, the caller's locals are read through ``sys._getframe``): prior pose / shape, root orientation before the
  solve, placement indices (marker stage), candidate joints and marker permutation (part stage).

``tests/test_gpu_fullsize.py`` evaluates the HIP closure at the recorded points (loss to 2e-5, indices exactly) and the
oracle closure at the HIP fit's own final points.  Products: ``tests/golden/e2e_<case>.npz`` (data only) and
``profiles/r4_cpu_full_fit_<case>.json`` (wall time of the CPU fit, threads, closures per stage type).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import p3d_ref  # noqa: E402
from oracle.make_golden import seq_inputs  # noqa: E402
from oracle.shim.install import install  # noqa: E402
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence  # noqa: E402

CASES = {
    # name: (yaml, frames, markers, seed, limb_only, stride of the ground-truth vertices kept, output file stem)
    "headline": ("video_mocap", 300, 50, 24, False, 97, "e2e_headline_300x50"),
    "config0": ("video_mocap", 30, 41, 11, False, 13, "e2e_config0"),
    "config1": ("hmr_full", 300, 50, 21, False, 97, "e2e_config1"),
    "hmr_part": ("hmr_part", 60, 10, 22, True, 13, "e2e_hmr_part"),
    "mht_rotation": ("mht_rotation", 30, 41, 23, False, 13, "e2e_mht_rotation"),
    "tiny": ("mht_rotation", 8, 12, 5, False, 13, "_e2e_tiny"),   # self-test of this script; product not committed
}

STAGE_FUNCTIONS = ("optim_chamfer", "optim_markers", "find_best_part_fits", "optim_reprojection", "optim_root")
_last_nn = {"idx": None}
_real_knn1_forward = p3d_ref._knn1_forward


def _recording_knn1_forward(p1, p2, *a, **k):
    dists, idx = _real_knn1_forward(p1, p2, *a, **k)
    _last_nn["idx"] = idx
    return dists, idx


def _np(t):
    return t.detach().cpu().numpy().copy() if isinstance(t, torch.Tensor) else np.asarray(t).copy()


class RecLBFGS(torch.optim.LBFGS):
    """torch.optim.LBFGS that records every ``.step``: losses, time, the caller's constants, the converged point."""

    records = []
    progress_path = None

    def step(self, closure):
        # the stage function that called .step (torch wraps Optimizer.step in a profiling hook: walk up to it)
        fr, caller = sys._getframe(1), {}
        while fr is not None:
            if fr.f_code.co_name in STAGE_FUNCTIONS:
                caller = fr.f_locals
                break
            fr = fr.f_back
        assert caller, "no stage function on the stack of LBFGS.step"
        params = self.param_groups[0]["params"]
        rec = {"losses": [], "n": int(sum(p.numel() for p in params)), "closure_s": 0.0, "const": {}}
        RecLBFGS.records.append(rec)
        c = rec["const"]
        is_param = {id(p) for p in params}
        for name in ("o_pose_body", "o_betas", "root_orient", "pose_body"):
            if isinstance(caller.get(name), torch.Tensor) and id(caller[name]) not in is_param:
                c[name] = _np(caller[name])
        for name in ("initial_angle", "repeat"):
            if name in caller:
                c[name] = np.float64(caller[name])
        if isinstance(caller.get("barycentric_coords_one_hot"), torch.Tensor):   # optim_markers: placement
            one_hot = caller["barycentric_coords_one_hot"]
            assert bool(((one_hot == 0) | (one_hot == 1)).all()) and bool((one_hot.sum(-1) == 1).all())
            c["placement_idx"] = _np(torch.argmax(one_hot, dim=-1)).astype(np.int32)
        if "subtree" in caller:                                                  # find_best_part_fits: candidate
            c["subtree"] = np.asarray(caller["subtree"], dtype=np.int32)
            c["marker_indices"] = np.asarray(_np(caller["indices"]), dtype=np.int32).reshape(-1)
            c["chain"] = np.asarray(caller["chain"], dtype=np.int32)
            c["n_vertex_indices"] = np.int64(caller["vertex_indices"].numel())

        def wrapped():
            t0 = time.perf_counter()
            loss = closure()
            rec["closure_s"] += time.perf_counter() - t0
            rec["losses"].append(float(loss))
            return loss

        t0 = time.perf_counter()
        out = super().step(wrapped)
        rec["wall_s"] = time.perf_counter() - t0
        # the converged point: one more evaluation of the reference's closure, outside the counts
        _last_nn["idx"] = None
        with torch.enable_grad():
            rec["loss_at_final"] = float(closure())
        rec["nn_at_final"] = None if _last_nn["idx"] is None else _np(_last_nn["idx"]).astype(np.int32)
        g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
        rec["grad_at_final_stats"] = np.array([float(g.abs().max()), float(g.abs().sum()), float(g.norm())])
        rec["final_params"] = [_np(p) for p in params]
        if RecLBFGS.progress_path:
            with open(RecLBFGS.progress_path, "a") as fh:
                fh.write("solve %d n=%d evals=%d first=%.6g last=%.6g at_final=%.6g wall=%.0fs\n" % (
                    len(RecLBFGS.records) - 1, rec["n"], len(rec["losses"]), rec["losses"][0], rec["losses"][-1],
                    rec["loss_at_final"], rec["wall_s"]))
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True, choices=sorted(CASES))
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--progress", default=None, help="append one line per finished solve to this file")
    args = ap.parse_args()
    yaml_name, F_, M_, seed, limb_only, stride, stem = CASES[args.case]
    torch.manual_seed(0)
    torch.set_num_threads(args.threads)
    tables = install(synthetic_smpl(0))
    p3d_ref._knn1_forward = _recording_knn1_forward
    import video_mocap.multimodal as ref_mm

    def stage_of(n: int) -> str:
        return {211 * F_ + 10: "chamfer", 219 * F_ + 10: "marker", 3 * F_ + 11: "part"}.get(n, "other")

    real = torch.optim.LBFGS
    torch.optim.LBFGS = RecLBFGS
    RecLBFGS.progress_path = args.progress
    cfg = packaged_config(yaml_name)
    seq = make_sequence(tables, seed=seed, num_frames=F_, num_markers=M_, limb_only=limb_only)
    inp = seq_inputs(seq)
    t0 = time.perf_counter()
    out = ref_mm.multimodal_video_mocap(seq.img_smpl, copy.deepcopy(seq.markers), torch.device("cpu"), cfg,
                                        offset=0, print_options=[], save_stages=True)
    wall = time.perf_counter() - t0
    torch.optim.LBFGS = real
    p3d_ref._knn1_forward = _real_knn1_forward
    recs = RecLBFGS.records
    stages = np.array([stage_of(r["n"]) for r in recs])
    per_solve = {}
    hmr_pose = inp["hmr_pose_body"]
    for k, r in enumerate(recs):
        for j, p in enumerate(r["final_params"]):
            per_solve["s%d_p%d" % (k, j)] = p
        if r["nn_at_final"] is not None:
            nn = r["nn_at_final"]
            per_solve["s%d_nn" % k] = nn.astype(np.int16) if nn.max() < 32768 else nn
        for name, v in r["const"].items():
            if name in ("o_pose_body", "pose_body") and v.shape == hmr_pose.shape and np.array_equal(v, hmr_pose):
                per_solve["s%d_%s_is_hmr" % (k, name)] = np.bool_(True)   # the HMR prior itself: not stored twice
            else:
                per_solve["s%d_%s" % (k, name)] = v
    extra = {"gt_verts_stride13": seq.gt["verts"][:, ::13].astype(np.float32)} if args.case == "config0" else {}
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", stem + ".npz"), **inp, seed=seed, yaml=yaml_name,
        limb_only=limb_only, n_solves=len(recs), solve_stage=stages,
        n_evals=np.array([len(r["losses"]) for r in recs]),
        first_losses=np.array([r["losses"][0] for r in recs]),
        final_losses=np.array([r["losses"][-1] for r in recs]),
        loss_at_final=np.array([r["loss_at_final"] for r in recs]),
        grad_at_final_stats=np.array([r["grad_at_final_stats"] for r in recs]),
        out_trans=out["trans"].numpy(), out_root_orient=out["root_orient"].numpy(),
        out_pose_body=out["pose_body"].numpy(), out_betas=out["betas"].numpy(),
        out_markers_labels=np.asarray(out["markers_labels"]), out_chain=out["chain"],
        stage_keys=np.array(sorted(out["stages"].keys())), gt_stride=stride,
        gt_verts_strided=seq.gt["verts"][:, ::stride].astype(np.float32), **extra, **per_solve,
    )
    per_stage = {}
    for r, s in zip(recs, stages):
        d = per_stage.setdefault(str(s), {"solves": 0, "evals": 0, "closure_s": 0.0, "wall_s": 0.0})
        d["solves"] += 1
        d["evals"] += len(r["losses"])
        d["closure_s"] += r["closure_s"]
        d["wall_s"] += r["wall_s"]
    for d in per_stage.values():
        d["seconds_per_eval"] = d["closure_s"] / max(1, d["evals"])
    prof = {
        "what": "reference's own multimodal_video_mocap (%s.yaml as shipped) on CPU over the restated smplx/pytorch3d "
                "primitives; synthetic sequence seed %d%s" % (yaml_name, seed, ", markers on one limb" if limb_only else ""),
        "frames": F_, "markers": M_, "torch_threads": args.threads, "nproc": os.cpu_count(),
        "wall_s": wall, "frames_per_s": F_ / wall,
        "solver_wall_s": float(sum(r["wall_s"] for r in recs)),
        "closure_wall_s": float(sum(r["closure_s"] for r in recs)),
        "per_stage": per_stage,
        "torch": torch.__version__,
    }
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r4_cpu_full_fit_%s.json" % args.case), "w") as fh:
        json.dump(prof, fh, indent=1, sort_keys=True)
    print(json.dumps(prof, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
