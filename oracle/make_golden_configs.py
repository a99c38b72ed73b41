"""End-to-end fixtures of the remaining BASELINE configurations: the reference's OWN ``multimodal_video_mocap`` at sizes the
CPU can finish.

TEST INFRASTRUCTURE; runs ONLY in the build container (needs /root/reference).

    python -m oracle.make_golden_configs --case config1|hmr_part|mht_rotation [--threads N]

Same recipe as ``make_golden_config0.py``: the reference's orchestrator (``/root/reference/src/video_mocap/multimodal.py:38-710``)
is executed on the CPU over the restated third-party primitives (oracle/shim/install.py) with the shipped YAML of the case:

* ``config1``       -- ``config/hmr_full.yaml`` (BASELINE ``configs[1]``) at the BASELINE size, 300 frames x 50 markers:
                        full-skeleton part stage + the 4-yaw selection (chamfer / marker stages are off there);
* ``hmr_part``      -- ``config/hmr_part.yaml`` (BASELINE ``configs[2]``), 60 frames x 10 markers on one limb, EVERY candidate
                        sub-hierarchy solved (the candidate count is data-dependent: 100-200 L-BFGS solves);
* ``mht_rotation``  -- ``config/mht_rotation.yaml`` (the reference side of BASELINE ``configs[4]``), 30 frames x 41 markers,
                        one yaw hypothesis through all stages.

Each writes ``tests/golden/e2e_<case>.npz`` (inputs + converged outputs + per-solve evaluation counts, first / final losses:
data only) and ``profiles/r3_cpu_full_fit_<case>.json`` (wall time of the CPU fit, thread count, closures per stage type).
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

from oracle.make_golden import seq_inputs  # noqa: E402
from oracle.shim.install import install  # noqa: E402
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence  # noqa: E402

CASES = {
    # name: (yaml, frames, markers, seed, limb_only, stride of the ground-truth vertices kept)
    "config1": ("hmr_full", 300, 50, 21, False, 97),
    "hmr_part": ("hmr_part", 60, 10, 22, True, 13),
    "mht_rotation": ("mht_rotation", 30, 41, 23, False, 13),
}


class TimedLBFGS(torch.optim.LBFGS):
    """torch.optim.LBFGS that records losses, parameter count and wall time of every ``.step``."""

    records = []

    def step(self, closure):
        params = self.param_groups[0]["params"]
        rec = {"losses": [], "n": int(sum(p.numel() for p in params)), "closure_s": 0.0}
        TimedLBFGS.records.append(rec)

        def wrapped():
            t0 = time.perf_counter()
            loss = closure()
            rec["closure_s"] += time.perf_counter() - t0
            rec["losses"].append(float(loss))
            return loss

        t0 = time.perf_counter()
        out = super().step(wrapped)
        rec["wall_s"] = time.perf_counter() - t0
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True, choices=sorted(CASES))
    ap.add_argument("--threads", type=int, default=1)
    args = ap.parse_args()
    yaml_name, F_, M_, seed, limb_only, stride = CASES[args.case]
    torch.manual_seed(0)
    torch.set_num_threads(args.threads)
    tables = install(synthetic_smpl(0))
    import video_mocap.multimodal as ref_mm

    def stage_of(n: int) -> str:
        return {211 * F_ + 10: "chamfer", 219 * F_ + 10: "marker", 3 * F_ + 11: "part"}.get(n, "other")

    real = torch.optim.LBFGS
    torch.optim.LBFGS = TimedLBFGS
    cfg = packaged_config(yaml_name)
    seq = make_sequence(tables, seed=seed, num_frames=F_, num_markers=M_, limb_only=limb_only)
    inp = seq_inputs(seq)
    t0 = time.perf_counter()
    out = ref_mm.multimodal_video_mocap(seq.img_smpl, copy.deepcopy(seq.markers), torch.device("cpu"), cfg,
                                        offset=0, print_options=[], save_stages=True)
    wall = time.perf_counter() - t0
    torch.optim.LBFGS = real
    recs = TimedLBFGS.records
    stages = np.array([stage_of(r["n"]) for r in recs])
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "e2e_%s.npz" % args.case), **inp, seed=seed, yaml=yaml_name,
        limb_only=limb_only, n_solves=len(recs), solve_stage=stages,
        n_evals=np.array([len(r["losses"]) for r in recs]),
        first_losses=np.array([r["losses"][0] for r in recs]),
        final_losses=np.array([r["losses"][-1] for r in recs]),
        out_trans=out["trans"].numpy(), out_root_orient=out["root_orient"].numpy(),
        out_pose_body=out["pose_body"].numpy(), out_betas=out["betas"].numpy(),
        out_markers_labels=np.asarray(out["markers_labels"]), out_chain=out["chain"],
        stage_keys=np.array(sorted(out["stages"].keys())), gt_stride=stride,
        gt_verts_strided=seq.gt["verts"][:, ::stride].astype(np.float32),
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
    with open(os.path.join(ROOT, "profiles", "r3_cpu_full_fit_%s.json" % args.case), "w") as fh:
        json.dump(prof, fh, indent=1, sort_keys=True)
    print(json.dumps(prof, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
