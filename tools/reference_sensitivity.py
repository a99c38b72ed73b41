#!/usr/bin/env python3
"""How reproducible is the *reference algorithm itself*?  Runs the CPU oracle's chamfer and marker stages twice
on the golden inputs, the second time with the initial translation perturbed by 1e-6 m, and reports how far the
two converged fits end up from each other.  The L-BFGS trajectories are chaotic in fp32,
so this spread -- not 1e-4 -- is the meaningful yardstick for comparing two converged fits."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stages_ref  # noqa: E402
from oracle.smpl_ref import SmplInferenceRef  # noqa: E402
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402

torch.set_num_threads(int(os.environ.get("THREADS", "8")))
tb = synthetic_smpl(0)
osm = SmplInferenceRef(tb)
T = lambda a: torch.from_numpy(np.asarray(a)).clone()
cfg = packaged_config("video_mocap")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "chamfer_stage.npz"))


def run(eps):
    pose = T(g["hmr_pose_body"]).requires_grad_(True)
    root = T(g["hmr_root_orient"]).requires_grad_(True)
    betas = T(g["o_betas"]).requires_grad_(True)
    trans = (T(g["trans0"]) + eps).requires_grad_(True)
    tr = []
    stages_ref.optim_chamfer(T(g["markers"]), pose, T(g["hmr_pose_body"]), betas, T(g["o_betas"]), root, trans, osm,
                             cfg, trace=tr)
    with torch.no_grad():
        v = osm(stages_ref.normalize_rot(pose), betas.expand(pose.shape[0], 10), stages_ref.normalize_rot(root), trans)[
            "vertices"]
    return tr, trans.detach(), v


tr_a, t_a, v_a = run(0.0)
tr_b, t_b, v_b = run(1e-6)
err = (v_a - v_b).norm(dim=-1)
print("chamfer stage, reference algorithm vs itself (+1e-6 m on the initial translation):")
print("  evals %d vs %d, final loss %.6f vs %.6f" % (len(tr_a), len(tr_b), tr_a[-1], tr_b[-1]))
print("  vertex distance between the two converged fits: mean %.2e m, median %.2e m, max %.2e m" %
      (err.mean(), err.median(), err.max()))
print("  translation difference: median %.2e m, max %.2e m" % ((t_a - t_b).abs().median(), (t_a - t_b).abs().max()))
