#!/usr/bin/env python3
"""Runs N back-to-back chamfer-stage closure evaluations (F=300, M=50) and one short solve: the target of
`rocprofv3 --kernel-trace --stats` / `--pmc` passes whose summaries are kept under profiles/."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.engine import ChamferProblem, MarkerProblem  # noqa: E402
from uuo_mocap_amd.smpl import SmplInference  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--markers", type=int, default=50)
ap.add_argument("--evals", type=int, default=50)
ap.add_argument("--solve-iters", type=int, default=0)
ap.add_argument("--lib", default=None, help="alternative build of the library (kernel tuning experiments)")
args = ap.parse_args()
if args.lib:
    from uuo_mocap_amd import _lib as _l

    _l.LIB_PATH = os.path.abspath(args.lib)

dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
seq = make_sequence(tables, seed=0, num_frames=args.frames, num_markers=args.markers)
cfg = packaged_config("video_mocap")
markers = torch.from_numpy(seq.markers.get_points()).float().to(dev)
o_pose = seq.img_smpl.pose_body.to(dev)
o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
root = seq.img_smpl.root_orient.to(dev)
trans = torch.median(markers, dim=1)[0]
prob = ChamferProblem(smpl, markers, o_pose, o_betas, root, cfg)
x = prob.pack(trans, torch.zeros(args.frames, 1, 1, device=dev), o_betas, o_pose)
ms_skin = prob.time_closure(x, iters=args.evals, dominant_only=True)
ms_all = prob.time_closure(x, iters=args.evals, dominant_only=False)
print("chamfer closure: %.1f us/eval, k_skin %.1f us/launch (HIP events)" % (1e3 * ms_all, 1e3 * ms_skin))
assign = torch.from_numpy(seq.gt["marker_vids"]).to(dev)
mprob = MarkerProblem(smpl, markers, o_pose, o_betas, assign, cfg)
xm = mprob.pack(o_pose, o_betas, root, trans)
print("marker closure: %.1f us/eval" % (1e3 * mprob.time_closure(xm, iters=args.evals)))
if args.solve_iters > 0:
    st = prob.solve(x, max_iter=args.solve_iters, lr=0.1)
    print("solve:", st)
    print("chamfer closure at the solved point: %.1f us/eval" % (1e3 * prob.time_closure(x, iters=args.evals)))
