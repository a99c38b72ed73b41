#!/usr/bin/env python3
"""Measured on recorded L-BFGS trajectories BEFORE building anything: how much of the chamfer stage's
skinning could be skipped without changing a single assignment?

Records every evaluated parameter vector of a chamfer-stage solve (F x M, one yaw hypothesis), re-skins them and replays two
policies on the exact vertices:

 (a) UNIT CULL (the VERDICT's proposal): keep the 16-vertex unit u in frame f iff its box, grown by the frame's vertex motion
     since the last evaluation that skinned everything, can hold a vertex closer than the marker's upper bound.  Reports the
     surviving fraction of (frame, unit) pairs and of (16-frame tile, unit) tasks -- the granule k_skin2 works on.
 (b) TRACKED CANDIDATES: at an anchor evaluation every marker column m gets a list L_m of K vertices (the most frequent
     per-frame winners, filled up with the nearest on average), and r_out[f, m] = the distance to the nearest vertex NOT in
     L_m.  A later evaluation may skin only the lists (M units of K vertices instead of 431 units of 16) if for every
     (f, m): d_in[f, m] < r_out[f, m] - delta_f, delta_f = motion of the frame's vertices since the anchor (any vertex
     outside the list is then provably farther than the best one inside).  Otherwise it is evaluated densely and becomes
     the new anchor.  Reports the fraction of evaluations that pass, with delta_f exact and inflated (a computable bound
     is looser than the exact motion).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.engine import ChamferProblem  # noqa: E402
from uuo_mocap_amd.smpl import SmplInference  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence  # noqa: E402
from uuo_mocap_amd.transforms import compute_root_orient_z, normalize_rot  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--markers", type=int, default=50)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--yaw", type=float, default=0.0, help="yaw hypothesis (radians) applied to the HMR root")
ap.add_argument("--k", type=int, default=16, help="vertices per tracked list")
ap.add_argument("--inflate", type=float, nargs="*", default=[1.0, 2.0, 4.0])
ap.add_argument("--out", default=None)
args = ap.parse_args()

dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
F, M, K = args.frames, args.markers, args.k
seq = make_sequence(tables, seed=args.seed, num_frames=F, num_markers=M)
cfg = packaged_config("video_mocap")
markers = torch.nan_to_num(torch.from_numpy(seq.markers.get_points()).float()).to(dev)
o_pose = seq.img_smpl.pose_body.to(dev)
o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
root = seq.img_smpl.root_orient.to(dev)
if args.yaw != 0.0:
    root = compute_root_orient_z(torch.full((F, 1, 1), args.yaw, device=dev)) @ root
trans = torch.median(markers, dim=1)[0]
prob = ChamferProblem(smpl, markers, o_pose, o_betas, root, cfg)
x = prob.pack(trans, torch.zeros(F, 1, 1, device=dev), o_betas, o_pose)
points = []
st = prob.solve(x, max_iter=int(cfg["stages"]["chamfer"]["num_iters"]), lr=0.1,
                point_callback=lambda i, loss, xe: points.append(xe.clone()))
print("solve:", st, "recorded", len(points))
valid = (markers.abs().sum(-1) != 0)  # [F, M]
V = smpl.device_model.V
nU = (V + 15) // 16
nT = (F + 15) // 16


def verts_at(xe):
    t_, z_, b_, p_ = prob.unpack(xe.to(dev))
    r = normalize_rot(compute_root_orient_z(z_) @ root)
    return smpl(normalize_rot(p_), b_.expand(F, 10), r, t_)["vertices"]


def unit_boxes(v):
    pad = nU * 16 - V
    vp = torch.cat([v, v[:, -1:].expand(F, pad, 3)], dim=1).reshape(F, nU, 16, 3)
    return vp.min(dim=2)[0], vp.max(dim=2)[0]


res = {"frames": F, "markers": M, "evals": len(points), "solve": st, "K": K}
# ---------------------------------------------------------------------------------------------------------------- replay
prev_v = None
prev_win = None
unit_frac, task_frac = [], []
anchors = {g: {"v": None, "L": None, "rout": None, "dense": 0, "sparse": 0} for g in args.inflate}
delta_log = []
for e, xe in enumerate(points):
    v = verts_at(xe)                                   # [F, V, 3]
    d = torch.cdist(markers, v)                        # [F, M, V]
    dmin, win = d.min(dim=2)
    # (a) unit cull against the previous evaluation's boxes grown by the exact motion since then
    if prev_v is not None:
        delta = (v - prev_v).norm(dim=-1).max(dim=1)[0]          # [F] exact motion of the frame's vertices
        lo, hi = unit_boxes(prev_v)
        lo = lo - delta[:, None, None]
        hi = hi + delta[:, None, None]
        ub = torch.gather(d, 2, prev_win[:, :, None])[:, :, 0]     # [F, M] previous winner re-skinned now: exact upper bound
        gap = torch.clamp(torch.maximum(lo[:, None] - markers[:, :, None], markers[:, :, None] - hi[:, None]), min=0.0)
        lb = gap.norm(dim=-1)                                     # [F, M, nU]
        keep = ((lb <= ub[:, :, None]) & valid[:, :, None]).any(dim=1)   # [F, nU]
        unit_frac.append(float(keep.float().mean()))
        kt = torch.nn.functional.pad(keep, (0, 0, 0, nT * 16 - F)).reshape(nT, 16, nU).any(dim=1)
        task_frac.append(float(kt.float().mean()))
    prev_v, prev_win = v, win
    # (b) tracked candidate lists
    for g, a in anchors.items():
        ok = False
        if a["v"] is not None:
            delta = (v - a["v"]).norm(dim=-1).max(dim=1)[0] * g   # [F]
            d_in = torch.gather(d, 2, a["L"][None].expand(F, M, K)).min(dim=2)[0]
            cert = (d_in < a["rout"] - delta[:, None]) | ~valid
            ok = bool(cert.all())
            if g == args.inflate[0]:
                delta_log.append(float(delta.max()) / g)
        if ok:
            a["sparse"] += 1
        else:
            a["dense"] += 1
            # new anchor: per marker the most frequent winners over the valid frames, then the nearest on average
            score = torch.zeros(M, V, device=dev)
            score.scatter_add_(1, win.t(), valid.t().float())
            dm = (d * valid[:, :, None]).sum(0) / valid.sum(0).clamp_min(1)[:, None]     # [M, V] mean distance
            order = score * 1e3 - dm                                                          # winners first, then nearest
            L = order.topk(K, dim=1)[1]                                                       # [M, K]
            dd = d.clone()
            dd.scatter_(2, L[None].expand(F, M, K), float("inf"))
            a.update(v=v, L=L, rout=dd.min(dim=2)[0])
res["unit_survive_mean"] = sum(unit_frac) / max(1, len(unit_frac))
res["task_survive_mean"] = sum(task_frac) / max(1, len(task_frac))
res["unit_survive_last100"] = sum(unit_frac[-100:]) / max(1, len(unit_frac[-100:]))
res["task_survive_last100"] = sum(task_frac[-100:]) / max(1, len(task_frac[-100:]))
res["tracked"] = {str(g): {"dense": a["dense"], "sparse": a["sparse"],
                           "sparse_frac": a["sparse"] / max(1, a["dense"] + a["sparse"])} for g, a in anchors.items()}
qs = sorted(delta_log)
if qs:
    res["delta_since_anchor_m"] = {"median": qs[len(qs) // 2], "p90": qs[int(0.9 * len(qs))], "max": qs[-1]}
print(json.dumps(res, indent=1))
if args.out:
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
