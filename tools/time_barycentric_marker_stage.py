"""The marker stage on a three-corner (barycentric) placement, 300 frames x 50 markers, 100 iterations: the fused closure
(k_bary_fwd + k_bwd_items under the device solver; execution.marker_bary_fused, default) against the closure composed from
the operators (SmplInference forward + einsum + uuo_smpl_backward under DeviceLBFGS).  And the one-hot stage of the shipped configs on the same
shapes.  Prints wall ms, evaluations, losses."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from uuo_mocap_amd.body_model import synthetic_smpl
from uuo_mocap_amd.config import packaged_config
from uuo_mocap_amd.optimization import last_stats, optim_markers
from uuo_mocap_amd.smpl import SmplInference
from uuo_mocap_amd.synthetic import make_sequence

dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
F, M = 300, 50
seq = make_sequence(tables, seed=0, num_frames=F, num_markers=M)
markers = torch.from_numpy(seq.markers.get_points()).float().nan_to_num().to(dev)
faces = torch.from_numpy(np.asarray(tables.faces).astype(np.int64))
vids = torch.from_numpy(np.asarray(seq.gt["marker_vids"])).long()
gen = torch.Generator().manual_seed(1)
coords = torch.zeros(M, 6890)
for m in range(M):
    tri = faces[(faces == vids[m]).any(1).nonzero()[0, 0]]
    w = torch.rand(3, generator=gen) + 0.05
    coords[m, tri] = w / w.sum()
coords = coords.to(dev)
o_pose = seq.img_smpl.pose_body.to(dev)
o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
root0 = seq.img_smpl.root_orient.to(dev)
trans0 = torch.median(markers, dim=1)[0]
one = torch.zeros(M, 6890, device=dev)
one[torch.arange(M), vids.to(dev)] = 1.0
for fused, mat, tag in ((True, coords, "fused"), (False, coords, "operators"), (True, one, "one-hot placement (shipped configs)")):
    cfg = packaged_config("video_mocap")
    cfg["stages"]["marker"]["num_iters"] = 100
    cfg["execution"] = dict(cfg.get("execution") or {}, marker_bary_fused=fused)
    for rep in range(2):
        leaves = [x.clone().requires_grad_(True) for x in (o_pose, o_betas, root0, trans0)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        optim_markers(markers=markers, pose_body=leaves[0], o_pose_body=o_pose, betas=leaves[1], o_betas=o_betas, root_orient=leaves[2],
                      trans=leaves[3], barycentric_coords_one_hot=mat, img_mask=torch.ones(F, device=dev), smpl_inference=smpl, config=cfg)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
    st = last_stats("marker")
    print("%s: %.1f ms, %d evaluations (%.3f ms each), loss %.6f -> %.6f" % (
        tag, ms, st["n_eval"], ms / st["n_eval"], st.get("first_loss", st.get("loss_first")),
        st.get("final_loss", st.get("loss_final"))))
