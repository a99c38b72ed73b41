"""k_skin3 (the chamfer closure's skinning on the fp16 matrix pipe, split operands) against k_skin2 (fp32 pipe) at 300 frames:
HIP-event time per launch of each, and of the whole chamfer closure (forward + backward) as the product runs it."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from uuo_mocap_amd.body_model import synthetic_smpl
from uuo_mocap_amd.config import packaged_config
from uuo_mocap_amd.engine import ChamferProblem
from uuo_mocap_amd.smpl import SmplInference
from uuo_mocap_amd.synthetic import make_sequence

dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
F, M = int(os.environ.get("F", 300)), 50
seq = make_sequence(tables, seed=0, num_frames=F, num_markers=M)
markers = torch.from_numpy(seq.markers.get_points()).float().to(dev)
o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
prob = ChamferProblem(smpl, markers, seq.img_smpl.pose_body.to(dev), o_betas, seq.img_smpl.root_orient.to(dev), packaged_config("video_mocap"))
x = prob.pack(torch.median(markers, dim=1)[0], torch.zeros(F, 1, 1, device=dev), o_betas, seq.img_smpl.pose_body.to(dev))
for rep in range(3):
    k2 = prob.time_closure(x, iters=200, dominant_only=1)
    k3 = prob.time_closure(x, iters=200, dominant_only=2)
    cl = prob.time_closure(x, iters=200, dominant_only=False)
    print("F=%d: k_skin2 (fp32 pipe) %.2f us, k_skin3 (fp16 split) %.2f us per launch; chamfer closure %.2f us per evaluation" % (F, 1e3 * k2, 1e3 * k3, 1e3 * cl))
