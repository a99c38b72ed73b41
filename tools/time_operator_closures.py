import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from uuo_mocap_amd.body_model import synthetic_smpl
from uuo_mocap_amd.smpl import SmplInference
from uuo_mocap_amd.synthetic import make_sequence
from uuo_mocap_amd.losses import soft_weighted_chamfer_distance, weighted_chamfer_distance
dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
F, M = 300, 50
seq = make_sequence(tables, seed=0, num_frames=F, num_markers=M)
markers = torch.from_numpy(seq.markers.get_points()).float().nan_to_num().to(dev)
pose = seq.img_smpl.pose_body.to(dev).requires_grad_(True)
betas = seq.img_smpl.betas.to(dev)[:1].clone().requires_grad_(True)
root = seq.img_smpl.root_orient.to(dev).requires_grad_(True)
trans = torch.median(markers, dim=1)[0].clone().requires_grad_(True)
mask = (markers.abs().sum(-1) != 0).float()
def ev(soft):
    out = smpl(pose, betas.expand(F, 10), root, trans)
    if soft:
        l = soft_weighted_chamfer_distance(markers, out["vertices"], mask, 1e-3)[0]
    else:
        l = weighted_chamfer_distance(markers, out["vertices"], mask)[0]
    l.backward()
    return l
for soft in (False, True):
    for _ in range(3): ev(soft)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): ev(soft)
    torch.cuda.synchronize(); print("soft" if soft else "hard", "operator-composed fwd+bwd: %.3f ms per evaluation" % (1e3 * (time.perf_counter() - t0) / n))
# backward alone
out = smpl(pose, betas.expand(F, 10), root, trans)
g = torch.randn_like(out["vertices"])
for _ in range(3): torch.autograd.grad(out["vertices"], (pose, betas, root, trans), g, retain_graph=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): torch.autograd.grad(out["vertices"], (pose, betas, root, trans), g, retain_graph=True)
torch.cuda.synchronize(); print("uuo_smpl_backward (dense upstream): %.3f ms" % (1e3 * (time.perf_counter() - t0) / 20))
