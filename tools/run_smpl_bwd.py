import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from uuo_mocap_amd.body_model import synthetic_smpl
from uuo_mocap_amd.smpl import SmplInference
from oracle import p3d_ref
dev = torch.device("cuda:0")
smpl = SmplInference(dev, tables=synthetic_smpl(0))
F = 300
g = torch.Generator().manual_seed(7)
rot = p3d_ref.rotation_6d_to_matrix(torch.randn(F, 24, 6, generator=g))
args = [t.to(dev) for t in (rot[:, 1:].contiguous(), torch.randn(1, 10, generator=g), rot[:, :1].contiguous(), torch.randn(F, 3, generator=g), torch.randn(F, 6890, 3, generator=g))] + [None]
for _ in range(12): smpl.device_model.smpl_backward(*args)
torch.cuda.synchronize()
