import sys, os, time, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from uuo_mocap_amd.body_model import synthetic_smpl
from uuo_mocap_amd.smpl import SmplInference
from uuo_mocap_amd.synthetic import make_sequence
from uuo_mocap_amd.config import packaged_config
from uuo_mocap_amd.engine import ChamferProblem
dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
F, M = 300, 50
seq = make_sequence(tables, seed=0, num_frames=F, num_markers=M)
markers = torch.from_numpy(seq.markers.get_points()).float().nan_to_num().to(dev)
o_pose = seq.img_smpl.pose_body.to(dev)
o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
root = seq.img_smpl.root_orient.to(dev)
trans = torch.median(markers, dim=1)[0]
for soft in (False, True):
    cfg = copy.deepcopy(packaged_config("video_mocap"))
    if soft:
        del cfg["stages"]["chamfer"]["losses"]["full_chamfer"]
        cfg["stages"]["chamfer"]["losses"]["soft_chamfer"] = 10.0
        cfg["stages"]["chamfer"]["soft_tau"] = 1e-3
    prob = ChamferProblem(smpl, markers, o_pose, o_betas, root, cfg)
    x = prob.pack(trans, torch.zeros(F, 1, 1, device=dev), o_betas, o_pose)
    for _ in range(3): prob.evaluate(x, want_nn=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    import ctypes
    from uuo_mocap_amd.engine import _ptr, current_stream, check
    loss = torch.empty(1, device=dev); grad = torch.empty(prob.n, device=dev)
    for _ in range(n):
        check(prob.lib.uuo_closure_eval(prob.fit, current_stream(dev), ctypes.byref(prob.problem), _ptr(x), _ptr(loss), _ptr(grad), None), "eval")
    torch.cuda.synchronize()
    print(("soft" if soft else "hard") + " fused chamfer closure (uuo_closure_eval, incl. mask read-back): %.3f ms per evaluation" % (1e3 * (time.perf_counter() - t0) / n))
    xs = x.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = prob.solve(xs, max_iter=60, lr=0.1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("  solve: %d evaluations in %.1f ms = %.3f ms per evaluation, loss %.5f -> %.5f" % (st["n_eval"], 1e3 * dt, 1e3 * dt / st["n_eval"], st["first_loss"], st["final_loss"]))
