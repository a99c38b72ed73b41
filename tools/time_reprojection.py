"""Times one yaw hypothesis of the 2D-prior fit (stages.reprojection_part; off in every shipped config) at F x M on the GPU:
the fused closure (uuo_reprojection_*) against the closure composed from the package's differentiable operators.
    gpurun -- python tools/time_reprojection.py [--frames 300] [--markers 50] [--iters 60]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.parallel import limit_host_threads  # noqa: E402
from uuo_mocap_amd.reprojection import optim_reprojection, reprojection_problem  # noqa: E402
from uuo_mocap_amd.smpl import SmplInference  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence, synthetic_hmr_camera  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--markers", type=int, default=50)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--out", default="")
    ap.add_argument("--fused-only", action="store_true", help="skip the operator-composed solve (tens of thousands of tiny "
                                                              "launches: too slow under a PMC pass)")
    a = ap.parse_args()
    limit_host_threads()
    dev = torch.device("cuda:0")
    tables = synthetic_smpl(0)
    smpl = SmplInference(dev, tables=tables)
    F, M = a.frames, a.markers
    seq = make_sequence(tables, seed=4, num_frames=F, num_markers=M)
    cfg = packaged_config("video_mocap")
    cfg["stages"]["reprojection_part"]["num_iters"] = a.iters
    markers = torch.from_numpy(seq.markers.get_points()).float().to(dev)
    img = seq.img_smpl
    betas = (img.betas.sum(0, keepdim=True) / img.img_mask.sum()).to(dev)
    trans = torch.median(markers, dim=1)[0].clone()
    pred_cam, center, size, scale = [t.to(dev) for t in synthetic_hmr_camera(F)]
    args = dict(markers=markers, pose_body=img.pose_body.to(dev), betas=betas, hmr_betas=img.betas.to(dev),
                root_orient=img.hmr_root_orient.to(dev), trans=trans, pred_cam=pred_cam, cam_center=center, cam_size=size,
                cam_scale=scale, angle=torch.tensor(0.0), smpl_inference=smpl, config=cfg)
    prob, x0 = reprojection_problem(**args)
    # closure alone: back-to-back evaluations
    loss = torch.empty(1, device=dev)
    prob.evaluate(x0)
    torch.cuda.synchronize()
    n = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from ctypes import c_void_p
    lib = prob.lib
    grad = torch.empty(prob.n, device=dev)
    st = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    e0.record()
    for _ in range(n):
        lib.uuo_reprojection_eval(prob.handle, st, c_void_p(x0.data_ptr()), c_void_p(loss.data_ptr()), c_void_p(grad.data_ptr()),
                                  None, None)
    e1.record()
    torch.cuda.synchronize()
    us_eval = e0.elapsed_time(e1) / n * 1e3
    res = {"F": F, "M": M, "fused_closure_us": us_eval}
    for driver in (("fused",) if a.fused_only else ("fused", "operators")):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = optim_reprojection(img_mask=img.img_mask.to(dev), num_iters=a.iters, driver=driver, **args)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
        s = out["solver"]
        res[driver] = {"ms": ms, "n_eval": s["n_eval"], "n_iter": s["n_iter"], "first_loss": s["first_loss"],
                       "final_loss": s["final_loss"], "solve_device_ms": s["device_ms"], "stop": s["stop_reason"],
                       "output_angle": out["output_angle"], "metrics": out["metrics"]}
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
