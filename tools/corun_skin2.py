"""What stretches k_skin2 inside a fit?  The dominant kernel alone against the dominant kernel with ONE
other stream busy with a given kind of work, all at the bench's size (F = 300, M = 50):

    python tools/corun_skin2.py            ->  one JSON line (profiles/r4_corun_skin2.json)

Stream A: `uuo_time_closure(dominant_only=1)` -- isolated k_skin2 launches, one HIP-event pair each, as in bench.py's
roofline.  Stream B (another host thread, its own workspace): nothing | another chain's k_skin2 launches | marker-stage
closures (k_bwd_sparse with its fused finalize: latency-bound, 300 blocks x 4 waves at 168 VGPRs) | whole chamfer closures
(pose preparation, k_skin2, pruned search, backward) | an L-BFGS solve of the marker stage (history passes + the
coefficient kernel + backward: no k_skin2 at all).  Reported: k_skin2's average duration on stream A in each case, and how
much of B's work ran meanwhile.
"""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.engine import ChamferProblem, MarkerProblem, set_workspace_group, set_workspace_slot  # noqa: E402
from uuo_mocap_amd.smpl import SmplInference  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence  # noqa: E402

F, M = 300, 50
dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
cfg = packaged_config("video_mocap")


def problems(seed, group):
    set_workspace_group(group)
    set_workspace_slot(0)
    seq = make_sequence(tables, seed=seed, num_frames=F, num_markers=M)
    markers = torch.from_numpy(seq.markers.get_points()).float().to(dev)
    o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
    pose, root = seq.img_smpl.pose_body.to(dev), seq.img_smpl.root_orient.to(dev)
    med = torch.median(markers, dim=1)[0]
    pc = ChamferProblem(smpl, markers, pose, o_betas, root, cfg)
    xc = pc.pack(med, torch.zeros(F, 1, 1, device=dev), o_betas, pose)
    pm = MarkerProblem(smpl, markers, pose, o_betas, torch.from_numpy(seq.gt["marker_vids"]).to(dev), cfg)
    xm = pm.pack(pose, o_betas, root, med)
    return pc, xc, pm, xm


pc_a, xc_a, _, _ = problems(0, 0)
pc_a.time_closure(xc_a, iters=50, dominant_only=True)  # code objects, workspaces
results = {}
for kind in ("nothing", "k_skin2", "marker_closures", "chamfer_closures", "marker_solve"):
    stop = threading.Event()
    done = {"n": 0}

    def other():
        pc_b, xc_b, pm_b, xm_b = problems(1, 1)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            while not stop.is_set():
                if kind == "k_skin2":
                    pc_b.time_closure(xc_b, iters=20, dominant_only=True)
                    done["n"] += 20
                elif kind == "marker_closures":
                    pm_b.time_closure(xm_b, iters=20, dominant_only=False)
                    done["n"] += 20
                elif kind == "chamfer_closures":
                    pc_b.time_closure(xc_b, iters=20, dominant_only=False)
                    done["n"] += 20
                else:
                    x = xm_b.clone()
                    done["n"] += pm_b.solve(x, max_iter=150, lr=1.0)["n_eval"]
            st.synchronize()

    th = None
    if kind != "nothing":
        th = threading.Thread(target=other)
        th.start()
        time.sleep(1.5)  # B's workspaces and first launches
    n0, t0 = done["n"], time.perf_counter()
    ms = pc_a.time_closure(xc_a, iters=400, dominant_only=True)
    dt, n1 = time.perf_counter() - t0, done["n"]
    stop.set()
    if th is not None:
        th.join()
    results[kind] = {"k_skin2_us": 1e3 * ms, "other_stream_units_per_ms": (n1 - n0) / (1e3 * dt) if kind != "nothing" else 0.0}
    print(kind, results[kind], flush=True)
base = results["nothing"]["k_skin2_us"]
for k in results:
    results[k]["stretch"] = results[k]["k_skin2_us"] / base
print(json.dumps({"what": "k_skin2 (F=300) on one stream, HIP-event pairs, while ONE other stream runs the named work",
                  "results": results}))
