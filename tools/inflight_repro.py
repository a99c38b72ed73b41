"""Are fits in flight bit-reproducible?  N synthetic 300 x 50 sequences fitted with several in flight (parallel.fit_many), R times
over; every output tensor of every fit must be bitwise equal between the repetitions -- a sporadic wrong value in ANY kernel of
any chain (they share CUs and SIMDs while in flight) would send the affected solve down another trajectory.  Product library."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np, torch
from uuo_mocap_amd.body_model import synthetic_smpl
from uuo_mocap_amd.config import packaged_config
from uuo_mocap_amd.multimodal import multimodal_video_mocap
from uuo_mocap_amd.parallel import fit_many, limit_host_threads
from uuo_mocap_amd.smpl import SmplInference
from uuo_mocap_amd.synthetic import make_sequence

limit_host_threads()
dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
n_seq, inflight, reps = int(os.environ.get("N_SEQ", 8)), int(os.environ.get("INFLIGHT", 4)), int(os.environ.get("REPS", 3))
seqs = [make_sequence(tables, seed=300 + i, num_frames=300, num_markers=50) for i in range(n_seq)]
cfg = packaged_config("video_mocap")
def fit(sq):
    out = multimodal_video_mocap(sq.img_smpl, copy.deepcopy(sq.markers), dev, cfg, offset=0, print_options=[], save_stages=False, smpl_inference=smpl)
    return {k: out[k].detach().cpu().numpy().copy() for k in ("trans", "pose_body", "root_orient", "betas")}
runs = []
t0 = time.perf_counter()
for r in range(reps):
    runs.append(fit_many(seqs, fit, inflight=inflight, device=dev))
torch.cuda.synchronize()
diff = 0
for r in range(1, reps):
    for a, b in zip(runs[0], runs[r]):
        diff += sum(int(not np.array_equal(a[k], b[k])) for k in a)
print("%d fits x %d repetitions, %d in flight, %.1f s: output tensors that differ between repetitions: %d" % (n_seq, reps, inflight, time.perf_counter() - t0, diff))
sys.exit(0 if diff == 0 else 1)
