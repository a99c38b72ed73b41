"""k_skin3 under the conditions of a dataset run: whole fits of 300 x 50 sequences, several in flight (parallel.fit_many), on the
DEBUG flavour of the library with UUO_SKIN_F16_CHECK=1 -- every launch of the fp16-split skinning kernel is followed by the fp32
kernel on the same operands and a device-side count of the vertex / box values that differ by more than 1e-5 m.  Prints the
number of launches checked and the two counts (both must be 0)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
os.environ["UUO_SKIN_F16_CHECK"] = "1"
import torch
from uuo_mocap_amd import _lib
_lib.LIB_PATH = _lib.LIB_DEBUG_PATH  # the check exists in the debug flavour only
from uuo_mocap_amd.body_model import synthetic_smpl
from uuo_mocap_amd.config import packaged_config
from uuo_mocap_amd.multimodal import multimodal_video_mocap
from uuo_mocap_amd.parallel import fit_many, limit_host_threads
from uuo_mocap_amd.smpl import SmplInference
from uuo_mocap_amd.synthetic import make_sequence
import copy

limit_host_threads()
dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
n_seq, inflight = int(os.environ.get("N_SEQ", 8)), int(os.environ.get("INFLIGHT", 4))
seqs = [make_sequence(tables, seed=100 + i, num_frames=300, num_markers=50) for i in range(n_seq)]
cfg = packaged_config("video_mocap")
lib = _lib.load_debug()
out = (ctypes.c_ulonglong * 3)()
assert lib.uuo_debug_skin16_check(out, 1) == 0
def fit(sq):
    return multimodal_video_mocap(sq.img_smpl, copy.deepcopy(sq.markers), dev, cfg, offset=0, print_options=[], save_stages=False, smpl_inference=smpl)
t0 = time.perf_counter()
fit_many(seqs, fit, inflight=inflight, device=dev)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert lib.uuo_debug_skin16_check(out, 0) == 0
print("%d fits, %d in flight, %.1f s: %d launches of k_skin3 checked against k_skin2; vertex values off by > 1e-5 m: %d; box values off: %d"
      % (n_seq, inflight, dt, out[0], out[1], out[2]))
sys.exit(0 if (out[0] > 0 and out[1] == 0 and out[2] == 0) else 1)
