#!/usr/bin/env python3
"""Attribution of k_bwd_sparse's latency (debug library only): the marker-stage closure is k_bwd_sparse + k_finalize, so
its HIP-event time with the kernel cut short after a phase (UUO_BWD_STOP = 1 prologue, 2 item loop, 3 slot reduction,
0 whole kernel) shows where the microseconds go.  One process per setting (the knob is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from uuo_mocap_amd import _lib
_lib.LIB_PATH = _lib.LIB_DEBUG_PATH
from uuo_mocap_amd.body_model import synthetic_smpl
from uuo_mocap_amd.config import packaged_config
from uuo_mocap_amd.engine import MarkerProblem, ChamferProblem
from uuo_mocap_amd.smpl import SmplInference
from uuo_mocap_amd.synthetic import make_sequence
dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
F, M = int(sys.argv[1]), int(sys.argv[2])
seq = make_sequence(tables, seed=0, num_frames=F, num_markers=M)
cfg = packaged_config("video_mocap")
markers = torch.from_numpy(seq.markers.get_points()).float().to(dev)
o_pose = seq.img_smpl.pose_body.to(dev)
o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
root = seq.img_smpl.root_orient.to(dev)
trans = torch.median(markers, dim=1)[0]
assign = torch.from_numpy(seq.gt["marker_vids"]).to(dev)
mprob = MarkerProblem(smpl, markers, o_pose, o_betas, assign, cfg)
xm = mprob.pack(o_pose, o_betas, root, trans)
mprob.time_closure(xm, iters=50)
print("%%.2f" %% (1e3 * mprob.time_closure(xm, iters=300)))
if os.environ.get("UUO_BWD_STOP") == "9":
    import ctypes, numpy as np
    lib = _lib.load()
    st = np.zeros((4096, 12), np.uint64)
    lib.uuo_debug_bwd_stamps.argtypes = [ctypes.c_void_p]
    torch.cuda.synchronize()
    assert lib.uuo_debug_bwd_stamps(st.ctypes.data) == 0
    st = st[:min(F, 4096), :11].astype(np.int64)
    d = np.diff(st, axis=1)
    names = ["frame state + LDS init", "item loop", "barrier", "slot reduction", "tree loads + init", "sweep (9 levels)",
             "shape gradient", "rotations: loads, GS backward, stores", "barrier", "stats + frame_part stores"]
    print("STAMPS cycles (median over %%d blocks): total %%d" %% (F, int(np.median(st[:, 10] - st[:, 0]))))
    for n_, v in zip(names, np.median(d, axis=0)):
        print("STAMPS   %%-40s %%7d" %% (n_, int(v)))
    order = np.argsort(st[:, 0])
    tot = (st[:, 10] - st[:, 0])[order]
    nb = min(F, 4096)
    print("STAMPS first block start -> last block end: %%d cycles" %% int(st[:, 10].max() - st[:, 0].min()))
    print("STAMPS block total by start order: first 128 blocks median %%d, blocks 256..511 median %%d, last 25%%%% median %%d" %%
          (int(np.median(tot[:128])), int(np.median(tot[256:512])) if nb > 300 else -1, int(np.median(tot[-nb // 4:]))))
    for lo, hi in ((0, 128), (nb - nb // 4, nb)):
        dd = np.median(d[order][lo:hi], axis=0)
        print("STAMPS   phases of blocks %%d..%%d: %%s" %% (lo, hi, " ".join(str(int(v)) for v in dd)))
''' % ROOT

if __name__ == "__main__":
    F, M = (sys.argv[1:3] + ["300", "50"])[:2] if len(sys.argv) >= 3 else ("300", "50")
    for stop, label in ((1, "prologue (frame state, LDS zero)"), (2, "+ item loop"), (3, "+ slot reduction"),
                        (0, "whole kernel (+ kinematic tail, outputs)")):
        env = dict(os.environ, UUO_BWD_STOP=str(stop))
        out = subprocess.run([sys.executable, "-c", CHILD, F, M], env=env, capture_output=True, text=True, timeout=600)
        print("UUO_BWD_STOP=%d  %-42s marker closure %s us/eval (k_bwd_sparse + k_finalize + 2 launch gaps)" %
              (stop, label, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED: " + out.stderr[-300:]))
    env = dict(os.environ, UUO_BWD_STOP="9")
    out = subprocess.run([sys.executable, "-c", CHILD, F, M], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout if out.stdout.strip() else "FAILED: " + out.stderr[-600:])
