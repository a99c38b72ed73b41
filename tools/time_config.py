#!/usr/bin/env python3
"""Sequential fits of one shipped configuration on synthetic sequences, one line per fit with the orchestrator's phase times
(multimodal_video_mocap's own timeline): where the milliseconds of a latency-bound configuration (hmr_full / hmr_part) go."""
import argparse
import contextlib
import copy
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.multimodal import last_run_stats, multimodal_video_mocap  # noqa: E402
from uuo_mocap_amd.smpl import SmplInference  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="hmr_full")
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--markers", type=int, default=50)
ap.add_argument("--fits", type=int, default=8)
args = ap.parse_args()
from uuo_mocap_amd.parallel import limit_host_threads  # noqa: E402
limit_host_threads()
dev = torch.device("cuda:0")
tables = synthetic_smpl(0)
smpl = SmplInference(dev, tables=tables)
cfg = packaged_config(args.config)
limb = args.config in ("hmr_part", "hmr_part_soft")
seqs = [make_sequence(tables, seed=1000 + i, num_frames=args.frames, num_markers=10 if limb else args.markers, limb_only=limb)
        for i in range(args.fits)]
for i, sq in enumerate(seqs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        multimodal_video_mocap(sq.img_smpl, copy.deepcopy(sq.markers), dev, cfg, offset=0, print_options=[], save_stages=False,
                               smpl_inference=smpl)
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    st = last_run_stats()
    tl = st["timeline"]
    phases = {l: round(1e3 * (t - p), 2) for (l, t), p in zip(tl, [0.0] + [t for _, t in tl[:-1]])}
    ev = sum(s["n_eval"] for s in st.get("part", []))
    print("fit %d: %.2f ms  part evals %d (%d solves)  %s" % (i, dt, ev, len(st.get("part", [])), phases), flush=True)
