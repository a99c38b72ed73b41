"""cProfile of one fit of a shipped configuration on a synthetic sequence (after one warm-up fit): where the HOST time of
a fit goes.   python tools/pyprof_fit.py --config hmr_part --markers 10 [--top 35]"""
import argparse
import contextlib
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uuo_mocap_amd.body_model import synthetic_smpl  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402
from uuo_mocap_amd.smpl import SmplInference  # noqa: E402
from uuo_mocap_amd.synthetic import make_sequence  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="hmr_part")
ap.add_argument("--markers", type=int, default=10)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--top", type=int, default=35)
a = ap.parse_args()
from uuo_mocap_amd.parallel import limit_host_threads  # noqa: E402
limit_host_threads()
dev = torch.device("cuda:0")
tables = synthetic_smpl()
smpl = SmplInference(dev, tables=tables)
cfg = packaged_config(a.config)
seqs = [make_sequence(tables, seed=1000 + i, num_frames=a.frames, num_markers=a.markers, limb_only=a.config == "hmr_part")
        for i in range(3)]
with contextlib.redirect_stdout(io.StringIO()):
    bench.fit_once(smpl, seqs[0], cfg, dev)
    bench.fit_once(smpl, seqs[1], cfg, dev)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    out, st = bench.fit_once(smpl, seqs[2], cfg, dev)
    torch.cuda.synchronize()
    pr.disable()
tl = st["timeline"]
print("timeline ms:", {l: round(1e3 * (t - p), 2) for (l, t), p in zip(tl, [0.0] + [t for _, t in tl[:-1]])})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(a.top)
print(s.getvalue()[:9000])
