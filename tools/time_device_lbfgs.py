"""Wall time of torch.optim.LBFGS vs DeviceLBFGS (uuo_lbfgs_minimize, host-composed closure) on the same closure: a
coupled quadratic of the chamfer stage's size (n = 211 F + 10 at F = 300), history 100.
    python tools/time_device_lbfgs.py [--n 63310] [--iters 150]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uuo_mocap_amd.device_lbfgs import DeviceLBFGS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=63310)
ap.add_argument("--iters", type=int, default=150)
a = ap.parse_args()
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
w = (1.0 + 99.0 * torch.rand(a.n, generator=gen)).to(dev)
c = torch.randn(a.n, generator=gen).to(dev)


def run(make):
    x = torch.zeros(a.n, device=dev, requires_grad=True)
    opt = make([x])
    n_eval = [0]

    def closure():
        opt.zero_grad()
        loss = 0.5 * (w * (x - c) ** 2).sum() + 0.05 * ((x[1:] - x[:-1]) ** 2).sum()
        loss.backward()
        n_eval[0] += 1
        return loss

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.step(closure)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, n_eval[0], float(closure())


kw = dict(max_iter=a.iters, tolerance_grad=1e-12, tolerance_change=1e-14, lr=1.0, line_search_fn="strong_wolfe")
for name, make in (("warm-up", lambda p: DeviceLBFGS(p, **kw)), ("torch.optim.LBFGS", lambda p: torch.optim.LBFGS(p, **kw)),
                   ("DeviceLBFGS", lambda p: DeviceLBFGS(p, **kw))):
    dt, ne, fl = run(make)
    print("%-18s %8.1f ms  %4d evals  %.3f ms/eval  final loss %.6g" % (name, 1e3 * dt, ne, 1e3 * dt / ne, fl))
