"""Timeline summary of a rocprofv3 --kernel-trace csv: busy (union of kernel intervals) vs idle time of the GPU inside the
window that holds the last `--frac` of the dispatches, per-kernel totals, and the distribution of idle gaps.
  python tools/trace_timeline.py gpurun_out/<tag>_kernel_trace.csv [--frac 0.5]"""
import argparse
import csv

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--frac", type=float, default=0.5)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
nm = np.array([r["Kernel_Name"].split("(")[0][:40] for r in rows])
o = np.argsort(st)
st, en, nm = st[o], en[o], nm[o]
k0 = int(len(st) * (1 - a.frac))
st, en, nm = st[k0:], en[k0:], nm[k0:]
span = (en.max() - st[0]) / 1e6
cur_end = np.maximum.accumulate(en)
gaps = np.maximum(st[1:] - cur_end[:-1], 0)
busy = span - gaps.sum() / 1e6
print("window %.2f ms  dispatches %d  busy %.2f ms (%.1f%%)  idle %.2f ms" % (span, len(st), busy, 100 * busy / span, gaps.sum() / 1e6))
g = gaps[gaps > 0] / 1e3
if len(g):
    print("idle gaps: n %d  median %.1f us  p90 %.1f us  max %.1f us;  sum of gaps > 50 us: %.2f ms (%d)" % (
        len(g), np.median(g), np.percentile(g, 90), g.max(), g[g > 50].sum() / 1e3, (g > 50).sum()))
print("%-42s %7s %10s %9s" % ("kernel", "calls", "total ms", "avg us"))
tot = {}
for n, s, e in zip(nm, st, en):
    t = tot.setdefault(n, [0, 0])
    t[0] += 1
    t[1] += e - s
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-42s %7d %10.2f %9.1f" % (n, c, t / 1e6, t / 1e3 / c))
print("sum of kernel durations %.2f ms" % (sum(t for _, t in tot.values()) / 1e6))
