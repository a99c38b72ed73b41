"""Where the time of a fit with several solves in flight goes, chain by chain (rocprofv3 --kernel-trace [--hip-trace]).

    python tools/trace_chains.py <dir with *_kernel_trace.csv [and *_hip_api_trace.csv]> [--frac 0.5]

A "chain" is one HIP stream (one L-BFGS solve at a time: kernels strictly one after the other, the host in the loop once
per closure evaluation).  For the last `frac` of the trace this prints

* per hardware queue: dispatches, busy time (union of its kernels), sum of durations -- a sum above the union means kernels
  of different streams overlap INSIDE a queue, equality means the queue runs the streams mapped onto it one kernel at a time;
* the stream -> queue mapping;
* per kernel name: calls, average duration, and the average wait between "ready" and "started" (below);
* per chain and in total, the decomposition of wall time into
    kernel   the chain's own kernels running,
    host     the previous kernel of the chain has ended and the next one has not been ENQUEUED yet (the solver thread is
             deciding / launching); without a HIP API trace: the part of the gap during which the chain's queue was idle,
    wait     the next kernel is enqueued and its predecessor has ended, but it has not started: the queue (or the chip) is
             busy with other chains' kernels.
"""
import argparse
import collections
import csv
import glob
import sys

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--frac", type=float, default=0.5, help="analyse the last FRAC of the traced time span")
ap.add_argument("--dense", action="store_true",
                help="analyse the stretch of the trace with the highest dispatch rate instead (the timed fits of bench.py, "
                     "several sequences in flight): 50-ms bins, the run of bins around the densest one holding >= 60 %% of its count")
args = ap.parse_args()

kt = glob.glob(args.dir + "/**/*kernel_trace.csv", recursive=True)
if not kt:
    sys.exit("no kernel trace under " + args.dir)
rows = list(csv.DictReader(open(kt[0])))
api = glob.glob(args.dir + "/**/*hip_api_trace.csv", recursive=True)
enq = {}
if api:
    for r in csv.DictReader(open(api[0])):
        fn = r.get("Function", "")
        if "Launch" in fn or "launch" in fn:
            enq[r["Correlation_Id"]] = int(r["End_Timestamp"])
st = np.array([int(r["Start_Timestamp"]) for r in rows])
en = np.array([int(r["End_Timestamp"]) for r in rows])
lo = st.min() + int((en.max() - st.min()) * (1.0 - args.frac))
hi = en.max()
if args.dense:
    bin_ns = 50_000_000
    b = ((st - st.min()) // bin_ns).astype(np.int64)
    cnt = np.bincount(b)
    top = int(np.argmax(cnt))
    a_, z_ = top, top
    while a_ > 0 and cnt[a_ - 1] >= 0.6 * cnt[top]:
        a_ -= 1
    while z_ + 1 < len(cnt) and cnt[z_ + 1] >= 0.6 * cnt[top]:
        z_ += 1
    lo, hi = st.min() + a_ * bin_ns, st.min() + (z_ + 1) * bin_ns
keep = [i for i in range(len(rows)) if st[i] >= lo and st[i] < hi]
span = en[keep].max() - st[keep].min()
print("window %.1f ms, %d dispatches, HIP API trace: %s" % (span / 1e6, len(keep), "yes (%d launches)" % len(enq) if enq else "no"))


def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


by_q = collections.defaultdict(list)
by_s = collections.defaultdict(list)
s2q = collections.defaultdict(set)
for i in keep:
    r = rows[i]
    by_q[r["Queue_Id"]].append((st[i], en[i]))
    by_s[r["Stream_Id"]].append(i)
    s2q[r["Stream_Id"]].add(r["Queue_Id"])
print("\nhardware queues:")
for q, iv in sorted(by_q.items()):
    u = union(iv)
    print("  queue %-3s dispatches %6d  busy %5.1f %% of the window  sum of durations / busy = %.2f  streams on it: %d" % (
        q, len(iv), 100.0 * u / span, sum(e - s for s, e in iv) / max(u, 1),
        sum(1 for s_, qs in s2q.items() if q in qs)))
print("  all queues: busy (any kernel running) %.1f %%" % (100.0 * union([x for iv in by_q.values() for x in iv]) / span))

# busy time of every queue before time t (its kernels never overlap each other: see the ratio above), for splitting a chain's
# idle gaps into "its queue was running other chains' kernels" and "its queue was idle too" (= the host had not enqueued yet)
qtab = {}
for q, iv in by_q.items():
    iv = sorted(iv)
    s_arr = np.array([a for a, _ in iv], dtype=np.int64)
    e_arr = np.array([b for _, b in iv], dtype=np.int64)
    qtab[q] = (s_arr, e_arr, np.concatenate([[0], np.cumsum(e_arr - s_arr)]))


def q_busy_before(q, t):
    s_arr, e_arr, cum = qtab[q]
    j = int(np.searchsorted(s_arr, t, side="right"))  # kernels started at or before t
    if j == 0:
        return 0
    return int(cum[j - 1]) + int(min(t, e_arr[j - 1]) - s_arr[j - 1])


tot = collections.Counter()
per_kernel = collections.defaultdict(lambda: [0, 0, 0, 0])  # calls, duration, wait, host
chains = []
for s_, idx in by_s.items():
    idx.sort(key=lambda i: st[i])
    if len(idx) < 200:
        continue  # not a solve chain (the orchestrator's own small streams)
    k = h = w = 0
    prev_end = None
    for i in idx:
        d = en[i] - st[i]
        k += d
        name = rows[i]["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
        pk = per_kernel[name]
        pk[0] += 1
        pk[1] += d
        if prev_end is not None:
            e_ = enq.get(rows[i]["Correlation_Id"])
            if e_ is None:
                # no API trace: the part of the gap during which the chain's queue ran other chains' kernels is queue wait,
                # the rest (queue idle, kernel not started) is the host deciding / launching (+ dispatch latency)
                gap = max(0, st[i] - prev_end)
                q_ = rows[i]["Queue_Id"]
                ww = min(gap, max(0, q_busy_before(q_, st[i]) - q_busy_before(q_, prev_end))) if gap else 0
                hw = gap - ww
            else:
                ready = max(prev_end, e_)
                hw = max(0, e_ - prev_end)
                ww = max(0, st[i] - ready)
            # idle stretches of a stream between solves (the orchestrator is busy elsewhere) are not part of a chain's life
            if st[i] - prev_end < 2_000_000:
                h += hw
                w += ww
                pk[2] += ww
                pk[3] += hw
        prev_end = max(prev_end or 0, en[i])
    chains.append((s_, len(idx), k, h, w, sorted(s2q[s_])))
    tot["kernel"] += k
    tot["host"] += h
    tot["wait"] += w
print("\nchains (streams with >= 200 dispatches): %d" % len(chains))
for s_, n, k, h, w, qs in sorted(chains, key=lambda c: -c[2])[:16]:
    t = k + h + w
    print("  stream %-4s queues %-8s dispatches %6d  kernel %6.1f ms (%4.1f %%)  host %6.1f ms (%4.1f %%)  wait %6.1f ms (%4.1f %%)" % (
        s_, ",".join(qs), n, k / 1e6, 100.0 * k / t, h / 1e6, 100.0 * h / t, w / 1e6, 100.0 * w / t))
t = sum(tot.values())
print("  all chains: kernel %.1f %%, host %.1f %%, wait %.1f %% of %.1f chain-ms (= %.2f x the window)" % (
    100.0 * tot["kernel"] / t, 100.0 * tot["host"] / t, 100.0 * tot["wait"] / t, t / 1e6, t / span))
print("\n%-30s %8s %9s %10s %10s" % ("kernel", "calls", "avg us", "wait us", "host us"))
for name, (c, d, w, h) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-30s %8d %9.1f %10.1f %10.1f" % (name, c, d / c / 1e3, w / c / 1e3, h / c / 1e3))


# ---- per kernel: how much of the window has at least one / at least two of its dispatches running (a dispatch's interval
# includes the time its blocks wait for room on the CUs, so the SUM of durations says little about a kernel that cannot share a CU
# with another launch of itself: k_skin2)
def depth_times(iv):
    ev = []
    for s_, e_ in iv:
        ev.append((s_, 1))
        ev.append((e_, -1))
    ev.sort()
    t_prev, d, acc = None, 0, collections.defaultdict(int)
    for t, k in ev:
        if t_prev is not None and d > 0:
            acc[min(d, 4)] += t - t_prev
        d += k
        t_prev = t
    return acc


by_k = collections.defaultdict(list)
for i in keep:
    by_k[rows[i]["Kernel_Name"].split("(")[0].replace("void ", "")[:28]].append((st[i], en[i]))
print("\nkernel                          union %   >=2 at once %   >=3 %   sum of durations %")
for k, iv in sorted(by_k.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:9]:
    acc = depth_times(iv)
    tot = sum(acc.values())
    print("%-30s %7.1f %12.1f %10.1f %14.1f" % (k, 100.0 * tot / span, 100.0 * (tot - acc[1]) / span,
                                              100.0 * (acc[3] + acc[4]) / span, 100.0 * sum(e - s for s, e in iv) / span))
