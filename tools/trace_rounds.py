"""Per-launch durations of one kernel over a rocprofv3 --kernel-trace csv, in launch order (how a lock-step batch's rounds
shrink as its problems finish):  python tools/trace_rounds.py <kernel_trace.csv> <kernel-name-substring> [--last N]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("name")
ap.add_argument("--last", type=int, default=400)
a = ap.parse_args()
rows = [r for r in csv.DictReader(open(a.csv)) if a.name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-a.last:]
t0 = int(rows[0]["Start_Timestamp"])
print("launches", len(rows))
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%4d  t=%9.3f ms  dur=%8.1f us  grid=%s" % (i, (s - t0) / 1e6, (e - s) / 1e3, r.get("Grid_Size_X", "?")))
