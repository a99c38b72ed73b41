# SQ / TCC counters of the latency-bound kernels (solver passes, backward, finalize): where their wave-cycles go.
#   gpurun -- bash tools/pmc_sq.sh   ->  gpurun_out/pmc_sq_summary.json
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/p_sq1 -- python3 $R/tools/profile_closure.py --evals 2 --solve-iters 150 > $R/gpurun_out/p_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/p_sq2 -- python3 $R/tools/profile_closure.py --evals 2 --solve-iters 150 > $R/gpurun_out/p_sq2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for tag in ("p_sq1", "p_sq2"):
    fs = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in fs:
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            tail = v[len(v) // 2:]
            out.setdefault(k, {})[c] = {"mean_last_half": sum(tail) / len(tail), "launches": len(v)}
json.dump(out, open("gpurun_out/pmc_sq_summary.json", "w"), indent=1, sort_keys=True)
for k in sorted(out):
    if k.startswith("k_"):
        print(k, {c: round(v["mean_last_half"], 1) for c, v in out[k].items()})
PY
rm -rf gpurun_out/p_sq1 gpurun_out/p_sq2
