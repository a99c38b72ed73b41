set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_pmc_lb1 -- python3 $R/tools/profile_closure.py --evals 2 --solve-iters 150 > $R/gpurun_out/p_pmc_lb1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_pmc_lb2 -- python3 $R/tools/profile_closure.py --evals 2 --solve-iters 150 > $R/gpurun_out/p_pmc_lb2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for tag in ("p_pmc_lb1", "p_pmc_lb2"):
    fs = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in fs:
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            tail = v[len(v) // 2:]  # second half of the solve: history (nearly) full
            out.setdefault(k, {})[c] = {"mean_per_launch": sum(v) / len(v), "mean_last_half": sum(tail) / len(tail), "launches": len(v)}
json.dump(out, open("gpurun_out/pmc_lbfgs_summary.json", "w"), indent=1, sort_keys=True)
for k in out:
    if k.startswith("k_lb") or k.startswith("k_bwd") or k.startswith("k_nn_cull"):
        print(k, {c: round(v["mean_last_half"], 1) for c, v in out[k].items()})
PY
rm -rf gpurun_out/p_pmc_lb1 gpurun_out/p_pmc_lb2
