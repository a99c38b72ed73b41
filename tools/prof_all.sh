set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_bench -- python3 $R/bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/p_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_closure -- python3 $R/tools/profile_closure.py --solve-iters 300 > $R/gpurun_out/p_closure.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_pmc1 -- python3 $R/tools/profile_closure.py --evals 10 > $R/gpurun_out/p_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_pmc2 -- python3 $R/tools/profile_closure.py --evals 10 > $R/gpurun_out/p_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/p_pmc3 -- python3 $R/tools/profile_closure.py --evals 10 > $R/gpurun_out/p_pmc3.log 2>&1
cd $R
for d in p_bench p_closure; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${d}_kernel_stats.csv; done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for tag in ("p_pmc1", "p_pmc2", "p_pmc3"):
    fs = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in fs:
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1, sort_keys=True)
print("kernels:", list(out)[:12])
PY
rm -rf gpurun_out/p_bench gpurun_out/p_closure gpurun_out/p_pmc1 gpurun_out/p_pmc2 gpurun_out/p_pmc3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_roof -- python3 $R/bench.py --roofline-only > $R/gpurun_out/p_roof.log 2>&1
cd $R
f=$(find gpurun_out/p_roof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/p_roof_kernel_stats.csv; rm -rf gpurun_out/p_roof
python3 bench.py --roofline-only > gpurun_out/roof_plain.log 2>&1
