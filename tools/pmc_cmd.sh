# PMC counters of one python command, one rocprofv3 pass per counter group (groups separated by '/'), summarised per kernel
# (mean over the last half of each kernel's launches):
#   gpurun -- bash tools/pmc_cmd.sh <tag> "<C1 C2 .. / C3 C4 ..>" <script.py> [args...]  ->  gpurun_out/<tag>_pmc.json
set -e
TAG=$1; shift
GROUPS_="$1"; shift
SCRIPT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
IFS='/' read -ra GS <<< "$GROUPS_"
for g in "${GS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/p_${TAG}_$i -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/${TAG}_pmc_$i.log 2>&1
done
cd $R
python3 - $TAG <<'PY'
import csv, glob, json, collections, sys
tag = sys.argv[1]
out = {}
for d in sorted(glob.glob("gpurun_out/p_%s_*" % tag)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            tail = v[len(v) // 2:]
            out.setdefault(k, {})[c] = {"mean_last_half": sum(tail) / len(tail), "launches": len(v)}
json.dump(out, open("gpurun_out/%s_pmc.json" % tag, "w"), indent=1, sort_keys=True)
for k in sorted(out):
    if "k_" in k:
        print(k[:40], {c: round(v["mean_last_half"], 1) for c, v in out[k].items()})
PY
rm -rf gpurun_out/p_${TAG}_*
