# rocprofv3 kernel-trace summary of an arbitrary python command:
#   gpurun -- bash tools/prof_cmd.sh <tag> <script.py> [args...]   ->  gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SCRIPT=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_$TAG -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/${TAG}.log 2>&1
cd $R
f=$(find gpurun_out/p_$TAG -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/p_$TAG
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:14]:
    print("%-34s calls %6s avg %9.1f us total %9.1f ms %5s%%" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
