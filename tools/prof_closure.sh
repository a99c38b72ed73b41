# rocprofv3 kernel trace of back-to-back closures + one 300-iteration chamfer solve (single stream):
#   gpurun -- bash tools/prof_closure.sh <tag>      ->  gpurun_out/<tag>_closure_kernel_stats.csv
set -e
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_closure -- python3 $R/tools/profile_closure.py --solve-iters 300 > $R/gpurun_out/${TAG}_closure.log 2>&1
cd $R
f=$(find gpurun_out/p_closure -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${TAG}_closure_kernel_stats.csv
rm -rf gpurun_out/p_closure
head -12 gpurun_out/${TAG}_closure_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-260
