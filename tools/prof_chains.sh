# kernel trace of the default bench (rocprofv3's --hip-trace segfaults under this workload: kernel trace only), reduced by tools/trace_chains.py:
#   gpurun -- bash tools/prof_chains.sh <tag> [bench args...]   ->  gpurun_out/r4/<tag>_chains[_api].log
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$TAG -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $R/gpurun_out/r4/${TAG}_bench.log 2>&1
python3 $R/tools/trace_chains.py /tmp/p_$TAG --dense > $R/gpurun_out/r4/${TAG}_chains.log
rm -rf /tmp/p_$TAG
cd $R
cat gpurun_out/r4/${TAG}_chains.log
