# rocprofv3 kernel trace (every dispatch) of a python command, then tools/trace_timeline.py over it:
#   gpurun -- bash tools/prof_trace.sh <tag> <script.py> [args...]   ->  gpurun_out/<tag>_timeline.log
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SCRIPT=$1; shift
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p_$TAG -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/${TAG}.log 2>&1
cd $R
f=$(find gpurun_out/p_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f --frac ${FRAC:-0.5} > gpurun_out/${TAG}_timeline.log
if [ -n "$ROUNDS" ]; then python3 tools/trace_rounds.py $f "$ROUNDS" --last ${ROUNDS_LAST:-200} > gpurun_out/${TAG}_rounds.log; fi
rm -rf gpurun_out/p_$TAG
cat gpurun_out/${TAG}_timeline.log
