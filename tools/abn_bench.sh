# A/B/n of builds of the library inside ONE gpurun call (same box, same clocks): alternates scratch_libs/<name>.so
#   gpurun -- bash tools/abn_bench.sh "a b c" [rounds] [bench args...]      (the FIRST name's library is restored at the end)
set -e
NAMES=$1; R=${2:-2}; shift 2 || true
mkdir -p gpurun_out/r4
for r in $(seq 1 $R); do
  for v in $NAMES; do
    cp scratch_libs/$v.so uuo_mocap_amd/libuuo_hip.so
    python bench.py --no-other-configs --no-cpu-baseline --steps 9 --warmup 3 "$@" > gpurun_out/r4/ab_${v}_$r.json 2> gpurun_out/r4/ab.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r4/ab_${v}_$r.json").read().strip().splitlines()[-1])
print("$v round $r: %.1f frames/s  %.1f ms/step  %.0f evals/step  %.3f M frame-evals/s  latency %.1f ms" % (d["value"], d["ms_per_step"], d["closure_evals_per_step"], d["frame_evals_per_s"] / 1e6, d.get("latency_one_sequence", {}).get("ms_per_fit", 0)))
PY
  done
done
set -- $NAMES
cp scratch_libs/$1.so uuo_mocap_amd/libuuo_hip.so
