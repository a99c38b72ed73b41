# A/B of two builds of the library inside ONE gpurun call (same box, same clocks): alternates scratch_libs/<a>.so and <b>.so
#   gpurun -- bash tools/ab_bench.sh old new [rounds] [bench args...]
set -e
A=$1; B=$2; R=${3:-2}; shift 3 || true
for r in $(seq 1 $R); do
  for v in $A $B; do
    cp scratch_libs/$v.so uuo_mocap_amd/libuuo_hip.so
    python bench.py --no-other-configs --no-cpu-baseline --steps 6 "$@" > gpurun_out/ab_${v}_$r.json 2> gpurun_out/ab.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_${v}_$r.json").read().strip().splitlines()[-1])
print("$v round $r: %.1f frames/s  %.1f ms/step  %.0f evals/step  %.3f M frame-evals/s" % (d["value"], d["ms_per_step"], d["closure_evals_per_step"], d["frame_evals_per_s"] / 1e6))
PY
  done
done
