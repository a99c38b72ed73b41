# sequences in flight per GPU: the bench at --inflight 2..6, alternating, one call
#   gpurun -- bash tools/sweep_inflight.sh [rounds]
set -e
mkdir -p gpurun_out/r4
for r in $(seq 1 ${1:-2}); do
  for n in 3 4 5 6 2; do
    python bench.py --no-other-configs --no-cpu-baseline --steps 12 --warmup $n --inflight $n > gpurun_out/r4/inflight_${n}_$r.json 2> gpurun_out/r4/inflight.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r4/inflight_${n}_$r.json").read().strip().splitlines()[-1])
print("inflight $n round $r: %.1f frames/s  %.1f ms/step  %.3f M frame-evals/s  cpu-s %.1f throttled %s" % (d["value"], d["ms_per_step"], d["frame_evals_per_s"] / 1e6, d["host"]["cpu_seconds_timed"], d["host"]["nr_throttled_timed"]))
PY
  done
done
