# What do the two L-BFGS history passes cost a fit?  Same library, alternating runs of the bench with torch's history (100) and
# with a quarter of it (the passes' bytes scale with it; the solves take other trajectories, so compare frame-evals/s).
#   gpurun -- bash tools/ablate_history.sh [rounds]
set -e
mkdir -p gpurun_out/r4
for r in $(seq 1 ${1:-2}); do
  for h in 100 25; do
    python bench.py --no-other-configs --no-cpu-baseline --steps 9 --warmup 3 --history-size $h > gpurun_out/r4/hist_${h}_$r.json 2> gpurun_out/r4/hist.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r4/hist_${h}_$r.json").read().strip().splitlines()[-1])
print("history $h round $r: %.1f ms/step  %.0f evals/step  %.3f M frame-evals/s" % (d["ms_per_step"], d["closure_evals_per_step"], d["frame_evals_per_s"] / 1e6))
PY
  done
done
