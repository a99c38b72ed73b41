# lock-step vs threaded yaw hypotheses at 1 / 2 / 3 sequences in flight:  gpurun -- bash tools/bench_variants.sh
set -e
for v in "0 1" "1 1" "1 2" "1 3" "0 3"; do
  set -- $v
  python bench.py $([ "$1" = 1 ] && echo --hypothesis-lockstep) --steps 6 --warmup 1 --inflight $2 --no-cpu-baseline --no-other-configs > gpurun_out/bv_$1_$2.log 2>gpurun_out/bv_$1_$2.err || { tail -5 gpurun_out/bv_$1_$2.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bv_$1_$2.log").read().strip().splitlines()[-1])
print("lockstep=$1 inflight=$2: %.1f frames/s  %.1f ms/step  evals/step %.0f  v2v %.2f mm fit_frac %.3f" % (d["value"], d["ms_per_step"], d["closure_evals_per_step"], d["fit_quality"]["mean"]["v2v_mm"], d["roofline"]["fit_frac"]))
PY
done
