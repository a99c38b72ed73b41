# Rehearsal of the multi-rank modes of bench.py on a ONE-GPU box: two ranks share the device and rendezvous over gloo
# (UUO_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device).  Shows that the modes run and what their protocol costs, not
# scaling.   gpurun -- bash tools/rehearse_modes.sh   ->  gpurun_out/r3_mode_2rank_<mode>[_lanes<n>].json
set -e
run() {  # mode, extra flags, tag
  UUO_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $((29520 + RANDOM % 400)) bench.py --gpus 2 --steps 4 --warmup 1 --inflight 1 --mode $1 $2 \
    --no-cpu-baseline --no-other-configs > gpurun_out/r3_mode_2rank_$3.json 2> gpurun_out/r3_mode_$3.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r3_mode_2rank_$3.json").read().strip().splitlines()[-1])
print("$3: %.1f frames/s  %.1f ms/step  v2v %.2f mm" % (d["value"], d["ms_per_step"], d["fit_quality"]["mean"]["v2v_mm"]))
PY
}
run sequences "" sequences
run hypotheses "" hypotheses
for l in 4 0; do
  run shared_betas "--collective-lanes $l" shared_betas_lanes$l
  run frames "--collective-lanes $l" frames_lanes$l
done
