# Rehearsal of the multi-rank modes of bench.py on a ONE-GPU box: the ranks share the device and rendezvous over gloo
# (UUO_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device).  Shows that the modes run and what their protocol costs, not
# scaling.   gpurun -- bash tools/rehearse_modes.sh   ->  gpurun_out/r4/mode_<ranks>rank_<mode>[_<transport>].json
set -e
mkdir -p gpurun_out/r4
run() {  # ranks, mode, extra flags, tag
  UUO_BENCH_SHARE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 \
    --master-port $((29520 + RANDOM % 400)) bench.py --gpus $1 --steps ${STEPS:-4} --warmup 1 --mode $2 $3 \
    --no-cpu-baseline --no-other-configs > gpurun_out/r4/mode_${1}rank_$4.json 2> gpurun_out/r4/mode_${1}rank_$4.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4/mode_${1}rank_$4.json").read().strip().splitlines()[-1])
c = d.get("collective") or {}
pr = d["host"].get("per_rank") or []
print("${1} ranks $4: %.1f frames/s  %.1f ms/step  v2v %.2f mm  %s  throttled per rank %s  cpu-s per rank %s" % (
    d["value"], d["ms_per_step"], d["fit_quality"]["mean"]["v2v_mm"],
    ("%s, %d gathers, %.1f us each" % (c["transport"], c["gathers_timed"], c["mean_gather_us"])) if c.get("gathers_timed") else "",
    [p["nr_throttled_timed"] for p in pr], [round(p["cpu_seconds_timed"], 1) for p in pr]))
PY
}
run 2 sequences "--inflight 1" sequences
run 2 shared_betas "--collective-lanes 4 --collective-transport shm" shared_betas_shm
run 2 shared_betas "--collective-lanes 4 --collective-transport gloo" shared_betas_gloo
run 2 frames "--collective-lanes 4 --collective-transport shm" frames_shm
run 2 frames "--collective-lanes 4 --collective-transport gloo" frames_gloo
run 2 hypotheses "" hypotheses
# the launch SCALE uses (--mode sequences, three sequences in flight per rank), with as many ranks as the pool's process guard
# lets one box hold on its GPU with margin (the guard counts the launcher too: 6 ranks + torchrun = 7 > 6 was killed; the 8-rank case is the driver's to run on a real node)
STEPS=2 run 4 sequences "" sequences_inflight3
