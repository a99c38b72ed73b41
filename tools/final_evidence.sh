# The round-end evidence run: default bench line, the driver-flags line, tools/prof_all.sh (kernel stats + PMC), the L-BFGS
# passes' PMC bytes, the pass microbenchmark, k_skin3 against k_skin2 (times; every launch of eight fits in flight checked).
#   gpurun --timeout 1200 -- bash tools/final_evidence.sh [round tag, default r4]   ->  gpurun_out/<tag>/...
set -e
T=${1:-r4}
mkdir -p gpurun_out/$T
python bench.py > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/bench_driver_flags.json 2> gpurun_out/$T/bench_driver_flags.err
bash tools/prof_all.sh > gpurun_out/$T/prof_all.log 2>&1
bash tools/pmc_lbfgs.sh > gpurun_out/$T/pmc_lbfgs.log 2>&1
python tools/time_skin16.py > gpurun_out/$T/skin16_times.log 2>&1
python tools/skin16_stress.py > gpurun_out/$T/skin16_stress_full.log 2>&1; tail -1 gpurun_out/$T/skin16_stress_full.log > gpurun_out/$T/skin16_stress.log
for f in p_bench_kernel_stats.csv p_closure_kernel_stats.csv p_roof_kernel_stats.csv pmc_summary.json pmc_lbfgs_summary.json roof_plain.log; do cp gpurun_out/$f gpurun_out/$T/$f; done
ls gpurun_out/$T
