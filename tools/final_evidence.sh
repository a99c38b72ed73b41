# The round-end evidence run: default bench line, the driver-flags line, tools/prof_all.sh (kernel stats + PMC).
#   gpurun --timeout 1200 -- bash tools/final_evidence.sh   ->  gpurun_out/r3_bench_*.json, p_*_kernel_stats.csv, pmc_summary.json
set -e
python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_driver_flags.json 2> gpurun_out/r3_bench_driver_flags.err
bash tools/prof_all.sh > gpurun_out/r3_prof_all.log 2>&1
ls gpurun_out/p_*kernel_stats.csv gpurun_out/pmc_summary.json gpurun_out/roof_plain.log
