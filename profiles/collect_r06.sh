#!/bin/bash
# Round-6 artefacts, run ON THE GPU BOX from the repo root:  bash profiles/collect_r06.sh
#   1. profiles/run_profile.sh r06: rocprofv3 kernel stats of the default bench command (replayed + eager) and the four counter passes of the hot path
#   2. FETCH_SIZE / WRITE_SIZE passes over the fine-stage step with the optimiser -> profiles/r06_pmc_fine_traffic.csv
#   3. profiles/make_roofline_table.py -> profiles/r06_roofline_table.md (reads both counter summaries)
set -u
export TMPDIR=/tmp
bash profiles/run_profile.sh r06 > gpurun_out/r06_run_profile.log 2>&1
OUT=$PWD/gpurun_out/r06prof
for f in stats_kernel_stats.csv stats_eager_kernel_stats.csv; do [ -f "$OUT/$f" ] && cp "$OUT/$f" "profiles/r06_bench_$( [ $f = stats_kernel_stats.csv ] && echo kernel_stats || echo eager_kernel_stats ).csv"; done
cp "$OUT/bench_line_driver_flags.json" profiles/r06_bench_line_driver_flags.json 2>/dev/null
cp "$OUT/bench_line.json" profiles/r06_bench_line.json 2>/dev/null
cp "$OUT/bench_profiled.json" profiles/r06_bench_line_profiled.json 2>/dev/null
FO=$PWD/gpurun_out/r06fine; mkdir -p $FO
P="python3 profiles/bench_full_step.py --fine --adam"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$FO" -o fetch -- $P > /dev/null 2> "$FO/pmc.err"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$FO" -o write -- $P > /dev/null 2>> "$FO/pmc.err"
python3 profiles/make_pmc_summary.py "$FO" | sed 's#python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --eager --settle-ms 0 --repeats 0#python3 profiles/bench_full_step.py --fine --adam (60 eager steps)#' > profiles/r06_pmc_fine_traffic.csv
rm -f "$FO"/*_kernel_trace.csv "$FO"/*_counter_collection.csv
timeout 900 python3 profiles/make_roofline_table.py > profiles/r06_roofline_table.md 2> gpurun_out/r06_table.err
mkdir -p gpurun_out/r06_keep; cp profiles/r06_* gpurun_out/r06_keep/
tail -5 gpurun_out/r06_table.err; wc -l profiles/r06_roofline_table.md
