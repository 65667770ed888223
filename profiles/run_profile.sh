#!/bin/bash
# Profile passes of one round, run ON THE GPU BOX from the repo root:  bash profiles/run_profile.sh r02
#   1. timeout 420 rocprofv3 --kernel-trace --stats over the default bench command (graph replay) and over --eager
#   2. four --pmc passes (FETCH_SIZE, WRITE_SIZE, two SQ groups), each with --kernel-trace only, over the eager step
#   3. the summaries profiles/make_pmc_summary.py / make_valu_summary.py condense them into
# Everything lands in gpurun_out/<round>prof/; the summaries to be judged are then copied into profiles/.
set -u
R=${1:-r03}
OUT=$PWD/gpurun_out/${R}prof
mkdir -p "$OUT"
export TMPDIR=/tmp
# (round 5) every pass runs the DRIVER'S flags, so that the kernel averages, the counter bytes and the algorithmic bytes of the JSON line all describe
# the same frames (5..24 of the clip); the counter passes add --eager --settle-ms 0 --repeats 0: without the settle replays of frame 0 and the
# repeats, 20 of the 26 forward and 20 of the 26 backward dispatches of a pass are the timed steps (the rest: one sizing step and five warm-up steps on
# frames 0..4; the five pair-statistics steps run a differently named instantiation of K7 and re-run timed frames for the other kernels)
# (--no-fine-stage under rocprofv3: the fine_stage block uses torch.profiler, i.e. roctracer, which must not run inside a traced process)
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fine-stage"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $B > "$OUT/bench_profiled.json" 2> "$OUT/stats.err"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats_eager -- $B --eager > "$OUT/bench_profiled_eager.json" 2> "$OUT/stats_eager.err"
P="$B --eager --settle-ms 0 --repeats 0"
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o fetch -- $P > /dev/null 2> "$OUT/pmc.err"
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT" -o write -- $P > /dev/null 2>> "$OUT/pmc.err"
timeout 420 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$OUT" -o sq1 -- $P > /dev/null 2>> "$OUT/pmc.err"
timeout 420 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT" -o sq2 -- $P > /dev/null 2>> "$OUT/pmc.err"
python3 profiles/make_pmc_summary.py "$OUT" > "$OUT/${R}_pmc_hbm_traffic.csv"
python3 profiles/make_valu_summary.py "$OUT" > "$OUT/${R}_pmc_valu.csv"
cp "$OUT/${R}_pmc_hbm_traffic.csv" "$OUT/${R}_pmc_valu.csv" profiles/
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_line_driver_flags.json" 2> "$OUT/bench_line.err"
timeout 600 python3 bench.py > "$OUT/bench_line.json" 2>> "$OUT/bench_line.err"
# keep the merge-back small: the raw traces are large
rm -f "$OUT"/*_kernel_trace.csv "$OUT"/*_counter_collection.csv
ls -la "$OUT"
