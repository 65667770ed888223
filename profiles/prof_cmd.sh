#!/bin/bash
# on the GPU box:  bash profiles/prof_cmd.sh TAG STEPS -- python3 <script> <args>     kernel time per step from rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
TAG=$1; STEPS=$2; shift 3
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- "$@" > $OUT/line.json 2> $OUT/err
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/stats_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
steps=$STEPS
print("kernel time per step (ms):", round(tot/steps/1e6,3), " over", steps, "steps (warm-up and set-up launches included)")
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/steps/1e6:8.3f} ms  x{int(r['Calls'])/steps:6.1f}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
rm -f $OUT/*_kernel_trace.csv
