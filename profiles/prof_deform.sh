#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/deformprof_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 profiles/bench_deform.py 2000000 --hip-only > $OUT/line.json 2> $OUT/err
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/stats_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
steps=24
print("total per step ms", tot/steps/1e6)
for r in rows[:14]:
    print(f"{float(r['TotalDurationNs'])/steps/1e6:8.3f} ms  x{int(r['Calls'])/steps:5.1f}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
PY
rm -f $OUT/*_kernel_trace.csv
