#!/bin/bash
# on the GPU box: fine-stage artefacts of round 3
O=gpurun_out/r03fine; mkdir -p $O
python3 profiles/bench_hexplane.py 2000000 2>/dev/null | tail -1 > $O/r03_hexplane_microbench.json
python3 profiles/bench_deform.py 2000000 2>/dev/null | tail -1 > $O/r03_deform_microbench.json
bash profiles/prof_deform.sh r03 > $O/r03_deform_step_breakdown.txt 2>&1
OUT=$PWD/gpurun_out/hexprof bash profiles/prof_hexplane.sh > $O/r03_hexplane_counters.txt 2>&1
: > $O/r03_full_step_variants.jsonl
for v in "" "--graph" "--fine" "--fine --graph" "--fine --feat --graph" "--fine --feat-separate --graph" "--adam --graph" "--fine --adam --graph"; do
  timeout 300 python3 profiles/bench_full_step.py $v 2>/dev/null | tail -1 >> $O/r03_full_step_variants.jsonl
done
