#!/bin/bash
# on the GPU box: fine-stage artefacts of a round:  bash profiles/collect_fine_stage.sh r04
R=${1:-r04}
O=gpurun_out/${R}fine; mkdir -p $O
python3 profiles/bench_hexplane.py 2000000 2>/dev/null | tail -1 > $O/${R}_hexplane_microbench.json
python3 profiles/bench_deform.py 2000000 2>/dev/null | tail -1 > $O/${R}_deform_microbench.json
bash profiles/prof_deform.sh $R > $O/${R}_deform_step_breakdown.txt 2>&1
OUT=$PWD/gpurun_out/hexprof bash profiles/prof_hexplane.sh > $O/${R}_hexplane_counters.txt 2>&1
bash profiles/prof_hexplane_traffic.sh >> $O/${R}_hexplane_counters.txt 2>&1
: > $O/${R}_full_step_variants.jsonl
for v in "" "--graph" "--fine" "--fine --graph" "--fine --feat --graph" "--fine --feat-separate --graph" "--adam --graph" "--fine --adam --graph"; do
  timeout 300 python3 profiles/bench_full_step.py $v 2>/dev/null | tail -1 >> $O/${R}_full_step_variants.jsonl
done
bash profiles/prof_cmd.sh fine_$R 60 -- python3 profiles/bench_full_step.py --fine > $O/${R}_fine_step_kernels.txt 2>&1
