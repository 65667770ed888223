#!/bin/bash
# A/B of library builds on the FINE-STAGE step (profiles/bench_full_step.py --fine --graph), inside one gpurun call: bash profiles/ab_fine.sh NAME...
for v in "$@"; do
  if [ "$v" != "base" ]; then export EMD_LIB_PATH=$PWD/emd_amd/csrc/variants/lib_$v.so; else unset EMD_LIB_PATH; fi
  timeout 300 python3 profiles/bench_full_step.py --fine --graph 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
done
