#!/bin/bash
# A/B of ENVIRONMENT switches on the fine-stage step inside one gpurun call: bash profiles/ab_fine_env.sh "VAR=0" "VAR=1" ...   ("-" = nothing set)
# Each entry runs in its own subshell, so a variable exported for one entry is NOT set for the next ("-" really is the unmodified environment).
for v in "$@"; do
  (
    if [ "$v" != "-" ]; then export "$v"; fi
    timeout 300 python3 profiles/bench_full_step.py --fine --graph 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  )
done
