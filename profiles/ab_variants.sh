#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes differ by +-2.5 %): bash profiles/ab_variants.sh [bench flags --] NAME1 NAME2 ...
# Each NAME is emd_amd/csrc/variants/lib_NAME.so (EMD_LIB_PATH selects the build); "base" is the in-tree library.  The list is run in the
# order given; put "base" first and last to see the drift of the box.  Everything in front of a literal `--` is passed to bench.py.
FLAGS="--steps 20 --warmup 5 --no-cpu-baseline --no-fine-stage"
NAMES=()
seen_sep=0
for a in "$@"; do [ "$a" = "--" ] && seen_sep=1; done
if [ $seen_sep = 1 ]; then
  EXTRA=()
  while [ "$1" != "--" ]; do EXTRA+=("$1"); shift; done
  shift
  FLAGS="$FLAGS ${EXTRA[*]}"
fi
NAMES=("$@")
for v in "${NAMES[@]}"; do
  (   # a subshell per variant: nothing exported for one build leaks into the next
    if [ "$v" != "base" ]; then export EMD_LIB_PATH=$PWD/emd_amd/csrc/variants/lib_$v.so; fi
    python3 bench.py $FLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), d['repeats_ms_per_step']['median'], {k:v['ms'] for k,v in d['roofline']['stages'].items()})"
  )
done
