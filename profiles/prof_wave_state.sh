#!/bin/bash
# on the GPU box: wave-state counters of the bench step's kernels (one --pmc pass, eager step), as profiles/r02_wave_state_counters.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/wavestate
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT -o ws -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --eager > /dev/null 2> $OUT/ws.err
python3 profiles/summarise_counters.py $OUT "k_[a-z_0-9]+"
rm -f $OUT/*_kernel_trace.csv $OUT/*counter_collection.csv
