#!/bin/bash
# on the GPU box: wave-state counters + instruction counts of the hexplane kernels
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/hexprof
rm -rf $OUT; mkdir -p $OUT
P="python3 profiles/bench_hexplane.py 2000000 --hip-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $P > /dev/null 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT -o ws -- $P > /dev/null 2> $OUT/ws.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o sq -- $P > /dev/null 2> $OUT/sq.err
python3 profiles/summarise_counters.py $OUT "k_hexplane\w+"
grep hexplane $OUT/stats_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
rm -f $OUT/*_kernel_trace.csv $OUT/*counter_collection.csv
