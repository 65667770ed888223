#!/bin/bash
# on the GPU box: fabric bytes per launch of the HexPlane kernels (separate --pmc passes, as profiles/run_profile.sh does for the bench)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/hextraffic
rm -rf $OUT; mkdir -p $OUT
P="python3 profiles/bench_hexplane.py 2000000 --hip-only"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $P > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- $P > /dev/null 2> $OUT/write.err
python3 profiles/summarise_counters.py $OUT "k_hexplane\w+"
rm -f $OUT/*_kernel_trace.csv $OUT/*counter_collection.csv
