#!/bin/bash
# round 6, first GPU call: sanity of the dp changes + two probes
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_bench_multirank_gpu.py tests/test_dp_nccl_gpu.py tests/test_exchange_rows_gpu.py -x -q -m gpu > gpurun_out/r6/c1_tests.txt 2>&1
tail -3 gpurun_out/r6/c1_tests.txt
timeout 300 python profiles/probe_density_event.py > gpurun_out/r6/c1_event.txt 2>&1
head -60 gpurun_out/r6/c1_event.txt
timeout 400 python profiles/probe_torch_profiler.py > gpurun_out/r6/c1_profiler.txt 2>&1
tail -40 gpurun_out/r6/c1_profiler.txt
