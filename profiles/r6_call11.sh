#!/bin/bash
mkdir -p gpurun_out/r6
bash profiles/probe_exchange_block.sh > gpurun_out/r6/c11_exchange.txt 2>&1; cat gpurun_out/r6/c11_exchange.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_dp_nccl_gpu.py tests/test_bench_multirank_gpu.py tests/test_step_inputs_gpu.py -q -m gpu > gpurun_out/r6/c11_tests.txt 2>&1
tail -6 gpurun_out/r6/c11_tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fine-stage 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
