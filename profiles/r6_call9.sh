#!/bin/bash
mkdir -p gpurun_out/r6
timeout 2700 python -m pytest tests/ -q -m gpu > gpurun_out/r6/c9_tests.txt 2>&1
tail -15 gpurun_out/r6/c9_tests.txt
