#!/bin/bash
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_vanilla_refine_gpu.py tests/test_gaussian_model_gpu.py -x -q -m gpu > gpurun_out/r6/c2_tests.txt 2>&1
tail -15 gpurun_out/r6/c2_tests.txt
timeout 300 python profiles/probe_density_event.py > gpurun_out/r6/c2_event.txt 2>&1
head -70 gpurun_out/r6/c2_event.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r6/c2_bench.json 2> gpurun_out/r6/c2_bench.err
tail -3 gpurun_out/r6/c2_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/c2_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("fine_stage"), indent=1)[:3500])
PY
timeout 300 python profiles/probe_glue.py > gpurun_out/r6/c2_glue.txt 2>&1
tail -65 gpurun_out/r6/c2_glue.txt
