#!/bin/bash
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_vanilla_refine_gpu.py tests/test_gaussian_model_gpu.py tests/test_step_inputs_gpu.py tests/test_bench_multirank_gpu.py -x -q -m gpu > gpurun_out/r6/c3_tests.txt 2>&1
tail -8 gpurun_out/r6/c3_tests.txt
timeout 300 python profiles/probe_density_event.py > gpurun_out/r6/c3_event.txt 2>&1
head -45 gpurun_out/r6/c3_event.txt | cut -c1-150
timeout 600 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r6/c3_bench_config4.json 2> gpurun_out/r6/c3_bench_config4.err
tail -3 gpurun_out/r6/c3_bench_config4.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/c3_bench_config4.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("density_control_event"), indent=1)[:2500])
PY
timeout 300 python profiles/render_loop_trips.py > gpurun_out/r6/r06_render_loop_trips.txt 2> gpurun_out/r6/c3_trips.err
cat gpurun_out/r6/r06_render_loop_trips.txt; tail -3 gpurun_out/r6/c3_trips.err
bash profiles/ab_variants.sh base k7_noflush k7_noprologue base > gpurun_out/r6/c3_ablations.txt 2>&1
cat gpurun_out/r6/c3_ablations.txt
