#!/bin/bash
mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6/c5_tests.txt 2>&1
tail -8 gpurun_out/r6/c5_tests.txt
timeout 600 bash profiles/ab_variants.sh base k7_row12 base k7_row12 > gpurun_out/r6/c5_row12.txt 2>&1
cat gpurun_out/r6/c5_row12.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r6/c5_bench_config4.json 2> gpurun_out/r6/c5_bench_config4.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/c5_bench_config4.json").read().strip().splitlines()[-1])
e=d["density_control_event"]; print(d["value"], {k:e[k] for k in ("event_ms","re_record_ms","threshold_quantile_ms","n_before","n_after")}, {k:e["first_event"][k] for k in ("event_ms","re_record_ms")})
PY
