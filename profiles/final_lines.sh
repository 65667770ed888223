#!/bin/bash
# the JSON lines of the final build, one gpurun call:  bash profiles/final_lines.sh r06
R=${1:-r06}; O=gpurun_out/${R}_final; mkdir -p $O
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench_line_driver_flags.json 2> $O/err1
timeout 600 python3 bench.py > $O/${R}_bench_line.json 2> $O/err2
timeout 600 python3 bench.py --config 3 --no-cpu-baseline > $O/${R}_bench_line_config3.json 2> $O/err3
timeout 600 python3 bench.py --config 4 --no-cpu-baseline > $O/${R}_bench_line_config4.json 2> $O/err4
timeout 600 python3 bench.py --config 1 --no-cpu-baseline > $O/${R}_bench_line_config1.json 2> $O/err5
timeout 600 python3 bench.py --config 0 > $O/${R}_bench_line_config0.json 2> $O/err6
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["value"],2), d["unit"], round(d["ms_per_step"],4), (d.get("fine_stage") or {}).get("ms_per_step"), {k:(d.get("density_control_event") or {}).get(k) for k in ("event_ms","re_record_ms")})
PY
