#!/bin/bash
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 EMD_DP_FORCE=1 EMD_DP_INIT_WORLD1=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout 300 python bench.py --gpus 1 --factored-sh --steps 4 --warmup 2 --no-cpu-baseline --gaussians 60000 --height 128 --width 192 2>/dev/null | grep '^{' | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['exchange'])"
timeout 300 python bench.py --gpus 1 --factored-sh --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['exchange'])"
