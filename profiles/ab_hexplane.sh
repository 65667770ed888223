#!/bin/bash
# A/B of library builds on the HexPlane micro-benchmark (profiles/bench_hexplane.py 2 M points, HIP only), inside one gpurun call: bash profiles/ab_hexplane.sh NAME...
for v in "$@"; do
  if [ "$v" != "base" ]; then export EMD_LIB_PATH=$PWD/emd_amd/csrc/variants/lib_$v.so; else unset EMD_LIB_PATH; fi
  python3 profiles/bench_hexplane.py 2000000 --hip-only 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'fwd', d['hip_forward_ms'], 'bwd', d['hip_backward_ms'])"
done
