#!/bin/bash
# A/B of library builds by KERNEL time (rocprofv3 --kernel-trace --stats over profiles/bench_hexplane.py, 2 M points), inside one gpurun call:
#   bash profiles/ab_hexplane_kernels.sh NAME...        ("base" = the in-tree library; the micro-benchmark's own forward figure includes host time)
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != "base" ]; then export EMD_LIB_PATH=$PWD/emd_amd/csrc/variants/lib_$v.so; else unset EMD_LIB_PATH; fi
  O=$PWD/gpurun_out/abk_$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 profiles/bench_hexplane.py 2000000 --hip-only > /dev/null 2> $O/err
  python3 - "$v" "$O" <<'PY'
import csv, sys
v, o = sys.argv[1], sys.argv[2]
rows = {r['Name']: float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(o + '/s_kernel_stats.csv')) if 'hexplane' in r['Name']}
print(v, ' '.join(f"{k.split('k_hexplane_')[1].split('(')[0][:14]}={t:.1f}us" for k, t in sorted(rows.items())))
PY
  rm -rf $O
done
