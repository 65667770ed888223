"""Condense two rocprofv3 SQ counter passes into profiles/r01_pmc_valu.csv.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d DIR -o sq1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d DIR -o sq2 -- (same)
    python profiles/make_valu_summary.py DIR > profiles/r01_pmc_valu.csv
"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", n)[0]


def main(d):
    rows = {}
    for f in ("sq1", "sq2"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f"{d}/{f}_counter_collection.csv")):
            k = short(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            if k.startswith("k_radix"):
                k += "[N]" if int(r["Grid_Size"]) < 400_000 else "[D]"
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                acc[k]["dur_us"].append(dur * 1e6)
                acc[k]["valu_util"].append(float(r["Counter_Value"]) * 4 / (1024 * 2.4e9 * dur))
        for k in acc:
            rows.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in acc[k].items()})
    cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "dur_us", "valu_util"]
    print("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES / SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
    print("# (two passes) -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline; wave-level instruction counts per dispatch, averaged over dispatches.")
    print("# dur_us = duration of the same dispatches (kernel trace of the counter run); valu_util = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x dur),")
    print("# averaged per dispatch: the fraction of the fp32 vector issue slots the kernel fills (a wave64 VALU instruction occupies its SIMD for 4 cycles).")
    print("# A value slightly above 1 (K6) means the kernel issues a VALU instruction practically every cycle of every SIMD and the nominal")
    print("# 2.4 GHz x 4-cycle model is a little conservative; it is reported raw.")
    print("kernel," + ",".join(cols))
    for k in sorted(rows):
        print(k + "," + ",".join(f"{rows[k].get(c, 0):.4g}" for c in cols))


if __name__ == "__main__":
    main(sys.argv[1])
