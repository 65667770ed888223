"""Condense two rocprofv3 SQ counter passes into profiles/rNN_pmc_valu.csv.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d DIR -o sq1 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --eager --settle-ms 0 --repeats 0
    rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d DIR -o sq2 -- (same)
    python profiles/make_valu_summary.py DIR > profiles/r02_pmc_valu.csv

The last column prices every wave-level VALU instruction at the MEASURED issue time of a plain fp32 instruction with eight waves resident
(1.32 ns per SIMD, profiles/r02_issue_rate_microbench.txt); packed, DPP and cross-lane instructions take 2.0 ns and transcendentals 3.4 ns,
so for kernels rich in those (K7) the true utilisation is higher than this lower bound (bench.py's `roofline.issue` applies the mix).
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
                acc[k]["valu_util"].append(float(r["Counter_Value"]) * 1.32e-9 / (1024 * dur))
        for k in acc:
            rows.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in acc[k].items()})
    cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "dur_us", "valu_util"]
    print("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES / SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
    print("# (two passes) -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --eager --settle-ms 0 --repeats 0; wave-level instruction counts per dispatch, averaged over dispatches.")
    print("# dur_us = duration of the same dispatches (kernel trace of the counter run); valu_util = SQ_INSTS_VALU x 1.32 ns / (1024 SIMDs x dur):")
    print("# every instruction priced as a plain fp32 VALU instruction at its MEASURED issue time with 8 waves resident (2.35 cycles at the 1.78 GHz the")
    print("# chip holds under that load, profiles/r02_issue_rate_microbench.txt) -- a LOWER bound: v_pk_*, DPP and v_readlane take 2.0 ns, transcendentals 3.4 ns.")
    print("kernel," + ",".join(cols))
    for k in sorted(rows):
        print(k + "," + ",".join(f"{rows[k].get(c, 0):.4g}" for c in cols))


if __name__ == "__main__":
    main(sys.argv[1])
