"""Condense rocprofv3 --pmc counter_collection.csv files: per kernel (name matched by a regex) the mean counter values per launch; wave-state
counters are printed as fractions of SQ_WAVE_CYCLES.    python3 profiles/summarise_counters.py DIR REGEX"""
import collections
import csv
import glob
import re
import sys

out, pat = sys.argv[1], re.compile(sys.argv[2])
for f in sorted(glob.glob(f"{out}/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        m = pat.search(r["Kernel_Name"])
        if not m:
            continue
        k = m.group(0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        d = {a: b / n[(k, a)] for a, b in v.items()}
        if "SQ_WAVE_CYCLES" in d:
            wc = d["SQ_WAVE_CYCLES"]
            print(k, {a.replace("SQ_", ""): round(b / wc, 3) for a, b in d.items() if a != "SQ_WAVE_CYCLES"}, f"wave_cycles {wc:.3g}")
        else:
            print(k, {a.replace("SQ_", ""): f"{b:.4g}" for a, b in d.items()})
