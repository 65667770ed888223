"""Condense two rocprofv3 counter passes into profiles/rNN_pmc_hbm_traffic.csv.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -o fetch -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --eager --settle-ms 0 --repeats 0
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIR -o write -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --eager --settle-ms 0 --repeats 0
    python profiles/make_pmc_summary.py DIR > profiles/rNN_pmc_hbm_traffic.csv

FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of
the bytes of wide coalesced reads -> column fetch_MB_x2.  The radix kernels run on two very different sizes (the N Gaussians
of the depth sort, the D duplicates of the tile sort); they are split by grid size.
"""
import csv
import collections
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", name)[0]


def load(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        if k.startswith("k_radix"):
            k += "[N]" if int(r["Grid_Size"]) < 400_000 else "[D]"
        acc[k].append(float(r["Counter_Value"]))
    return acc


def main(d):
    f, w = load(f"{d}/fetch_counter_collection.csv"), load(f"{d}/write_counter_collection.csv")
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --eager --settle-ms 0 --repeats 0")
    print("# KiB per dispatch averaged over dispatches, in MB.  fetch_MB_x2 = gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md);")
    print("# WRITE_SIZE is exact for 16 B/lane stores and float atomics.  Infinity-Cache hits are counted: fabric traffic, an upper bound on HBM.")
    print("# k_radix_*[N]: passes of the Gaussian depth sort (grid over N), [D]: passes of the tile sort (grid over the duplicate capacity).")
    print("# per_iter: launches of the kernel per training iteration (calls / calls of k_preprocess, resp. of k_render_backward_q for backward kernels)")
    print("kernel,calls,fetch_MB_raw,fetch_MB_x2,write_MB,per_iter")
    n_fwd, n_bwd = max(len(f.get("k_preprocess", [])), 1), max(len(f.get("k_render_backward_q", [])), 1)
    for k in sorted(set(f) | set(w)):
        fm = sum(f.get(k, [0])) / max(len(f.get(k, [0])), 1) * 1024 / 1e6
        wm = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1) * 1024 / 1e6
        bwd = "backward" in k or k.endswith("_bwd") or k == "k_l1_loss"
        calls = len(f.get(k, []))
        print(f"{k},{calls},{fm:.1f},{2 * fm:.1f},{wm:.1f},{calls / (n_bwd if bwd else n_fwd):.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
