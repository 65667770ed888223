"""Analysis tool (test infrastructure, not product): (pixel, list entry) pair statistics of the bench scene on the CPU oracle.
    python -m tests.analysis.pair_stats [frame]
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(frame=10, n=2_000_000):
    from oracle import cpu_oracle as co
    from tests.test_parity_gpu import _bench_scene_case
    from tests.helpers import oracle_settings, oracle_scene
    here = os.path.dirname(os.path.abspath(__file__))
    os.makedirs(os.path.join(here, "_build"), exist_ok=True)
    so = os.path.join(here, "_build", "libpair_stats.so")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(here, "pair_stats.c"), "-lm"])
    lib = C.CDLL(so)
    case, sc = _bench_scene_case(n, frame=frame, actors=True)
    case["scales"], case["opacities"] = torch.exp(sc.log_scales), torch.sigmoid(sc.opacity_logits)
    case["rotations"] = torch.nn.functional.normalize(sc.quats, dim=1)
    S = oracle_settings(case)
    pre, b, img = co.forward(S, oracle_scene(case), case["flags"])
    out = np.zeros(32, np.float64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.pair_stats(case["W"], case["H"], p(b["ranges"]), p(b["ids"]), p(pre["means2D"]), p(pre["conic_opacity"]),
                   p(img["n_contrib"]), p(out))
    names = ["D", "tile_pairs", "quad_entries", "quad_pairs", "sub_entries", "sub_pairs", "hits", "quad_entries_with_hit",
             "rowspan_pairs", "row_box_pairs", "sub_pairs_own_depth", "half_pairs"]
    for k, v in zip(names, out):
        print(f"{k:24s} {v:14.0f}")
    for k, v in zip(["8x8", "8x4", "4x4", "8x2", "4x2", "8x1", "2x2"], out[16:23]):
        print(f"exact {k:19s} {v:14.0f}  ({v / out[3]:.3f} of quad_pairs)")
    print(f"tile entries whose exact footprint reaches the tile: {out[24]:.0f} of D = {out[0]:.0f} ({out[24] / out[0]:.3f}); in front of the tile's deepest contributor: {out[25]:.0f}")
    print("hit fraction of quadrant pairs", out[6] / out[3], " of sub-block pairs", out[6] / out[5], " of row spans", out[6] / out[8])


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
