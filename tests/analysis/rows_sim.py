"""Analysis tool: simulate the K7 'rows' formulation (16-entry batches per DPP row / 4x4 sub-block) on the bench scene.
    python -m tests.analysis.rows_sim [frame]"""
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
    so = os.path.join(here, "_build", "librows_sim.so")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(here, "rows_sim.c"), "-lm"])
    lib = C.CDLL(so)
    case, sc = _bench_scene_case(n, frame=frame, actors=True)
    case["scales"], case["opacities"] = torch.exp(sc.log_scales), torch.sigmoid(sc.opacity_logits)
    case["rotations"] = torch.nn.functional.normalize(sc.quats, dim=1)
    S = oracle_settings(case)
    pre, b, img = co.forward(S, oracle_scene(case), case["flags"])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    print("R  STEP  steps   rounds  row_batches  kept  row_fill  forced  waves   pair_iters(M)")
    for R, STEP in ((80, 32), (96, 32), (96, 64), (128, 64), (160, 64), (256, 64)):
        out = np.zeros(8, np.float64)
        lib.rows_sim(case["W"], case["H"], p(b["ranges"]), p(b["ids"]), p(pre["means2D"]), p(pre["conic_opacity"]), p(img["n_contrib"]), R, STEP, p(out))
        print(f"{R:4d} {STEP:3d} {out[0]:9.0f} {out[1]:9.0f} {out[2]:9.0f} {out[3]:10.0f} {out[4] / max(out[2], 1) / 16:6.3f} {out[5]:8.0f} {out[6]:8.0f} {out[1] * 8 / 1e6:8.3f}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
