"""Analysis tool (test infrastructure): simulate K7's per-sub-block queue policy on the bench scene.
    python -m tests.analysis.queue_sim [frame]"""
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
    so = os.path.join(here, "_build", "libqueue_sim.so")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(here, "queue_sim.c"), "-lm"])
    lib = C.CDLL(so)
    case, sc = _bench_scene_case(n, frame=frame, actors=True)
    case["scales"], case["opacities"] = torch.exp(sc.log_scales), torch.sigmoid(sc.opacity_logits)
    case["rotations"] = torch.nn.functional.normalize(sc.quats, dim=1)
    S = oracle_settings(case)
    pre, b, img = co.forward(S, oracle_scene(case), case["flags"])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    print("NS  R    steps   batches  forced  final_partial  kept   fill   pair_iters(M)  est_instr(1e8)")
    for NS, per in ((1, 32), (2, 16), (4, 8)):
        for R in ((64,) if NS == 1 else (96, 128, 160, 192, 256)):
            out = np.zeros(8, np.float64)
            lib.queue_sim(case["W"], case["H"], p(b["ranges"]), p(b["ids"]), p(pre["means2D"]), p(pre["conic_opacity"]),
                          p(img["n_contrib"]), R, NS, p(out))
            steps, batches, forced, kept, fill, finalp = out[0], out[1], out[2], out[3], out[4], out[5]
            iters = batches * per
            # instruction model: ~115 per pair-iteration, ~100 per scan step (+60 for the 4-bit exact mask per 64 kept), ~60 per batch
            # (record loads, LDS accumulate), ~150 per 64 flushed rows
            instr = iters * 115 + steps * 100 + (kept / 64) * (60 if NS > 1 else 0) + batches * (60 if NS > 1 else 30) + kept / 64 * 150
            print(f"{NS:2d} {R:4d} {steps:8.0f} {batches:9.0f} {forced:7.0f} {finalp:9.0f} {kept:10.0f} {fill / batches / 64:6.3f} {iters / 1e6:10.2f} {instr / 1e8:10.3f}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
