"""Micro-benchmark of the optimiser step on the reference's Gaussian parameter groups at N = 2 M
(xyz 3, f_dc 3, f_rest 45, opacity 1, scaling 3, rotation 4, embedding 4 floats per Gaussian = 504 MB; gaussian_model.py:188-199):
emd_amd.optim.Adam (one HIP launch) beside torch.optim.Adam as the reference constructs it (default: multi-tensor) and with
fused=True, HIP events, everything resident in HBM.  Algorithmic traffic: 28 B per element (read p, g, m, v; write p, m, v).
    python profiles/bench_adam.py > profiles/r01_adam_microbench.json"""
import json
import sys

import torch

sys.path.insert(0, ".")
from emd_amd.optim import Adam  # noqa: E402

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
shapes = {"xyz": (N, 3), "f_dc": (N, 1, 3), "f_rest": (N, 15, 3), "opacity": (N, 1), "scaling": (N, 3), "rotation": (N, 4), "embedding": (N, 4)}
elements = sum(int(torch.tensor(s).prod()) for s in shapes.values())


def run(make):
    ps = {k: torch.nn.Parameter(torch.randn(*s, device=dev)) for k, s in shapes.items()}
    opt = make([{"params": [p], "lr": 1e-3 * (i + 1), "name": k} for i, (k, p) in enumerate(ps.items())])
    for p in ps.values():
        p.grad = torch.randn_like(p)
    for _ in range(3):
        opt.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


hip = run(lambda g: Adam(g, lr=0.0, eps=1e-15))
stock = run(lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15))
fused = run(lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15, fused=True))
print(json.dumps({"op": "Adam step over the seven Gaussian parameter groups", "gaussians": N, "elements": elements, "hip_ms": round(hip, 4),
                  "alg_GBps": round(28 * elements / hip / 1e6, 1), "hbm_peak_GBps": 8000.0, "torch_adam_default_ms": round(stock, 4),
                  "torch_adam_fused_ms": round(fused, 4)}))
