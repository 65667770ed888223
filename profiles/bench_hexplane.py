"""Micro-benchmark of the fused HexPlane lookup at the reference's configuration: 32 channels,
resolution [64,64,64,25], multires [1,2,4,8] (arguments/gaussian_options.py:136-156), N points in the box; forward + backward
(gradients to every plane and to the points), HIP events, inputs resident in HBM.  The same computation written with
F.grid_sample exactly as scene/hexplane.py does (run on the GPU) is timed beside it.  Prints one JSON line.
    python profiles/bench_hexplane.py [N] > profiles/r01_hexplane_microbench.json"""
import json
import sys

import torch

sys.path.insert(0, ".")
import itertools  # noqa: E402

import torch.nn.functional as F  # noqa: E402

from emd_amd.hexplane import HexPlaneField  # noqa: E402


def grid_sample_formulation(pts, timestamps, aabb, planes):
    """The lookup as the reference's PyTorch code issues it (scene/hexplane.py:18-110,150-183): per scale six F.grid_sample
    launches, five products, then a concat -- the "what it replaces" timing."""
    q = torch.cat(((pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0, timestamps), dim=-1)
    outs = []
    for scale in planes:
        feat = 1.0
        for grid, pair in zip(scale, itertools.combinations(range(4), 2)):
            interp = F.grid_sample(grid, q[:, list(pair)].view(1, 1, -1, 2), align_corners=True, mode="bilinear", padding_mode="border")
            feat = feat * interp.view(grid.shape[1], -1).t()
        outs.append(feat)
    return torch.cat(outs, dim=-1)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
HIP_ONLY = "--hip-only" in sys.argv
dev = torch.device("cuda", 0)
cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [64, 64, 64, 25]}
field = HexPlaneField(1.6, cfg, [1, 2, 4, 8]).to(dev)
g = torch.Generator().manual_seed(0)
pts = ((torch.rand(N, 3, generator=g) * 3.2) - 1.6).to(dev).requires_grad_(True)
t = torch.full((N, 1), 0.37, device=dev)                    # one frame per step: every point carries the same time
gout = torch.randn(N, 128, generator=g).to(dev)


def run(fn, n):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        pts.grad = None
        for p in field.parameters():
            p.grad = None
        e[0].record()
        f = fn()
        e[1].record()
        f.backward(gout)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    return tf / n, tb / n


hip = lambda: field(pts, t)
ref = lambda: grid_sample_formulation(pts, t, field.aabb, [[p for p in gp] for gp in field.grids])
run(hip, 2)
hf, hb = run(hip, 10)
if HIP_ONLY:
    rf = rb = None
else:
    run(ref, 1)
    rf, rb = run(ref, 3)
print(json.dumps({"op": "HexPlane lookup, 4 scales x 6 planes x 32 channels, forward / backward (planes + points)", "N": N,
                  "hip_forward_ms": round(hf, 3), "hip_backward_ms": round(hb, 3), "grid_sample_forward_ms": None if rf is None else round(rf, 3),
                  "grid_sample_backward_ms": None if rb is None else round(rb, 3), "out_bytes": N * 128 * 4}))
