"""Micro-benchmark of the sky cube-map + blend launch pair at the reference's sizes: 1066 x 1600 view,
6 x 1024 x 1024 x 3 cube map.  HIP events on the launch stream, inputs resident in HBM.  Prints one JSON line.
Algorithmic bytes per pixel: forward 4 (acc) + 12 (fg) + 12 (sky) + 12 (out) = 40; backward 12 (dL/dout) + 4 + 12 (acc, fg re-read)
+ 12 (dL/dfg) + 4 (dL/dacc) = 44 (+ the 75 MB memset of dL/dcube); texel gathers / atomics hit L2 (the view covers ~1.7 M texels).
    python profiles/bench_sky.py > profiles/r01_sky_microbench.json"""
import json
import sys
import types

import numpy as np
import torch

sys.path.insert(0, ".")
from emd_amd.sky import SkyCubeMap, composite_s3g  # noqa: E402

dev = torch.device("cuda", 0)
H, W, RES = 1066, 1600, 1024
K = torch.tensor([[1700.0, 0, 800.0], [0, 1700.0, 533.0], [0, 0, 1]], device=dev)
wvt = torch.eye(4)
wvt[:3, :3] = torch.tensor([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]])
cam = types.SimpleNamespace(image_height=H, image_width=W, intrinsic=K, world_view_transform=wvt.to(dev))
m = SkyCubeMap(types.SimpleNamespace(sky_resolution=RES, sky_white_background=False, white_background=False), device=dev)
g = torch.Generator().manual_seed(0)
m.sky_cube_map.data = torch.rand(6, RES, RES, 3, generator=g).to(dev)
acc = (torch.rand(1, H, W, generator=g) * 0.9).to(dev).requires_grad_(True)
render = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
gout = torch.rand(3, H, W, generator=g).to(dev)


def run(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        m.sky_cube_map.grad = None; acc.grad = None; render.grad = None
        ev[0].record()
        out, sky = composite_s3g(m, cam, render, acc)
        ev[1].record()
        out.backward(gout)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    return tf / n, tb / n


run(5)
f_ms, b_ms = run(50)
P = H * W
print(json.dumps({"op": "sky cube-map lookup + S3G blend, fwd and bwd (incl. autograd glue and the dL/dcube memset)", "H": H, "W": W,
                  "cube_resolution": RES, "forward_ms": round(f_ms, 4), "backward_ms": round(b_ms, 4),
                  "forward_alg_GBps": round(40 * P / f_ms / 1e6, 1), "backward_alg_GBps": round((44 * P + 6 * RES * RES * 12) / b_ms / 1e6, 1),
                  "hbm_peak_GBps": 8000.0}))
