"""Micro-benchmark of the fused image-loss tail at 1066 x 1600: L1 + depth L2 + D-SSIM + sky BCE with
all gradients, HIP events on the launch stream, inputs resident in HBM; the same loss in plain PyTorch (the reference's
formulas, on the GPU) is timed beside it.  Prints one JSON line.
Algorithmic bytes per pixel: pointwise 24 (image, gt) + 12 (depth, gt, mask) + 5 (weight, sky) + 12 + 4 + 4 (gradients) = 61;
SSIM forward 24 + 36 (three derivative maps x 3 channels); SSIM backward 36 + 24 + 24 (gradient read-modify-write) + 8 = 92.
    python profiles/bench_loss.py > profiles/r01_loss_microbench.json"""
import json
import sys

import torch

sys.path.insert(0, ".")
from emd_amd.loss import image_loss  # noqa: E402
from oracle import loss_oracle as lo  # noqa: E402  (the reference's formulas, run on the GPU as the "what it replaces" timing)

dev = torch.device("cuda", 0)
H, W = 1066, 1600
g = torch.Generator().manual_seed(0)
gt = torch.rand(3, H, W, generator=g).to(dev)
image = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
gt_depth = (torch.rand(1, H, W, generator=g) * 90).to(dev)
depth = (torch.rand(1, H, W, generator=g) * 90).to(dev).requires_grad_(True)
sky = (torch.rand(1, H, W, generator=g) < 0.3).to(dev)
weight = torch.rand(1, H, W, generator=g).to(dev).requires_grad_(True)
mask = (~sky).float()


def run(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = 0.0
    for _ in range(n):
        image.grad = depth.grad = weight.grad = None
        e0.record()
        loss = fn()
        loss.backward()
        e1.record()
        torch.cuda.synchronize()
        t += e0.elapsed_time(e1)
    return t / n


hip = lambda: image_loss(image, gt, depth, gt_depth, mask, weight, sky)[0]


def torch_ref():
    win = lo.window_1d().to(dev)
    lo.window_1d = lambda *a, **k: win.cpu()          # oracle builds the window on CPU; keep its formulas, move inputs
    return lo.loss_tail(image, gt, depth, gt_depth, mask, weight, sky)[0]


run(hip, 5)
t_hip = run(hip, 50)
try:
    import oracle.loss_oracle as _lo
    _orig = _lo.ssim

    def ssim_dev(a, b, size=11):
        C = a.shape[-3]
        w1 = _lo.window_1d(size).unsqueeze(1)
        win = w1.mm(w1.t()).float()[None, None].expand(C, 1, size, size).contiguous().to(a.device)
        import torch.nn.functional as F
        x, y = a[None], b[None]
        conv = lambda t: F.conv2d(t, win, padding=size // 2, groups=C)
        mu1, mu2 = conv(x), conv(y)
        s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
        m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
        return m.mean()
    _lo.ssim = ssim_dev
    ref = lambda: _lo.loss_tail(image, gt, depth, gt_depth, mask, weight, sky)[0]
    run(ref, 3)
    t_ref = run(ref, 10)
except Exception as e:      # MIOpen may not serve the grouped 11x11 convolution
    t_ref = None
P = H * W
print(json.dumps({"op": "image-loss tail: L1 + depth L2 + D-SSIM 11x11 + sky BCE, values and all gradients (autograd glue included)",
                  "H": H, "W": W, "hip_ms": round(t_hip, 4), "alg_GBps": round((61 + 60 + 92) * P / t_hip / 1e6, 1),
                  "pytorch_same_formulas_on_gpu_ms": None if t_ref is None else round(t_ref, 3), "hbm_peak_GBps": 8000.0}))
