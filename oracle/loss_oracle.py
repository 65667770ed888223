"""CPU restatement of the image-loss tail of the training step.  TEST INFRASTRUCTURE ONLY.

Follows S3Gaussian/train.py:226-363 and S3Gaussian/utils/loss_utils.py:
    loss = l1_loss(image, gt)                                            loss_utils.py:50-51, train.py:226
         + lambda_depth * compute_depth("l2", depth * mask, gt_depth * mask)   loss_utils.py:21-45, train.py:346-349
         + lambda_dssim * (1 - ssim(image, gt))                          loss_utils.py:56-98 (11x11 Gaussian window, sigma 1.5,
                                                                          zero padding, C1 = 0.01^2, C2 = 0.03^2), train.py:351-355
         + lambda_sky * mean(where(sky, -log(1 - w), -log(w))), w = clamp(weight, 1e-6, 1 - 1e-6)     train.py:357-361
Gradients come from torch autograd on these formulas.  Pinned by tests/golden/s3g_loss.npz (the reference's own functions,
imported on CPU, values and gradients)."""
import math

import torch
import torch.nn.functional as F


def window_1d(size=11, sigma=1.5):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])
    return g / g.sum()


def ssim(img1, img2, size=11):
    C = img1.shape[-3]
    w1 = window_1d(size).unsqueeze(1)
    win = w1.mm(w1.t()).float()[None, None].expand(C, 1, size, size).contiguous()
    x, y = img1.reshape(-1, C, *img1.shape[-2:]), img2.reshape(-1, C, *img2.shape[-2:])
    conv = lambda t: F.conv2d(t, win, padding=size // 2, groups=C)
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return m.mean()


def depth_l2(depth, gt_depth, mask, max_depth=80.0):
    p, g = (depth * mask).squeeze(), (gt_depth * mask).squeeze()
    valid = (g > 0.01) & (g < max_depth)
    p = torch.clamp(p[valid] / max_depth, 0.0, 1.0)
    g = torch.clamp(g[valid] / max_depth, 0.0, 1.0)
    return ((p - g) ** 2).mean()


def sky_bce(weight, sky_mask):
    w = torch.clamp(weight, min=1e-6, max=1.0 - 1e-6)
    return torch.where(sky_mask, -torch.log(1 - w), -torch.log(w)).mean()


def loss_tail(image, gt, depth=None, gt_depth=None, mask=None, weight=None, sky_mask=None, lambda_dssim=0.2, lambda_depth=0.5,
              lambda_sky=0.05):
    """-> (total, dict of the four terms); terms whose inputs are None are skipped, as train.py does."""
    l1 = (image - gt).abs().mean()
    terms = {"l1": l1}
    total = l1
    if depth is not None and lambda_depth != 0:
        m = mask if mask is not None else torch.ones_like(depth)
        terms["depth"] = depth_l2(depth, gt_depth, m)
        total = total + lambda_depth * terms["depth"]
    if lambda_dssim != 0:
        terms["ssim"] = ssim(image, gt)
        total = total + lambda_dssim * (1.0 - terms["ssim"])
    # train.py:360: `if args.lambda_sky > 0 and sky_mask is not None and sky_mask.sum() > 0`
    if weight is not None and sky_mask is not None and lambda_sky > 0 and bool(sky_mask.sum() > 0):
        terms["sky"] = sky_bce(weight, sky_mask)
        total = total + lambda_sky * terms["sky"]
    return total, terms
