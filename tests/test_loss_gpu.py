"""-m gpu: the fused HIP image-loss tail (emd_image_loss, through the C ABI) against the golden vectors of the reference's
loss functions and against the CPU oracle at training size.  fp32: loss terms within 2e-6 relative (atomic block sums
reorder the additions), gradients within 1e-4 of the largest oracle entry (separable vs outer-product window)."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _close_grad(a, b, what):
    a, b = a.detach().cpu().numpy(), np.asarray(b)
    assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-12, (what, np.abs(a - b).max(), np.abs(b).max())


def test_image_loss_matches_reference_golden():
    from emd_amd.loss import image_loss
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "s3g_loss.npz"))
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    image, depth, weight = t("image").requires_grad_(True), t("depth").requires_grad_(True), t("weight").requires_grad_(True)
    sky = t("sky_mask").bool()
    lam = g["lambdas"]
    total, terms = image_loss(image, t("gt"), depth, t("gt_depth"), ~sky, weight, sky, float(lam[0]), float(lam[1]), float(lam[2]))
    total.backward()
    for k, ref in (("l1", "l1"), ("ssim", "ssim"), ("depth", "depth_l2"), ("sky", "sky")):
        np.testing.assert_allclose(terms[k].item(), g[ref], rtol=5e-6, err_msg=k)
    np.testing.assert_allclose(total.item(), g["total"], rtol=5e-6)
    _close_grad(image.grad, g["g_image"], "image")
    _close_grad(depth.grad, g["g_depth"], "depth")
    _close_grad(weight.grad, g["g_weight"], "weight")
    assert torch.all(depth.grad[:, ::3] == 0) and torch.all(weight.grad[0, 0, :5] == 0)


@pytest.mark.parametrize("H,W,with_depth,with_sky,lam_dssim", [(1066, 1600, True, True, 0.2), (123, 77, False, False, 0.2),
                                                               (64, 96, True, False, 0.0), (17, 19, False, True, 0.2)])
def test_image_loss_vs_oracle(H, W, with_depth, with_sky, lam_dssim):
    from emd_amd.loss import image_loss
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(H * 7 + W)
    gt = torch.rand(3, H, W, generator=gen)
    image = (gt + 0.2 * torch.randn(3, H, W, generator=gen)).clamp(0, 1)
    gt_depth = torch.rand(1, H, W, generator=gen) * 90.0 * (torch.rand(1, H, W, generator=gen) > 0.4)
    depth = (gt_depth + 2.0 * torch.randn(1, H, W, generator=gen)).abs()
    sky = torch.rand(1, H, W, generator=gen) < 0.3
    weight = torch.rand(1, H, W, generator=gen)
    kw = dict(lambda_dssim=lam_dssim, lambda_depth=0.5, lambda_sky=0.05)
    i0, d0, w0 = image.clone().requires_grad_(True), depth.clone().requires_grad_(True), weight.clone().requires_grad_(True)
    tot0, t0 = lo.loss_tail(i0, gt, d0 if with_depth else None, gt_depth if with_depth else None, (~sky).float() if with_depth else None,
                            w0 if with_sky else None, sky if with_sky else None, **kw)
    tot0.backward()
    i1, d1, w1 = image.to(dev).requires_grad_(True), depth.to(dev).requires_grad_(True), weight.to(dev).requires_grad_(True)
    tot1, t1 = image_loss(i1, gt.to(dev), d1 if with_depth else None, gt_depth.to(dev) if with_depth else None,
                          (~sky).to(dev) if with_depth else None, w1 if with_sky else None, sky.to(dev) if with_sky else None, **kw)
    (tot1 * 2.0).backward()                                   # a non-unit upstream gradient
    np.testing.assert_allclose(tot1.item(), tot0.item(), rtol=1e-5)
    for k in t0:
        np.testing.assert_allclose(t1[k].item(), t0[k].item(), rtol=1e-5, err_msg=k)
    _close_grad(i1.grad / 2.0, i0.grad.numpy(), "image")
    if with_depth:
        _close_grad(d1.grad / 2.0, d0.grad.numpy(), "depth")
    if with_sky:
        _close_grad(w1.grad / 2.0, w0.grad.numpy(), "weight")


@pytest.mark.parametrize("lam_dssim", [0.2, 0.0])
def test_sky_term_skipped_without_sky_pixels(lam_dssim):
    """S3Gaussian/train.py:360 applies the sky BCE only when the mask holds at least one sky pixel; and a weight image without a
    mask (load_sky_mask off, train.py:219) takes no sky term at all.  Loss and dL/dweight must follow (zero gradient)."""
    from emd_amd.loss import image_loss
    dev = torch.device("cuda", 0)
    H, W = 70, 90
    gen = torch.Generator().manual_seed(3)
    gt, image = torch.rand(3, H, W, generator=gen), torch.rand(3, H, W, generator=gen)
    weight = torch.rand(1, H, W, generator=gen)
    no_sky = torch.zeros(1, H, W, dtype=torch.bool)
    kw = dict(lambda_dssim=lam_dssim, lambda_depth=0.5, lambda_sky=0.05)
    i0, w0 = image.clone().requires_grad_(True), weight.clone().requires_grad_(True)
    tot0, t0 = lo.loss_tail(i0, gt, None, None, None, w0, no_sky, **kw)
    assert "sky" not in t0
    for mask in (no_sky.to(dev), None):
        i1, w1 = image.to(dev).requires_grad_(True), weight.to(dev).requires_grad_(True)
        tot1, t1 = image_loss(i1, gt.to(dev), None, None, None, w1, mask, **kw)
        tot1.backward()
        np.testing.assert_allclose(tot1.item(), tot0.item(), rtol=1e-5)
        assert t1["sky"].item() == 0.0
        assert w1.grad is None or float(w1.grad.abs().max()) == 0.0
    # one sky pixel switches the term on for the whole image
    one = no_sky.clone()
    one[0, 3, 4] = True
    i0, w0 = image.clone().requires_grad_(True), weight.clone().requires_grad_(True)
    tot0, t0 = lo.loss_tail(i0, gt, None, None, None, w0, one, **kw)
    tot0.backward()
    i1, w1 = image.to(dev).requires_grad_(True), weight.to(dev).requires_grad_(True)
    tot1, t1 = image_loss(i1, gt.to(dev), None, None, None, w1, one.to(dev), **kw)
    tot1.backward()
    np.testing.assert_allclose(t1["sky"].item(), t0["sky"].item(), rtol=1e-5)
    _close_grad(w1.grad, w0.grad.numpy(), "weight")


def test_abs_mean_regulariser_matches_torch():
    """mean |x| (the L1 residual regularisers of S3Gaussian/train.py:242-310) and its gradient under a non-trivial upstream gradient,
    odd sizes included."""
    import torch
    from emd_amd.model import abs_mean
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(4)
    for shape in ((1000, 16, 3), (777, 3), (5,), (4096, 1)):
        x = torch.randn(*shape, generator=g)
        x.view(-1)[::7] = 0.0                                       # sign(0) = 0
        a, b = x.clone().to(dev).requires_grad_(True), x.clone().double().requires_grad_(True)
        la, lb = 0.37 * abs_mean(a) + 1.0, 0.37 * b.abs().mean() + 1.0
        la.backward(); lb.backward()
        assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb))
        assert float((a.grad.double().cpu() - b.grad).abs().max()) <= 1e-6 * float(b.grad.abs().max())
