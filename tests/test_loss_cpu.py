"""CPU: the loss oracle against values and gradients captured from the reference's own loss functions
(tests/golden/s3g_loss.npz: utils/loss_utils.py l1_loss / ssim / compute_depth and the sky BCE of train.py)."""
import os

import numpy as np
import torch

from oracle import loss_oracle as lo

G = os.path.join(os.path.dirname(__file__), "golden")


def test_loss_tail_matches_reference_values_and_gradients():
    g = np.load(os.path.join(G, "s3g_loss.npz"))
    t = lambda k: torch.from_numpy(g[k])
    image, depth, weight = t("image").requires_grad_(True), t("depth").requires_grad_(True), t("weight").requires_grad_(True)
    sky = t("sky_mask").bool()
    lam = g["lambdas"]
    total, terms = lo.loss_tail(image, t("gt"), depth, t("gt_depth"), (~sky).float(), weight, sky, float(lam[0]), float(lam[1]), float(lam[2]))
    total.backward()
    np.testing.assert_allclose(terms["l1"].item(), g["l1"], rtol=1e-6)
    np.testing.assert_allclose(terms["ssim"].item(), g["ssim"], rtol=1e-6)
    np.testing.assert_allclose(terms["depth"].item(), g["depth_l2"], rtol=1e-6)
    np.testing.assert_allclose(terms["sky"].item(), g["sky"], rtol=1e-6)
    np.testing.assert_allclose(total.item(), g["total"], rtol=1e-6)
    np.testing.assert_allclose(image.grad.numpy(), g["g_image"], atol=1e-9, rtol=1e-5)
    np.testing.assert_allclose(depth.grad.numpy(), g["g_depth"], atol=1e-9, rtol=1e-5)
    np.testing.assert_allclose(weight.grad.numpy(), g["g_weight"], atol=1e-9, rtol=1e-5)
    # edge cases the fixture contains: no lidar return and beyond max depth -> no gradient; clamped weights -> no gradient
    assert np.all(g["g_depth"][:, ::3] == 0) and np.all(g["g_depth"][:, :, :4] == 0)
    assert np.all(g["g_weight"][0, 0, :5] == 0) and np.all(g["g_weight"][0, 1, :5] == 0)
