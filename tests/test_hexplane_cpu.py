"""CPU: the HexPlane oracle against values and gradients produced by the reference's own HexPlaneField
(tests/golden/s3g_hexplane.npz, S3Gaussian/scene/hexplane.py imported on CPU)."""
import os

import numpy as np
import torch

from oracle import hexplane_oracle as ho

G = os.path.join(os.path.dirname(__file__), "golden")


def load_case(req_grad=True):
    g = np.load(os.path.join(G, "s3g_hexplane.npz"))
    planes = [[torch.from_numpy(g[f"plane_{s}_{p}"]).clone().requires_grad_(req_grad) for p in range(6)] for s in range(len(g["multires"]))]
    return g, planes


def test_hexplane_oracle_matches_reference_values_and_gradients():
    g, planes = load_case()
    pts = torch.from_numpy(g["pts"]).requires_grad_(True)
    feat = ho.hexplane_features(pts, torch.from_numpy(g["times"]), torch.from_numpy(g["aabb"]), planes)
    np.testing.assert_allclose(feat.detach().numpy(), g["feat"], rtol=1e-6, atol=1e-7)
    (feat * torch.from_numpy(g["gout"])).sum().backward()
    np.testing.assert_allclose(pts.grad.numpy(), g["g_pts"], rtol=1e-5, atol=1e-6)
    for s, sc in enumerate(planes):
        for p, pl in enumerate(sc):
            np.testing.assert_allclose(pl.grad.numpy(), g[f"g_plane_{s}_{p}"], rtol=1e-5, atol=1e-6, err_msg=f"plane {s} {p}")
    # the fixture exercises the border: points outside the box and times outside [-1, 1]
    q = (g["pts"] - g["aabb"][0]) * (2.0 / (g["aabb"][1] - g["aabb"][0])) - 1.0
    assert (np.abs(q) > 1).any() and (np.abs(g["times"]) > 1).any()


def test_visiting_orders_are_permutations_with_inverses_and_defer_the_fine_scales():
    """Host logic of the aggregating backward's orders (emd_amd.hexplane.VisitingOrders, EmdHexGrads.order2d / pos2d / defer_mask):
    every order is a permutation, pos2d is its inverse, consecutive points of a plane order are close in that plane's two coordinates,
    and the scales a 256-point run of the main kernel overflows the 7 x 7 windows of are the deferred ones (reference configuration: 128, 256 and 512 at 2 M points)."""
    from emd_amd.hexplane import VisitingOrders, morton_order, plane_order
    g = torch.Generator().manual_seed(0)
    N = 50_000
    aabb = torch.tensor([[1.6, 1.6, 1.6], [-1.6, -1.6, -1.6]])
    pts = torch.rand(N, 3, generator=g) * 3.2 - 1.6
    res = [[64, 64, 64, 25], [128, 128, 128, 25], [256, 256, 256, 25], [512, 512, 512, 25]]
    vo = VisitingOrders.build(pts, aabb, res)
    assert sorted(vo.order.tolist()) == list(range(N)) and torch.equal(vo.order, morton_order(pts, aabb))
    assert vo.defer_mask == 0b1111                               # 50 k points: a run spans 0.29 of the box, 19 cells even at resolution 64 (> the 7-cell window)
    for k, (ax, ay) in enumerate(((0, 1), (0, 2), (1, 2))):
        o, inv = vo.order2d[k].long(), vo.pos2d[k].long()
        assert sorted(o.tolist()) == list(range(N))
        assert torch.equal(inv[o], torch.arange(N)) and torch.equal(o[inv], torch.arange(N))
        assert torch.equal(vo.order2d[k], plane_order(pts, aabb, ax, ay))
        run = pts[o][: (N // 256) * 256].reshape(-1, 256, 3)
        ext = (run.max(dim=1)[0] - run.min(dim=1)[0]) / 3.2     # extent of a run as a fraction of the box
        other = 3 - ax - ay
        assert float(ext[:, ax].median()) < 0.15 and float(ext[:, ay].median()) < 0.15 and float(ext[:, other].median()) > 0.9
    # the mask follows N: at 2 M points resolution 64 stays in the main kernel's windows (5.5 cells), 128 (11 cells) and finer are deferred,
    # and a small cloud defers nothing at coarse resolutions
    big = VisitingOrders.build(torch.rand(2_000_000, 3, generator=g) * 3.2 - 1.6, aabb, res)
    assert big.defer_mask == 0b1110
    small = VisitingOrders.build(pts[:300], aabb, [[4, 4, 4, 2], [6, 6, 6, 2]])
    assert small.defer_mask == 0 and small.order2d is None
