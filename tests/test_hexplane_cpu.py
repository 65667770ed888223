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
