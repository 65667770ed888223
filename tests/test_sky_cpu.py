"""CPU: the sky oracle against the golden vectors captured from the reference's SkyCubeMap / EnvLight / render() blend
(tests/gen_golden.py, nvdiffrast's dr.texture replaced by the oracle lookup), and the size-independent properties of the
cube lookup itself (the part of the path whose parity is unpinned: nvdiffrast is absent from the reference tree)."""
import os

import numpy as np
import torch

from oracle import sky_oracle as so

G = os.path.join(os.path.dirname(__file__), "golden")


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_rays_mask_clamp_layout_blend_match_reference():
    g = np.load(os.path.join(G, "s3g_sky.npz"))
    H, W = int(g["H"]), int(g["W"])
    w2c = _t(g["world_view_transform"]).T
    rays = so.rays(H, W, _t(g["K"]), w2c[:3, :3], w2c[:3, 3])
    np.testing.assert_allclose(rays.numpy(), g["rays"], atol=2e-6)
    for tag, fill in (("white", 1.0), ("black", 0.0)):
        np.testing.assert_allclose(_t(g[f"{tag}_dirs_all"]).numpy(), g["rays"], atol=0)      # what the reference handed to dr.texture
        cube, acc = _t(g[f"{tag}_cube"]), _t(g[f"{tag}_acc"])
        np.testing.assert_allclose(so.sky_s3g(cube, rays).numpy(), g[f"{tag}_sky_all"], atol=1e-5)
        sky = so.sky_s3g(cube, rays, acc, fill=fill)
        np.testing.assert_allclose(sky.numpy(), g[f"{tag}_sky_masked"], atol=1e-5)
        assert int(((1 - acc[0]) > 1e-3).sum()) == int(g[f"{tag}_n_masked_dirs"])
        assert np.all(sky.numpy()[:, :10] == fill)                                           # fully covered rows are not sampled
        np.testing.assert_allclose(so.blend_s3g(_t(g[f"{tag}_render"]), acc, sky).numpy(), g[f"{tag}_blended"], atol=1e-5)
        assert g[f"{tag}_sky_all"].min() >= 0 and g[f"{tag}_sky_all"].max() <= 1


def test_envlight_matches_reference():
    g = np.load(os.path.join(G, "or_envlight.npz"))
    d = _t(g["viewdirs"]).reshape(-1, 3) @ _t(g["to_opengl"]).T
    np.testing.assert_allclose(d.numpy(), g["lookup_dirs"], atol=1e-7)
    light = so.cube_lookup(_t(g["base"]), d).reshape(g["light"].shape)
    np.testing.assert_allclose(light.numpy(), g["light"], atol=1e-6)
    np.testing.assert_allclose(so.blend_add(_t(g["rgb"]), _t(g["opacity"]), light).numpy(), g["blended"], atol=1e-6)


def test_cube_lookup_properties():
    torch.manual_seed(0)
    res = 16
    cube = torch.rand(6, res, res, 3)
    d = torch.randn(50000, 3)
    # constant texture -> constant colour (weights sum to one everywhere, corners included)
    np.testing.assert_allclose(so.cube_lookup(torch.full((6, res, res, 3), 0.37), d).numpy(), 0.37, atol=1e-6)
    # the six axes hit the centre of faces +x,-x,+y,-y,+z,-z
    for f, ax in enumerate([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]):
        v = so.cube_lookup(cube, torch.tensor([ax], dtype=torch.float32))[0]
        np.testing.assert_allclose(v.numpy(), cube[f, res // 2 - 1:res // 2 + 1, res // 2 - 1:res // 2 + 1].mean((0, 1)).numpy(), atol=1e-6)
    # OpenGL orientation: on +z, u grows with x and v grows with -y
    face, u, v = so.index_cube(torch.tensor([[0.5, 0.0, 1.0], [0.0, 0.5, 1.0]]))
    assert face.tolist() == [4, 4] and u[0] > 0.5 and abs(v[0] - 0.5) < 1e-6 and v[1] < 0.5
    # scale invariance
    np.testing.assert_allclose(so.cube_lookup(cube, d * 7.5).numpy(), so.cube_lookup(cube, d).numpy(), atol=1e-5)
    # seamless: moving a direction by 1e-5 never changes the colour by more than the texture gradient allows,
    # in particular not across the 12 edges (a seam would jump by O(1))
    out = so.cube_lookup(cube, d)
    d2 = d / d.norm(dim=1, keepdim=True)
    step = 1e-4 * torch.randn_like(d2)
    jump = (so.cube_lookup(cube, d2 + step) - out).abs().max().item()
    assert jump < 0.02, jump
    # directions ON edges and corners are finite and inside the range of the texture
    e = torch.tensor([[1.0, 1.0, 0.3], [1.0, -1.0, -0.2], [-1.0, 0.4, 1.0], [1.0, 1.0, 1.0], [-1.0, 1.0, -1.0], [1.0, -1.0, 1.0]])
    v = so.cube_lookup(cube, e)
    assert torch.isfinite(v).all() and v.min() >= cube.min() and v.max() <= cube.max()
    idx, w = so.cube_taps(e, res)
    np.testing.assert_allclose(w.sum(1).numpy(), 1.0, atol=1e-6)
    assert (w[3:] == 0).sum() >= 3            # each corner direction drops its fourth tap
