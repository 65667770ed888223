"""-m gpu: the HIP sky cube-map path (through the C ABI, emd_amd/sky.py) against the CPU oracle and the golden vectors
captured from the reference's SkyCubeMap / EnvLight.  Floating point: colours within 2e-5 of the oracle (fp32: the bilinear
weights are fractions of u * res, so their rounding error grows with res -- ~res * 2^-23; the face / tap decisions are
identical), texture gradients within 1e-4 of the largest entry (float atomics reorder the sums)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import sky_oracle as so

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5


def _cam(g, dev):
    return types.SimpleNamespace(image_height=int(g["H"]), image_width=int(g["W"]), intrinsic=torch.from_numpy(g["K"]).to(dev),
                                 world_view_transform=torch.from_numpy(g["world_view_transform"]).to(dev))


def test_skycubemap_matches_reference_golden():
    from emd_amd.sky import SkyCubeMap, composite_s3g
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "s3g_sky.npz"))
    cam = _cam(g, dev)
    for tag, white in (("white", True), ("black", False)):
        cfg = types.SimpleNamespace(sky_resolution=16, sky_white_background=white, white_background=False)
        m = SkyCubeMap(cfg, device=dev)
        m.sky_cube_map.data = torch.from_numpy(g[f"{tag}_cube"]).to(dev)
        acc = torch.from_numpy(g[f"{tag}_acc"]).to(dev)
        np.testing.assert_allclose(m(cam).detach().cpu().numpy(), g[f"{tag}_sky_all"], atol=1e-5)
        np.testing.assert_allclose(m(cam, acc=acc).detach().cpu().numpy(), g[f"{tag}_sky_masked"], atol=1e-5)
        out, sky = composite_s3g(m, cam, torch.from_numpy(g[f"{tag}_render"]).to(dev), acc)
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"{tag}_blended"], atol=1e-5)
        np.testing.assert_allclose(sky.detach().cpu().numpy(), g[f"{tag}_sky_masked"], atol=1e-5)


def test_envlight_matches_reference_golden():
    from emd_amd.sky import EnvLight, composite_add
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "or_envlight.npz"))
    env = EnvLight(resolution=8, device=dev)
    env.base.data = torch.from_numpy(g["base"]).to(dev)
    infos = {"viewdirs": torch.from_numpy(g["viewdirs"]).to(dev)}
    np.testing.assert_allclose(env(infos).detach().cpu().numpy(), g["light"], atol=1e-6)
    rgb, sky = composite_add(env, infos, torch.from_numpy(g["rgb"]).to(dev), torch.from_numpy(g["opacity"]).to(dev))
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), g["blended"], atol=1e-6)


@pytest.mark.parametrize("res,P,seed", [(8, 20000, 0), (64, 200000, 1), (33, 50000, 2)])
def test_lookup_forward_backward_vs_oracle(res, P, seed):
    from emd_amd.sky import EnvLight, composite_add
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(seed)
    base = torch.rand(6, res, res, 3, generator=gen)
    dirs = torch.randn(P, 3, generator=gen)
    dirs[:64] = torch.tensor([1.0, 1.0, 1.0]) * torch.sign(torch.randn(64, 3, generator=gen))        # cube corners
    dirs[64:256, 0] = dirs[64:256, 1].abs()                                                           # on an edge
    rgb, op = torch.rand(P, 3, generator=gen), torch.rand(P, 1, generator=gen)
    g_out, g_sky = torch.randn(P, 3, generator=gen), torch.randn(P, 3, generator=gen)
    # oracle (torch autograd on CPU); EnvLight rotates to OpenGL axes first, undo that so both see `dirs`
    b0 = base.clone().requires_grad_(True); r0 = rgb.clone().requires_grad_(True); o0 = op.clone().requires_grad_(True)
    sky0 = so.cube_lookup(b0, dirs)
    out0 = so.blend_add(r0, o0, sky0)
    ((out0 * g_out).sum() + (sky0 * g_sky).sum()).backward()
    env = EnvLight(resolution=res, device=dev)
    env.base.data = base.to(dev)
    r1 = rgb.to(dev).requires_grad_(True); o1 = op.to(dev).requires_grad_(True)
    infos = {"viewdirs": (dirs @ env.to_opengl.cpu()).to(dev)}          # (d M) M^T = d for the orthonormal axis swap
    out1, sky1 = composite_add(env, infos, r1, o1)
    ((out1 * g_out.to(dev)).sum() + (sky1 * g_sky.to(dev)).sum()).backward()
    np.testing.assert_allclose(sky1.detach().cpu().numpy(), sky0.detach().numpy(), atol=TOL)
    np.testing.assert_allclose(out1.detach().cpu().numpy(), out0.detach().numpy(), atol=2 * TOL)
    gb0, gb1 = b0.grad.numpy(), env.base.grad.cpu().numpy()
    assert np.abs(gb1 - gb0).max() <= 1e-4 * np.abs(gb0).max(), np.abs(gb1 - gb0).max()
    np.testing.assert_allclose(r1.grad.cpu().numpy(), r0.grad.numpy(), atol=1e-6)
    np.testing.assert_allclose(o1.grad.cpu().numpy(), o0.grad.numpy(), atol=1e-4)   # -sum_c g_c sky_c with |g| up to ~4


def test_full_size_properties():
    """1066 x 1600 view, 1024^2 faces (the reference's sizes): colours stay inside the texture's range, a constant map
    gives a constant image, and the texture gradient conserves mass (bilinear weights sum to one)."""
    from emd_amd.sky import SkyCubeMap, composite_s3g
    dev = torch.device("cuda", 0)
    H, W = 1066, 1600
    K = torch.tensor([[1700.0, 0, 800.0], [0, 1700.0, 533.0], [0, 0, 1]], device=dev)
    yaw = 0.3
    R = torch.tensor([[np.sin(yaw), -np.cos(yaw), 0], [0, 0, -1], [np.cos(yaw), np.sin(yaw), 0]], dtype=torch.float32)
    wvt = torch.eye(4); wvt[:3, :3] = R.T; wvt[3, :3] = torch.tensor([0.3, -1.5, 0.2])
    cam = types.SimpleNamespace(image_height=H, image_width=W, intrinsic=K, world_view_transform=wvt.to(dev))
    cfg = types.SimpleNamespace(sky_resolution=1024, sky_white_background=False, white_background=False)
    m = SkyCubeMap(cfg, device=dev)
    m.sky_cube_map.data.fill_(0.25)
    assert torch.all((m(cam) - 0.25).abs() < 1e-6)
    gen = torch.Generator().manual_seed(5)
    m.sky_cube_map.data = (torch.rand(6, 1024, 1024, 3, generator=gen) * 0.8 + 0.1).to(dev)
    acc = torch.rand(1, H, W, generator=gen).to(dev).requires_grad_(True)
    render = torch.rand(3, H, W, generator=gen).to(dev).requires_grad_(True)
    out, sky = composite_s3g(m, cam, render, acc)
    sampled = (1 - acc.detach()[0]) > 1e-3                      # fully covered pixels are not looked up (fill colour)
    assert sky[:, sampled].min() >= 0.1 - 1e-6 and sky[:, sampled].max() <= 0.9 + 1e-6 and torch.all(sky[:, ~sampled] == 0)
    torch.testing.assert_close(out, render * acc + sky.detach() * (1 - acc), atol=1e-6, rtol=0)
    out.sum().backward()
    # d(sum out)/d texel summed over texels = sum over pixels of (1 - acc) * 3 channels... per channel: sum (1 - acc)
    expect = ((1 - acc.detach()[0]) * sampled).sum().item()
    got = m.sky_cube_map.grad.sum(dim=(0, 1, 2)).cpu().numpy()
    np.testing.assert_allclose(got, expect, rtol=2e-4)
    torch.testing.assert_close(render.grad, acc.detach().expand(3, H, W), atol=1e-6, rtol=0)
