"""-m gpu: the OmniRe call site (`rasterization(...)`, base.py:393-408) served by the HIP rasterizer, checked against the CPU oracle
configured the same way (near plane 0.1, intrinsics with an off-centre principal point, RGB + expected depth, absgrad) -- images to
1e-4, radii exact, and EVERY gradient of the call (means, quats through the adapter's normalisation, scales, opacities, colours / SH
coefficients, `means2d.grad`, `means2d.absgrad`) to the gradient bar of tests/helpers.py; precomputed colours and SH colours
(sh_degree = 3); one camera and a two-camera batch.  `postprocess_per_train_step` (base.py:279-297) reads exactly these."""
import numpy as np
import pytest
import torch

from emd_amd import camera, gsplat_api
from oracle import cpu_oracle as co
from tests.helpers import make_case, assert_grad_close, END2END_ATOL_FRAC, END2END_REL_L2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _camera(case, yaw_deg=0.0, dx=3.5, dy=-2.25):
    H, W = case["H"], case["W"]
    K = torch.tensor([[118.0, 0, W / 2 + dx], [0, 112.0, H / 2 + dy], [0, 0, 1]])
    c2w = torch.linalg.inv(case["cam"].world_view_transform.t())
    if yaw_deg:
        a = np.deg2rad(yaw_deg)
        R = torch.tensor([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
        c2w = R @ c2w
    return c2w, K, camera.from_c2w_K(c2w, K, W, H)


def _oracle_view(case, cam, quats_raw, sh):
    """Forward + backward inputs of one view on the oracle, the quaternions normalised as the adapter does."""
    H, W = case["H"], case["W"]
    S = co.make_settings(H, W, cam.tanfovx, cam.tanfovy, [0, 0, 0], cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                         3 if sh else 0, cam.camera_center.numpy(), 1.0, near_plane=0.1)
    qn = torch.nn.functional.normalize(quats_raw, dim=-1).numpy()          # fp32, as the adapter normalises them
    kw = dict(shs=case["shs"].numpy()) if sh else dict(colors_precomp=case["colors_precomp"].numpy())
    sc = co.Scene(case["means3D"].numpy(), case["opacities"].numpy(), scales=case["scales"].numpy(), rotations=qn, **kw)
    pre, b, img = co.forward(S, sc, 0)
    return S, sc, pre, b, img


def _through_normalize(q_raw, g_unit):
    """dL/dq_raw from dL/dq_unit (float64): d(q / |q|) = (I - u u^T) / |q|."""
    q = q_raw.double().numpy()
    n = np.linalg.norm(q, axis=1, keepdims=True)
    u = q / n
    g = np.asarray(g_unit, np.float64)
    return (g - u * (u * g).sum(1, keepdims=True)) / n


@pytest.mark.parametrize("sh,cams", [(False, 1), (True, 1), (False, 2), (True, 2)], ids=["rgb-1cam", "sh3-1cam", "rgb-2cams", "sh3-2cams"])
def test_rasterization_matches_oracle_and_reference_call_convention(sh, cams):
    case = make_case(n=3000, H=72, W=104, seed=51, colors_precomp=not sh)
    H, W, N = case["H"], case["W"], case["N"]
    views = [_camera(case)] + ([_camera(case, yaw_deg=9.0, dx=-1.5, dy=4.0)] if cams == 2 else [])
    d = lambda t: t.to(DEV).clone().requires_grad_(True)
    quats_raw = case["rotations"] * (0.6 + 1.4 * torch.rand(N, 1, generator=torch.Generator().manual_seed(2)))       # un-normalised, as OmniRe stores them
    means, quats, scales, opac = d(case["means3D"]), d(quats_raw), d(case["scales"]), d(case["opacities"])
    colors = d(case["shs"]) if sh else d(case["colors_precomp"])
    renders, alphas, info = gsplat_api.rasterization(
        means=means, quats=quats, scales=scales, opacities=opac.squeeze(), colors=colors,
        viewmats=torch.stack([torch.linalg.inv(c2w) for c2w, _, _ in views]).to(DEV), Ks=torch.stack([K for _, K, _ in views]).to(DEV),
        width=W, height=H, packed=False, absgrad=True, sparse_grad=False, rasterize_mode="classic", near_plane=0.1, far_plane=1e10,
        render_mode="RGB+ED", radius_clip=0.0, sh_degree=3 if sh else None)
    assert renders.shape == (cams, H, W, 4) and alphas.shape == (cams, H, W, 1)
    assert info["means2d"].shape == (cams, N, 2) and info["radii"].shape == (cams, N)
    info["means2d"].retain_grad()
    rng = np.random.default_rng(3)
    gC = rng.standard_normal((cams, 3, H, W)).astype(np.float32)
    gA = (0.3 * rng.standard_normal((cams, 1, H, W))).astype(np.float32)
    loss = (renders[..., :3] * torch.tensor(gC).to(DEV).permute(0, 2, 3, 1)).sum() + (alphas * torch.tensor(gA).to(DEV).permute(0, 2, 3, 1)).sum()
    loss.backward()
    tot = {k: 0.0 for k in ("means3D", "scales", "rotations", "opacities", "colors")}
    for c, (c2w, K, cam) in enumerate(views):
        S, sc, pre, b, img = _oracle_view(case, cam, quats_raw, sh)
        np.testing.assert_array_equal(info["radii"][c].cpu().numpy(), pre["radii"])
        rgb = renders[c, ..., :3].detach().cpu().numpy().transpose(2, 0, 1)
        assert np.abs(rgb - img["color"]).max() <= 1e-4
        assert np.abs(alphas[c, ..., 0].detach().cpu().numpy() - img["alpha"][0]).max() <= 1e-4
        ed = img["depth"][0] / np.maximum(img["alpha"][0], 1e-10)
        m = img["alpha"][0] > 1e-3
        assert np.abs(renders[c, ..., 3].detach().cpu().numpy() - ed)[m].max() <= 1e-3 * max(1.0, ed[m].max())
        g = co.backward(S, sc, pre, b, img, gC[c], None, gA[c], None, co.F_ABSGRAD)
        # per view: the pixel-space mean gradients of THIS camera (what postprocess_per_train_step scales by W/2, H/2 and accumulates)
        assert_grad_close(info["means2d"].grad[c].cpu().numpy(), g["render_grads"]["mean2D"], f"means2d.grad[{c}]")
        assert_grad_close(info["means2d"].absgrad[c].cpu().numpy(), g["render_grads"]["abs"], f"means2d.absgrad[{c}]")
        tot["means3D"] = tot["means3D"] + np.asarray(g["means3D"], np.float64)
        tot["scales"] = tot["scales"] + np.asarray(g["scales"], np.float64)
        tot["rotations"] = tot["rotations"] + np.asarray(g["rotations"], np.float64)
        tot["opacities"] = tot["opacities"] + np.asarray(g["opacities"], np.float64).reshape(-1)
        tot["colors"] = tot["colors"] + np.asarray(g["shs"] if sh else g["colors"], np.float64)
    # the shared tensors receive the SUM over the cameras of the batch
    assert_grad_close(means.grad.cpu().numpy(), tot["means3D"], "means")
    assert_grad_close(colors.grad.cpu().numpy(), tot["colors"], "colors / sh coefficients")
    assert_grad_close(opac.grad.cpu().numpy().reshape(-1), tot["opacities"], "opacities")
    # scales / rotations: the conditioned chain (tests/helpers.py), end-to-end floor; the rotation gradient through F.normalize
    assert_grad_close(scales.grad.cpu().numpy(), tot["scales"], "scales", atol_frac=END2END_ATOL_FRAC, rel_l2=END2END_REL_L2)
    assert_grad_close(quats.grad.cpu().numpy(), _through_normalize(quats_raw, tot["rotations"]), "quats (through the normalisation)",
                      atol_frac=END2END_ATOL_FRAC, rel_l2=END2END_REL_L2)


def test_rasterization_rejects_unsupported_options():
    z = torch.zeros(4, 3, device=DEV)
    kw = dict(means=z, quats=torch.ones(4, 4, device=DEV), scales=z + 1, opacities=torch.ones(4, device=DEV), colors=z,
              viewmats=torch.eye(4, device=DEV)[None], Ks=torch.eye(3, device=DEV)[None], width=32, height=32)
    with pytest.raises(NotImplementedError):
        gsplat_api.rasterization(**kw, rasterize_mode="antialiased")
    with pytest.raises(NotImplementedError):
        gsplat_api.rasterization(**kw, packed=True)
    with pytest.raises(NotImplementedError):            # a finite far plane would have to cull: refused, not ignored
        gsplat_api.rasterization(**kw, far_plane=80.0)


def test_after_train_running_sums_on_top_of_means2d_absgrad():
    """SURVEY row a17 end to end: `rasterization(..., absgrad=True)` -> `info["means2d"].absgrad` scaled to pixels as the trainer does
    (OmniRe/models/trainers/base.py:279-286: x (W/2 B), y (H/2 B), B = 1) -> `VanillaGaussians.after_train` over three views with different
    visibility.  Expected: vanilla.py:163-191 written out in numpy (first call: the norm of EVERY row and a count of one everywhere; later calls
    only where radii > 0; max_2Dsize = max over views of radii / max(W, H)) applied to the CPU ORACLE's absgrad and radii of the same views --
    so the chain kernel (K7 |grad| column) -> adapter -> statistics kernel is pinned against a path that shares none of its code."""
    from emd_amd.vanilla import VanillaGaussians
    case = make_case(n=2500, H=72, W=104, seed=77, colors_precomp=True)
    H, W, N = case["H"], case["W"], case["N"]
    views = [_camera(case), _camera(case, yaw_deg=25.0, dx=-1.5, dy=4.0), _camera(case, yaw_deg=-40.0, dx=0.5, dy=0.0)]
    quats_raw = case["rotations"]
    node = VanillaGaussians("Background", dict(sh_degree=3, refine_interval=100), device=DEV)
    node._means = torch.nn.Parameter(case["means3D"].to(DEV))
    exp_norm = exp_vis = exp_m2d = None
    rng = np.random.default_rng(5)
    for v, (c2w, K, cam) in enumerate(views):
        d = lambda t: t.to(DEV).clone().requires_grad_(True)
        means, quats, scales, opac, colors = d(case["means3D"]), d(quats_raw), d(case["scales"]), d(case["opacities"]), d(case["colors_precomp"])
        renders, alphas, info = gsplat_api.rasterization(
            means=means, quats=quats, scales=scales, opacities=opac.squeeze(), colors=colors, viewmats=torch.linalg.inv(c2w)[None].to(DEV),
            Ks=K[None].to(DEV), width=W, height=H, packed=False, absgrad=True, sparse_grad=False, rasterize_mode="classic", near_plane=0.1,
            far_plane=1e10, render_mode="RGB+ED", radius_clip=0.0, sh_degree=None)
        info["means2d"].retain_grad()
        gC = rng.standard_normal((3, H, W)).astype(np.float32)
        (renders[0, ..., :3] * torch.tensor(gC).to(DEV).permute(1, 2, 0)).sum().backward()
        grads = info["means2d"].absgrad.clone()                      # base.py:281-286
        grads[..., 0] *= info["width"] / 2.0 * 1
        grads[..., 1] *= info["height"] / 2.0 * 1
        node.after_train(info["radii"][0], grads[0], last_size=max(info["width"], info["height"]))
        # the same view on the oracle
        S, sc, pre, b, img = _oracle_view(case, cam, quats_raw, False)
        g = co.backward(S, sc, pre, b, img, gC, None, None, None, co.F_ABSGRAD)
        ab = np.asarray(g["render_grads"]["abs"], np.float64) * np.array([W / 2.0, H / 2.0])
        norm = np.sqrt((ab ** 2).sum(1))
        vis = pre["radii"] > 0
        if exp_norm is None:
            exp_norm, exp_vis, exp_m2d = norm.copy(), np.ones(N), np.zeros(N)
        else:
            exp_norm[vis] += norm[vis]
            exp_vis[vis] += 1
        exp_m2d[vis] = np.maximum(exp_m2d[vis], pre["radii"][vis] / float(max(W, H)))
        np.testing.assert_array_equal(info["radii"][0].cpu().numpy(), pre["radii"])
    assert (exp_vis > 1).any() and (exp_vis == 1).any()
    np.testing.assert_array_equal(node.vis_counts.cpu().numpy(), exp_vis.astype(np.float32))
    np.testing.assert_allclose(node.max_2Dsize.cpu().numpy(), exp_m2d.astype(np.float32), rtol=1e-7, atol=0)
    assert_grad_close(node.xys_grad_norm.cpu().numpy(), exp_norm, "xys_grad_norm after three views")
