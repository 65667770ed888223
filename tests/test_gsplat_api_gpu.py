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
