"""-m gpu: the OmniRe call site (`rasterization(...)`, base.py:393-408) served by the HIP rasterizer, checked
against the CPU oracle configured the same way (near plane 0.1, intrinsics with an off-centre principal point,
precomputed colours, RGB + expected depth, absgrad)."""
import numpy as np
import pytest
import torch

from emd_amd import camera, gsplat_api
from oracle import cpu_oracle as co
from tests.helpers import make_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_rasterization_matches_oracle_and_reference_call_convention():
    case = make_case(n=3000, H=72, W=104, seed=51, colors_precomp=True)
    H, W = case["H"], case["W"]
    K = torch.tensor([[118.0, 0, W / 2 + 3.5], [0, 112.0, H / 2 - 2.25], [0, 0, 1]])
    c2w = torch.linalg.inv(case["cam"].world_view_transform.t())
    cam = camera.from_c2w_K(c2w, K, W, H)
    d = lambda t: t.to(DEV).clone().requires_grad_(True)
    means, quats, scales, opac, colors = d(case["means3D"]), d(case["rotations"] * 1.3), d(case["scales"]), d(case["opacities"]), d(case["colors_precomp"])
    renders, alphas, info = gsplat_api.rasterization(
        means=means, quats=quats, scales=scales, opacities=opac.squeeze(), colors=colors,
        viewmats=torch.linalg.inv(c2w.to(DEV))[None], Ks=K.to(DEV)[None], width=W, height=H, packed=False, absgrad=True,
        sparse_grad=False, rasterize_mode="classic", near_plane=0.1, far_plane=1e10, render_mode="RGB+ED", radius_clip=0.0)
    assert renders.shape == (1, H, W, 4) and alphas.shape == (1, H, W, 1)
    assert info["means2d"].shape == (1, case["N"], 2) and info["radii"].shape == (1, case["N"])
    info["means2d"].retain_grad()
    # oracle with the same camera
    S = co.make_settings(H, W, cam.tanfovx, cam.tanfovy, [0, 0, 0], cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                         0, cam.camera_center.numpy(), 1.0, near_plane=0.1)
    qn = (case["rotations"] * 1.3)
    qn = (qn / qn.norm(dim=1, keepdim=True)).numpy()
    sc = co.Scene(case["means3D"].numpy(), case["opacities"].numpy(), colors_precomp=case["colors_precomp"].numpy(),
                  scales=case["scales"].numpy(), rotations=qn)
    pre, b, img = co.forward(S, sc, 0)
    np.testing.assert_array_equal(info["radii"][0].cpu().numpy(), pre["radii"])
    rgb = renders[0, ..., :3].detach().cpu().numpy().transpose(2, 0, 1)
    assert np.abs(rgb - img["color"]).max() <= 1e-4
    assert np.abs(alphas[0, ..., 0].detach().cpu().numpy() - img["alpha"][0]).max() <= 1e-4
    ed = img["depth"][0] / np.maximum(img["alpha"][0], 1e-10)
    got_ed = renders[0, ..., 3].detach().cpu().numpy()
    m = img["alpha"][0] > 1e-3
    assert np.abs(got_ed - ed)[m].max() <= 1e-3 * max(1.0, ed[m].max())
    # gradients: loss on rgb only -> compare with the oracle backward; means2d grads come back in pixel units
    gC = np.random.default_rng(3).standard_normal((3, H, W)).astype(np.float32)
    (renders[0, ..., :3] * torch.tensor(gC).to(DEV).permute(1, 2, 0)).sum().backward()
    g = co.backward(S, sc, pre, b, img, gC, None, None, None, co.F_ABSGRAD)
    rel = lambda a, r: np.abs(a - r).max() / max(np.abs(r).max(), 1e-12)
    assert rel(means.grad.cpu().numpy(), g["means3D"]) < 2e-3
    assert rel(colors.grad.cpu().numpy(), g["colors"]) < 2e-3
    assert rel(info["means2d"].grad[0].cpu().numpy(), g["render_grads"]["mean2D"]) < 2e-3
    assert rel(info["means2d"].absgrad[0].cpu().numpy(), g["render_grads"]["abs"]) < 2e-3
    # quaternion gradient flows through the adapter's normalisation
    assert quats.grad is not None and torch.isfinite(quats.grad).all()


def test_rasterization_rejects_unsupported_options():
    z = torch.zeros(4, 3, device=DEV)
    kw = dict(means=z, quats=torch.ones(4, 4, device=DEV), scales=z + 1, opacities=torch.ones(4, device=DEV), colors=z,
              viewmats=torch.eye(4, device=DEV)[None], Ks=torch.eye(3, device=DEV)[None], width=32, height=32)
    with pytest.raises(NotImplementedError):
        gsplat_api.rasterization(**kw, rasterize_mode="antialiased")
    with pytest.raises(NotImplementedError):
        gsplat_api.rasterization(**kw, packed=True)
