"""-m gpu: one S3Gaussian-style training step assembled from every HIP piece of this repository, chained through autograd:
fused-motion rasterizer -> sky cube map + blend -> image-loss tail (L1 + depth + D-SSIM + sky BCE) -> backward -> per-view
densification statistics.  Checks the composition (values against the oracles evaluated on the rasterizer's outputs, gradients
reaching every parameter group), not the pieces (they have their own tests)."""
import types

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo
from oracle import sky_oracle as so

pytestmark = pytest.mark.gpu


def test_s3g_style_step_composes():
    from emd_amd import dp, scenes
    from emd_amd.loss import image_loss
    from emd_amd.model import StreetGaussians, render
    from emd_amd.sky import SkyCubeMap, composite_s3g
    dev = torch.device("cuda", 0)
    N, H, W = 30000, 96, 160
    scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=3, pts_per_actor=2000, num_frames=5, seed=1)
    model = StreetGaussians(scene, dev)
    cam = scenes.rig_camera(2, 0, H, W)
    K = torch.tensor([[cam.tanfovx and (W / (2 * cam.tanfovx)), 0, W / 2], [0, H / (2 * cam.tanfovy), H / 2], [0, 0, 1]], dtype=torch.float32)
    skycam = types.SimpleNamespace(image_height=H, image_width=W, intrinsic=K.to(dev), world_view_transform=cam.world_view_transform.to(dev))
    sky = SkyCubeMap(types.SimpleNamespace(sky_resolution=32, sky_white_background=False, white_background=False), device=dev)
    g = torch.Generator().manual_seed(9)
    sky.sky_cube_map.data = torch.rand(6, 32, 32, 3, generator=g).to(dev)
    gt = torch.rand(3, H, W, generator=g).to(dev)
    gt_depth = (torch.rand(1, H, W, generator=g) * 60).to(dev)
    sky_mask = (torch.rand(1, H, W, generator=g) < 0.3).to(dev)

    out = render(model, cam, torch.zeros(3), frame=2)
    image, sky_color = composite_s3g(sky, skycam, out["render"], out["weight"])
    loss, terms = image_loss(image, gt, out["depth"], gt_depth, ~sky_mask, out["weight"], sky_mask)
    loss.backward()

    # composition: the same tail evaluated by the CPU oracles on the rasterizer's outputs
    r, w, d = out["render"].detach().cpu(), out["weight"].detach().cpu(), out["depth"].detach().cpu()
    w2c = cam.world_view_transform.T
    sky_o = so.sky_s3g(sky.sky_cube_map.detach().cpu(), so.rays(H, W, K, w2c[:3, :3], w2c[:3, 3]), w)
    np.testing.assert_allclose(sky_color.detach().cpu().numpy(), sky_o.numpy(), atol=2e-5)
    img_o = so.blend_s3g(r, w, sky_o)
    tot_o, _ = lo.loss_tail(img_o, gt.cpu(), d, gt_depth.cpu(), (~sky_mask).float().cpu(), w, sky_mask.cpu())
    np.testing.assert_allclose(loss.item(), tot_o.item(), rtol=2e-5)
    # gradients reach every group
    for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features", "instances_quats", "instances_trans"):
        gr = getattr(model, name).grad
        assert gr is not None and torch.isfinite(gr).all() and gr.abs().sum() > 0, name
    assert sky.sky_cube_map.grad is not None and sky.sky_cube_map.grad.abs().sum() > 0
    # per-view densification statistics, fused
    accum, denom, maxr = (torch.zeros(N, device=dev) for _ in range(3))
    dp.add_densification_stats(out["viewspace_points"].grad, out["radii"], accum, denom, maxr)
    vis = out["radii"] > 0
    assert torch.equal(denom.bool(), vis) and torch.all(maxr[vis] == out["radii"][vis].float()) and accum[vis].sum() > 0


def test_fine_stage_trains_through_the_rasterizer():
    """S3Gaussian's fine stage as a loop (train.py after coarse_iterations): the EMD deformation network (HexPlane + temporal row +
    fused MLP kernels) in front of the rasterizer, trained through it with emd_amd.optim.Adam on the reference's parameter groups.
    The photometric loss to a target rendered from displaced Gaussians must fall, every group must receive finite gradients, and the
    first step's gradients must not depend on which formulation of the MLP ran (fused fp32-MFMA kernels vs hipBLASLt GEMMs)."""
    from emd_amd import scenes
    from emd_amd.deformation import DeformOptions, deform_network
    from emd_amd.model import StreetGaussians, l1_loss, render
    from emd_amd.optim import Adam
    dev = torch.device("cuda", 0)
    N, H, W, F = 20000, 96, 160, 4
    scene = scenes.make_static_scene(N, seed=2)
    model = StreetGaussians(scene, dev)
    torch.manual_seed(3)
    opt = DeformOptions()
    deform = deform_network(opt).to(dev)
    deform.deformation_net.set_aabb([120.0, 30.0, 10.0], [0.0, -30.0, -2.0])
    for n_, p_ in deform.named_parameters():
        if p_.dim() > 1 and "grid" not in n_:
            p_.data.mul_(0.05)
    emb = torch.nn.Parameter(0.1 * torch.randn(N, 4, device=dev))
    cam = scenes.rig_camera(1, 0, H, W)
    bg = torch.zeros(3)
    with torch.no_grad():            # the target: the same scene with its Gaussians pushed half a metre sideways
        model._xyz.data[:, 1] += 0.5
        target = render(model, cam, bg, frame=0)["render"].clone()
        model._xyz.data[:, 1] -= 0.5

    def step_grads(fused):
        opt.fused_mlp = fused
        for p in list(deform.parameters()) + [emb]:
            p.grad = None
        out = render(model, cam, bg, frame=0, deformation=deform, embeddings=emb, iteration=12000, time=0.4)
        l1_loss(out["render"], target).backward()
        return {n: p.grad.clone() for n, p in deform.named_parameters() if p.grad is not None}, emb.grad.clone()
    gf, ef = step_grads(True)
    gg, eg = step_grads(False)
    assert set(gf) == set(gg)
    for n in gf:
        scale = max(float(gg[n].abs().max()), 1e-12)
        assert float((gf[n] - gg[n]).abs().max()) <= 2e-4 * scale, n
    assert float((ef - eg).abs().max()) <= 2e-4 * max(float(eg.abs().max()), 1e-12)
    opt.fused_mlp = True

    groups = [{"params": deform.get_mlp_parameters(), "lr": 2e-3, "name": "deformation"},
              {"params": deform.get_grid_parameters(), "lr": 2e-2, "name": "grid"}, {"params": [emb], "lr": 1e-2, "name": "embedding"}]
    adam = Adam(groups, lr=0.0, eps=1e-15)
    losses = []
    for it in range(40):
        adam.zero_grad(set_to_none=True)
        out = render(model, cam, bg, frame=0, deformation=deform, embeddings=emb, iteration=12000 + it, time=0.4)
        loss = l1_loss(out["render"], target)
        loss.backward()
        for g_ in groups:
            for p in g_["params"]:
                assert p.grad is None or bool(torch.isfinite(p.grad).all()), g_["name"]
        adam.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.9 * losses[0], (losses[0], losses[-1])


def test_decomposition_passes_of_the_reference_render():
    """render_decomposition: the evaluation passes of gaussian_renderer/__init__.py:203-294 (no combine_dynamic_static) -- per level the
    0.5 % farthest-moving Gaussians rendered alone through a boolean-mask subset, and the scene coloured by |dx| / max |dx| -- against
    the same passes issued by hand (a fresh rasterizer per call, activated parameters, the oracle for the colour pass)."""
    from emd_amd import GaussianRasterizer, scenes
    from emd_amd.deformation import DeformOptions, deform_network
    from emd_amd.model import StreetGaussians, raster_settings_for, render, render_decomposition
    from oracle import cpu_oracle as co
    dev = torch.device("cuda", 0)
    N, H, W = 12000, 80, 128
    model = StreetGaussians(scenes.make_static_scene(N, seed=5), dev)
    torch.manual_seed(7)
    deform = deform_network(DeformOptions()).to(dev)
    deform.deformation_net.set_aabb([120.0, 30.0, 10.0], [0.0, -30.0, -2.0])
    for n_, p_ in deform.named_parameters():
        if p_.dim() > 1 and "grid" not in n_:
            p_.data.mul_(0.05)
    emb = 0.1 * torch.randn(N, 4, device=dev)
    cam, bg = scenes.rig_camera(2, 0, H, W), torch.zeros(3)
    with torch.no_grad():
        out = render(model, cam, bg, frame=0, deformation=deform, embeddings=emb, iteration=12000, time=0.3)
        dec = render_decomposition(out)
    assert set(dec) == {"coarse_render", "fine_render", "coarse_fine_render"}
    bd = {k_: (v.detach() if isinstance(v, torch.Tensor) else v) for k_, v in out["boundary"].items()}
    s_act, q_act, o_act = torch.exp(bd["scales"]), torch.nn.functional.normalize(bd["rotations"]), torch.sigmoid(bd["opacities"])
    assert bd["shs_residuals"] is not None and len(bd["shs_residuals"]) == 2        # render() hands the dshs residuals to K1 unsummed
    shs_all = (bd["shs"] + bd["shs_residuals"][0].detach()) + bd["shs_residuals"][1].detach()
    rs = raster_settings_for(cam, bg, model.active_sh_degree)
    for lvl, d in (("coarse", out["ddict"]["coarse"]["dx"]), ("fine", out["ddict"]["fine"]["dx"]),
                   ("coarse_fine", out["ddict"]["coarse"]["dx"] - out["ddict"]["fine"]["dx"])):
        got = dec[lvl + "_render"]
        d_abs = d.abs()
        mask = torch.zeros(N, dtype=torch.bool, device=dev)
        mask[torch.topk(d_abs.norm(dim=1), int(N * 0.005))[1]] = True
        assert int(mask.sum()) == 60
        with torch.no_grad():
            ref = GaussianRasterizer(rs)(means3D=bd["means3D"][mask], means2D=torch.zeros(60, 3, device=dev), shs=shs_all[mask], colors_precomp=None,
                                         opacities=o_act[mask], scales=s_act[mask], rotations=q_act[mask], cov3Ds_precomp=None, extra_attrs=None)
        torch.testing.assert_close(got["render"], ref[0], rtol=0, atol=2e-6)       # (fused activations vs activated inputs: last-bit differences)
        torch.testing.assert_close(got["weight"], ref[3], rtol=0, atol=2e-6)
        # the |dx| colour map against the oracle
        col = (d_abs / d_abs.max(dim=0, keepdim=True)[0]).cpu().numpy()
        S = co.make_settings(H, W, cam.tanfovx, cam.tanfovy, [0, 0, 0], cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), 0,
                             cam.camera_center.numpy(), 1.0)
        sc = co.Scene(bd["means3D"].cpu().numpy(), o_act.cpu().numpy(), colors_precomp=col, scales=s_act.cpu().numpy(), rotations=q_act.cpu().numpy())
        _, _, img = co.forward(S, sc, 0)
        assert float(np.abs(got["color"].cpu().numpy() - img["color"]).max()) <= 1e-4
        assert torch.equal(got["dx"], d)


def test_combine_dynamic_static_render_and_decomposition_reproduce_the_reference_calls():
    """render(..., combine_dynamic_static=True, convert_SHs_python=True) + render_decomposition against the seven rasterizer calls the
    reference's render() makes with that flag (tests/golden/s3g_render_combined.npz: gaussian_renderer/__init__.py:118-138,203-294; the
    fixture's residuals are replayed by a stand-in deformation): every boundary tensor of every call, in the reference's order -- and the
    SH path's single call (shs mixed by the activated opacities)."""
    import os
    from emd_amd import camera, model as M, scenes
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "s3g_render_combined.npz"))
    dev = torch.device("cuda", 0)
    t = lambda k: torch.tensor(z[k]).to(dev)
    N = z["xyz"].shape[0]
    sc = scenes.make_static_scene(N, seed=0)
    mdl = M.StreetGaussians(sc, dev)
    with torch.no_grad():
        mdl._xyz.copy_(t("xyz")); mdl._scaling.copy_(t("scaling")); mdl._rotation.copy_(t("rotation")); mdl._opacity.copy_(t("opacity"))
        mdl._features.copy_(t("features"))
    mdl.active_sh_degree = int(z["active_sh_degree"])
    dd = {lvl: {k: t(f"ddict_{lvl}_{k}") for k in ("dx", "do", "dshs")} for lvl in ("coarse", "fine")}

    def deformation(means3D, scales, rotations, opacity, shs, *a, **k):
        p, s_, r_, o_, h_ = M.apply_deform(means3D, scales, rotations, opacity, shs, dd["coarse"], dd["fine"])
        return p, s_, r_, o_, h_, {"coarse": dict(dd["coarse"]), "fine": dict(dd["fine"])}
    cam = camera.make_camera(z["R"].astype(np.float64), z["T"].astype(np.float64), float(z["fovx"]), float(z["fovy"]), int(z["H"]), int(z["W"]))
    calls = []
    real = M.GaussianRasterizer

    class Recording(real):
        def forward(self, **kw):
            calls.append({k: (None if v is None else v.detach().clone()) for k, v in kw.items() if k in
                          ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations")})
            return super().forward(**kw)
    M.GaussianRasterizer = Recording
    try:
        with torch.no_grad():
            out = M.render(mdl, cam, torch.tensor(z["bg"]), deformation=deformation, embeddings=None, iteration=3000, time=0.3,
                           combine_dynamic_static=True, convert_SHs_python=True)
            dec = M.render_decomposition(out)
            assert len(calls) == 7 and set(dec) == {"coarse_render", "fine_render", "coarse_fine_render"}
            names = ("main", "coarse_set", "coarse_dx", "fine_set", "fine_dx", "coarse_fine_set", "coarse_fine_dx")
            for nm, kw in zip(names, calls):
                assert kw["shs"] is None, nm
                for k in ("means3D", "colors_precomp", "opacities", "scales", "rotations"):
                    got, want = kw[k].cpu().numpy().reshape(z[f"{nm}_{k}"].shape), z[f"{nm}_{k}"]
                    np.testing.assert_allclose(got, want, rtol=2e-5, atol=5e-6, err_msg=f"{nm}.{k}")
            assert float(calls[0]["opacities"].max()) > 1.0 and bool(torch.isfinite(out["render"]).all())
            # the SH path: ONE call on the mixed coefficients
            calls.clear()
            out = M.render(mdl, cam, torch.tensor(z["bg"]), deformation=deformation, embeddings=None, iteration=3000, time=0.3,
                           combine_dynamic_static=True)
            assert len(calls) == 1 and calls[0]["colors_precomp"] is None
            for k in ("means3D", "shs", "opacities", "scales", "rotations"):
                np.testing.assert_allclose(calls[0][k].cpu().numpy().reshape(z[f"sh_main_{k}"].shape), z[f"sh_main_{k}"], rtol=2e-5, atol=5e-6,
                                           err_msg="sh_main." + k)
            # ... and its decomposition passes run too (the reference raises NameError there): coarse = the dynamic copies, fine = the static ones
            calls.clear()
            dec = M.render_decomposition(out)
            assert len(calls) == 6 and calls[0]["shs"] is not None
            np.testing.assert_allclose(calls[0]["means3D"].cpu().numpy(), z["sh_main_means3D"], rtol=2e-5, atol=5e-6)
            np.testing.assert_allclose(calls[2]["means3D"].cpu().numpy(), z["xyz"], atol=0)
    finally:
        M.GaussianRasterizer = real


def test_fused_sh_residuals_and_regulariser_equal_the_reference_formulation():
    """The fine-stage step with the SH residuals handed to the projection kernel unsummed and their L1 regulariser folded into the residuals'
    gradient (deform_network(..., fused_shs_residuals=True), GaussianRasterizer(shs_residuals=...), model.residual_pair_l1) against the
    formulation of the reference: shs + dshs_c + dshs_f as tensors (scene/deformation.py:468-481) and torch.abs(dshs).mean() per level
    (train.py:238-310).  Same image bit for bit, same loss, same gradients of every parameter."""
    from emd_amd import GaussianRasterizer, scenes
    from emd_amd.deformation import DeformOptions, deform_network
    from emd_amd.model import StreetGaussians, l1_loss, raster_settings_for, render, residual_abs_mean
    dev = torch.device("cuda", 0)
    N, H, W = 15000, 80, 128
    model = StreetGaussians(scenes.make_static_scene(N, seed=8), dev)
    torch.manual_seed(9)
    deform = deform_network(DeformOptions()).to(dev)
    deform.deformation_net.set_aabb([120.0, 30.0, 10.0], [0.0, -30.0, -2.0])
    for n_, p_ in deform.named_parameters():
        if p_.dim() > 1 and "grid" not in n_:
            p_.data.mul_(0.05)
    emb = torch.nn.Parameter(0.1 * torch.randn(N, 4, device=dev))
    cam, bg = scenes.rig_camera(3, 0, H, W), torch.zeros(3)
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    params = list(model.parameters()) + list(deform.parameters()) + [emb]

    def grads():
        return [None if p.grad is None else p.grad.clone() for p in params]

    # (a) the fused path (what render() does with an emd_amd network)
    for p in params:
        p.grad = None
    out = render(model, cam, bg, frame=0, deformation=deform, embeddings=emb, iteration=12000, time=0.4)
    assert out["ddict"].get("shs_residuals") is not None and "dshs_abs_mean" in out["ddict"]["fine"]
    loss_a = l1_loss(out["render"], target) + 0.01 * (residual_abs_mean(out["ddict"]["coarse"], "dshs") + residual_abs_mean(out["ddict"]["fine"], "dshs"))
    loss_a.backward()
    img_a, g_a = out["render"].detach().clone(), grads()
    # (b) the reference's formulation, from the same network: explicit sums and torch means
    for p in params:
        p.grad = None
    z = torch.zeros_like(model._xyz).requires_grad_(True)
    m3, sc_, rot_, op_, shs_sum, dd = deform(model._xyz, model._scaling, model._rotation, model._opacity, model._features,
                                            torch.full((N, 1), 0.4, device=dev), emb, 12000, 0, 0.0, True)
    assert "shs_residuals" not in dd and torch.equal(shs_sum, (model._features + dd["coarse"]["dshs"]) + dd["fine"]["dshs"])
    img_b = GaussianRasterizer(raster_settings_for(cam, bg, model.active_sh_degree))(
        means3D=m3, means2D=z, shs=shs_sum, colors_precomp=None, opacities=op_, scales=sc_, rotations=rot_, cov3Ds_precomp=None, raw_params=True)[0]
    loss_b = l1_loss(img_b, target) + 0.01 * (dd["coarse"]["dshs"].abs().mean() + dd["fine"]["dshs"].abs().mean())
    loss_b.backward()
    g_b = grads()
    assert torch.equal(img_a, img_b.detach())
    assert abs(float(loss_a.detach()) - float(loss_b.detach())) <= 1e-6 * abs(float(loss_b.detach()))
    for p, a_, b_ in zip(params, g_a, g_b):
        assert (a_ is None) == (b_ is None)
        if a_ is not None:
            assert float((a_ - b_).abs().max()) <= 2e-5 * max(float(b_.abs().max()), 1e-20), tuple(p.shape)


def test_residual_regularisers_folded_into_the_head_kernels_equal_the_separate_ops():
    """render(..., fused_l1=("dx", "do")): the regularisers mean |dx|, mean |do| of train.py:238-310 formed by the head kernels themselves
    (ddict[level]["dx_abs_mean"], picked up by model.residual_abs_mean) against abs_mean() on the residual tensors: same image, same loss,
    same gradients of every parameter -- without an abs-mean launch each way per residual and level, and without the add autograd needs to
    join the regulariser's gradient with the rasterizer's."""
    from emd_amd import scenes
    from emd_amd.deformation import DeformOptions, deform_network
    from emd_amd.model import StreetGaussians, abs_mean, l1_loss, render, residual_abs_mean
    dev = torch.device("cuda", 0)
    N, H, W = 15000, 80, 128
    model = StreetGaussians(scenes.make_static_scene(N, seed=8), dev)
    torch.manual_seed(9)
    deform = deform_network(DeformOptions()).to(dev)
    deform.deformation_net.set_aabb([120.0, 30.0, 10.0], [0.0, -30.0, -2.0])
    for n_, p_ in deform.named_parameters():
        if p_.dim() > 1 and "grid" not in n_:
            p_.data.mul_(0.05)
    emb = torch.nn.Parameter(0.1 * torch.randn(N, 4, device=dev))
    cam, bg = scenes.rig_camera(3, 0, H, W), torch.zeros(3)
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    params = list(model.parameters()) + list(deform.parameters()) + [emb]

    def run(folded):
        for p in params:
            p.grad = None
        out = render(model, cam, bg, frame=0, deformation=deform, embeddings=emb, iteration=12000, time=0.4, fused_l1=("dx", "do") if folded else ())
        loss = l1_loss(out["render"], target)
        for lvl in ("coarse", "fine"):
            d = out["ddict"][lvl]
            assert ("dx_abs_mean" in d) == folded and ("do_abs_mean" in d) == folded
            if folded:
                loss = loss + 0.02 * residual_abs_mean(d, "dx") + 0.03 * residual_abs_mean(d, "do") + 0.01 * residual_abs_mean(d, "dshs")
            else:
                loss = loss + 0.02 * abs_mean(d["dx"]) + 0.03 * abs_mean(d["do"]) + 0.01 * residual_abs_mean(d, "dshs")
        loss.backward()
        return out["render"].detach().clone(), float(loss.detach()), [None if p.grad is None else p.grad.clone() for p in params]
    img_a, loss_a, g_a = run(True)
    img_b, loss_b, g_b = run(False)
    assert torch.equal(img_a, img_b)
    assert abs(loss_a - loss_b) <= 1e-6 * abs(loss_b)
    for p, a_, b_ in zip(params, g_a, g_b):
        assert (a_ is None) == (b_ is None)
        if a_ is not None:
            assert float((a_ - b_).abs().max()) <= 2e-5 * max(float(b_.abs().max()), 1e-20), tuple(p.shape)


def test_step_graphs_replay_the_fine_stage_step_per_frame():
    """emd_amd.graphs.StepGraphs: one hipGraph per frame of a fine-stage step (deformation network -> rasterizer -> sky blend -> loss -> backward
    -> densification statistics -> capturable Adam), all in one pool.  (a) Without the optimiser every replay leaves the gradients of the eager
    step of ITS frame; (b) with it, a sequence of replays follows the same sequence of eager steps."""
    import copy
    import types
    from emd_amd import RasterOptions, dp, scenes
    from emd_amd.deformation import DeformOptions, deform_network
    from emd_amd.graphs import StepGraphs
    from emd_amd.loss import image_loss
    from emd_amd.model import StreetGaussians, abs_mean, render, residual_abs_mean
    from emd_amd.optim import Adam
    from emd_amd.sky import SkyCubeMap, _camera_rays_params, composite_s3g
    dev = torch.device("cuda", 0)
    N, H, W, F = 12000, 64, 96, 3
    g = torch.Generator().manual_seed(5)
    gt, gt_depth = torch.rand(3, H, W, generator=g).to(dev), (torch.rand(1, H, W, generator=g) * 90).to(dev)
    sky_mask = (torch.rand(1, H, W, generator=g) < 0.2).to(dev)
    cams, skycams = {}, {}
    for f in range(F):
        cam = scenes.rig_camera(f, 0, H, W)
        K = torch.tensor([[W / (2 * cam.tanfovx), 0, W / 2], [0, H / (2 * cam.tanfovy), H / 2], [0, 0, 1]], dtype=torch.float32)
        cams[f] = cam
        skycams[f] = types.SimpleNamespace(image_height=H, image_width=W, intrinsic=K.to(dev), world_view_transform=cam.world_view_transform.to(dev))

    def build(with_adam):
        torch.manual_seed(11)
        model = StreetGaussians(scenes.make_static_scene(N, seed=4), dev)
        deform = deform_network(DeformOptions()).to(dev)
        deform.deformation_net.set_aabb([120.0, 30.0, 10.0], [0.0, -30.0, -2.0])
        for n_, p_ in deform.named_parameters():
            if p_.dim() > 1 and "grid" not in n_:
                p_.data.mul_(0.05)
        deform.deformation_net.grid.reorder_min_points = 1000          # (the aggregating HexPlane backward and its cached orders at this size)
        emb = torch.nn.Parameter(0.1 * torch.randn(N, 4, device=dev))
        sky = SkyCubeMap(types.SimpleNamespace(sky_resolution=64, sky_white_background=False, white_background=False), device=dev)
        params = list(model.parameters()) + list(deform.parameters()) + [emb, sky.sky_cube_map]
        opt = Adam([{"params": [p for p in params if p.requires_grad], "lr": 1e-3}], lr=0.0, eps=1e-15, capturable=True) if with_adam else None
        stats = [torch.zeros(N, device=dev) for _ in range(3)]
        opts = RasterOptions(no_sync=True, capacity_hint=400000)

        def step(f):
            for p in params:
                p.grad = None
            out = render(model, cams[f], torch.zeros(3), frame=f, deformation=deform, embeddings=emb, iteration=12000, time=f / 2.0, options=opts,
                         need_feat=False)
            image, _ = composite_s3g(sky, skycams[f], out["render"], out["weight"])
            loss, _ = image_loss(image, gt, out["depth"], gt_depth, ~sky_mask, out["weight"], sky_mask)
            for lvl in ("coarse", "fine"):
                d = out["ddict"][lvl]
                loss = loss + 0.001 * (abs_mean(d["dx"]) + abs_mean(d["do"]) + residual_abs_mean(d, "dshs"))
            loss.backward()
            dp.add_densification_stats(out["viewspace_points"].grad, out["radii"], *stats)
            if opt is not None:
                opt.step()
        return model, deform, emb, sky, params, opt, stats, step

    # ---- (a) gradients of every frame's replay against the eager step of that frame
    model, deform, emb, sky, params, opt, stats, step = build(False)
    want = {}
    for f in range(F):
        step(f)
        want[f] = [None if p.grad is None else p.grad.clone() for p in params]
    held = {}

    def recorded(f):
        step(f)
        held[f] = [p.grad for p in params]            # (keeps every graph's gradient tensors alive: they are compared below)
    sg = StepGraphs(recorded, range(F), prime=lambda f: _camera_rays_params(skycams[f]), freeze=[deform.deformation_net.grid], warmup=1)
    assert deform.deformation_net.grid.reorder_every >= 1 << 60
    for f in (2, 0, 1, 2):
        sg.replay(f)
        torch.cuda.synchronize()
        for p, got, ref in zip(params, held[f], want[f]):
            assert (got is None) == (ref is None)
            if got is not None:
                assert float((got - ref).abs().max()) <= 2e-4 * max(float(ref.abs().max()), 1e-20) + 1e-12, (f, tuple(p.shape))
    sg.release()
    assert deform.deformation_net.grid.reorder_every < 1 << 60
    # ---- (b) with the optimiser inside: replays follow eager steps
    seq = [0, 1, 2, 1, 0, 2]
    m1 = build(True)
    for f in [0] + seq:                                   # (the recorder's one eager warm-up step of frame 0, then the sequence)
        m1[-1](f)
    m2 = build(True)
    with pytest.raises(RuntimeError):                     # no eager step yet: the visiting orders would be built inside a capture
        StepGraphs(m2[-1], range(F), prime=lambda f: _camera_rays_params(skycams[f]), freeze=[m2[1].deformation_net.grid], optimizers=[m2[5]], warmup=0)
    sg2 = StepGraphs(m2[-1], range(F), prime=lambda f: _camera_rays_params(skycams[f]), freeze=[m2[1].deformation_net.grid], optimizers=[m2[5]], warmup=1)
    for f in seq:
        sg2.replay(f)
    torch.cuda.synchronize()
    assert float(m2[5].state[m2[0]._xyz]["step"]) == len(seq) + 1
    # Adam divides by sqrt(v): where a gradient is rounding noise (the render backward's float atomics reorder sums between runs) its step is
    # +-lr whatever the noise says, so single elements may differ by up to steps x lr; all but a few must agree closely
    for a_, b_ in zip(m1[4], m2[4]):
        diff = (a_.detach() - b_.detach()).abs()
        assert float(diff.max()) <= (len(seq) + 1) * 1e-3 * 2.0 + 1e-7, tuple(a_.shape)
        assert float((diff > 1e-4 * max(float(a_.detach().abs().max()), 1e-20) + 1e-7).float().mean()) <= 0.02, tuple(a_.shape)
    with pytest.raises(ValueError):
        StepGraphs(lambda f: None, [0], optimizers=[Adam([{"params": [m2[2]], "lr": 1e-3}], lr=0.0, eps=1e-15)], warmup=0)
