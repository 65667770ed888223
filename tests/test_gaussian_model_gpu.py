"""-m gpu: adaptive density control on the device (emd_amd.gaussian_model.GaussianModel: densify / prune / reset_opacity through
emd_densify_*) against the reference's own GaussianModel methods run on CPU (tests/golden/s3g_surgery.npz: every parameter, both
Adam moments, the statistics and the deformation table after each call, with the reference's torch.normal draw recorded), the
Philox split samples, and a training loop that densifies and prunes unattended."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
G = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "embedding")


def _train_args(**kw):
    a = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                              position_lr_max_steps=30000, deformation_lr_init=1.6e-5, deformation_lr_final=1.6e-6, deformation_lr_delay_mult=0.01,
                              grid_lr_init=1.6e-3, grid_lr_final=1.6e-5, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3,
                              sky_cube_map_lr_init=0.01, sky_cube_map_lr_final=1e-4, sky_cube_map_max_steps=30000)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _model_from(z, tag):
    from emd_amd.gaussian_model import GaussianModel
    m = GaussianModel(sh_degree=3, gaussian_embedding_dim=4, device=DEV)
    P = lambda k: torch.nn.Parameter(torch.tensor(z[f"{tag}_{k}"]).to(DEV).contiguous())
    for n in NAMES:
        setattr(m, m._ATTR[n], P(n))
    m._deformation_table = torch.tensor(z[f"{tag}_table"]).to(DEV)
    m.spatial_lr_scale = 5.0
    m.training_setup(_train_args(percent_dense=float(z["percent_dense"])))
    for n in NAMES:
        p = getattr(m, m._ATTR[n])
        m.optimizer.state[p] = {"step": torch.tensor(2.0), "exp_avg": torch.tensor(z[f"{tag}_m_{n}"]).to(DEV).contiguous(),
                                "exp_avg_sq": torch.tensor(z[f"{tag}_v_{n}"]).to(DEV).contiguous()}
    m.xyz_gradient_accum = torch.tensor(z[f"{tag}_accum"]).to(DEV).contiguous()
    m.denom = torch.tensor(z[f"{tag}_denom"]).to(DEV).contiguous()
    m.max_radii2D = torch.tensor(z[f"{tag}_maxr"]).to(DEV).contiguous()
    return m


def _check(m, z, tag, exact=True):
    for n in NAMES:
        p = getattr(m, m._ATTR[n])
        got, ref = p.detach().cpu().numpy(), z[f"{tag}_{n}"]
        assert got.shape == ref.shape, (tag, n, got.shape, ref.shape)
        if exact and n not in ("xyz", "scaling"):
            np.testing.assert_array_equal(got, ref, err_msg=f"{tag} {n}")
        else:
            np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6, err_msg=f"{tag} {n}")
        st = m.optimizer.state[p]
        np.testing.assert_array_equal(st["exp_avg"].cpu().numpy(), z[f"{tag}_m_{n}"], err_msg=f"{tag} exp_avg {n}")
        np.testing.assert_array_equal(st["exp_avg_sq"].cpu().numpy(), z[f"{tag}_v_{n}"], err_msg=f"{tag} exp_avg_sq {n}")
        assert any(p is q for g_ in m.optimizer.param_groups for q in g_["params"]), n          # the optimiser holds the NEW leaf
    np.testing.assert_array_equal(m.xyz_gradient_accum.cpu().numpy(), z[f"{tag}_accum"])
    np.testing.assert_array_equal(m.denom.cpu().numpy(), z[f"{tag}_denom"])
    np.testing.assert_array_equal(m.max_radii2D.cpu().numpy(), z[f"{tag}_maxr"])
    np.testing.assert_array_equal(m._deformation_table.cpu().numpy(), z[f"{tag}_table"])


def test_densify_prune_reset_match_the_reference_gaussian_model():
    z = np.load(os.path.join(G, "s3g_surgery.npz"))
    m = _model_from(z, "in")
    ns = z["normal_z"].shape[0] // 2
    samples = torch.tensor(z["normal_z"]).view(2, ns, 3)                       # the reference's draw: repeat(N=2) order = [replica][selected]
    n_keep, n_clone, n_split = m.densify(float(z["max_grad"]), 0.005, float(z["extent"]), None, 5, 5, samples=samples)
    assert n_split == ns and n_clone > 0 and n_keep + n_clone + 2 * n_split == z["dens_xyz"].shape[0]
    _check(m, z, "dens")
    # statistics accumulated between the events, then prune (opacity, screen size and world size criteria all fire)
    m.max_radii2D = torch.tensor(z["prune_in_maxr"]).to(DEV)
    m.xyz_gradient_accum = torch.tensor(z["prune_in_accum"]).to(DEV)
    m.denom = torch.tensor(z["prune_in_denom"]).to(DEV)
    m.prune(float(z["max_grad"]), float(z["min_opacity"]), float(z["prune_extent"]), float(z["max_screen_size"]))
    assert m._xyz.shape[0] == z["prune_xyz"].shape[0] < z["dens_xyz"].shape[0]
    _check(m, z, "prune")
    m.reset_opacity()
    for n in NAMES:
        p = getattr(m, m._ATTR[n])
        tol = dict(rtol=2e-6, atol=2e-6) if n in ("opacity", "xyz", "scaling") else dict(rtol=0, atol=0)     # (xyz / scaling: the split's fp32 arithmetic)
        np.testing.assert_allclose(p.detach().cpu().numpy(), z[f"reset_{n}"], err_msg=n, **tol)
        np.testing.assert_array_equal(m.optimizer.state[p]["exp_avg"].cpu().numpy(), z[f"reset_m_{n}"])
        np.testing.assert_array_equal(m.optimizer.state[p]["exp_avg_sq"].cpu().numpy(), z[f"reset_v_{n}"])
    # the optimiser still steps on the new leaves
    for n in NAMES:
        p = getattr(m, m._ATTR[n])
        p.grad = torch.ones_like(p)
    before = m._xyz.detach().clone()
    m.update_learning_rate(3)
    m.optimizer.step()
    assert not torch.equal(before, m._xyz.detach())
    cap = m.capture()
    assert len(cap) == int(z["capture_len"]) == 16


def test_philox_split_samples_are_rank_independent_and_standard_normal():
    from emd_amd.gaussian_model import GaussianModel
    g = torch.Generator().manual_seed(5)
    N = 60000

    def build(seed):
        m = GaussianModel(device=DEV, densify_seed=seed)
        m.create_from_tensors(torch.zeros(N, 3), torch.rand(N, 3, generator=torch.Generator().manual_seed(1)), torch.zeros(N, 1), spatial_lr_scale=1.0)
        m.training_setup(_train_args())
        m.xyz_gradient_accum = torch.ones(N, 1, device=DEV)                 # every Gaussian is "hot" and large (scale 1): all are split
        m.denom = torch.ones(N, 1, device=DEV)
        return m
    a, b, c = build(11), build(11), build(12)
    for m in (a, b, c):
        k, cl, sp = m.densify(0.5, 0.005, 1.0, None)
        assert (k, cl, sp) == (0, 0, N)
    assert torch.equal(a._xyz, b._xyz), "same seed (another rank): identical samples without communication"
    assert not torch.equal(a._xyz, c._xyz)
    x = a._xyz.detach()                                                       # xyz = R(identity) (1 * n) + 0 = n
    assert abs(float(x.mean())) < 0.01 and abs(float(x.std()) - 1.0) < 0.01
    assert abs(float((x[:N] * x[N:]).mean())) < 0.01                          # the two replicas of a Gaussian are independent draws
    assert torch.allclose(a._scaling, torch.full_like(a._scaling, float(np.log(1 / 1.6))), atol=1e-6)


def test_training_loop_densifies_and_prunes_unattended():
    """A config-5-style loop at small size: render -> L1 -> backward -> statistics -> Adam, with a densification every 50 steps, a
    prune every 100 and an opacity reset at 150 -- no host-side masks, the point count moves, the loss falls."""
    from emd_amd import GaussianRasterizationSettings, GaussianRasterizer, RasterOptions, scenes
    from emd_amd.gaussian_model import GaussianModel
    from emd_amd.model import l1_loss
    torch.manual_seed(0)
    H, W, N = 96, 128, 6000
    sc = scenes.make_static_scene(N, seed=3)
    cam = scenes.small_camera(H, W)
    means = sc.means.clone()
    means[:, 0] = means[:, 0] * 0.25 + 1.0
    means[:, 1] *= 0.3
    means[:, 2] = means[:, 2] * 0.3 + 1.0
    m = GaussianModel(device=DEV, densify_seed=1)
    m.create_from_tensors(means, torch.rand(N, 3), sc.log_scales + 1.0, spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    m.training_setup(_train_args(position_lr_init=1.6e-3))
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(9)).to(DEV) * 0.5 + 0.25
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=DEV), 1.0, cam.world_view_transform.to(DEV),
                                       cam.full_proj_transform.to(DEV), 3, cam.camera_center.to(DEV), False, False)
    opts = RasterOptions(compute_normal=False)
    counts, losses = [m._xyz.shape[0]], []
    for it in range(1, 201):
        m.update_learning_rate(it)
        sp = torch.zeros_like(m._xyz, requires_grad=True)
        img, _, _, _, radii, _ = GaussianRasterizer(rs, options=opts)(means3D=m._xyz, means2D=sp, shs=m.get_features, opacities=m._opacity,
                                                                      scales=m._scaling, rotations=m._rotation, raw_params=True)
        loss = l1_loss(img, target)
        loss.backward()
        losses.append(float(loss))
        with torch.no_grad():
            m.add_densification_stats(sp.grad, radii)
            if it % 50 == 0:
                m.densify(2e-4, 0.005, 4.0, None)
            if it % 100 == 0:
                m.prune(2e-4, 0.005, 4.0, 20)
            if it == 150:
                m.reset_opacity()
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=True)
        counts.append(m._xyz.shape[0])
    assert len(set(counts)) >= 3, counts[::25]                                  # grew and shrank
    assert all(np.isfinite(losses)) and np.mean(losses[-10:]) < np.mean(losses[:10])
    n = m._xyz.shape[0]
    for t in (m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation, m._embedding, m.xyz_gradient_accum, m.denom, m.max_radii2D,
              m._deformation_table):
        assert t.shape[0] == n
    for g_ in m.optimizer.param_groups:
        st = m.optimizer.state.get(g_["params"][0])
        if st:                                                   # (the embedding never receives a gradient in this loop: no Adam state)
            assert st["exp_avg"].shape == g_["params"][0].shape and st["exp_avg_sq"].shape == g_["params"][0].shape
    assert m.optimizer.state.get(m._xyz) and m.optimizer.state[m._xyz]["exp_avg"].shape == m._xyz.shape


def test_ply_and_checkpoint_round_trip(tmp_path):
    from emd_amd.gaussian_model import GaussianModel, read_ply
    z = np.load(os.path.join(G, "s3g_surgery.npz"))
    m = _model_from(z, "in")
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    m.save_ply(path)
    d = read_ply(path)
    assert list(d.keys()) == [str(a) for a in z["attributes"]]                # the reference's construct_list_of_attributes()
    m2 = GaussianModel(device=DEV)
    m2.load_ply(path)
    for n in NAMES:
        assert torch.equal(getattr(m2, m2._ATTR[n]).detach(), getattr(m, m._ATTR[n]).detach()), n
    assert m2.active_sh_degree == 3
    ck = str(tmp_path / "chkpnt.pth")
    torch.save((m.capture(), 123), ck)
    model_args, it = torch.load(ck, weights_only=False)
    m3 = GaussianModel(device=DEV)
    m3.restore(model_args, _train_args(percent_dense=float(z["percent_dense"])))
    assert it == 123
    for n in NAMES:
        p, q = getattr(m3, m3._ATTR[n]), getattr(m, m._ATTR[n])
        assert torch.equal(p.detach(), q.detach())
        assert torch.equal(m3.optimizer.state[p]["exp_avg"], m.optimizer.state[q]["exp_avg"])
    assert torch.equal(m3.xyz_gradient_accum, m.xyz_gradient_accum)


def test_graph_replayed_training_loop_with_density_control():
    """The same unattended loop with every iteration REPLAYED from a hipGraph (emd_amd.StepGraphs: render -> L1 -> backward -> statistics ->
    capturable Adam recorded once), learning rates uploaded by update_learning_rate, and the graphs released and recorded again around
    every densification / prune / opacity reset (the point count and the parameter tensors change there).  The optimiser's step counts
    and moments survive the surgery, the point count moves, the loss falls."""
    from emd_amd import GaussianRasterizationSettings, GaussianRasterizer, RasterOptions, StepGraphs, scenes
    from emd_amd.gaussian_model import GaussianModel
    from emd_amd.model import l1_loss
    torch.manual_seed(0)
    H, W, N = 96, 128, 6000
    sc = scenes.make_static_scene(N, seed=3)
    cam = scenes.small_camera(H, W)
    means = sc.means.clone()
    means[:, 0] = means[:, 0] * 0.25 + 1.0
    means[:, 1] *= 0.3
    means[:, 2] = means[:, 2] * 0.3 + 1.0
    m = GaussianModel(device=DEV, densify_seed=1)
    m.create_from_tensors(means, torch.rand(N, 3), sc.log_scales + 1.0, spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    m.training_setup(_train_args(position_lr_init=1.6e-3, capturable_optimizer=True))
    assert m.optimizer.capturable
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(9)).to(DEV) * 0.5 + 0.25
    rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=DEV), 1.0, cam.world_view_transform.to(DEV),
                                       cam.full_proj_transform.to(DEV), 3, cam.camera_center.to(DEV), False, False)
    opts = RasterOptions(compute_normal=False, no_sync=True, capacity_hint=3_000_000)
    loss_buf = torch.zeros((), device=DEV)
    state = {}

    def iteration(_key):
        m.optimizer.zero_grad(set_to_none=True)
        sp = state["sp"]
        sp.grad = None
        img, _, _, _, radii, _ = GaussianRasterizer(rs, options=opts)(means3D=m._xyz, means2D=sp, shs=m.get_features, opacities=m._opacity,
                                                                      scales=m._scaling, rotations=m._rotation, raw_params=True)
        loss = l1_loss(img, target)
        loss.backward()
        with torch.no_grad():
            m.add_densification_stats(sp.grad, radii)
            m.optimizer.step()
            loss_buf.copy_(loss.detach())

    def record():
        state["sp"] = torch.zeros_like(m._xyz, requires_grad=True)
        return StepGraphs(iteration, [0], optimizers=[m.optimizer], warmup=1)       # (one eager iteration, then the recording)
    graphs = record()
    counts, losses, it = [m._xyz.shape[0]], [], 1
    while it <= 200:
        m.update_learning_rate(it)                 # host-side schedule -> the groups' device-resident rates
        graphs.replay(0)
        losses.append(loss_buf.clone())
        it += 1
        if it % 50 == 0 or it == 150:
            graphs.release()
            with torch.no_grad():
                if it % 50 == 0:
                    m.densify(2e-4, 0.005, 4.0, None)
                if it % 100 == 0:
                    m.prune(2e-4, 0.005, 4.0, 20)
                if it == 150:
                    m.reset_opacity()
            graphs = record()
            it += 1                                # (the recorder's eager iteration)
        counts.append(m._xyz.shape[0])
    losses = [float(x) for x in losses]
    assert len(set(counts)) >= 3, counts[::25]
    assert all(np.isfinite(losses)) and np.mean(losses[-10:]) < np.mean(losses[:10])
    st = m.optimizer.state[m._xyz]
    assert st["exp_avg"].shape == m._xyz.shape and st["step"].device.type == "cuda" and float(st["step"]) >= 195.0
    lr_dev = float(m.optimizer._lr_dev[0])
    assert abs(lr_dev - m.optimizer.param_groups[0]["lr"]) <= 1e-9 + 1e-6 * lr_dev      # the schedule reached the device copy


def test_one_graph_for_all_views_equals_the_graphs_per_view():
    """StepGraphs(..., inputs=StepInputs(...)): ONE recorded step serves every view of a rig / clip -- camera block, field-of-view tangents,
    background and frame index come from device tables through the graph's first node (emd_select_step_inputs) -- where round 3 recorded
    one graph per view.  Ten views: every view's replay of the one graph leaves the gradients, the image and the densification statistics
    the per-view graphs leave (image bit for bit), in any order of replays; then an unattended loop over the views with density control
    every 40 iterations re-records ONE graph per event (S3Gaussian/train.py:203-229,404-423)."""
    from emd_amd import GaussianRasterizer, RasterOptions, StepGraphs, StepInputs, scenes
    from emd_amd.gaussian_model import GaussianModel
    from emd_amd.model import l1_loss, raster_settings_for
    torch.manual_seed(0)
    H, W, N, NV = 96, 128, 6000, 10
    sc = scenes.make_static_scene(N, seed=3)
    means = sc.means.clone()
    means[:, 0] = means[:, 0] * 0.25 + 1.0
    means[:, 1] *= 0.3
    means[:, 2] = means[:, 2] * 0.3 + 1.0
    # (two focal lengths in the rig: the field-of-view tangents must come from the selected row too)
    cams = [scenes.small_camera(H, W, yaw=-30.0 + 7.0 * v, focal=(1.0625 if v % 2 else 0.9) * W) for v in range(NV)]
    bg = torch.tensor([0.05, 0.1, 0.15])
    targets = (torch.rand(NV, 3, H, W, generator=torch.Generator().manual_seed(9)) * 0.5 + 0.25).to(DEV)

    def build():
        m = GaussianModel(device=DEV, densify_seed=1)
        m.create_from_tensors(means, torch.rand(N, 3, generator=torch.Generator().manual_seed(4)), sc.log_scales + 1.0, spatial_lr_scale=1.0)
        m.active_sh_degree = 3
        m.training_setup(_train_args(position_lr_init=1.6e-3, capturable_optimizer=True))
        return m
    opts = RasterOptions(compute_normal=False, no_sync=True, capacity_hint=3_000_000)
    out = {}
    rs_of = []                                       # the per-view graphs' settings: uploaded before any capture
    for c in cams:
        rs = raster_settings_for(c, bg.to(DEV), 3)
        rs_of.append(rs._replace(viewmatrix=rs.viewmatrix.to(DEV), projmatrix=rs.projmatrix.to(DEV), campos=rs.campos.to(DEV)))

    def make_iteration(m, state, optimise):
        def iteration(view):
            """view: a row index (per-view graphs: the camera is a host constant of the capture) or the StepInputs (one graph)"""
            m.optimizer.zero_grad(set_to_none=True)
            sp = state["sp"]
            sp.grad = None
            if isinstance(view, StepInputs):
                rs = raster_settings_for(view.camera, view.bg, 3)
                target = targets.index_select(0, view.frame.long())[0]      # (the frame table names the view's target image)
            else:
                rs, target = rs_of[view], targets[view]
            img, _, _, _, radii, _ = GaussianRasterizer(rs, options=opts)(means3D=m._xyz, means2D=sp, shs=m.get_features, opacities=m._opacity,
                                                                          scales=m._scaling, rotations=m._rotation, raw_params=True)
            loss = l1_loss(img, target)
            loss.backward()
            with torch.no_grad():
                m.add_densification_stats(sp.grad, radii)
                if optimise:
                    m.optimizer.step()
                out["img"], out["loss"] = img.detach(), loss.detach()
                # (every recorded graph has its OWN output and gradient tensors in the shared pool: `p.grad` names those of the last capture)
                out[view if not isinstance(view, StepInputs) else "one"] = (img.detach(), m._xyz.grad, m._features_dc.grad, m._opacity.grad, sp.grad)
        return iteration
    # ---- (a) one graph against ten: same image (bit for bit), same gradients, for every view, replayed out of order
    m = build()
    state = {"sp": torch.zeros_like(m._xyz, requires_grad=True)}
    it = make_iteration(m, state, optimise=False)
    per_view = StepGraphs(it, list(range(NV)), warmup=1)
    want = {}
    for v in (3, 0, 9, 5, 1, 2, 4, 6, 7, 8):
        per_view.replay(v)
        want[v] = tuple(t.clone() for t in out[v])
    per_view.release()
    inputs = StepInputs(cams, bg, frames=list(range(NV)), device=DEV)
    one = StepGraphs(it, list(range(NV)), warmup=1, inputs=inputs)
    assert len(one.graphs) == 1
    for v in (7, 7, 2, 9, 0, 4, 1, 3, 5, 6, 8, 2):
        one.replay(v)
        got = out["one"]
        assert torch.equal(got[0], want[v][0]), v
        for a_, b_ in zip(got[1:], want[v][1:]):
            assert float((a_ - b_).abs().max()) <= 1e-5 * max(float(b_.abs().max()), 1e-12), v
    assert int(inputs.frame) == 2
    one.release()
    # ---- (b) the unattended loop on ONE graph: re-recorded once per density-control event
    m = build()
    state = {}

    def record():
        state["sp"] = torch.zeros_like(m._xyz, requires_grad=True)
        return StepGraphs(make_iteration(m, state, optimise=True), list(range(NV)), optimizers=[m.optimizer], warmup=1, inputs=inputs)
    graphs, captures = record(), 1
    counts, losses = [m._xyz.shape[0]], []
    for i in range(1, 161):
        m.update_learning_rate(i)
        graphs.replay((i * 7) % NV)
        losses.append(out["loss"].clone())
        if i % 40 == 0:
            graphs.release()
            with torch.no_grad():
                m.densify(2e-4, 0.005, 4.0, None)
                if i % 80 == 0:
                    m.prune(2e-4, 0.005, 4.0, 20)
            graphs, captures = record(), captures + 1
        counts.append(m._xyz.shape[0])
    losses = [float(x) for x in losses]
    assert captures == 5 and len(set(counts)) >= 3, (captures, counts[::20])
    assert all(np.isfinite(losses)) and np.mean(losses[-20:]) < np.mean(losses[:20])


def test_segmented_step_graphs_are_cut_inside_backward():
    """StepGraphs(..., segmented=True): the recorded step calls `graphs.cut` from RasterCall.on_sh_factor -- i.e. from autograd's device thread,
    between the render backward and the projection backward -- and is recorded as TWO graphs; `replay(key, between=fn)` runs fn on the host
    between them (where a view-parallel loop issues the SH-factor gathers, which then travel under the projection backward of a REPLAYED
    step: DESIGN section 7).  What the pair leaves -- image, factor, every gradient of the slab -- equals the one-graph step, for three views
    of one StepInputs, and the colour factor is complete when `between` runs."""
    from emd_amd import GaussianRasterizer, RasterCall, RasterOptions, StepGraphs, StepInputs, scenes
    from emd_amd.model import l1_loss, raster_settings_for
    torch.manual_seed(0)
    H, W, N, NV = 96, 128, 5000, 3
    sc = scenes.make_static_scene(N, seed=5)
    means = sc.means.clone()
    means[:, 0] = means[:, 0] * 0.25 + 1.0
    means[:, 1] *= 0.3
    means[:, 2] = means[:, 2] * 0.3 + 1.0
    cams = [scenes.small_camera(H, W, yaw=-20.0 + 15.0 * v, focal=W) for v in range(NV)]
    bg = torch.tensor([0.05, 0.1, 0.15])
    targets = (torch.rand(NV, 3, H, W, generator=torch.Generator().manual_seed(2)) * 0.5 + 0.25).to(DEV)
    P = {"xyz": means.to(DEV).requires_grad_(True), "shs": (torch.randn(N, 16, 3, generator=torch.Generator().manual_seed(3)) * 0.2).to(DEV).requires_grad_(True),
         "opacity": torch.full((N, 1), 0.5, device=DEV).requires_grad_(True), "scaling": (sc.log_scales + 1.0).exp().to(DEV).requires_grad_(True),
         "rotation": torch.nn.functional.normalize(torch.randn(N, 4, generator=torch.Generator().manual_seed(6)), dim=1).to(DEV).requires_grad_(True)}
    opts = RasterOptions(compute_normal=False, no_sync=True, capacity_hint=2_000_000, factored_sh_grad=True)
    inputs = StepInputs(cams, bg, frames=list(range(NV)), device=DEV)
    out, holder = {}, {}

    def step(view):
        for p in P.values():
            p.grad = None
        rs = raster_settings_for(view.camera, view.bg, 3)
        rec = RasterCall()
        rec.on_sh_factor = holder["graphs"].cut if holder.get("graphs") is not None else None
        img = GaussianRasterizer(rs, options=opts)(means3D=P["xyz"], means2D=torch.zeros_like(P["xyz"], requires_grad=True), shs=P["shs"],
                                                   opacities=P["opacity"], scales=P["scaling"], rotations=P["rotation"], record=rec)[0]
        l1_loss(img, targets.index_select(0, view.frame.long())[0]).backward()
        out["img"], out["rec"] = img.detach(), rec
        out["grads"] = (P["xyz"].grad, P["opacity"].grad, P["scaling"].grad, P["rotation"].grad)

    # a proxy the step can name before the StepGraphs object exists (its constructor already runs the step)
    class _Cut:
        target = None
        def cut(self, *a):
            if self.target is not None:
                self.target.cut(*a)
    proxy = _Cut()
    holder["graphs"] = None
    one = StepGraphs(step, list(range(NV)), warmup=1, inputs=inputs)
    want = {}
    for v in (2, 0, 1):
        one.replay(v)
        want[v] = (out["img"].clone(), out["rec"].sh_color_grad.clone()) + tuple(g.clone() for g in out["grads"])
    one.release()
    holder["graphs"] = proxy
    two = StepGraphs.__new__(StepGraphs)
    proxy.target = two
    two.__init__(step, list(range(NV)), warmup=1, inputs=inputs, segmented=True)
    assert two.segments() == 2
    seen = []
    for v in (1, 2, 0, 2):
        def between(i, v=v):
            # host code between the graphs: the colour factor of THIS view is complete on the stream (a collective issued here reads it)
            seen.append((v, i, float((out["rec"].sh_color_grad - want[v][1]).abs().max())))
        two.replay(v, between=between)
        torch.cuda.synchronize()
        got = (out["img"], out["rec"].sh_color_grad) + tuple(out["grads"])
        assert torch.equal(got[0], want[v][0])                     # the image bit for bit
        for a, b in zip(got[1:], want[v][1:]):                     # (gradients are sums of float atomics: equal up to their order; the
            # rotation chain amplifies a one-ulp move of its inputs -- DESIGN section 5 -- and sat at 1.1e-5 of the largest entry once)
            assert float((a - b).abs().max()) <= 5e-5 * float(b.abs().max()) + 1e-12
    assert [s_[:2] for s_ in seen] == [(1, 0), (2, 0), (0, 0), (2, 0)]
    assert all(s_[2] <= 1e-5 * float(want[s_[0]][1].abs().max()) + 1e-12 for s_ in seen)
    two.release()


def test_density_control_on_the_bench_parameter_store_leaves_the_actors_alone():
    """emd_amd.model.density_control (bench.py --config 4's event, S3Gaussian/train.py:404-423 -> gaussian_model.py:442-556 on a StreetGaussians store): the
    background Gaussians are cloned / split / pruned by the device-side engine exactly as a GaussianModel holding only them would be; the actors' points
    (in front, contiguous, at their fixed budget) keep their values, ids and order; the same call on identical inputs gives identical parameters (Philox)."""
    from emd_amd import scenes
    from emd_amd.gaussian_model import GaussianModel
    from emd_amd.model import StreetGaussians, density_control
    N, A, P = 30000, 3, 2000
    sc = scenes.add_actors(scenes.make_static_scene(N, seed=2), num_actors=A, pts_per_actor=P, num_frames=4, seed=1)
    g = torch.Generator().manual_seed(4)
    accum, denom, radii = torch.rand(N, 1, generator=g) * 4e-4, torch.randint(0, 3, (N, 1), generator=g).float(), torch.rand(N, generator=g) * 30
    n_dyn = A * P

    def run():
        m = StreetGaussians(sc, DEV)
        before = {k: getattr(m, k).detach().clone() for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features")}
        ev = density_control(m, accum.to(DEV), denom.to(DEV), radii.to(DEV), max_grad=2e-4, min_opacity=0.005, extent=4.0, percent_dense=0.01, seed=5, event=2)
        return m, before, ev
    m, before, ev = run()
    assert ev["n_before"] == N and ev["n_after"] == N + ev["cloned"] + ev["split"] - ev["pruned"] and ev["cloned"] + ev["split"] > 0
    assert m._xyz.shape[0] == ev["n_after"] == m.actor_id.shape[0]
    for k, v in before.items():                                   # the actors' rows: untouched, still in front
        assert torch.equal(getattr(m, k).detach()[:n_dyn], v[:n_dyn]), k
    assert bool((m.actor_id[:n_dyn] >= 0).all()) and bool((m.actor_id[n_dyn:] == -1).all())
    # the background against a GaussianModel that holds only the background
    ref = GaussianModel(device=DEV, densify_seed=5)
    ref.densify_events = 2
    Pm = lambda t: torch.nn.Parameter(t[n_dyn:].to(DEV).contiguous())
    ref._xyz, ref._scaling, ref._rotation, ref._opacity = Pm(sc.means), Pm(sc.log_scales), Pm(sc.quats), Pm(sc.opacity_logits)
    ref._features_dc, ref._features_rest = Pm(sc.shs[:, :1]), Pm(sc.shs[:, 1:])
    ref._embedding = torch.nn.Parameter(torch.zeros(N - n_dyn, 4, device=DEV))
    ref._deformation_table = torch.ones(N - n_dyn, dtype=torch.bool, device=DEV)
    ref.xyz_gradient_accum, ref.denom, ref.max_radii2D = accum[n_dyn:].to(DEV).contiguous(), denom[n_dyn:].to(DEV).contiguous(), radii[n_dyn:].to(DEV).contiguous()
    ref.percent_dense = 0.01
    with torch.no_grad():
        ref.densify(2e-4, 0.005, 4.0, None)
        ref.prune(2e-4, 0.005, 4.0, None)
    assert torch.equal(m._xyz.detach()[n_dyn:], ref._xyz.detach()) and torch.equal(m._scaling.detach()[n_dyn:], ref._scaling.detach())
    assert torch.equal(m._features.detach()[n_dyn:], torch.cat([ref._features_dc, ref._features_rest], 1).detach())
    m2, _, ev2 = run()                                            # a second replica: the same event, bit for bit
    assert ev2 == ev and torch.equal(m2._xyz.detach(), m._xyz.detach()) and torch.equal(m2._rotation.detach(), m._rotation.detach())


@pytest.mark.parametrize("opt_kind", ["hip_adam", "torch_adam"])
def test_density_control_carries_the_callers_optimizer_state(opt_kind):
    """density_control(..., optimizer=...): the optimiser's groups point at the NEW parameters and both Adam moments travel with the rows -- a survivor's
    moments are its old ones (found again through its xyz, which a clone shares with its source: the FIRST row with that position is the survivor),
    new rows start at zero, the actors' rows are untouched -- as cat_tensors_to_optimizer / _prune_optimizer do (gaussian_model.py:454-500);
    a step of the optimiser afterwards works on the new tensors."""
    from emd_amd import scenes
    from emd_amd.model import StreetGaussians, density_control
    from emd_amd.optim import Adam
    N, A, P = 20000, 2, 1500
    n_dyn = A * P
    sc = scenes.add_actors(scenes.make_static_scene(N, seed=3), num_actors=A, pts_per_actor=P, num_frames=4, seed=1)
    m = StreetGaussians(sc, DEV)
    g = torch.Generator().manual_seed(9)
    names = {"xyz": "_xyz", "f": "_features", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}
    opt = (Adam if opt_kind == "hip_adam" else torch.optim.Adam)([{"params": [getattr(m, a)], "lr": 1e-3, "name": n} for n, a in names.items()], lr=0.0, eps=1e-15)
    for _ in range(2):
        for a in names.values():
            p = getattr(m, a)
            p.grad = (torch.randn(p.shape, generator=g) * 1e-2).to(DEV)
        opt.step()
    old = {n: (getattr(m, a).detach().clone(), opt.state[getattr(m, a)]["exp_avg"].clone(), opt.state[getattr(m, a)]["exp_avg_sq"].clone()) for n, a in names.items()}
    accum, denom, radii = (torch.rand(N, 1, generator=g) * 4e-4).to(DEV), torch.randint(0, 3, (N, 1), generator=g).float().to(DEV), torch.zeros(N, device=DEV)
    ev = density_control(m, accum, denom, radii, max_grad=2e-4, min_opacity=0.005, extent=4.0, percent_dense=0.01, seed=1, event=0, optimizer=opt)
    M = ev["n_after"]
    assert ev["cloned"] > 0 and M != N
    # which old row every new row came from: survivors keep their xyz (split samples move, clones share their source's: the first hit is the survivor)
    new_xyz, old_xyz = m._xyz.detach(), old["xyz"][0]
    n_keep = N - ev["split"] - ev["pruned"] if ev["pruned"] == 0 else None
    for n, a in names.items():
        p = getattr(m, a)
        grp = [g_ for g_ in opt.param_groups if g_["name"] == n][0]
        assert grp["params"][0] is p and p.shape[0] == M and p in opt.state
        st = opt.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
        assert torch.equal(st["exp_avg"][:n_dyn], old[n][1][:n_dyn]) and torch.equal(st["exp_avg_sq"][:n_dyn], old[n][2][:n_dyn])      # the actors' rows
    # survivors are the leading background rows in their old order (minus the split / pruned ones): match by position
    keep_mask = torch.zeros(N, dtype=torch.bool, device=DEV)
    key_old = {tuple(r): i for i, r in enumerate(old_xyz[n_dyn:].cpu().tolist())}
    rows_new = new_xyz[n_dyn:].cpu().tolist()
    seen, survivors = set(), []
    for j, r in enumerate(rows_new):
        i = key_old.get(tuple(r))
        if i is not None and i not in seen:
            seen.add(i)
            survivors.append((j, i))
    assert len(survivors) >= (N - n_dyn) - ev["split"] - ev["pruned"] - 1
    jj = torch.tensor([n_dyn + j for j, _ in survivors], device=DEV)
    ii = torch.tensor([n_dyn + i for _, i in survivors], device=DEV)
    for n, a in names.items():
        st = opt.state[getattr(m, a)]
        assert torch.equal(st["exp_avg"][jj], old[n][1][ii]) and torch.equal(st["exp_avg_sq"][jj], old[n][2][ii]), n
        rest = torch.ones(M, dtype=torch.bool, device=DEV)
        rest[jj] = False
        rest[:n_dyn] = False
        assert float(st["exp_avg"][rest].abs().max()) == 0.0 and float(st["exp_avg_sq"][rest].abs().max()) == 0.0, n          # new rows start at zero
    for a in names.values():
        p = getattr(m, a)
        p.grad = torch.ones_like(p) * 1e-3
    before = m._xyz.detach().clone()
    opt.step()
    assert not torch.equal(before, m._xyz.detach())
