"""-m gpu: re-entrancy and host-synchronisation behaviour of the drop-in boundary.

  - options belong to the rasterizer instance and results to the call's record: two rasterizers on two streams plus three
    calls of one object in one autograd graph (absgrad on one of them) must give exactly the single-call results;
  - with device-resident camera settings (what the reference passes: `.cuda()` tensors, gaussian_renderer/__init__.py:54-59) and
    `no_sync`, a forward + backward performs NO stream synchronisation and no device-to-host copy: run under
    torch.cuda.set_sync_debug_mode("error"), which raises on any synchronising call."""
import numpy as np
import pytest
import torch

from tests.helpers import make_case

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _settings(case, dev=DEV):
    from emd_amd import GaussianRasterizationSettings
    cam = case["cam"]
    return GaussianRasterizationSettings(image_height=case["H"], image_width=case["W"], tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                         bg=case["bg"].to(dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
                                         projmatrix=cam.full_proj_transform.to(dev), sh_degree=case["sh_degree"],
                                         campos=cam.camera_center.to(dev), prefiltered=False, debug=False)


def _leaves(case):
    t = {k: case[k].to(DEV).clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    t["means2D"] = torch.zeros(case["N"], 3, device=DEV, requires_grad=True)
    return t


def _call(rast, t, **kw):
    return rast(means3D=t["means3D"], means2D=t["means2D"], shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
                scales=t["scales"], rotations=t["rotations"], cov3Ds_precomp=None, extra_attrs=None, **kw)


def test_two_streams_and_three_calls_in_one_graph_with_absgrad_on_one():
    from emd_amd import GaussianRasterizer, RasterCall
    case_a, case_b = make_case(n=6000, H=80, W=112, seed=3), make_case(n=5000, H=64, W=96, seed=4, yaw=20.0)
    rs_a, rs_b = _settings(case_a), _settings(case_b)
    gen = torch.Generator().manual_seed(1)
    Ga = [torch.randn(3, case_a["H"], case_a["W"], generator=gen).to(DEV) for _ in range(3)]
    Gb = torch.randn(3, case_b["H"], case_b["W"], generator=gen).to(DEV)

    # single-call references
    def single(rs, case, G, **opt):
        t = _leaves(case)
        r = GaussianRasterizer(rs, **opt)
        img = _call(r, t)[0]
        (img * G).sum().backward()
        return img.detach(), {k: v.grad.clone() for k, v in t.items()}, r.last_call
    ref_imgs, ref_grads = [], None
    for w in range(3):
        img, gr, call = single(rs_a, case_a, Ga[w], absgrad=(w == 1))
        ref_imgs.append(img)
        ref_grads = gr if ref_grads is None else {k: ref_grads[k] + gr[k] for k in gr}
        if w == 1:
            ref_abs = call.absgrad.clone()
    ref_b_img, ref_b_grads, _ = single(rs_b, case_b, Gb)

    # now: stream 1 runs three calls of ONE rasterizer object pair in one graph, stream 2 another rasterizer, interleaved
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    ta, tb = _leaves(case_a), _leaves(case_b)
    plain, with_abs = GaussianRasterizer(rs_a), GaussianRasterizer(rs_a, absgrad=True)
    other = GaussianRasterizer(rs_b)
    recs = [RasterCall() for _ in range(3)]
    with torch.cuda.stream(s1):
        i0 = _call(plain, ta, record=recs[0])[0]
    with torch.cuda.stream(s2):
        ib = _call(other, tb)[0]
    with torch.cuda.stream(s1):
        i1 = _call(with_abs, ta, record=recs[1])[0]
        i2 = _call(plain, ta, record=recs[2])[0]
        (sum((img * Ga[w]).sum() for w, img in enumerate((i0, i1, i2)))).backward()
    with torch.cuda.stream(s2):
        (ib * Gb).sum().backward()
    torch.cuda.synchronize()
    for w, img in enumerate((i0, i1, i2)):
        assert torch.equal(img.detach(), ref_imgs[w]), w
    assert torch.equal(ib.detach(), ref_b_img)
    for k, v in ta.items():
        err = float((v.grad - ref_grads[k]).abs().max())
        assert err <= 1e-4 * max(float(ref_grads[k].abs().max()), 1e-12), (k, err)
    for k, v in tb.items():
        err = float((v.grad - ref_b_grads[k]).abs().max())
        assert err <= 1e-5 * max(float(ref_b_grads[k].abs().max()), 1e-12), (k, err)
    # absgrad: only the call that asked for it has it, and it is that call's own
    assert recs[0].absgrad is None and recs[2].absgrad is None
    err = float((recs[1].absgrad - ref_abs).abs().max())
    assert err <= 1e-5 * float(ref_abs.abs().max()), err
    assert plain.last_call is recs[2] and with_abs.last_call is recs[1]


def test_reference_call_site_with_device_settings_never_synchronises():
    """S3Gaussian/gaussian_renderer/__init__.py:49-62 builds the settings from CUDA tensors.  With `no_sync` such a call (forward and
    backward, main pass + a colors_precomp feature pass as at :145-201) must not synchronise: sync-debug mode raises if it does."""
    from emd_amd import GaussianRasterizer, rasterizer
    case = make_case(n=8000, H=96, W=128, seed=11)
    rs = _settings(case)                       # bg / viewmatrix / projmatrix / campos live on the device
    t = _leaves(case)
    feat = torch.rand(case["N"], 3, device=DEV, requires_grad=True)
    G = torch.randn(3, case["H"], case["W"], device=DEV)
    # a synchronising warm-up call sizes the binning workspace (capacity hint) and builds the reference result
    warm = GaussianRasterizer(rs, no_sync=False)
    ref = _call(warm, t)[0].detach().clone()
    D = warm.last_status()["num_rendered"]
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        rast = GaussianRasterizer(rs, no_sync=True)
        img = _call(rast, t)[0]
        img_f = rast(means3D=t["means3D"], means2D=t["means2D"], shs=None, colors_precomp=feat, opacities=t["opacities"],
                     scales=t["scales"], rotations=t["rotations"], cov3Ds_precomp=None, extra_attrs=None)[0]
        ((img * G).sum() + (img_f * G).sum()).backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert torch.equal(img.detach(), ref)
    assert rast.last_call.num_rendered == -1 and rast.last_call.settings_dev is not None      # count never read, settings by pointer
    assert rast.last_call.last_status()["num_rendered"] == D
    assert t["means3D"].grad is not None and feat.grad is not None and torch.isfinite(t["means3D"].grad).all()


def test_no_sync_overflow_is_reported_by_a_later_call_and_heals():
    """no_sync never waits for the duplicate count; an overflowing call yields a blank image.  Its status word is copied to pinned
    memory asynchronously and examined by a later forward: that call warns and raises the capacity hint, so the NEXT call is right."""
    import warnings
    from emd_amd import GaussianRasterizer, rasterizer
    case = make_case(n=6000, H=96, W=128, seed=41)
    rs = _settings(case)
    t = _leaves(case)
    good = _call(GaussianRasterizer(rs, no_sync=False), t)[0].detach().clone()
    key = (0, case["H"], case["W"])
    rasterizer._watch.pop(key, None)
    rasterizer._capacity_hint[key] = 64                       # far too small
    r = GaussianRasterizer(rs, no_sync=True, min_capacity=64)
    with torch.no_grad():
        blank = _call(r, t)[0]
        assert r.last_call.last_status()["overflow"] == 1 and not torch.equal(blank, good)
        torch.cuda.synchronize()                              # (only so that the pinned copy has certainly landed for the test)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            second = _call(r, t)[0]                           # polls the earlier status: warns + raises the hint; this call may still be short
            assert any("overflowed" in str(x.message) for x in w)
        torch.cuda.synchronize()
        third = _call(r, t)[0]
        if not torch.equal(second, good):
            torch.cuda.synchronize()
            third = _call(r, t)[0]
    assert torch.equal(third, good)


@pytest.mark.parametrize("n_extra", [1, 2])
def test_one_binning_many_colour_sets(n_extra):
    """The reference's fine stage calls the rasterizer three times with the same Gaussians: main pass, colors_precomp = coarse feat,
    colors_precomp = fine feat (gaussian_renderer/__init__.py:145-201).  `colors_extra` serves the feature passes from the main
    call's projection, sort and list walk: every extra image equals the separate call's image BIT FOR BIT (and the oracle's), and
    the gradients equal those of the three separate calls / the sum of three oracle backward passes."""
    from emd_amd import GaussianRasterizer
    from tests.helpers import run_oracle, assert_grad_close
    case = make_case(n=7000, H=96, W=128, seed=23)
    rs = _settings(case)
    gen = torch.Generator().manual_seed(2)
    feats = [torch.rand(case["N"], 3, generator=gen) for _ in range(n_extra)]
    G = [torch.randn(3, case["H"], case["W"], generator=gen) for _ in range(1 + n_extra)]
    Gd, Ga = 0.1 * torch.randn(1, case["H"], case["W"], generator=gen), torch.randn(1, case["H"], case["W"], generator=gen)
    # (a) separate calls, as the reference issues them
    t = _leaves(case)
    f_sep = [f.to(DEV).clone().requires_grad_(True) for f in feats]
    r = GaussianRasterizer(rs)
    img, depth, _, alpha, _, _ = _call(r, t)
    imgs_sep = [r(means3D=t["means3D"], means2D=t["means2D"], shs=None, colors_precomp=f, opacities=t["opacities"], scales=t["scales"],
                  rotations=t["rotations"], cov3Ds_precomp=None, extra_attrs=None)[0] for f in f_sep]
    loss = (img * G[0].to(DEV)).sum() + (depth * Gd.to(DEV)).sum() + (alpha * Ga.to(DEV)).sum() + sum((i * g.to(DEV)).sum() for i, g in zip(imgs_sep, G[1:]))
    loss.backward()
    # (b) one call
    t1 = _leaves(case)
    f_one = [f.to(DEV).clone().requires_grad_(True) for f in feats]
    r1 = GaussianRasterizer(rs)
    img1, depth1, _, alpha1, _, extra = _call(r1, t1, colors_extra=f_one)
    assert isinstance(extra, list) and len(extra) == n_extra
    assert torch.equal(img1, img) and torch.equal(depth1, depth) and torch.equal(alpha1, alpha)
    for a, b in zip(extra, imgs_sep):
        assert torch.equal(a.detach(), b.detach())
    loss1 = (img1 * G[0].to(DEV)).sum() + (depth1 * Gd.to(DEV)).sum() + (alpha1 * Ga.to(DEV)).sum() + sum((i * g.to(DEV)).sum() for i, g in zip(extra, G[1:]))
    loss1.backward()
    # (c) the oracle: three forward / backward passes, gradients of the shared inputs summed
    case_o = dict(case, dL_dcolor=G[0].numpy(), dL_ddepth=Gd.numpy(), dL_dalpha=Ga.numpy())
    orc = run_oracle(case_o, backward=True)
    total = {k: np.asarray(orc["grads"][k], np.float64).copy() for k in ("means3D", "means2D", "opacities", "scales", "rotations")}
    for k, f in enumerate(feats):
        case_f = dict(case, shs=None, colors_precomp=f, dL_dcolor=G[1 + k].numpy(), dL_ddepth=None, dL_dalpha=None)
        of = run_oracle(case_f, backward=True)
        got_img = extra[k].detach().cpu().numpy()
        assert int((got_img.view(np.uint32) != of["img"]["color"].view(np.uint32)).sum()) == 0, "extra image differs from the oracle's"
        assert_grad_close(f_one[k].grad.cpu().numpy(), of["grads"]["colors"], f"colors_extra[{k}]")
        assert_grad_close(f_one[k].grad.cpu().numpy(), f_sep[k].grad.cpu().numpy(), f"colors_extra[{k}] vs separate call")
        for name in total:
            total[name] += np.asarray(of["grads"][name], np.float64).reshape(total[name].shape)
    from tests.helpers import END2END_ATOL_FRAC, END2END_REL_L2
    for name in total:          # (scales / rotations: end-to-end floor of the ill-conditioned conic -> covariance chain, tests/helpers.py)
        loose = name in ("scales", "rotations")
        assert_grad_close(t1[name].grad.cpu().numpy(), total[name], name, atol_frac=END2END_ATOL_FRAC if loose else None, rel_l2=END2END_REL_L2 if loose else None)
    assert_grad_close(t1["shs"].grad.cpu().numpy(), orc["grads"]["shs"], "shs")


def test_step_replayed_from_a_hip_graph_equals_the_eager_step():
    """bench.py replays the whole step (pose table with learned track offsets -> fused-motion rasterizer -> L1 -> backward) from ONE
    captured hipGraph; camera, frame index and frame time are device-resident inputs selected by a device index.  A replay for
    another view must give exactly the eager step of that view: image bit for bit, every gradient."""
    import types
    from emd_amd import RasterOptions, scenes
    from emd_amd.model import StreetGaussians, l1_loss, render
    N, H, W, F = 30000, 96, 128, 6
    scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=4, pts_per_actor=2000, num_frames=F, seed=1)
    model = StreetGaussians(scene, DEV, track_heads=True)
    params = list(model.parameters())
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(DEV)
    bg = torch.zeros(3)
    views = [(f, c, scenes.rig_camera(f, c, H, W)) for f, c in ((0, 0), (3, 1), (5, 2), (2, 0))]
    opts = RasterOptions(no_sync=True)

    def eager(i):
        f, c, cam = views[i]
        for p in params:
            p.grad = None
        o = render(model, cam, bg, frame=f, iteration=0, options=opts)
        l1_loss(o["render"], target).backward()
        return o["render"].detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in params]
    # a synchronising call sizes the binning workspace for all views
    dmax = 0
    for f, c, cam in views:
        with torch.no_grad():
            o = render(model, cam, bg, frame=f, iteration=0, options=opts.replace(no_sync=False))
        dmax = max(dmax, o["raster_call"].last_status()["num_rendered"])
    opts.capacity_hint = int(dmax * 1.3) + 1024
    ref = [eager(i) for i in range(len(views))]
    blocks = torch.stack([torch.cat([bg, c_.world_view_transform.reshape(-1), c_.full_proj_transform.reshape(-1), c_.camera_center.reshape(-1)])
                          for _, _, c_ in views]).to(DEV)
    frame_of = torch.tensor([f for f, _, _ in views], dtype=torch.int32, device=DEV)
    sel = torch.zeros(1, dtype=torch.int64, device=DEV)
    cam0 = views[0][2]
    out = {}

    def body():
        for p in params:
            p.grad = None
        blk = blocks.index_select(0, sel)[0]
        cam_g = types.SimpleNamespace(image_height=H, image_width=W, tanfovx=cam0.tanfovx, tanfovy=cam0.tanfovy,
                                      world_view_transform=blk[3:19].view(4, 4), full_proj_transform=blk[19:35].view(4, 4), camera_center=blk[35:38])
        o = render(model, cam_g, blk[0:3], frame=frame_of.index_select(0, sel), iteration=0, options=opts)
        l1_loss(o["render"], target).backward()
        out["img"] = o["render"].detach()          # (no reference to the autograd graph may outlive the body: its AccumulateGrad nodes are bound to a stream)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        body()
    for i in (1, 3, 0, 2):
        sel.fill_(i)
        g.replay()
        torch.cuda.synchronize()
        img_ref, grads_ref = ref[i]
        assert torch.equal(out["img"].detach(), img_ref), f"view {i}: replayed image differs from the eager one"
        for p, gr in zip(params, grads_ref):
            if gr is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0
                continue
            err = float((p.grad - gr).abs().max())
            bad = ((p.grad - gr).abs() > 2e-5 * max(float(gr.abs().max()), 1e-12))
            names = [n for n, q in model.named_parameters() if q is p]
            assert err <= 2e-5 * max(float(gr.abs().max()), 1e-12), (i, names, tuple(p.shape), err, int(bad.sum()), bad.nonzero()[:5].tolist(),
                                                                     p.grad[bad][:5].tolist(), gr[bad][:5].tolist())


def test_accumulator_rows_are_handed_back_clean():
    """EMD_FLAG_BWD_WS_CLEAN (round 3): the binding keeps ONE accumulator workspace per (device, stream, size); the projection backward
    zeroes every row it reads, so consecutive backward passes -- of different views, with different visible sets, with an extra colour
    set whose gradient nobody asked for -- need no zero fill in between: gradients equal those of calls on a freshly cleared
    buffer (keep_render_grads=True takes that path) bit for bit, and the kept workspace is all zero after every backward."""
    from emd_amd import GaussianRasterizer
    from emd_amd import rasterizer as rz
    n = 7000
    cases = [make_case(n=n, H=80, W=112, seed=31), make_case(n=n, H=80, W=112, seed=31, yaw=35.0), make_case(n=n, H=80, W=112, seed=31, yaw=-20.0)]
    gen = torch.Generator().manual_seed(8)
    G = [torch.randn(3, 80, 112, generator=gen).to(DEV) for _ in cases]
    feat = torch.rand(n, 3, generator=gen).to(DEV)

    def run(case, g, fresh, extra):
        t = _leaves(case)
        r = GaussianRasterizer(_settings(case), keep_render_grads=fresh)
        out = _call(r, t, colors_extra=[feat] if extra else None)
        (out[0] * g).sum().backward()                       # (the extra image takes no gradient: its accumulator columns still get written)
        return {k: v.grad.clone() for k, v in t.items()}

    rz._clean_ws.clear()
    for rep in range(2):
        for extra in (False, True):
            for case, g in zip(cases, G):
                ref = run(case, g, True, extra)
                got = run(case, g, False, extra)
                for k in ref:
                    # (two launches of the same backward differ by the order of K7's float atomics, amplified for scales / rotations: a row
                    #  left dirty would show up as an O(1) error, not at the 1e-4 level)
                    assert float((got[k] - ref[k]).abs().max()) <= 2e-4 * float(ref[k].abs().max()), k
                assert len(rz._clean_ws) >= 1
                for ws in rz._clean_ws.values():
                    assert int(torch.count_nonzero(ws)) == 0, "a row was left dirty"


def test_colour_half_on_an_auxiliary_stream_gives_the_same_call():
    """RasterOptions.aux_stream (EmdFwdArgs.aux_stream): the projection kernel split into its geometry half on the call's stream and
    its colour half on a second stream beside the binning stage, forked and joined by events -- the images of the call are bit for bit
    those of the one-stream call, with and without fused motion, and so are radii and the status words; gradients agree to the order of
    the backward's float atomics.  Also inside a captured graph (the fork / join become graph edges)."""
    from emd_amd import GaussianRasterizer
    for motion in (False, True):
        case = make_case(n=9000, H=96, W=144, seed=57, motion=motion)
        G = torch.randn(3, case["H"], case["W"], generator=torch.Generator().manual_seed(3)).to(DEV)
        kw = {}
        if motion:
            kw = dict(actor_ids=case["actor_ids"].to(DEV), actor_pose=case["actor_pose"].to(DEV))

        def run(aux):
            t = _leaves(case)
            r = GaussianRasterizer(_settings(case), aux_stream=aux)
            out = _call(r, t, **kw)
            (out[0] * G).sum().backward()
            return [o.detach().clone() for o in out[:5]], {k: v.grad.clone() for k, v in t.items()}, r.last_call.status.clone()

        ref_out, ref_g, ref_st = run(False)
        got_out, got_g, got_st = run(True)
        for a, b in zip(got_out, ref_out):
            assert torch.equal(a, b)
        assert torch.equal(got_st[:3], ref_st[:3])
        for k in ref_g:
            assert float((got_g[k] - ref_g[k]).abs().max()) <= 2e-4 * float(ref_g[k].abs().max()) + 1e-12, k
    # captured: fork and join inside one hipGraph
    case = make_case(n=9000, H=96, W=144, seed=57)
    t = {k: case[k].to(DEV) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    t["means2D"] = torch.zeros(case["N"], 3, device=DEV)
    r = GaussianRasterizer(_settings(case), aux_stream=True, no_sync=True)
    with torch.no_grad():
        want = _call(GaussianRasterizer(_settings(case)), t)[0].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                img = _call(r, t)[0]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            img = _call(r, t)[0]
        for _ in range(3):
            img.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(img, want)
        graph.reset()


@pytest.mark.parametrize("sh_degree", [3, 2])
def test_shs_residuals_equal_the_summed_coefficients(sh_degree):
    """EmdFwdArgs.shs_residual (ABI 20): the call with shs_residuals=(a, b) gives, bit for bit, the image of the call with
    shs = (shs + a) + b -- the fine stage's `shs + dshs_coarse + dshs_fine` (S3Gaussian/scene/deformation.py:468-481) -- and each of the
    three terms receives the same dL/dshs; one residual alone and the row path without the LDS staging (9 coefficients) included."""
    from emd_amd import GaussianRasterizer
    case = make_case(n=6000, H=80, W=112, seed=11 + sh_degree, sh_degree=sh_degree)
    K = (sh_degree + 1) ** 2
    case["shs"] = case["shs"][:, :K].contiguous()
    rs = _settings(case)
    gen = torch.Generator().manual_seed(sh_degree)
    ra, rb = torch.randn(case["shs"].shape, generator=gen) * 0.1, torch.randn(case["shs"].shape, generator=gen) * 0.05
    gout = torch.rand(3, case["H"], case["W"], generator=gen).to(DEV)

    def run(mode):
        t = _leaves(case)
        res = [r_.to(DEV).clone().requires_grad_(True) for r_ in (ra, rb)]
        base = t["shs"]
        if mode == "summed":
            t = dict(t, shs=(base + res[0]) + res[1])
            img = _call(GaussianRasterizer(rs), t)[0]
        elif mode == "one":
            img = _call(GaussianRasterizer(rs), t, shs_residuals=[res[0]])[0]
        elif mode == "one_summed":
            t = dict(t, shs=base + res[0])
            img = _call(GaussianRasterizer(rs), t)[0]
        else:
            img = _call(GaussianRasterizer(rs), t, shs_residuals=res)[0]
        (img * gout).sum().backward()
        return img.detach(), [base.grad] + [r_.grad for r_ in res]
    img_sum, g_sum = run("summed")
    img_res, g_res = run("residuals")
    assert float(img_sum.abs().max()) > 0 and torch.equal(img_res, img_sum)
    close = lambda a_, b_: float((a_ - b_).abs().max()) <= 1e-5 * float(b_.abs().max())     # (the render backward's float atomics reorder sums between runs)
    for a_, b_ in zip(g_res, g_sum):
        assert a_ is not None and close(a_, b_)
    assert torch.equal(g_res[0], g_res[1]) and torch.equal(g_res[0], g_res[2])              # one dL/dshs for the three terms
    img_one, g_one = run("one")
    img_one_sum, g_one_sum = run("one_summed")
    assert torch.equal(img_one, img_one_sum) and close(g_one[0], g_one_sum[0]) and close(g_one[1], g_one_sum[1]) and g_one[2] is None
    with pytest.raises(ValueError):
        _call(GaussianRasterizer(rs), _leaves(case), shs_residuals=[ra[:10].to(DEV)])
