"""-m gpu: parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars: integer/key work (radii, tiles touched, sorted 64-bit keys, Gaussian ids, tile ranges, pixel means, depths)
bit-exact; images bit-exact (north_star asks L_inf <= 1e-4); gradients element-wise within tests/helpers.py's bar
(|hip - oracle| <= 1e-4 |oracle| + 1e-6 max|oracle| for every element, relative L2 <= 1e-5)."""
import numpy as np
import pytest
import torch

from tests.helpers import (IMAGE_TOL, compare_backward, compare_forward, make_case, run_hip, run_oracle)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [
    dict(n=2000, H=64, W=96, seed=0),
    dict(n=5000, H=100, W=150, seed=1),               # ragged: H, W not multiples of 16
    dict(n=3000, H=64, W=96, seed=2, sh_degree=0),
    dict(n=3000, H=64, W=96, seed=3, sh_degree=1),
    dict(n=3000, H=64, W=96, seed=4, sh_degree=2),
    dict(n=3000, H=80, W=80, seed=5, colors_precomp=True),
    dict(n=3000, H=80, W=80, seed=6, cov_precomp=True),
    dict(n=20000, H=128, W=192, seed=7, scale_mult=1.5),
    dict(n=4000, H=64, W=96, seed=8, yaw=30.0),
    dict(n=80000, H=96, W=128, seed=9, scale_mult=1.0),   # deep tile lists: thousands of entries, many LDS chunks / batches
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_forward_backward_parity(kw):
    case = make_case(**kw)
    orc = run_oracle(case, backward=True)
    hip = run_hip(case, backward=True)
    compare_forward(hip, orc, tol=IMAGE_TOL)
    checked = compare_backward(hip, orc)
    assert "means3D" in checked and "opacities" in checked
    # the same call with every (tile, Gaussian) pair kept: upstream's sorted list entry for entry, the same images and gradients
    full = run_hip(case, backward=True, keep_all_pairs=True)
    compare_forward(full, orc, tol=IMAGE_TOL)
    assert full["status"]["num_rendered"] == orc["bin"]["D"] >= hip["status"]["num_rendered"]
    for k in ("color", "depth", "alpha", "normal", "radii"):
        np.testing.assert_array_equal(full[k], hip[k], err_msg=k)
    compare_backward(full, orc)


@pytest.mark.parametrize("kw", [
    dict(n=4000, H=64, W=96, seed=11, motion=True),
    dict(n=4000, H=64, W=96, seed=12, motion=True, residual=True),
    dict(n=3000, H=64, W=96, seed=13, residual=True),        # S3G-style residual add only
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_fused_motion_parity(kw):
    case = make_case(**kw)
    orc = run_oracle(case, backward=True)
    hip = run_hip(case, backward=True)
    compare_forward(hip, orc, tol=IMAGE_TOL)
    checked = compare_backward(hip, orc)
    if kw.get("motion"):
        assert "actor_pose" in checked
    if kw.get("residual"):
        assert "residual_dx" in checked


@pytest.mark.parametrize("kw", [
    dict(n=300, H=16, W=16, seed=41),                  # one tile: zero tile passes, list = depth order
    dict(n=600, H=16, W=40, seed=42),                  # three tiles, ragged width
    dict(n=400, H=4112, W=4100, seed=43),              # 257 x 257 = 66 049 tiles: 17 tile bits -> three tile passes
], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_tile_pass_counts(kw):
    """The tile sort runs ceil(log2(T) / 8) passes: 0, 1 and 3 here (2 everywhere else in this file)."""
    case = make_case(**kw)
    orc = run_oracle(case, backward=False)
    for keep_all in (False, True):
        hip = run_hip(case, backward=False, keep_all_pairs=keep_all)
        compare_forward(hip, orc, tol=IMAGE_TOL)


def test_equal_depth_ties_keep_gaussian_order():
    """Gaussians at exactly the same view depth: the reference's stable sort leaves them in index order inside a tile.
    The depth-first sort must do the same (stable depth sort of the Gaussians, stable partition by tile)."""
    case = make_case(n=4000, H=64, W=96, seed=44)
    m = case["means3D"]
    m[:, 0] = torch.round(m[:, 0] * 2.0) / 2.0        # camera looks along +x: depth quantised to 0.5 m -> thousands of ties
    orc = run_oracle(case, backward=True)
    hip = run_hip(case, backward=True)
    compare_forward(hip, orc, tol=IMAGE_TOL)
    compare_backward(hip, orc)
    full = run_hip(case, backward=False, keep_all_pairs=True)
    compare_forward(full, orc, tol=IMAGE_TOL)
    for keys in (hip["keys"], full["keys"]):
        d = (keys & np.uint64(0xFFFFFFFF))
        assert (np.diff(d.astype(np.int64)) == 0).sum() > 100, "test did not produce depth ties"


def test_absgrad():
    case = make_case(n=3000, H=64, W=96, seed=21)
    orc = run_oracle(case, backward=True)
    hip = run_hip(case, backward=True, absgrad=True)
    compare_backward(hip, orc, names=("means2D", "means2D_abs"))
    assert np.all(hip["grads"]["means2D_abs"] >= np.abs(hip["grads"]["means2D"][:, :2]) * (1 - 1e-4) - 1e-6)


def test_empty_and_degenerate_inputs():
    # nothing visible: every Gaussian behind the camera
    case = make_case(n=500, H=48, W=64, seed=31)
    case["means3D"][:, 0] = -5.0
    orc = run_oracle(case, backward=True)
    assert orc["bin"]["D"] == 0
    hip = run_hip(case, backward=True)
    compare_forward(hip, orc)
    assert np.all(hip["radii"] == 0)
    np.testing.assert_allclose(hip["color"], np.broadcast_to(case["bg"].numpy()[:, None, None], hip["color"].shape), atol=1e-7)
    for k, v in hip["grads"].items():
        if v is not None:
            assert np.all(v == 0), k
    # a single huge Gaussian covering every tile; opaque -> early termination path
    case = make_case(n=64, H=64, W=64, seed=32)
    case["means3D"][:] = torch.tensor([3.0, 0.0, 1.5])
    case["means3D"][:, 0] += torch.linspace(0, 1, 64)
    case["scales"][:] = 2.0
    case["opacities"][:] = 0.999
    orc = run_oracle(case, backward=True)
    hip = run_hip(case, backward=True)
    compare_forward(hip, orc)
    compare_backward(hip, orc)
    assert (orc["img"]["n_contrib"] < 64).any(), "early termination should trigger"


def test_capacity_retry_and_determinism_of_forward():
    from emd_amd import rasterizer
    case = make_case(n=6000, H=96, W=128, seed=41)
    rasterizer._capacity_hint.clear()
    old = rasterizer.RasterConfig.min_capacity
    rasterizer.RasterConfig.min_capacity = 16   # process-wide DEFAULT read when run_hip builds its rasterizer: forces EMD_ERR_CAPACITY -> regrow -> retry
    try:
        rasterizer._capacity_hint[(0, 96, 128)] = 16
        a = run_hip(case)
    finally:
        rasterizer.RasterConfig.min_capacity = old
    b = run_hip(case)
    for k in ("color", "depth", "alpha", "keys", "ids", "ranges"):
        np.testing.assert_array_equal(a[k], b[k])   # forward is bit-reproducible run to run
    orc = run_oracle(case)
    compare_forward(a, orc)


def test_round_trip_properties_full_size():
    """Size-independent properties at 1066 x 1600 with 2 M Gaussians (on top of the oracle comparisons at that size below):
    keys sorted, every tile range consistent with the keys, alpha in [0,1], colour = linear in the features."""
    from emd_amd import scenes, GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    sc = scenes.make_static_scene(2_000_000, seed=0).to(dev)
    cam = scenes.rig_camera(0, 0)
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy,
                                       torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                       cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
    r = GaussianRasterizer(rs)
    m2 = torch.zeros(sc.N, 3, device=dev)
    args = dict(means3D=sc.means, means2D=m2, opacities=torch.sigmoid(sc.opacity_logits), scales=torch.exp(sc.log_scales),
                rotations=sc.quats)
    col = torch.rand(sc.N, 3, device=dev)
    c1, d1, _, a1, radii, _ = r(colors_precomp=col, **args)
    first_call = r.last_call
    keys, ids, ranges = r.export_binning()
    k = keys.cpu().numpy().view(np.uint64)
    assert np.all(k[1:] >= k[:-1]), "keys must be sorted"
    tiles = (k >> np.uint64(32)).astype(np.int64)
    rg = ranges.cpu().numpy().view(np.uint32).astype(np.int64)
    cnt = np.bincount(tiles, minlength=rg.shape[0])
    np.testing.assert_array_equal(rg[:, 1] - rg[:, 0], cnt)
    assert int(radii.gt(0).sum()) == first_call.last_status()["num_visible"]
    assert float(a1.min()) >= 0.0 and float(a1.max()) <= 1.0
    # linearity in colour: render(2 c) = 2 render(c) with bg = 0
    c2, _, _, a2, _, _ = r(colors_precomp=2 * col, **args)
    torch.testing.assert_close(c2, 2 * c1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a2, a1, rtol=0, atol=0)


@pytest.mark.parametrize("motion", [False, True], ids=["static", "motion"])
def test_fused_activations_raw_params(motion):
    """EMD_FLAG_RAW_PARAMS: exp / normalize / sigmoid (gaussian_renderer/__init__.py:99-101) fused into K1 / K8.
    The oracle is fed the activations exactly as the library computes them (emd_activations_forward), so the integer
    contract stays bit-exact; gradients are chained through the activations in numpy (tests/helpers.raw_params_parity)."""
    from tests.helpers import raw_params_parity
    case = make_case(n=4000, H=64, W=96, seed=61, motion=motion)
    g = torch.Generator().manual_seed(5)
    log_s = torch.log(case["scales"])
    raw_q = case["rotations"] * (0.5 + torch.rand(case["N"], 1, generator=g))
    logit = torch.logit(case["opacities"].clamp(1e-4, 1 - 1e-4))
    raw_params_parity(case, log_s, raw_q, logit)


def _bench_scene_case(n, frame, cam_idx=0, actors=True, seed_grad=17, num_actors=32, residual=False):
    """bench.py's scene (emd_amd.scenes) as a parity case at 1066 x 1600, raw parameters."""
    from emd_amd import scenes
    from emd_amd.motion import build_actor_pose
    H, W = 1066, 1600
    sc = scenes.make_static_scene(n, seed=0)
    pose = None
    if actors:
        sc = scenes.add_actors(sc, num_actors=num_actors, pts_per_actor=5000, num_frames=50, seed=1)
        pose = build_actor_pose(sc.actor_quats, sc.actor_trans, sc.actor_valid, frame)
    case = dict(N=n, H=H, W=W, sh_degree=3, bg=torch.zeros(3), cam=scenes.rig_camera(frame, cam_idx, H, W), means3D=sc.means,
                opacities=None, scales=None, rotations=None, shs=sc.shs, colors_precomp=None, cov3D_precomp=None,
                actor_ids=sc.actor_id if actors else None, actor_pose=pose, residual_dx=None, residual_dq=None,
                flags=1 | (2 if actors else 0))
    g = np.random.default_rng(seed_grad)
    case["dL_dcolor"] = g.standard_normal((3, H, W)).astype(np.float32)
    case["dL_ddepth"] = (0.01 * g.standard_normal((1, H, W))).astype(np.float32)
    case["dL_dalpha"] = g.standard_normal((1, H, W)).astype(np.float32)
    if residual:        # the learned per-Gaussian deformation residual as K1 / K8 consume it (positions for all, rotations for actor points)
        tg = torch.Generator().manual_seed(seed_grad + 1)
        case["residual_dx"] = 0.02 * torch.randn(n, 3, generator=tg)
        if actors:
            case["residual_dq"] = 0.02 * torch.randn(n, 4, generator=tg)
    return case, sc


def test_full_size_config1_static_1M():
    """BASELINE configs[1]: single static frame, 1 M Gaussians, 1066 x 1600, forward + backward, against the (OpenMP) C oracle:
    radii / keys / ids / ranges / images bit-exact, every gradient element within the bar of tests/helpers.py."""
    from tests.helpers import raw_params_parity
    case, sc = _bench_scene_case(1_000_000, frame=0, actors=False)
    res = raw_params_parity(case, sc.log_scales, sc.quats, sc.opacity_logits)
    assert res["V"] > 500_000 and res["D"] > 2_000_000, res
    print("config1", res)


@pytest.mark.parametrize("frame", [0, 24])
def test_full_size_config2_dynamic_2M(frame):
    """BASELINE configs[2] = the scene bench.py times: 2 M Gaussians of which 32 x 5000 ride on rigid actors, 50-frame clip,
    1066 x 1600, fused explicit-motion transform + fused activations (raw_params), at frames 0 and 24."""
    from tests.helpers import raw_params_parity
    case, sc = _bench_scene_case(2_000_000, frame=frame, actors=True)
    res = raw_params_parity(case, sc.log_scales, sc.quats, sc.opacity_logits)
    assert res["V"] > 1_000_000 and res["D"] > 4_000_000, res
    print("config2 frame", frame, res)


def test_full_size_config4_rank_workload_2M_with_deformation_residual():
    """What ONE rank of BASELINE configs[3] (4-camera rig, 2 M Gaussians + deformation residual, view-parallel on 4 GPUs) computes:
    camera 2 of the rig at frame 10, the residual as an input of the fused transform (gradients back to it).  The multi-GPU part
    of that configuration (the gradient exchange) is covered by tests/test_bench_multirank_gpu.py."""
    from tests.helpers import raw_params_parity
    case, sc = _bench_scene_case(2_000_000, frame=10, cam_idx=2, actors=True, residual=True)
    res = raw_params_parity(case, sc.log_scales, sc.quats, sc.opacity_logits)
    assert res["V"] > 100_000 and "residual_dx" in res and "residual_dq" in res, res
    print("config4 rank workload", res)


def test_full_size_config5_rank_workload_3M():
    """What one rank of BASELINE configs[4] (6-camera rig, 3 M Gaussians, 8 GPUs) computes per step: camera 5 of the rig, 3 M
    Gaussians of which 48 x 5000 ride on actors (densification, the other half of that configuration: tests/test_gaussian_model_gpu.py)."""
    from tests.helpers import raw_params_parity
    case, sc = _bench_scene_case(3_000_000, frame=30, cam_idx=5, actors=True, num_actors=48)
    res = raw_params_parity(case, sc.log_scales, sc.quats, sc.opacity_logits)
    assert res["V"] > 100_000, res
    print("config5 rank workload", res)


def test_parity_at_bench_resolution():
    """1066 x 1600 (6700 tiles, ragged bottom row) with the bench rig camera and a 150 k-Gaussian street scene:
    the C oracle finishes this in seconds, so the full contract (bit-exact keys, image L_inf, gradients) is
    checked at the metric's own image size."""
    from emd_amd import scenes
    from tests.helpers import oracle_settings  # noqa: F401
    n, H, W = 150_000, 1066, 1600
    sc = scenes.make_static_scene(n, seed=4)
    case = dict(N=n, H=H, W=W, sh_degree=3, bg=torch.tensor([0.0, 0.0, 0.0]), cam=scenes.rig_camera(3, 0, H, W),
                means3D=sc.means, opacities=torch.sigmoid(sc.opacity_logits), scales=torch.exp(sc.log_scales) * 2.0,
                rotations=sc.quats, shs=sc.shs, colors_precomp=None, cov3D_precomp=None, actor_ids=None, actor_pose=None,
                residual_dx=None, residual_dq=None, flags=1)
    g = np.random.default_rng(17)
    case["dL_dcolor"] = g.standard_normal((3, H, W)).astype(np.float32)
    case["dL_ddepth"] = (0.01 * g.standard_normal((1, H, W))).astype(np.float32)
    case["dL_dalpha"] = g.standard_normal((1, H, W)).astype(np.float32)
    orc = run_oracle(case, backward=True)
    assert orc["bin"]["D"] > 300_000
    hip = run_hip(case, backward=True)
    compare_forward(hip, orc, tol=IMAGE_TOL)
    compare_backward(hip, orc)


def test_one_rasterizer_object_called_three_times_in_one_graph():
    """The reference re-invokes the same rasterizer object up to nine times per render() -- main pass, feature passes with
    colors_precomp, decomposition passes on a boolean-mask subset (gaussian_renderer/__init__.py:145,172,247) -- and
    back-propagates through all of them at once.  The calls must not disturb each other's saved state: images equal the
    single-call images bit for bit, and the gradients of the shared tensors equal the sum of the single-call gradients."""
    from emd_amd import GaussianRasterizationSettings, GaussianRasterizer
    case = make_case(n=6000, H=80, W=112, seed=77)
    cam, dev = case["cam"], torch.device("cuda", 0)
    rs = GaussianRasterizationSettings(image_height=case["H"], image_width=case["W"], tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                       bg=case["bg"].to(dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
                                       projmatrix=cam.full_proj_transform.to(dev), sh_degree=case["sh_degree"],
                                       campos=cam.camera_center.to(dev), prefiltered=False, debug=False)
    gen = torch.Generator().manual_seed(5)
    feat0 = torch.rand(case["N"], 3, generator=gen)
    mask = (torch.arange(case["N"]) % 3 == 0).to(dev)
    G = [torch.randn(3, case["H"], case["W"], generator=gen).to(dev) for _ in range(3)]
    names = ("means3D", "shs", "opacities", "scales", "rotations")

    def leaves():
        t = {k: case[k].to(dev).clone().requires_grad_(True) for k in names}
        t["feat"] = feat0.to(dev).clone().requires_grad_(True)
        t["means2D"] = torch.zeros(case["N"], 3, device=dev, requires_grad=True)
        return t

    def call(rast, t, which):
        common = dict(opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3Ds_precomp=None, extra_attrs=None)
        if which == 0:
            return rast(means3D=t["means3D"], means2D=t["means2D"], shs=t["shs"], colors_precomp=None, **common)[0]
        if which == 1:
            return rast(means3D=t["means3D"], means2D=t["means2D"], shs=None, colors_precomp=t["feat"], **common)[0]
        return rast(means3D=t["means3D"][mask], means2D=t["means2D"][mask], shs=t["shs"][mask], colors_precomp=None,
                    opacities=t["opacities"][mask], scales=t["scales"][mask], rotations=t["rotations"][mask], cov3Ds_precomp=None,
                    extra_attrs=None)[0]

    singles, sums = [], None
    for w in range(3):
        t = leaves()
        img = call(GaussianRasterizer(rs), t, w)
        (img * G[w]).sum().backward()
        singles.append(img.detach())
        gr = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in t.items()}
        sums = gr if sums is None else {k: sums[k] + gr[k] for k in gr}
    t = leaves()
    rast = GaussianRasterizer(rs)                              # ONE object
    imgs = [call(rast, t, w) for w in range(3)]
    sum((imgs[w] * G[w]).sum() for w in range(3)).backward()
    for w in range(3):
        assert torch.equal(imgs[w].detach(), singles[w]), w
    for k, v in t.items():
        ref = sums[k]
        err = float((v.grad - ref).abs().max())
        assert err <= 1e-4 * max(float(ref.abs().max()), 1e-12), (k, err)


def test_nonfinite_gaussians_are_dropped_not_fatal():
    """Gaussians with NaN / Inf parameters (a diverged optimisation step) must neither fault nor disturb the rest of the image: they
    are invisible (radius 0), the image equals the image without them bit for bit, and the other Gaussians' gradients are unchanged."""
    from emd_amd import GaussianRasterizationSettings, GaussianRasterizer
    case = make_case(n=5000, H=80, W=112, seed=91)
    cam, dev = case["cam"], torch.device("cuda", 0)
    rs = GaussianRasterizationSettings(image_height=case["H"], image_width=case["W"], tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                       bg=case["bg"].to(dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
                                       projmatrix=cam.full_proj_transform.to(dev), sh_degree=case["sh_degree"],
                                       campos=cam.camera_center.to(dev), prefiltered=False, debug=False)
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    bad_rows = torch.tensor([3, 700, 1500, 2600, 4100])

    def run(poison, keep):
        t = {k: case[k].to(dev).clone() for k in names}
        if poison:
            t["means3D"][3, 0] = float("nan")
            t["scales"][700] = float("inf")
            t["rotations"][1500] = float("nan")
            t["opacities"][2600] = float("nan")
            t["means3D"][4100] = float("-inf")
        t = {k: v[keep].requires_grad_(True) for k, v in t.items()}
        means2D = torch.zeros(t["means3D"].shape[0], 3, device=dev, requires_grad=True)
        img, depth, normal, alpha, radii, _ = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=means2D, shs=t["shs"], colors_precomp=None,
                                                                     opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"],
                                                                     cov3Ds_precomp=None, extra_attrs=None)
        (img.sum() + depth.sum()).backward()
        return img.detach(), radii, t

    all_rows = torch.ones(case["N"], dtype=torch.bool)
    good = all_rows.clone()
    good[bad_rows] = False
    img_p, radii_p, tp = run(True, all_rows.to(dev))
    img_c, radii_c, tc = run(False, good.to(dev))
    assert torch.isfinite(img_p).all()
    assert (radii_p[torch.tensor([3, 700, 1500, 4100], device=dev)] == 0).all()      # poisoned geometry: culled; the NaN opacity
    # leaves the footprint valid (radius from geometry) and is skipped per pixel by `alpha >= 1/255`, which NaN fails
    assert torch.equal(img_p, img_c)
    for k in names:
        gp = tp[k].grad[good.to(dev)]
        assert torch.isfinite(gp).all(), k
        assert float((gp - tc[k].grad).abs().max()) <= 1e-5 * max(float(tc[k].grad.abs().max()), 1e-12), k


def test_depth_beyond_the_three_pass_range_falls_back_to_the_wide_sort():
    """The depth sort covers the 27 key bits above the near plane (depths up to 65 536 x near = 13 km) in three passes.  A visible
    Gaussian farther away makes the call fail over to the four-pass sort over all 32 bits (EMD_ERR_DEPTH_RANGE -> retry with
    EMD_FLAG_WIDE_DEPTH_SORT, remembered per camera size): the result is the oracle's, bit for bit, either way."""
    from emd_amd import rasterizer
    case = make_case(n=3000, H=64, W=96, seed=71)
    m = case["means3D"]
    m[10] = torch.tensor([20000.0, 0.0, 1.5])            # 20 km down the road, on the optical axis: visible (0.3 px dilation)
    m[11] = torch.tensor([90000.0, 3.0, 1.5])
    case["scales"][10:12] = 30.0
    case["opacities"][10:12] = 0.9
    key = (0, case["H"], case["W"])
    rasterizer._wide_depth.discard(key)
    orc = run_oracle(case, backward=True)
    assert orc["pre"]["radii"][10] > 0 and orc["pre"]["radii"][11] > 0
    hip = run_hip(case, backward=True)
    assert key in rasterizer._wide_depth                   # the narrow sort reported the range, the wide one produced the result
    compare_forward(hip, orc, tol=IMAGE_TOL)
    compare_backward(hip, orc)
    hip2 = run_hip(case, backward=False, keep_all_pairs=True)       # second call: wide from the start (and upstream's list, entry for entry)
    np.testing.assert_array_equal(hip2["keys"], orc["bin"]["keys"])
    rasterizer._wide_depth.discard(key)


def test_baseline_config0_exact_size_10k_static_256():
    """BASELINE configs[0] at exactly its size: 10 000 static Gaussians of the bench scene generator, 256 x 256, fx = fy = 272 -- the
    scene `bench.py --config 0` times -- forward + backward against the oracle: keys / ids / ranges / radii / images bit-exact, all
    gradients at the bars of tests/helpers.py."""
    from emd_amd import scenes
    n, H, W = 10_000, 256, 256
    sc = scenes.make_static_scene(n, seed=0)
    case = dict(N=n, H=H, W=W, sh_degree=3, bg=torch.tensor([0.0, 0.0, 0.0]), cam=scenes.rig_camera(0, 0, H, W, fx=272.0, fy=272.0),
                means3D=sc.means, opacities=torch.sigmoid(sc.opacity_logits), scales=torch.exp(sc.log_scales),
                rotations=torch.nn.functional.normalize(sc.quats, dim=1), shs=sc.shs, colors_precomp=None, cov3D_precomp=None, actor_ids=None,
                actor_pose=None, residual_dx=None, residual_dq=None, flags=1)
    g = np.random.default_rng(23)
    case["dL_dcolor"] = g.standard_normal((3, H, W)).astype(np.float32)
    case["dL_ddepth"] = (0.01 * g.standard_normal((1, H, W))).astype(np.float32)
    case["dL_dalpha"] = g.standard_normal((1, H, W)).astype(np.float32)
    orc = run_oracle(case, backward=True)
    assert orc["bin"]["D"] > 5_000 and int((orc["pre"]["radii"] > 0).sum()) > 3_000
    hip = run_hip(case, backward=True)
    compare_forward(hip, orc, tol=IMAGE_TOL)
    compare_backward(hip, orc)


def test_long_diagonal_needles_keep_upstreams_rectangle():
    """footprints more than 500 px long and under a pixel wide, on the diagonal.  The projection kernel cuts the rectangle it hands
    to the binning down to the alpha >= 1/255 box computed from the exact 2-D covariance, while the render kernels evaluate alpha with the float
    conic adj(cov) / fl(det): for a c / det above ~3e4 the box's margin no longer covers that rounding, so such a Gaussian must keep
    upstream's rectangle and all quadrant bits (as csrc/footprint.h refuses to cull it).  Every dropped pair / cleared quadrant bit is checked
    against the brute-force evaluation of the render loop's own skip test; images bit for bit in both list modes."""
    case = make_case(n=600, H=512, W=768, seed=91)
    g = torch.Generator().manual_seed(11)
    n = case["N"]
    case["scales"][: n // 2] = torch.tensor([30.0, 0.0015, 0.0015])       # needles: hundreds of pixels long at every depth of the scene
    case["scales"][n // 2:] = torch.tensor([8.0, 0.004, 0.004])
    q = torch.randn(n, 4, generator=g)
    case["rotations"] = q / q.norm(dim=1, keepdim=True)
    case["opacities"][:] = 0.9
    orc = run_oracle(case, backward=False)
    hip = run_hip(case, backward=False)
    compare_forward(hip, orc, tol=IMAGE_TOL)
    assert int(orc["pre"]["radii"].max()) > 500, int(orc["pre"]["radii"].max())
    st = hip["cull_stats"]
    assert st["contributing"] <= st["kept"] <= st["D"], st
    full = run_hip(case, backward=False, keep_all_pairs=True)
    compare_forward(full, orc, tol=IMAGE_TOL)
    for k in ("color", "depth", "alpha", "normal"):
        np.testing.assert_array_equal(full[k], hip[k], err_msg=k)


def test_elongated_and_faint_footprints_are_not_overculled():
    """The footprint test of the duplicate stage decides which (tile, Gaussian) pairs exist and which quadrants the render forward
    visits (csrc/footprint.h).  Needles hundreds of pixels long and a pixel wide -- where det = A C - B^2 cancels and the test must
    fall back to keeping the pair --, faint Gaussians just above 1/255 and footprints larger than the image: every dropped pair and
    every cleared quadrant bit is checked against the brute-force evaluation of the render loop's own skip test, images bit for bit."""
    case = make_case(n=3000, H=96, W=160, seed=83)
    g = torch.Generator().manual_seed(7)
    n = case["N"]
    s = case["scales"]
    s[: n // 3] = torch.tensor([4.0, 0.002, 0.002])                     # needles (the +0.3 px dilation is their width on screen)
    s[n // 3: n // 2] = torch.tensor([0.6, 0.6, 0.001])                 # discs seen at all angles
    q = torch.randn(n, 4, generator=g)
    case["rotations"] = q / q.norm(dim=1, keepdim=True)
    o = case["opacities"]
    o[: n // 4] = 0.0040 + 0.002 * torch.rand(n // 4, 1, generator=g).reshape(o[: n // 4].shape)      # around 1/255 = 0.0039
    o[n // 4: n // 2] = 0.99
    from tests.helpers import compare_render_grads
    orc = run_oracle(case, backward=True)
    hip = run_hip(case, backward=True)
    compare_forward(hip, orc, tol=IMAGE_TOL)
    st = hip["cull_stats"]
    assert st["kept"] < st["D"], st                                       # pairs are left out here ...
    assert st["contributing"] <= st["kept"]                               # ... and never one that contributes
    full = run_hip(case, backward=True, keep_all_pairs=True)
    compare_forward(full, orc, tol=IMAGE_TOL)
    for k in ("color", "depth", "alpha", "normal"):
        np.testing.assert_array_equal(full[k], hip[k], err_msg=k)
    # The render backward's per-Gaussian sums at the strict bar, for both lists (the same survivors reach the backward either way).
    # The projection backward behind them is as ill-conditioned as it gets on needles (aspect 2000 : 1 -- a one-ulp move of a conic
    # gradient moves dL/dmean by percents), so its end-to-end comparison belongs to the well-conditioned scenes of this file.
    compare_render_grads(hip["render_grads"], orc["grads"]["render_grads"])
    compare_render_grads(full["render_grads"], orc["grads"]["render_grads"])
