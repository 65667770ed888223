"""The S3Gaussian-style training step at the headline size, assembled from every HIP piece of this repository, as an importable builder:
`bench.py` reports its `fine_stage` block from here, `profiles/bench_full_step.py` is the command line over it.

    fused-motion rasterizer -> sky cube map (1024^2 faces) + blend -> L1 + depth L2 + D-SSIM + sky BCE -> backward to all Gaussian parameters,
    actor poses and the cube map -> per-view densification statistics                                   (S3Gaussian/train.py:203-229,366-430)
    fine=True: the "fine" stage (train.py after coarse_iterations; 50 000 of the 55 000 iterations, arguments/gaussian_options.py:67-68): no
    actors, the self-supervised EMD deformation network (HexPlane 4 x 6 planes x 32 channels, temporal table, heads dx / do / dshs; run-script
    flags) in front of the rasterizer on all Gaussians and trained through it, plus the residual regularisers of train.py.

`kernel_models()` prices the kernels of that step (algorithmic bytes or FLOPs per launch, and the ceiling each is measured against) so that
`measure()` can report the longest ones with a fraction -- the same table `profiles/make_roofline_table.py` prints for every kernel."""
import gc
import json
import os
import re
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
SPLIT_BF16_PEAK_TFLOPS = 2500.0 / 6.0   # fp32-equivalent dense peak of the six-product split-bf16 formulation on the bf16 MFMA peak (2.5 PFLOP/s)
SPLIT_BF16_MEASURED_TFLOPS = 252.0      # ... and what a loop of nothing but those MFMAs + conversions reaches (profiles/r04_mfma_split_bf16_microbench.txt)
FP32_MFMA_PEAK_TFLOPS = 157.3


def build(dev, N=2_000_000, H=1066, W=1600, F=50, fine=True, feat=False, feat_separate=False, fused_l1=False, adam=None, capturable=False):
    """-> namespace(step(s), params, model, deform, frames F, ...).  `adam`: None | "hip" | "torch" (the optimiser step of train.py:428 inside step)."""
    from emd_amd import dp, scenes, RasterOptions
    from emd_amd.loss import image_loss
    from emd_amd.model import StreetGaussians, render, residual_abs_mean
    from emd_amd.sky import SkyCubeMap, composite_s3g
    S = types.SimpleNamespace(N=N, H=H, W=W, F=F, fine=fine, feat=bool(fine and (feat or feat_separate)), feat_separate=feat_separate, dev=dev)
    scene = scenes.make_static_scene(N, seed=0)
    if not fine:
        scene = scenes.add_actors(scene, num_actors=32, pts_per_actor=5000, num_frames=F, seed=1)
    model = StreetGaussians(scene, dev)
    params = list(model.parameters())
    deform = embeddings = None
    if fine:
        from emd_amd.deformation import DeformOptions, deform_network
        torch.manual_seed(5)
        deform = deform_network(DeformOptions()).to(dev)
        deform.deformation_net.set_aabb([120.0, 30.0, 10.0], [0.0, -30.0, -2.0])
        for n_, p_ in deform.named_parameters():          # non-zero heads so that the residuals (and their gradients) are live
            if p_.dim() > 1 and "grid" not in n_:
                p_.data.mul_(0.05)
        embeddings = torch.nn.Parameter(torch.zeros(N, 4, device=dev))
        params += list(deform.parameters()) + [embeddings]
    optimizer = None
    if adam:
        from emd_amd.optim import Adam
        groups = [{"params": [model._xyz], "lr": 1.6e-4, "name": "xyz"}, {"params": [model._features], "lr": 2.5e-3, "name": "f"},
                  {"params": [model._opacity], "lr": 0.05, "name": "opacity"}, {"params": [model._scaling], "lr": 5e-3, "name": "scaling"},
                  {"params": [model._rotation], "lr": 1e-3, "name": "rotation"}]
        if model.has_actors:
            groups.append({"params": [model.instances_quats, model.instances_trans], "lr": 1e-5, "name": "ins_pose"})
        if fine:
            groups += [{"params": deform.get_mlp_parameters(), "lr": 1.6e-5, "name": "deformation"},
                       {"params": deform.get_grid_parameters(), "lr": 1.6e-4, "name": "grid"}, {"params": [embeddings], "lr": 2.5e-3, "name": "embedding"}]
        optimizer = (Adam if adam == "hip" else torch.optim.Adam)(groups, lr=0.0, eps=1e-15, capturable=capturable)
    sky = SkyCubeMap(types.SimpleNamespace(sky_resolution=1024, sky_white_background=False, white_background=False), device=dev)
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(3, H, W, generator=g).to(dev)
    gt_depth = (torch.rand(1, H, W, generator=g) * 90).to(dev)
    gt_feat = torch.rand(3, H, W, generator=g).to(dev)
    # (the two masks are dataset constants of a frame: held in the dtypes the loss kernels read -- float32 for the depth mask, uint8 for the sky
    #  mask -- so that the step converts nothing; 17 us per step of image-sized copies before round 6)
    sky_bool = (torch.rand(1, H, W, generator=g) < 0.2).to(dev)
    sky_mask, not_sky = sky_bool.to(torch.uint8), (~sky_bool).float()
    accum, denom, maxr = (torch.zeros(N, device=dev) for _ in range(3))
    cams, skycams = {}, {}
    for f in range(F):
        cam = scenes.rig_camera(f, 0, H, W)
        K = torch.tensor([[W / (2 * cam.tanfovx), 0, W / 2], [0, H / (2 * cam.tanfovy), H / 2], [0, 0, 1]], dtype=torch.float32)
        cams[f] = cam
        skycams[f] = types.SimpleNamespace(image_height=H, image_width=W, intrinsic=K.to(dev), world_view_transform=cam.world_view_transform.to(dev))
    bg = torch.zeros(3)
    opts = [RasterOptions(no_sync=False)]

    def step(s):
        f = s % F
        for p in params:
            p.grad = None
        sky.sky_cube_map.grad = None
        out = render(model, cams[f], bg, frame=f, deformation=deform, embeddings=embeddings, iteration=12000 + s, time=f / (F - 1), options=opts[0],
                     render_feat=S.feat and not feat_separate, need_feat=S.feat, fused_l1=("dx", "do") if (fine and fused_l1) else ())
        if S.feat and feat_separate:        # the reference's three calls: main pass above + one call per feature set, same rasterizer object
            bd, dd = out["boundary"], out["ddict"]
            base = dict(means3D=bd["means3D"], means2D=out["viewspace_points"], opacities=bd["opacities"], scales=bd["scales"],
                        rotations=bd["rotations"], raw_params=bd["raw_params"])
            out["feat_c"] = out["rasterizer"](shs=None, colors_precomp=dd["coarse"]["feat"], **base)[0]
            out["feat_f"] = out["rasterizer"](shs=None, colors_precomp=dd["fine"]["feat"], **base)[0]
        image, _ = composite_s3g(sky, skycams[f], out["render"], out["weight"])
        loss, _ = image_loss(image, gt, out["depth"], gt_depth, not_sky, out["weight"], sky_mask)
        if fine:                                           # residual regularisers (train.py: lambda_dx / do / dshs on both levels)
            for lvl in ("coarse", "fine"):
                d = out["ddict"][lvl]
                loss = loss + 0.001 * (residual_abs_mean(d, "dx") + residual_abs_mean(d, "do") + residual_abs_mean(d, "dshs"))
        if S.feat:
            loss = loss + 0.001 * (((out["feat_c"] - gt_feat) ** 2).mean() + ((out["feat_f"] - gt_feat) ** 2).mean())
        loss.backward()
        dp.add_densification_stats(out["viewspace_points"].grad, out["radii"], accum, denom, maxr)
        if optimizer is not None:
            optimizer.step()

    # size the binning workspace once (synchronising forwards over a few frames); afterwards no step reads a count back
    dmax = 0
    for f in range(0, F, 7):
        with torch.no_grad():
            o = render(model, cams[f], bg, frame=f, deformation=deform, embeddings=embeddings, iteration=12000, time=f / (F - 1), options=opts[0])
        dmax = max(dmax, o["raster_call"].last_status()["num_rendered"])
    del o
    opts[0] = RasterOptions(no_sync=True, capacity_hint=int(dmax * 1.3) + 1024)
    S.step, S.params, S.model, S.deform, S.optimizer, S.skycams, S.sky, S.embeddings = step, params, model, deform, optimizer, skycams, sky, embeddings
    return S


def record_graphs(S, frames, check=True):
    """One hipGraph per frame in `frames` (camera, frame time and sky rays are host constants of a frame), all in one memory pool; with `check`
    the replay of the first frame is compared with the eager step of that frame (dL/dxyz and the finest plane's gradient)."""
    from emd_amd.graphs import StepGraphs
    from emd_amd.sky import _camera_rays_params
    frames = list(frames)
    want = {}
    if check and S.optimizer is None:
        S.step(frames[0])
        torch.cuda.synchronize()
        want["xyz"] = S.model._xyz.grad.clone()
        if S.fine:
            want["grid"] = S.deform.deformation_net.grid.grids[-1][0].grad.clone()
    keep0 = {}

    def recorded(f):
        S.step(f)
        if f == frames[0]:                  # the first graph's gradient tensors stay referenced for the self-check (later graphs recycle everything else)
            keep0["xyz"] = S.model._xyz.grad
            keep0["grid"] = S.deform.deformation_net.grid.grids[-1][0].grad if S.fine else None
    sg = StepGraphs(recorded, frames, prime=lambda f: _camera_rays_params(S.skycams[f]), freeze=[S.deform.deformation_net.grid] if S.fine else [],
                    optimizers=[S.optimizer] if S.optimizer is not None else [], warmup=0)
    if want:
        sg.graphs[frames[0]].replay()
        torch.cuda.synchronize()
        tol = lambda a, b: float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-12
        assert tol(keep0["xyz"], want["xyz"]), "graph replay: dL/dxyz differs from the eager step"
        if S.fine:
            assert tol(keep0["grid"], want["grid"]), "graph replay: plane gradients differ from the eager step"
    return sg


# ---- what the kernels of the step must move / compute (per launch), and the ceiling each is held against --------------------------------------
def kernel_models(N, V, D, HW, C_img=7, planes_bytes=None, sky_face=1024):
    """[(regex on the kernel name, label, bound, work per launch, formula)]: bound "hbm" -> work in bytes (compulsory traffic with perfect
    on-chip reuse: every input read once, every output written once), "mfma" -> useful FLOPs (2 x multiply-adds of the layer shapes).
    N points, V visible, D list entries, HW pixels.  HexPlane: 4 scales x 32 channels = 128 features per point, 3 spatial + 3 time planes per
    scale, resolutions 64 x [1, 2, 4, 8] (space) and 25 x (time)."""
    if planes_bytes is None:
        res = [64 * m for m in (1, 2, 4, 8)]
        planes_bytes = sum(3 * r * r + 3 * r * 25 * m for r, m in zip(res, (1, 2, 4, 8))) * 32 * 4
    P = planes_bytes
    P_def = 3 * (128 * 128 + 256 * 256 + 512 * 512) * 32 * 4          # spatial planes of the scales the main backward kernel defers (2 M points)
    mac = lambda *dims: 2.0 * N * sum(a * b for a, b in dims)          # FLOPs of Linear layers [in, out] over N rows
    hid, dx, do, dshs = (64, 64), (64, 3), (64, 1), (64, 48)
    M = [
        # HexPlane (csrc/hexplane.hip)
        (r"k_hexplane_bwd_agg", "HexPlane backward, main kernel", "hbm", N * (512 + 16 + 4 + 12) + 2 * P,
         "N (128 f32 dL/dfeat + xyzt + order + dL/dxyz) + planes read + plane gradients written (the deferred 128-B rows are this design's, not compulsory)"),
        (r"k_hexplane_bwd_plane", "HexPlane backward, per-plane pass of the fine scales", "hbm", N * 4 * 3 + P_def,
         "the deferred scales' spatial plane gradients written + one position per point and plane; the 128-B rows it reads are this design's, not compulsory"),
        (r"k_hexplane_fwd", "HexPlane forward", "hbm", N * (16 + 4 + 512) + P, "N (xyzt + order + 128 f32 features) + planes read once"),
        (r"k_hexplane_time_tables|k_hexplane_order", "HexPlane tables / orders", "hbm", None, ""),
        # fused MLP (csrc/mlp.hip); the template arguments after the name select the shapes: <DEPTH, NTO, ...>
        (r"k_mlp_trunk_fwd<4", "trunk forward 132 -> 64 (coarse level)", "mfma", mac((132, 64)), "2 N 132 x 64; HBM: x 528 B + h 256 B per row"),
        (r"k_mlp_trunk_bwd<4", "trunk backward (dL/dx + dW)", "mfma", 2 * mac((132, 64)), "2 x forward (data gradient + weight gradient)"),
        (r"k_mlp_embed_bwd", "trunk backward of the level without HexPlane features (vector pipe)", "hbm", N * (256 + 16 + 16), "dL/dh 256 B + xb + dL/dxb per row"),
        (r"k_mlp_branch_fwd<1, 2", "dshs head forward 64 -> 64 -> 48", "mfma", mac(hid, dshs), "2 N (64 x 64 + 64 x 48)"),
        (r"k_mlp_branch_bwd<1, 2", "dshs head backward", "mfma", 2 * mac(hid, dshs) + mac(hid), "2 x forward + the recomputed hidden layer"),
        (r"k_mlp_branch_fwd<1, 1", "dx / do head forward 64 -> 64 -> 3 | 1", "mfma", mac(hid, (64, 2)), "2 N (64 x 64 + 64 x 2 on average)"),
        (r"k_mlp_branch_bwd<1, 1", "dx / do head backward", "mfma", 2 * mac(hid, (64, 2)) + mac(hid), "2 x forward + the recomputed hidden layer"),
        (r"k_mlp_branch_fwd<2", "feature head forward (two hidden layers)", "mfma", mac(hid, hid, (64, 3)), ""),
        (r"k_mlp_branch_bwd<2", "feature head backward", "mfma", 2 * mac(hid, hid, (64, 3)) + mac(hid, hid), ""),
        # rasterizer (bench.py:algorithmic_bytes)
        (r"k_preprocess<", "projection + SH colour (K1)", "hbm", 68 * N + V * (216 + 4 * C_img) + (2 * 192 * N), "SURVEY 8d F1 + the two SH residual sets read (fine stage)"),
        (r"k_preprocess_backward", "projection backward (K8)", "hbm", V * ((24 + 16) + 48 + 192) + N * (236 + 12), "SURVEY 8d B2"),
        (r"k_render_forward_q", "tile compositing forward (K6)", "hbm", D * (4 + 24 + 4 * C_img) + HW * (4 * (C_img + 1) + 8), "SURVEY 8d F6"),
        (r"k_render_backward_q", "tile compositing backward (K7)", "hbm", D * (28 + 16) + HW * (4 * 5 + 8) + V * (24 + 16), "SURVEY 8d B1 (colour + depth + alpha gradients)"),
        # image-space tail
        (r"k_ssim_forward", "D-SSIM forward", "hbm", HW * 4 * (3 + 3 + 3 * 5), "image + target read, five 11 x 11-filtered maps per channel written"),
        (r"k_ssim_backward", "D-SSIM backward", "hbm", HW * 4 * (3 + 3 + 3 * 5 + 3), "the forward's maps read, dL/dimage written"),
        (r"k_loss_pointwise", "L1 + depth L2 + sky BCE (+ gradients)", "hbm", HW * 4 * (3 + 3 + 1 + 1 + 1 + 1 + 5), "image, target, depth, gt depth, weight, masks; three gradients"),
        (r"k_sky_forward", "sky cube map lookup + blend", "hbm", HW * 4 * (3 + 1 + 3 + 3), "render + weight read, sky colour + blended image written (the texels hit in cache)"),
        (r"k_sky_backward", "sky backward (texel scatter)", "hbm", HW * 4 * (3 + 1 + 3 + 1), "dL/dimage + weight + render read, dL/dweight; + 4 texel rows of atomics per pixel"),
        (r"k_adam", "Adam step", "hbm", None, "16 B read + 12 B written per parameter element"),
        (r"k_densification_stats", "densification statistics", "hbm", N * (4 + 12 + 3 * 8), "radii + dL/dmean2D read, three statistics updated"),
        # binning and small launches: priced per stage in section A of the table; latency-bound at 4-25 us each
        (r"k_radix_|k_duplicate|k_sorted_counts|k_tile_ranges|k_tile_order", "binning stage (depth sort, duplication, tile sort, ranges): see section A", "hbm", None,
         "launch-bound: 19 launches of 4-25 us"),
        (r"k_l1_loss|k_abs_mean|k_residual_l1|k_loss_finalize", "scalar reductions of the residual regularisers / loss terms", "hbm", None, "launch-bound"),
        (r"k_temporal_embed|k_tracked_pose|k_track_|k_select_step|k_zero_words|k_hexplane_order_keys", "small per-step launches", "hbm", None, "launch-bound"),
    ]
    return M


def price(name, ms, models):
    """-> dict for one kernel of a profile: the first model whose pattern matches its name."""
    for pat, label, bound, work, formula in models:
        if re.search(pat, name):
            e = {"kernel": short_name(name), "what": label, "ms": round(ms, 4), "bound": bound}
            if work:
                if bound == "hbm":
                    e.update(alg_GB=round(work / 1e9, 4), GBps=round(work / 1e9 / (ms * 1e-3), 1), frac=round(work / 1e9 / (ms * 1e-3) / HBM_PEAK_GBS, 4), peak="8000 GB/s HBM")
                else:
                    tf = work / 1e12 / (ms * 1e-3)
                    e.update(useful_GFLOP=round(work / 1e9, 2), TFLOPs=round(tf, 1), frac=round(tf / SPLIT_BF16_PEAK_TFLOPS, 4),
                             frac_of_measured_ceiling=round(tf / SPLIT_BF16_MEASURED_TFLOPS, 3),
                             peak="416.7 TFLOP/s fp32-equivalent = the 2.5 PFLOP/s bf16 MFMA peak / six products of the three-term split; measured ceiling 252")
                e["formula"] = formula
            return e
    return {"kernel": short_name(name), "what": "(HIP kernel without a model)" if re.search(r"\bk_[a-z]", name) else "(torch / library kernel)", "ms": round(ms, 4), "bound": None}


def short_name(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    s = m.group(1) if m else name
    return s if len(s) <= 90 else s[:87] + "..."


def profile_kernels(step_fn, steps, first=0):
    """Per-kernel device time of `steps` eager steps through torch.profiler (roctracer): {kernel name: (ms per step, launches per step)}."""
    from torch.profiler import ProfilerActivity, profile
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for s in range(steps):
                step_fn(first + s)
            torch.cuda.synchronize()
    agg = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            a = agg.setdefault(e.name, [0.0, 0])
            a[0] += float(getattr(e, "device_time", 0.0) or getattr(e, "cuda_time", 0.0))
            a[1] += 1
    return {k: (v[0] / steps / 1e3, v[1] / steps) for k, v in agg.items()}


def measure(dev, N=2_000_000, H=1066, W=1600, F=50, graph_frames=10, steps=50, profile_steps=5, top=5):
    """The `fine_stage` block of bench.py's JSON line: ms per step of the graph-replayed S3G fine-stage step (graphs of `graph_frames` frames of
    the clip, replayed round robin), and the `top` longest kernels of the same step issued eagerly under torch.profiler, each priced."""
    t_all = time.perf_counter()
    # (fused_l1: the dx / do regularisers formed by the head kernels, render(..., fused_l1=("dx", "do")) -- 11.15 / 11.08 ms without against 11.04 / 11.01
    #  with, alternated in one call in round 6; it lost in round 4, before the regularised narrow-head forward ran two waves per SIMD)
    S = build(dev, N, H, W, F, fine=True, fused_l1=True)
    for s in range(4):
        S.step(s)
    torch.cuda.synchronize()
    gc.collect()
    frames = [int(round(i * F / graph_frames)) % F for i in range(graph_frames)]
    t0 = time.perf_counter()
    sg = record_graphs(S, frames)
    t_record = time.perf_counter() - t0
    for i in range(10):
        sg.graphs[frames[i % len(frames)]].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        sg.graphs[frames[i % len(frames)]].replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same block once more (spread)
    t0 = time.perf_counter()
    for i in range(steps):
        sg.graphs[frames[i % len(frames)]].replay()
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t0
    sg.release()
    del sg
    gc.collect()
    # V, D of the profiled frames (synchronising forward) for the rasterizer kernels' models
    from emd_amd import RasterOptions
    from emd_amd.model import render
    prof = profile_kernels(S.step, profile_steps, first=frames[1])
    total = sum(ms for ms, _ in prof.values())
    glue = sum(ms for k, (ms, _) in prof.items() if not re.search(r"\bk_[a-z]", k))
    glue_n = sum(n for k, (_, n) in prof.items() if not re.search(r"\bk_[a-z]", k))
    st = None
    try:
        with torch.no_grad():
            from emd_amd import scenes
            cam = scenes.rig_camera(frames[1], 0, H, W)
            o = render(S.model, cam, torch.zeros(3), frame=frames[1], deformation=S.deform, embeddings=S.embeddings, iteration=12000, time=frames[1] / (F - 1),
                       options=RasterOptions(no_sync=False))
            st = o["raster_call"].last_status()
    except Exception:
        st = None
    V, D = (st["num_visible"], st["num_rendered"]) if st else (N // 2, 2 * N)
    models = kernel_models(N, V, D, H * W)
    ranked = sorted(prof.items(), key=lambda kv: -kv[1][0])
    kernels = []
    for name, (ms, n) in ranked[:top]:
        e = price(name, ms / max(n, 1e-9) if n >= 1 else ms, models)
        e["launches_per_step"] = round(n, 2)
        e["ms_per_step"] = round(ms, 4)
        kernels.append(e)
    groups = {}
    for name, (ms, n) in prof.items():
        key = ("hexplane" if "k_hexplane" in name else "mlp" if "k_mlp" in name else "rasterizer" if re.search(r"k_(preprocess|render|radix|sorted|scan|duplicate|tile_ranges|sh_)", name)
               else "loss_sky" if re.search(r"k_(ssim|loss|sky)", name) else "other_hip" if re.search(r"\bk_[a-z]", name) else "torch_glue")
        groups[key] = groups.get(key, 0.0) + ms
    return {"ms_per_step": round(dt / steps * 1e3, 4), "ms_per_step_repeat": round(dt2 / steps * 1e3, 4), "iters_per_s": round(steps / dt, 2), "steps": steps,
            "workload": f"S3Gaussian fine-stage step (50 000 of 55 000 iterations): EMD deformation network (HexPlane 4 x 6 planes x 32 ch + temporal table + dx / do / dshs heads, "
                        f"run-script flags; the dx / do regularisers formed inside the head kernels) -> raster -> sky cube map + blend -> L1 + depth L2 + D-SSIM + sky BCE + residual regularisers -> backward to Gaussians, planes, "
                        f"table, heads -> densification statistics; N={N}, {H}x{W}, no optimiser step (as `value`)",
            "step_issue": f"hipGraph replay, one graph per frame, {len(frames)} frames of the {F}-frame clip round robin", "graph_record_s": round(t_record, 2),
            "visible_V": V, "duplicates_D": D,
            "kernel_ms_per_step_eager": round(total, 3), "kernel_groups_ms": {k: round(v, 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])},
            "torch_glue": {"ms_per_step": round(glue, 3), "launches_per_step": round(glue_n, 1)},
            "kernels_measured_with": f"torch.profiler (roctracer) over {profile_steps} eager steps behind the replayed block: device time per launch",
            "longest_kernels": kernels, "outside_value": True, "wall_s": round(time.perf_counter() - t_all, 1)}


if __name__ == "__main__":
    print(json.dumps(measure(torch.device("cuda", 0))))
