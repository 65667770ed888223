"""An S3Gaussian-style training step assembled from every HIP piece of this repository at the headline size (2 M Gaussians,
32 actors, 1066 x 1600): fused-motion rasterizer -> sky cube map (1024^2 faces) + blend -> L1 + depth L2 + D-SSIM + sky BCE ->
backward to all Gaussian parameters, actor poses and the cube map -> per-view densification statistics.  (The bench.py metric is
the L1-only step of BASELINE.json; this is the same step with the reference's full loss and sky model.)  One JSON line.
    python profiles/bench_full_step.py > profiles/r01_full_step.json
    python profiles/bench_full_step.py --fine > profiles/r01_full_step_fine.json
--fine: the "fine" stage of S3Gaussian's training (train.py after coarse_iterations): no actors, the self-supervised EMD
deformation network (HexPlane 4 x 6 planes x 32 channels, temporal table, heads dx / do / dshs / feat; run-script flags) runs in
front of the rasterizer on all 2 M Gaussians and is trained through it, plus the residual regularisers of train.py.
--feat (with --fine): the reference's default feature head (feat_head=True, arguments/gaussian_options.py:163): the two feature
images rendered as extra colour sets of the main rasterizer call (one binning, three colour sets) and an L2 loss on each;
--feat-separate: the same step issued the way the reference issues it, as three rasterizer calls.
--graph: the step of every frame captured once into a hipGraph (one graph per frame of the clip, all in one memory pool: camera, frame time
and sky rays are host constants of a frame and are baked into its graph) and replayed -- the host then spends ~0.05 ms per step
instead of 15-19 ms issuing ~170 launches from Python, so the rate is the GPU's whatever the host is doing; one replay is checked
against the eager step of the same frame before the timed region.
--fused-l1 (with --fine): the dx / do regularisers formed by the head kernels (render(..., fused_l1=("dx", "do"))) instead of abs_mean launches --
measured SLOWER (12.38 against 12.30 ms: the regularised instantiation of the narrow heads' forward takes 172 us against 130-140), so not the default.
--adam / --torch-adam: also take the optimiser step (train.py:428) with emd_amd.optim.Adam / torch.optim.Adam over the groups of
gaussian_model.py:188-199 (per-group learning rates, eps 1e-15)."""
import json
import sys
import time
import types

import torch

sys.path.insert(0, ".")
from emd_amd import dp, scenes, RasterOptions  # noqa: E402
from emd_amd.loss import image_loss  # noqa: E402
from emd_amd.model import StreetGaussians, abs_mean, render, residual_abs_mean  # noqa: E402
from emd_amd.sky import SkyCubeMap, composite_s3g  # noqa: E402

dev = torch.device("cuda", 0)
FINE = "--fine" in sys.argv
FEAT_SEP = "--feat-separate" in sys.argv
FEAT = FINE and ("--feat" in sys.argv or FEAT_SEP)
N, H, W, F = 2_000_000, 1066, 1600, 50
scene = scenes.make_static_scene(N, seed=0)
if not FINE:
    scene = scenes.add_actors(scene, num_actors=32, pts_per_actor=5000, num_frames=F, seed=1)
model = StreetGaussians(scene, dev)
params = list(model.parameters())
deform = embeddings = None
if FINE:
    from emd_amd.deformation import DeformOptions, deform_network  # noqa: E402
    torch.manual_seed(5)
    deform = deform_network(DeformOptions()).to(dev)
    deform.deformation_net.set_aabb([120.0, 30.0, 10.0], [0.0, -30.0, -2.0])
    for n_, p_ in deform.named_parameters():          # non-zero heads so that the residuals (and their gradients) are live
        if p_.dim() > 1 and "grid" not in n_:
            p_.data.mul_(0.05)
    embeddings = torch.nn.Parameter(torch.zeros(N, 4, device=dev))
    params += list(deform.parameters()) + [embeddings]
optimizer = None
if "--adam" in sys.argv or "--torch-adam" in sys.argv:
    from emd_amd.optim import Adam  # noqa: E402
    groups = [{"params": [model._xyz], "lr": 1.6e-4, "name": "xyz"}, {"params": [model._features], "lr": 2.5e-3, "name": "f"},
              {"params": [model._opacity], "lr": 0.05, "name": "opacity"}, {"params": [model._scaling], "lr": 5e-3, "name": "scaling"},
              {"params": [model._rotation], "lr": 1e-3, "name": "rotation"}]
    if model.has_actors:
        groups.append({"params": [model.instances_quats, model.instances_trans], "lr": 1e-5, "name": "ins_pose"})
    if FINE:
        groups += [{"params": deform.get_mlp_parameters(), "lr": 1.6e-5, "name": "deformation"},
                   {"params": deform.get_grid_parameters(), "lr": 1.6e-4, "name": "grid"}, {"params": [embeddings], "lr": 2.5e-3, "name": "embedding"}]
    # (--graph: the optimiser step is recorded into the graphs too: step counts and learning rates on the device, `capturable`)
    optimizer = (Adam if "--adam" in sys.argv else torch.optim.Adam)(groups, lr=0.0, eps=1e-15, capturable="--graph" in sys.argv)
sky = SkyCubeMap(types.SimpleNamespace(sky_resolution=1024, sky_white_background=False, white_background=False), device=dev)
g = torch.Generator().manual_seed(3)
gt = torch.rand(3, H, W, generator=g).to(dev)
gt_depth = (torch.rand(1, H, W, generator=g) * 90).to(dev)
gt_feat = torch.rand(3, H, W, generator=g).to(dev)
sky_mask = (torch.rand(1, H, W, generator=g) < 0.2).to(dev)
not_sky = ~sky_mask
accum, denom, maxr = (torch.zeros(N, device=dev) for _ in range(3))
cams, skycams = {}, {}
for f in range(F):
    cam = scenes.rig_camera(f, 0, H, W)
    K = torch.tensor([[W / (2 * cam.tanfovx), 0, W / 2], [0, H / (2 * cam.tanfovy), H / 2], [0, 0, 1]], dtype=torch.float32)
    cams[f] = cam
    skycams[f] = types.SimpleNamespace(image_height=H, image_width=W, intrinsic=K.to(dev), world_view_transform=cam.world_view_transform.to(dev))
bg = torch.zeros(3)
OPTS = [RasterOptions(no_sync=False)]


def step(s):
    f = s % F
    for p in params:
        p.grad = None
    sky.sky_cube_map.grad = None
    out = render(model, cams[f], bg, frame=f, deformation=deform, embeddings=embeddings, iteration=12000 + s, time=f / (F - 1), options=OPTS[0],
                 render_feat=FEAT and not FEAT_SEP, need_feat=FEAT, fused_l1=("dx", "do") if (FINE and "--fused-l1" in sys.argv) else ())
    if FEAT and FEAT_SEP:        # the reference's three calls: main pass above + one call per feature set, same rasterizer object
        bd, dd = out["boundary"], out["ddict"]
        base = dict(means3D=bd["means3D"], means2D=out["viewspace_points"], opacities=bd["opacities"], scales=bd["scales"],
                    rotations=bd["rotations"], raw_params=bd["raw_params"])
        out["feat_c"] = out["rasterizer"](shs=None, colors_precomp=dd["coarse"]["feat"], **base)[0]
        out["feat_f"] = out["rasterizer"](shs=None, colors_precomp=dd["fine"]["feat"], **base)[0]
    image, _ = composite_s3g(sky, skycams[f], out["render"], out["weight"])
    loss, _ = image_loss(image, gt, out["depth"], gt_depth, not_sky, out["weight"], sky_mask)
    if FINE:                                           # residual regularisers (train.py: lambda_dx / do / dshs on both levels)
        for lvl in ("coarse", "fine"):
            d = out["ddict"][lvl]
            loss = loss + 0.001 * (residual_abs_mean(d, "dx") + residual_abs_mean(d, "do") + residual_abs_mean(d, "dshs"))
    if FEAT:
        loss = loss + 0.001 * (((out["feat_c"] - gt_feat) ** 2).mean() + ((out["feat_f"] - gt_feat) ** 2).mean())
    loss.backward()
    dp.add_densification_stats(out["viewspace_points"].grad, out["radii"], accum, denom, maxr)
    if optimizer is not None:
        optimizer.step()


dmax = 0
for f in range(0, F, 7):
    with torch.no_grad():
        o = render(model, cams[f], bg, frame=f, deformation=deform, embeddings=embeddings, iteration=12000, time=f / (F - 1), options=OPTS[0])
    dmax = max(dmax, o["raster_call"].last_status()["num_rendered"])
OPTS[0] = RasterOptions(no_sync=True, capacity_hint=int(dmax * 1.3) + 1024)
for s in range(10):
    step(s)
torch.cuda.synchronize()
K_STEPS = 50
graphs = None
if "--graph" in sys.argv:
    from emd_amd.graphs import StepGraphs
    from emd_amd.sky import _camera_rays_params
    step(0)
    torch.cuda.synchronize()
    want = {"xyz": model._xyz.grad.clone()}
    if FINE:
        want["grid"] = deform.deformation_net.grid.grids[-1][0].grad.clone()
    keep0 = {}

    def recorded(f):
        step(f)
        if f == 0:                          # graph 0's gradient tensors stay referenced for the self-check below (the later graphs
            keep0["xyz"] = model._xyz.grad  # recycle everything else)
            keep0["grid"] = deform.deformation_net.grid.grids[-1][0].grad if FINE else None
    sg = StepGraphs(recorded, range(F), prime=lambda f: _camera_rays_params(skycams[f]), freeze=[deform.deformation_net.grid] if FINE else [],
                    optimizers=[optimizer] if optimizer is not None else [], warmup=0)
    graphs = [sg.graphs[f] for f in range(F)]
    # self-check: the replay of frame 0 against the eager step of frame 0 (densification statistics accumulate: compare the increment)
    graphs[0].replay()
    torch.cuda.synchronize()
    tol = lambda a, b: float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-12
    if optimizer is None:                  # (with an optimiser in the step the parameters have moved since the eager step: nothing to compare)
        assert tol(keep0["xyz"], want["xyz"]), "graph replay: dL/dxyz differs from the eager step"
        if FINE:
            assert tol(keep0["grid"], want["grid"]), "graph replay: plane gradients differ from the eager step"
    for s in range(10):
        graphs[s % F].replay()
    torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(K_STEPS):
    if graphs is not None:
        graphs[(10 + s) % F].replay()
    else:
        step(10 + s)
t_host = time.perf_counter() - t0          # an eagerly issued step: on a slow host THIS, not the GPU, sets the rate
torch.cuda.synchronize()
dt = time.perf_counter() - t0
op = "S3G-style step: raster (fused motion) + sky cube map + blend + L1/depth/D-SSIM/sky-BCE + backward + densification stats"
if FEAT:
    op = ("S3G fine-stage step with the feature head, THREE rasterizer calls as the reference issues them: " if FEAT_SEP else
          "S3G fine-stage step with the feature head, main + feat_c + feat_f as ONE rasterizer call (one binning, three colour sets): ") + \
        "EMD deformation network -> raster -> sky + blend -> full loss + feature L2 + residual regularisers -> backward -> densification stats"
elif FINE:
    op = "S3G fine-stage step: EMD deformation network (HexPlane + temporal table + heads) -> raster -> sky + blend -> full loss + residual regularisers -> backward to Gaussians, planes, table, heads -> densification stats"
if optimizer is not None:
    op += " -> optimiser step (" + ("emd_amd.optim.Adam" if "--adam" in sys.argv else "torch.optim.Adam") + ")"
print(json.dumps({"op": op,
                  "gaussians": N, "height": H, "width": W, "steps": K_STEPS, "ms_per_step": round(dt / K_STEPS * 1e3, 4),
                  "iters_per_s": round(K_STEPS / dt, 1), "host_enqueue_ms_per_step": round(t_host / K_STEPS * 1e3, 4),
                  "host_bound": bool(t_host > 0.9 * dt),
                  "step_issue": "hipGraph replay (one graph per frame of the clip, one memory pool)" if graphs is not None else "eager (Python)"}))
