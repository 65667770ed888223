"""Micro-benchmark of the S3Gaussian deformation front-end at the reference's configuration
(arguments/gaussian_options.py:128-196 + run_dynamic_nvs.sh flags: 32-channel planes [64,64,64,25] x multires [1,2,4,8], 150 x 32
temporal table, 164 -> 64 trunk, heads dx / do / dshs / feat) on N Gaussians: forward + backward of deform_network, HIP events,
inputs resident in HBM.  Beside it the same step issued the way the reference's PyTorch code issues it (poc_fre, 24 grid_samples in
BOTH passes, F.interpolate + F.grid_sample + repeat for the temporal row, the [N,164] concat, one ReLU per head), on the GPU.
Prints one JSON line.
    python profiles/bench_deform.py [N] > profiles/r01_deform_microbench.json"""
import itertools
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from emd_amd.deformation import DeformOptions, deform_network  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2_000_000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
opt = DeformOptions()
net = deform_network(opt).to(dev)
net.deformation_net.set_aabb([120.0, 30.0, 10.0], [0.0, -30.0, -2.0])
g = torch.Generator().manual_seed(0)
point = (torch.rand(N, 3, generator=g) * torch.tensor([120.0, 60.0, 12.0]) + torch.tensor([0.0, -30.0, -2.0])).to(dev).requires_grad_(True)
scales, rotations = torch.randn(N, 3, generator=g).to(dev), torch.randn(N, 4, generator=g).to(dev)
opacity = torch.randn(N, 1, generator=g).to(dev).requires_grad_(True)
shs = torch.randn(N, 16, 3, generator=g).to(dev).requires_grad_(True)
emb = (torch.randn(N, 4, generator=g) * 0.1).to(dev).requires_grad_(True)
times = torch.full((N, 1), 0.37, device=dev)
gp_, go_, gs_ = torch.randn(N, 3, generator=g).to(dev), torch.randn(N, 1, generator=g).to(dev), torch.randn(N, 16, 3, generator=g).to(dev)
IT, CAM = 12000, 1


def hip_step():
    p, s, r, o, sh, dd = net(point, scales, rotations, opacity, shs, times, emb, IT, CAM, 0.1, True)
    return (p * gp_).sum() + (o * go_).sum() + (sh * gs_).sum() + dd["fine"]["feat"].sum()


def reference_formulation():
    """deform_network.forward as scene/deformation.py:187-527 issues it, on this module's parameters."""
    d = net.deformation_net
    aabb = d.grid.aabb

    def poc_fre(x, poc):
        e = (x.unsqueeze(-1) * poc).flatten(-2)
        return torch.cat([x, e.sin(), e.cos()], -1)

    def hexplane(pts, t):
        q = torch.cat(((pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0, t), dim=-1)
        outs = []
        for scale in d.grid.grids:
            feat = 1.0
            for grid, pair in zip(scale, itertools.combinations(range(4), 2)):
                interp = F.grid_sample(grid, q[:, list(pair)].view(1, 1, -1, 2), align_corners=True, mode="bilinear", padding_mode="border")
                feat = feat * interp.view(grid.shape[1], -1).t()
            outs.append(feat)
        return torch.cat(outs, dim=-1)

    def temporal(t, k):
        er = F.interpolate(d.weight[None, None], size=(k, 32), mode="bilinear", align_corners=True)
        grid = torch.cat([torch.arange(32, device=dev).unsqueeze(-1) / 31, torch.ones(32, 1, device=dev) * t[0, 0]], dim=-1)[None, None]
        e = F.grid_sample(er, (grid - 0.5) * 2, align_corners=True, mode="bilinear", padding_mode="reflection")
        return e.repeat(1, 1, t.shape[0], 1).squeeze()

    def level(pts_emb, t, coarse):
        hid = hexplane(pts_emb[:, :3], t)                       # evaluated in the fine pass too (deformation.py:256), then dropped
        k = 30 if coarse else d.int_lininterp(IT, 30, 150, 25000)
        feats = ([hid] if coarse else []) + [temporal(t, k), emb]
        h = (d.feature_out if coarse else d.feature_out_f)(torch.cat(feats, dim=-1))
        sfx = "" if coarse else "_f"
        return dict(dx=getattr(d, "pos_deform" + sfx)(h), do=getattr(d, "opacity_deform" + sfx)(h),
                    dshs=getattr(d, "shs_deform" + sfx)(h).reshape(-1, 16, 3), feat=d.dino_head(h))

    def step():
        t = times + d.time_offset[CAM]
        c = level(poc_fre(point, net.pos_poc), t, True)
        f = level(poc_fre(point + c["dx"], net.pos_poc), t, False)
        p = point.clone() + c["dx"] + f["dx"]
        o = opacity.clone() + c["do"] + f["do"]
        sh = shs.clone() + c["dshs"] + f["dshs"]
        return (p * gp_).sum() + (o * go_).sum() + (sh * gs_).sum() + f["feat"].sum()
    return step


def run(fn, n):
    """-> (forward ms, backward ms, ms per step): the split from events around each half with a sync per step (the forward's
    includes whatever the host takes to issue its ~60 launches), the step time from n steps issued back to back."""
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0

    def one(split):
        for t_ in (point, opacity, shs, emb):
            t_.grad = None
        for p in net.parameters():
            p.grad = None
        if split:
            e[0].record()
        loss = fn()
        if split:
            e[1].record()
        loss.backward()
        if split:
            e[2].record()
            torch.cuda.synchronize()
            return e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    for _ in range(n):
        a, b = one(True)
        tf += a; tb += b
    torch.cuda.synchronize()
    e[0].record()
    for _ in range(n):
        one(False)
    e[1].record()
    torch.cuda.synchronize()
    return tf / n, tb / n, e[0].elapsed_time(e[1]) / n


run(hip_step, 2)
hf, hb, hstep = run(hip_step, 10)
if "--hip-only" in sys.argv:       # for rocprofv3 --kernel-trace: only the product path
    print(json.dumps({"N": N, "hip_path_step_ms": round(hstep, 3), "hip_path_forward_ms": round(hf, 3), "hip_path_backward_ms": round(hb, 3)}))
    sys.exit(0)
# the same step with the trunk + heads as hipBLASLt GEMMs + element-wise launches (round 1's path) instead of the fused MFMA kernels
opt.fused_mlp = False
run(hip_step, 2)
gf, gb, gstep = run(hip_step, 10)
opt.fused_mlp = True
geometry = "uniform"
if "--street" in sys.argv:
    # the bench scene's geometry (ground / facade sheets) instead of points uniform in the box: what the LDS
    # aggregation of the HexPlane backward sees in training
    from emd_amd import scenes
    point = scenes.make_static_scene(N, seed=0).means.to(dev).requires_grad_(True)
    geometry = "street scene (emd_amd.scenes.make_static_scene)"
    run(hip_step, 2)
    hf, hb, hstep = run(hip_step, 10)
# parity of the two formulations on the spot (values of the loss and of the table / offset gradients)
lv = float(hip_step())
ref = reference_formulation()
lr = float(ref())
run(ref, 1)
rf, rb, rstep = run(ref, 3)
print(json.dumps({"op": "S3Gaussian deform_network forward / backward, reference configuration, run-script flags", "N": N, "points": geometry,
                  "hip_path_step_ms": round(hstep, 3), "hip_path_forward_ms": round(hf, 3), "hip_path_backward_ms": round(hb, 3),
                  "gemm_path_step_ms": round(gstep, 3), "gemm_path_forward_ms": round(gf, 3), "gemm_path_backward_ms": round(gb, 3),
                  "reference_formulation_step_ms": round(rstep, 3), "reference_formulation_forward_ms": round(rf, 3),
                  "reference_formulation_backward_ms": round(rb, 3), "loss_hip_path": lv, "loss_reference_formulation": lr}))
