#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: train iters/s (fwd+bwd) @1066x1600, 2M Gaussians; 1/2/4/8 GPU.

One step = one pass of the hot path over one camera view per GPU:
  activations + per-frame actor pose table -> fused explicit-motion transform + projection + SH colour ->
  tile duplication + radix sort -> per-tile alpha compositing (forward) -> L1 loss vs a fixed target ->
  backward to all 59 floats per Gaussian and the actor poses (-> RCCL all-reduce of the gradients for N > 1).
Optimiser, densification and data loading are excluded (they are outside the path).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


PMC_STAGE_KERNELS = {"preprocess": ("k_preprocess",), "scan_duplicate": ("k_sorted_counts", "k_scan_publish", "k_duplicate"),
                     "radix_sort": ("k_radix_hist[N]", "k_radix_scatter[N]", "k_radix_scan_bins[N]", "k_radix_hist[D]",
                                    "k_radix_scatter[D]"),
                     "tile_ranges": ("k_tile_ranges",), "render_forward": ("k_render_forward_q",),
                     "render_backward": ("k_render_backward_q",), "preprocess_backward": ("k_preprocess_backward",)}


def pmc_traffic(stage, passes):
    """HBM-side bytes per iteration of `stage` from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, separate passes, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md; produced by
    profiles/make_pmc_summary.py) -- collected on this exact workload; None when the summary is missing."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.csv")
    if not os.path.exists(path):
        return None
    tot = 0.0
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        f = line.rstrip("\n").split(",")
        name = f[0]
        if name in PMC_STAGE_KERNELS.get(stage, ()):
            per_launch = (float(f[3]) + float(f[4])) * 1e6        # fetch_MB_x2 + write_MB
            launches = 1
            if name.endswith("[N]"):
                launches = 4 + (passes if "scan_bins" in name else 0)      # depth passes (+ the tile passes' scans)
            elif name.endswith("[D]"):
                launches = passes
            tot += per_launch * launches
    return tot or None


def pmc_valu(stage):
    """Fraction of the fp32 vector issue slots the stage's dominant kernel fills (profiles/r01_pmc_valu.csv: rocprofv3 --pmc
    SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x duration of the same dispatch)); None when the summary is missing."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_valu.csv")
    if not os.path.exists(path):
        return None
    names = PMC_STAGE_KERNELS.get(stage, ())
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        f = line.rstrip("\n").split(",")
        if f[0] in names:
            return {"kernel": f[0], "SQ_INSTS_VALU": float(f[2]), "utilisation": float(f[-1]), "peak_Ginst_per_s": 614.4,
                    "source": "profiles/r01_pmc_valu.csv"}
    return None


def algorithmic_bytes(N, V, D, HW, T, C, p, residual=False, C_bwd=None):
    """Compulsory HBM bytes per stage (C blended channels, p radix passes)."""
    Cf, C = C, (C if C_bwd is None else C_bwd)
    return {
        "preprocess": 68 * N + (12 * N if residual else 0) + V * (216 + 4 * Cf),
        # depth-ordered Gaussians -> (rect, count) gather, scan, (tile id, Gaussian id) pairs
        "scan_duplicate": 28 * N + 8 * V + 8 * D,
        # 4 passes over the N (depth, id) pairs + p passes over the D (tile, id) pairs, 8 B read + 8 B written each
        "radix_sort": 4 * 16 * N + p * 16 * D,
        "tile_ranges": 4 * D + 8 * T,
        "render_forward": D * (4 + 24 + 4 * Cf) + HW * (4 * (Cf + 1) + 8),
        "render_backward": D * (28 + 4 * C) + HW * (4 * (C + 1) + 8) + V * (24 + 4 * C),
        "preprocess_backward": V * ((24 + 4 * C) + 48 + 192) + N * (236 + 12),
    }


def cpu_baseline(scene, cam, budget_s=15.0):
    """The reference's pure-PyTorch projection + cov3D + SH forward on the host cores.

    Thread count: measured on the MI355X host (256 hardware threads), torch's intra-op pool is fastest at 64 threads
    for these element-wise ops (70 M Gaussians/s at N = 1 M) and collapses beyond 128 (profiles/r01_cpu_thread_scan.txt),
    so min(cores, 64) threads are used and reported.
    """
    from oracle import torch_ref
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    means, scales, rots, shs = scene.means, torch.exp(scene.log_scales), scene.quats, scene.shs
    V, Pm, cp = cam.world_view_transform, cam.full_proj_transform, cam.camera_center
    times = []
    with torch.no_grad():
        for _ in range(2):   # warm-up (thread pool, allocator)
            t0 = time.perf_counter()
            torch_ref.reference_projection_cpu(means, scales, rots, shs, V, Pm, cp, 3)
            first = time.perf_counter() - t0
        reps = int(max(3, min(20, budget_s / max(first, 1e-3))))
        for _ in range(reps):
            t0 = time.perf_counter()
            torch_ref.reference_projection_cpu(means, scales, rots, shs, V, Pm, cp, 3)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "iters/s (projection+cov3D+SH forward stage only)", "cores": cores,
            "kind": "port", "ms_per_call": med * 1e3, "gaussians_per_s": scene.N / med,
            "sample": f"reference pure-PyTorch geom_transform_points + get_covariance + eval_sh(deg 3) forward, "
                      f"N={scene.N}, fp32, median of {len(times)} calls after 2 warm-ups, {cores} torch threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=2_000_000)
    ap.add_argument("--height", type=int, default=1066)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-count", action="store_true", help="read the duplicate count back every forward (reference behaviour)")
    ap.add_argument("--no-normal", action="store_true", help="skip the normal image (unused by the training loss)")
    ap.add_argument("--factored-sh", action="store_true", help="use the multi-GPU SH-gradient factor exchange at any world size (1 GPU: measures its local cost)")
    args = ap.parse_args()

    from emd_amd import dp, scenes, _lib
    from emd_amd import RasterConfig, GaussianRasterizer
    from emd_amd.model import StreetGaussians, render, l1_loss

    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("EMD_BENCH_SHARE_GPU"):        # functional test of the N > 1 path on a 1-GPU box (with EMD_DP_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)              # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local)
    rank, world, local = dp.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    N, H, W = args.gaussians, args.height, args.width
    num_frames, num_actors = 50, 32
    scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=num_actors, pts_per_actor=5000,
                              num_frames=num_frames, seed=1)
    model = StreetGaussians(scene, dev)
    params = [p for p in model.parameters()]
    bg = torch.zeros(3)
    g3 = torch.Generator().manual_seed(3)
    target = torch.rand(3, H, W, generator=g3).to(dev)

    RasterConfig.compute_normal = not args.no_normal
    RasterConfig.factored_sh_grad = world > 1 or args.factored_sh
    RasterConfig.no_sync = not args.sync_count
    cams, campos_dev = {}, {}

    def cam_for(step):
        # frame of the 50-frame clip; rank r looks through rig camera r at that timestamp
        f, c = step % num_frames, rank % len(scenes.RIG_YAWS)
        if (f, c) not in cams:
            cams[(f, c)] = scenes.rig_camera(f, c, H, W)
            campos_dev[(f, c)] = cams[(f, c)].camera_center.to(dev)      # once per camera: no per-step host-to-device copy
        return f, cams[(f, c)]

    def one_step(step):
        f, cam = cam_for(step)
        for p in params:
            p.grad = None
        out = render(model, cam, bg, frame=f)
        loss = l1_loss(out["render"], target)
        loss.backward()
        if RasterConfig.factored_sh_grad:
            # SH gradient (81 % of the gradient bytes): exchange the rank-one factors, 12 B per Gaussian and rank instead of
            # all-reducing 192 B per Gaussian; everything else: in-place RCCL all-reduce (emd_amd/dp.py)
            dp.exchange_sh_gradient(model._features, model._xyz, campos_dev[(f, rank % len(scenes.RIG_YAWS))], model.active_sh_degree,
                                    actor_ids=model.actor_id if model.has_actors else None, actor_pose=out["actor_pose"],
                                    also_allreduce=params)
        return out

    # Size the binning workspace once, with synchronising forwards over the clip (the duplicate count D moves with
    # the ego pose and the actors); afterwards the async path never reads D back.  Overflow of any timed step is
    # checked after the timed region from the per-step device status words.
    for s_ in range(num_frames):          # the clip's cameras are dataset state: built (and their centres uploaded) before timing
        cam_for(s_)
    RasterConfig.no_sync = False
    out = one_step(0)
    st = GaussianRasterizer.last_status()
    from emd_amd import rasterizer as _rz
    dmax = st["num_rendered"]
    with torch.no_grad():
        for f in sorted(set(list(range(0, num_frames, 7)) + [num_frames - 1])):
            render(model, cam_for(f)[1], bg, frame=f)
            dmax = max(dmax, GaussianRasterizer.last_status()["num_rendered"])
    _rz._capacity_hint[(dev.index, H, W)] = int(dmax * 1.3) + 1024
    RasterConfig.no_sync = not args.sync_count
    statuses = []
    for s in range(args.warmup):
        one_step(s)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    _lib.profile_enable(True)
    _lib.profile_read()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        out = one_step(args.warmup + s)
        statuses.append(GaussianRasterizer._last["status"])
    t_enqueue = time.perf_counter() - t0      # host time to enqueue the K steps (the GPU runs behind it)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    overflow = int(torch.stack(statuses)[:, 1].sum().item()) if statuses else 0
    assert overflow == 0, "binning workspace overflowed during the timed region"

    if rank == 0:
        V, D = st["num_visible"], st["num_rendered"]
        T = ((W + 15) // 16) * ((H + 15) // 16)
        C = 7 if RasterConfig.compute_normal else 4
        passes = (max(T - 1, 1).bit_length() + 7) // 8            # radix passes over the D duplicates (tile bits)
        ab = algorithmic_bytes(N, V, D, H * W, T, C, passes, C_bwd=4)   # the L1 loss sends no gradient into the normal image
        stages = {}
        for name, (ms, cnt) in prof.items():
            if cnt and name in ab:
                avg = ms / args.steps                                # per iteration (a stage may open twice per step)
                stages[name] = {"ms": round(avg, 4), "alg_GB": round(ab[name] / 1e9, 4),
                                "GBps": round(ab[name] / 1e9 / (avg * 1e-3), 1)}
        dom = max(stages, key=lambda k: stages[k]["ms"])
        kernel_ms = sum(v["ms"] for v in stages.values())
        total_alg = sum(ab.values())
        roofline = {"bound": "hbm", "kernel": dom, "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(stages[dom]["GBps"] / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic(dom, passes) if (N, H, W) == (2_000_000, 1066, 1600) else None,
                    "traffic_source": "profiles/r01_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)",
                    "secondary_bound": "fp32 vector issue rate: the render kernels are VALU-bound (DESIGN.md section 3)",
                    "valu": pmc_valu(dom) if (N, H, W) == (2_000_000, 1066, 1600) else None,
                    "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": stages[dom]["ms"],
                    "whole_iter": {"algorithmic_GB": round(total_alg / 1e9, 3), "kernel_ms": round(kernel_ms, 3),
                                   "GBps": round(total_alg / 1e9 / (kernel_ms * 1e-3), 1),
                                   "frac": round(total_alg / 1e9 / (kernel_ms * 1e-3) / HBM_PEAK_GBS, 4)},
                    "stages": stages}
        res = {
            "metric": "train iters/s (fwd+bwd) @1066x1600, 2M Gaussians; 1/2/4/8 GPU",
            "value": world * args.steps / dt, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "host_enqueue_ms_per_step": round(t_enqueue / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 50-frame dynamic clip, per-actor rigid motion on 2M Gaussians "
                                   "(32 actors x 5000), SH degree 3, one 1066x1600 view per GPU per step, L1 loss, "
                                   "fwd+bwd to all 59 floats/Gaussian + actor poses",
                       "gaussians": N, "height": H, "width": W, "visible_V": V, "duplicates_D": D, "tiles_T": T,
                       "radix_passes_depth_on_N": 4, "radix_passes_tile_on_D": passes, "blended_channels_C": C, "views_per_step": world,
                       "parallelism": f"view-parallel dp{world}", "count_readback": bool(args.sync_count),
                       "gradient_exchange": ("none (1 GPU)" if world == 1 else
                                             "SH gradient as rank-one factors: all-gather of 12 B per Gaussian and rank, dense average "
                                             "rebuilt locally; one RCCL all-reduce (AVG) of the remaining 44 B per Gaussian (one slab) + actor poses")},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:      # the CPU leg is timed on rank 0 of the 1-GPU run only
            res["cpu_baseline"] = cpu_baseline(scene, cam_for(0)[1])
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
