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


PMC_TRAFFIC_CSV = "r02_pmc_hbm_traffic.csv"
PMC_VALU_CSV = "r02_pmc_valu.csv"


def _pmc_path(name):
    for cand in (name, name.replace("r02_", "r01_")):
        path = os.path.join(ROOT, "profiles", cand)
        if os.path.exists(path):
            return path
    return None


def pmc_traffic(stage, passes):
    """HBM-side bytes per iteration of `stage` from the COMMITTED PMC summary (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, separate passes, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md; produced by
    profiles/make_pmc_summary.py on this exact workload).  A file constant, NOT measured in this run: counters cannot be read
    from inside the process; the JSON line says so ("static": true + the file).  None when the summary is missing."""
    path = _pmc_path(PMC_TRAFFIC_CSV)
    if path is None:
        return None
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
            launches = float(f[5]) if len(f) > 5 else 1.0           # launches per iteration, counted by make_pmc_summary.py
            tot += per_launch * launches
    return tot or None


def issue_bound(stage, avg_ms):
    """Secondary bound of the render kernels: the fraction of the MEASURED vector-issue capacity the stage's dominant kernel uses.
    Instruction counts per launch come from the committed counter summary (rocprofv3 --pmc SQ_INSTS_VALU ..., static, like
    `traffic`); the capacity per instruction class from profiles/microbench_issue_rate.hip as measured on MI355X
    (profiles/r02_issue_rate_microbench.txt): a plain fp32 VALU instruction occupies its SIMD for 1.32 ns with 8 waves resident
    (2.35 cycles at the 1.78 GHz the chip holds under that load -- not the 0.83 ns of "2 cycles at 2.4 GHz"), v_pk_* / DPP /
    v_readlane 2.0 ns, transcendentals 3.4 ns.  Counters do not split the classes, so two fractions are given: every instruction
    priced as a plain one (lower bound of the utilisation) and priced with the kernel's static instruction mix (from the ISA of
    its inner loop)."""
    path = _pmc_path(PMC_VALU_CSV)
    if path is None:
        return None
    mix = {"k_render_backward_q": 0.55, "k_render_forward_q": 0.12}          # fraction of 2-slot instructions (pk / DPP / readlane) in the inner loop
    names = PMC_STAGE_KERNELS.get(stage, ())
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        f = line.rstrip("\n").split(",")
        if f[0] in names:
            n = float(f[2])
            plain_ns, wide_ns = 1.32, 2.0
            lo = n * plain_ns / 1024.0 * 1e-6 / avg_ms
            w = mix.get(f[0], 0.0)
            hi = n * ((1 - w) * plain_ns + w * wide_ns) / 1024.0 * 1e-6 / avg_ms
            return {"kernel": f[0], "SQ_INSTS_VALU_per_launch": n, "static": True, "source": "profiles/" + os.path.basename(path),
                    "ceiling_source": "profiles/r02_issue_rate_microbench.txt", "ns_per_plain_valu_per_simd": plain_ns,
                    "ns_per_pk_dpp_valu_per_simd": wide_ns, "issue_utilisation_all_plain": round(lo, 3),
                    "issue_utilisation_with_mix": round(hi, 3), "wide_instruction_fraction": w}
    return None


def algorithmic_bytes(N, V, D, HW, T, C, p, residual=False, C_bwd=None):
    """Compulsory HBM bytes per stage (C blended channels, p radix passes)."""
    Cf, C = C, (C if C_bwd is None else C_bwd)
    return {
        "preprocess": 68 * N + (12 * N if residual else 0) + V * (216 + 4 * Cf),
        # depth-ordered Gaussians -> (rect, count) gather, scan, (tile id, Gaussian id) pairs
        "scan_duplicate": 28 * N + 8 * V + 8 * D,
        # depth sort: the first pass reads the N depth keys and writes the V visible (key, id) pairs, two more passes over those
        # pairs; then p passes over the D (tile, id) pairs -- 8 B read + 8 B written per element and pass
        "radix_sort": 4 * N + 8 * V + 2 * 16 * V + p * 16 * D,
        "tile_ranges": 4 * D + 8 * T,
        "render_forward": D * (4 + 24 + 4 * Cf) + HW * (4 * (Cf + 1) + 8),
        "render_backward": D * (28 + 4 * C) + HW * (4 * (C + 1) + 8) + V * (24 + 4 * C),
        "preprocess_backward": V * ((24 + 4 * C) + 48 + 192) + N * (236 + 12),
    }


def cpu_baseline(scene, cam, frame=0, budget_s=15.0):
    """The reference's pure-PyTorch per-actor rigid transform (RigidNodes.transform_means / transform_quats, batched) +
    projection + cov3D + SH forward on the host cores.

    Thread count: measured on the MI355X host (256 hardware threads), torch's intra-op pool is fastest at 64 threads
    for these element-wise ops (70 M Gaussians/s at N = 1 M) and collapses beyond 128 (profiles/r01_cpu_thread_scan.txt),
    so min(cores, 64) threads are used; both numbers are reported.
    """
    from oracle import torch_ref
    from emd_amd.motion import build_actor_pose
    avail = os.cpu_count() or 1
    cores = min(avail, 64)
    torch.set_num_threads(cores)
    means, scales, rots, shs = scene.means, torch.exp(scene.log_scales), scene.quats, scene.shs
    V, Pm, cp = cam.world_view_transform, cam.full_proj_transform, cam.camera_center
    has_actors = scene.actor_id is not None
    pose = build_actor_pose(scene.actor_quats, scene.actor_trans, scene.actor_valid, frame) if has_actors else None

    def one():
        m, q = means, rots
        if has_actors:       # explicit motion first, as the reference does (rigid.py:478-568), then the projection path
            m, q, _ = torch_ref.motion_transform(means, rots, None, scene.actor_id, pose)
        return torch_ref.reference_projection_cpu(m, scales, q, shs, V, Pm, cp, 3)

    times = []
    with torch.no_grad():
        for _ in range(2):   # warm-up (thread pool, allocator)
            t0 = time.perf_counter()
            one()
            first = time.perf_counter() - t0
        reps = int(max(3, min(20, budget_s / max(first, 1e-3))))
        for _ in range(reps):
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "iters/s (rigid transform + projection + cov3D + SH forward stage only)", "cores": cores,
            "cores_available": avail, "kind": "port", "ms_per_call": med * 1e3, "gaussians_per_s": scene.N / med,
            "sample": f"reference pure-PyTorch per-actor rigid transform ({'32 actors' if has_actors else 'none'}) + geom_transform_points + "
                      f"get_covariance + eval_sh(deg 3) forward, N={scene.N}, fp32, median of {len(times)} calls after 2 warm-ups, "
                      f"{cores} torch threads of {avail} hardware threads"}


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
    ap.add_argument("--no-track-heads", action="store_true", help="leave the learned per-actor track offsets out of the step")
    ap.add_argument("--densify-stats", action="store_true", help="also accumulate the per-view densification statistics every step (one launch)")
    ap.add_argument("--eager", action="store_true", help="issue every step from Python instead of replaying it from a hipGraph (1 GPU)")
    args = ap.parse_args()

    from emd_amd import dp, scenes, _lib
    from emd_amd import RasterCall, RasterOptions
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
    num_cams = dp.rig_size(world)             # 1 / 2 / 4 cameras on 1 / 2 / 4 GPUs (rank <-> camera of one timestamp), 6 on 8 GPUs
    scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=num_actors, pts_per_actor=5000,
                              num_frames=num_frames, seed=1)
    model = StreetGaussians(scene, dev, track_heads=not args.no_track_heads)
    params = [p for p in model.parameters()]
    bg = torch.zeros(3)
    g3 = torch.Generator().manual_seed(3)
    target = torch.rand(3, H, W, generator=g3).to(dev)

    factored = world > 1 or args.factored_sh
    # options of THIS run's rasterizer calls (an instance, handed to every call: nothing process-wide is written)
    opts = RasterOptions(compute_normal=not args.no_normal, factored_sh_grad=factored, no_sync=not args.sync_count)
    cams, campos_dev = {}, {}
    stats = None
    if args.densify_stats:
        stats = [torch.zeros(N, 1, device=dev), torch.zeros(N, 1, device=dev), torch.zeros(N, device=dev)]

    def cam_for(step):
        # views are ordered timestamp-major and dealt out by dp.view_for: every rank renders a DISTINCT (frame, camera);
        # with 8 ranks on the 6-camera rig two ranks hold cameras of the next timestamp (dp.frame_and_camera)
        f, c = dp.frame_and_camera(step, rank, world, num_frames, num_cams)
        if (f, c) not in cams:
            cams[(f, c)] = scenes.rig_camera(f, c, H, W)
            campos_dev[(f, c)] = cams[(f, c)].camera_center.to(dev)      # once per camera: no per-step host-to-device copy
        return f, c, cams[(f, c)]

    def one_step(step, options=opts):
        f, c, cam = cam_for(step)
        for p in params:
            p.grad = None
        rec = RasterCall()
        xchg = None
        if options.factored_sh_grad:
            # SH gradient (81 % of the gradient bytes): rank-one factors, 12 B per Gaussian and rank instead of all-reducing
            # 192 B per Gaussian; the collectives are issued from inside backward(), right behind K8 (emd_amd/dp.py)
            xchg = dp.GradientExchange(campos_dev[(f, c)], actor_ids=model.actor_id if model.has_actors else None)
            rec.on_backward = xchg.start
        out = render(model, cam, bg, frame=f, iteration=step, options=options, record=rec)
        if xchg is not None:
            xchg.actor_pose = None if out["actor_pose"] is None else out["actor_pose"].detach()   # (no reference into the autograd graph)
        loss = l1_loss(out["render"], target)
        loss.backward()
        if xchg is not None:
            xchg.finish(model._features, model._xyz, model.active_sh_degree, other_params=params)
            rec.on_backward = None
        elif world > 1:
            dp.allreduce_gradients(params)
        if stats is not None:
            dp.add_densification_stats(out["viewspace_points"].grad, out["radii"], *stats)
        return out

    # Size the binning workspace once, with synchronising forwards over the clip (the duplicate count D moves with
    # the ego pose and the actors); afterwards the async path never reads D back.  Overflow of any timed step is
    # checked after the timed region from the per-step device status words.
    for s_ in range(args.warmup + args.steps):          # the cameras are dataset state: built (and their centres uploaded) before timing
        cam_for(s_)
    from emd_amd import rasterizer as _rz
    sync_opts = opts.replace(no_sync=False)
    out = one_step(0, sync_opts)
    dmax = out["raster_call"].last_status()["num_rendered"]
    with torch.no_grad():
        for s_ in sorted(set(list(range(0, args.warmup + args.steps, 7)) + [args.warmup + args.steps - 1])):
            f, c, cam = cam_for(s_)
            o = render(model, cam, bg, frame=f, options=sync_opts)
            dmax = max(dmax, o["raster_call"].last_status()["num_rendered"])
    _rz._capacity_hint[(dev.index, H, W)] = int(dmax * 1.3) + 1024
    # ---- the step as a hipGraph: the ~45 launches of a step are captured once and replayed; everything that changes from
    # step to step (camera matrices, frame index, frame time) lives in device buffers selected by ONE device index `sel`, rewritten
    # before every replay.  The host then spends ~20 us per step instead of ~1-2 ms of Python + launch calls, i.e. the run is
    # GPU-bound whatever the host is doing.  (--eager, or a failed capture, issues the same step from Python.)
    total_steps = args.warmup + args.steps
    graph = None
    out = o = None          # no autograd graph of an eager step may be alive at capture time (its AccumulateGrad nodes are bound to the eager stream)
    import gc
    gc.collect()            # ... including graphs held only by reference cycles (RasterCall <-> GradientExchange)
    status_log = torch.zeros(max(total_steps, 1), 4, dtype=torch.int32, device=dev)
    gstate = {}
    if not args.eager and opts.no_sync:
        try:
            import types
            views = [cam_for(s_) for s_ in range(total_steps)]
            blocks = torch.stack([torch.cat([bg.reshape(-1).float(), c_.world_view_transform.reshape(-1), c_.full_proj_transform.reshape(-1),
                                             c_.camera_center.reshape(-1)]) for _, _, c_ in views]).to(dev)            # [steps, 38]
            frame_of = torch.tensor([f_ for f_, _, _ in views], dtype=torch.int32, device=dev)
            sel = torch.zeros(1, dtype=torch.int64, device=dev)
            cam0 = views[0][2]

            def graph_body():
                for p in params:
                    p.grad = None
                blk = blocks.index_select(0, sel)[0]
                frame_dev = frame_of.index_select(0, sel)
                cam_g = types.SimpleNamespace(image_height=H, image_width=W, tanfovx=cam0.tanfovx, tanfovy=cam0.tanfovy,
                                              world_view_transform=blk[3:19].view(4, 4), full_proj_transform=blk[19:35].view(4, 4),
                                              camera_center=blk[35:38])
                rec_g = RasterCall()
                o = render(model, cam_g, blk[0:3], frame=frame_dev, iteration=0, options=opts, record=rec_g)
                l1_loss(o["render"], target).backward()
                if stats is not None:
                    dp.add_densification_stats(o["viewspace_points"].grad, o["radii"], *stats)
                status_log.index_copy_(0, sel, o["raster_call"].status.view(1, 4))
                # what the gradient exchange of a multi-GPU step reads after the replay: graph-static tensors
                gstate["rec"], gstate["campos"] = rec_g, blk[35:38]
                gstate["pose"] = None if o["actor_pose"] is None else o["actor_pose"].detach()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i_ in range(3):
                    sel.fill_(i_)
                    graph_body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # (multi-rank: the process group's watchdog thread polls events while we capture; "thread_local" keeps its calls from
            #  invalidating the capture -- nothing of the exchange is captured)
            with torch.cuda.graph(graph, stream=side, capture_error_mode="global" if world == 1 else "thread_local"):
                graph_body()
            torch.cuda.synchronize()
        except Exception as e:          # capture is an optimisation of the host side only: fall back to issuing the step from Python
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None

    def timed_step(step):
        if graph is not None:
            sel.fill_(step)
            graph.replay()
            if opts.factored_sh_grad:
                # the exchange is issued behind the replay (RCCL collectives are not captured): what follows K8 inside the graph is
                # ~20 us of pose / track-head backward, so nothing is lost against starting it from inside backward()
                xchg = dp.GradientExchange(gstate["campos"], actor_ids=model.actor_id if model.has_actors else None, actor_pose=gstate["pose"])
                xchg.start(gstate["rec"])
                xchg.finish(model._features, model._xyz, model.active_sh_degree, other_params=params)
            elif world > 1:
                dp.allreduce_gradients(params)
        else:
            o = one_step(step)
            status_log[step].copy_(o["raster_call"].status)

    for s in range(args.warmup):
        timed_step(s)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    if graph is None:
        _lib.profile_enable(True)
        _lib.profile_read()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        timed_step(args.warmup + s)
    t_enqueue = time.perf_counter() - t0      # host time to enqueue the K steps (the GPU runs behind it)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    stage_region = "the timed region"
    if graph is not None:
        # Per-stage HIP events are recorded by host code, which does not run when a graph is replayed: the stage durations of the
        # roofline block come from the SAME steps issued eagerly right after the timed region (same kernels, same inputs).
        n_prof = min(args.steps, 20)
        _lib.profile_enable(True)
        _lib.profile_read()
        for s in range(n_prof):
            one_step(args.warmup + s)
        torch.cuda.synchronize()
        stage_region = f"{n_prof} eager repetitions of the timed steps, run right after the timed region (graph replays execute no host-side event records)"
        prof_steps = n_prof
    else:
        prof_steps = args.steps
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    if graph is not None:          # release the captured graph and its memory pool explicitly, in a quiet state
        torch.cuda.synchronize()
        graph.reset()
        graph = "released"
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    st_all = status_log[args.warmup:args.warmup + args.steps].cpu().numpy().astype("int64") & 0xFFFFFFFF
    overflow = int(st_all[:, 1].sum())
    assert overflow == 0, "binning workspace overflowed during the timed region"

    if rank == 0:
        # V and D of EVERY timed step (device status words of rank 0's views), not of one frame
        Ds, Vs = st_all[:, 0], st_all[:, 2]
        D, V = float(Ds.mean()), float(Vs.mean())
        T = ((W + 15) // 16) * ((H + 15) // 16)
        C = 7 if opts.compute_normal else 4
        passes = (max(T - 1, 1).bit_length() + 7) // 8            # radix passes over the D duplicates (tile bits)
        ab = algorithmic_bytes(N, V, D, H * W, T, C, passes, C_bwd=4)   # the L1 loss sends no gradient into the normal image
        stages = {}
        for name, (ms, cnt) in prof.items():
            if cnt and name in ab:
                avg = ms / prof_steps                                # per iteration (a stage may open twice per step)
                stages[name] = {"ms": round(avg, 4), "alg_GB": round(ab[name] / 1e9, 4),
                                "GBps": round(ab[name] / 1e9 / (avg * 1e-3), 1)}
        dom = max(stages, key=lambda k: stages[k]["ms"])
        kernel_ms = sum(v["ms"] for v in stages.values())
        total_alg = sum(ab.values())
        full = (N, H, W) == (2_000_000, 1066, 1600)
        traffic = pmc_traffic(dom, passes) if full else None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(stages[dom]["GBps"] / HBM_PEAK_GBS, 4),
                    "traffic": traffic,
                    "traffic_static": {"static": True, "source": None if _pmc_path(PMC_TRAFFIC_CSV) is None else "profiles/" + os.path.basename(_pmc_path(PMC_TRAFFIC_CSV)),
                                       "note": "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of the same command, bytes per launch; a committed file constant, not measured in this run"},
                    "secondary_bound": "fp32 vector issue rate: the render kernels are issue-bound (DESIGN.md section 6)",
                    "issue": issue_bound(dom, stages[dom]["ms"]) if full else None,
                    "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": stages[dom]["ms"],
                    "stage_durations_measured_over": stage_region,
                    "per_step": {"steps": int(len(Ds)), "D_min": int(Ds.min()), "D_mean": round(D, 1), "D_max": int(Ds.max()),
                                 "V_min": int(Vs.min()), "V_mean": round(V, 1), "V_max": int(Vs.max()),
                                 "note": "duplicates D and visible Gaussians V of every timed step, read from the device status words after the timed region; algorithmic bytes use the means"},
                    "whole_iter": {"algorithmic_GB": round(total_alg / 1e9, 3), "kernel_ms": round(kernel_ms, 3),
                                   "GBps": round(total_alg / 1e9 / (kernel_ms * 1e-3), 1),
                                   "frac": round(total_alg / 1e9 / (kernel_ms * 1e-3) / HBM_PEAK_GBS, 4)},
                    "stages": stages}
        if world == 1:
            mapping = "1 GPU: camera 0 of frame (step mod 50)"
        elif world == num_cams:
            mapping = f"{world} GPUs: rank r <-> camera r of the {num_cams}-camera rig, all ranks of a step share one timestamp"
        else:
            mapping = (f"{world} GPUs on the {num_cams}-camera rig: view (step * {world} + rank) of the timestamp-major view list, i.e. "
                       f"{world - num_cams} rank(s) per step render cameras of the NEXT timestamp; every rank renders a distinct view")
        res = {
            "metric": "train iters/s (fwd+bwd) @1066x1600, 2M Gaussians; 1/2/4/8 GPU",
            "value": world * args.steps / dt, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "host_enqueue_ms_per_step": round(t_enqueue / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 50-frame dynamic clip, per-actor rigid motion on 2M Gaussians "
                                   "(32 actors x 5000" + ("" if args.no_track_heads else ", learned per-actor track offsets") + "), SH degree 3, "
                                   "one 1066x1600 view per GPU per step, L1 loss, fwd+bwd to all 59 floats/Gaussian + actor poses"
                                   + ("" if args.no_track_heads else " + track heads"),
                       "gaussians": N, "height": H, "width": W, "visible_V": round(V, 1), "duplicates_D": round(D, 1), "tiles_T": T,
                       "radix_passes_depth": "3 x 9 bits above the near plane: first over the N keys (compacting to V), two over the V pairs",
                       "radix_passes_tile_on_D": passes, "blended_channels_C": C, "views_per_step": world,
                       "rig_cameras": num_cams, "rank_view_mapping": mapping, "track_heads": not args.no_track_heads,
                       "densification_stats_in_step": bool(args.densify_stats),
                       "step_issue": ("hipGraph replay (one capture, device-resident per-step inputs" + ("; the gradient exchange is issued behind each replay)" if world > 1 else ")"))
                                     if graph is not None else "eager (Python issues every launch)",
                       "parallelism": f"view-parallel dp{world}", "count_readback": bool(args.sync_count),
                       "gradient_exchange": ("none (1 GPU)" if world == 1 else
                                             "SH gradient as rank-one factors: all-gather of 12 B per Gaussian and rank + camera centres + per-view actor "
                                             "pose tables, dense average rebuilt locally; one RCCL all-reduce (AVG) of the remaining 44 B per Gaussian "
                                             "(one slab, started inside backward()) + actor poses / track heads")},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:      # the CPU leg is timed on rank 0 of the 1-GPU run only
            f0, _, cam0 = cam_for(0)
            res["cpu_baseline"] = cpu_baseline(scene, cam0, frame=f0)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
