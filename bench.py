#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: train iters/s (fwd+bwd) @1066x1600, 2M Gaussians; 1/2/4/8 GPU.

One step = one pass of the hot path over one camera view per GPU:
  activations + per-frame actor pose table -> fused explicit-motion transform + projection + SH colour ->
  tile duplication + radix sort -> per-tile alpha compositing (forward) -> L1 loss vs a fixed target ->
  backward to all 59 floats per Gaussian and the actor poses (-> RCCL all-reduce of the gradients for N > 1).
Optimiser, densification and data loading are excluded (they are outside the path).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W            # N > 1: starts its own N ranks as a fresh child process (before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              # ... or under an external torchrun (RANK / WORLD_SIZE from the environment)
    python bench.py --config 3 | 4                           # BASELINE configs[3] / [4] (at --gpus 1: the per-rank workload)

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


PMC_STAGE_KERNELS = {"preprocess": ("k_preprocess",), "scan_duplicate": ("k_sorted_counts", "k_duplicate"),
                     "radix_sort": ("k_radix_hist[N]", "k_radix_scatter[N]", "k_radix_scan_bins[N]", "k_radix_hist[D]",
                                    "k_radix_scatter[D]"),
                     "tile_ranges": ("k_tile_ranges", "k_tile_order"), "render_forward": ("k_render_forward_q",),
                     "render_backward": ("k_render_backward_q",), "preprocess_backward": ("k_preprocess_backward",)}


PMC_TRAFFIC_CSV = "r06_pmc_hbm_traffic.csv"
PMC_VALU_CSV = "r06_pmc_valu.csv"


def _pmc_path(name):
    for cand in (name, name.replace("r06_", "r05_"), name.replace("r06_", "r04_"), name.replace("r06_", "r03_"), name.replace("r06_", "r02_")):
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


ISA_MIX_TXT = "r06_render_isa_mix.txt"


def load_isa_mix():
    """{kernel: {valu, plain, pk, dpp, trans, wide_fraction, ns_per_valu, loop}} of each render kernel's hottest loop, from the committed disassembly summary
    (profiles/make_isa_mix.py -> profiles/r06_render_isa_mix.txt, its `MIX` lines).  Empty when the file is missing."""
    path = os.path.join(ROOT, "profiles", ISA_MIX_TXT)
    mix = {}
    if os.path.exists(path):
        for line in open(path):
            if line.startswith("MIX "):
                f = line.split()
                d = {}
                for kv in f[2:]:
                    k, v = kv.split("=", 1)
                    try:
                        d[k] = float(v)
                    except ValueError:
                        d[k] = v
                mix[f[1]] = d
    return mix


def issue_bound(stage, avg_ms):
    """Secondary bound of the render kernels: the fraction of the MEASURED vector-issue capacity the stage's dominant kernel uses.
    Instruction counts per launch come from the committed counter summary (rocprofv3 --pmc SQ_INSTS_VALU ..., static, like
    `traffic`); the capacity per instruction class from profiles/microbench_issue_rate.hip as measured on MI355X
    (profiles/r02_issue_rate_microbench.txt): a plain fp32 VALU instruction occupies its SIMD for 1.32 ns with 8 waves resident
    (2.35 cycles at the 1.78 GHz the chip holds under that load -- not the 0.83 ns of "2 cycles at 2.4 GHz"), v_pk_* / DPP /
    v_readlane 2.0 ns, transcendentals 3.4 ns.  Counters do not split the classes, so two fractions are given: every instruction
    priced as a plain one (lower bound of the utilisation) and priced with the instruction mix of the kernel's hottest loop, counted in the
    disassembly of the shipped build (profiles/make_isa_mix.py -> profiles/r06_render_isa_mix.txt; no literal in this file)."""
    path = _pmc_path(PMC_VALU_CSV)
    if path is None:
        return None
    mix = load_isa_mix()
    names = PMC_STAGE_KERNELS.get(stage, ())
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        f = line.rstrip("\n").split(",")
        if f[0] in names:
            n = float(f[2])
            plain_ns = 1.32
            lo = n * plain_ns / 1024.0 * 1e-6 / avg_ms
            m = mix.get(f[0])
            mixed_ns = float(m["ns_per_valu"]) if m else plain_ns
            hi = n * mixed_ns / 1024.0 * 1e-6 / avg_ms
            return {"kernel": f[0], "SQ_INSTS_VALU_per_launch": n, "static": True, "source": "profiles/" + os.path.basename(path),
                    "ceiling_source": "profiles/r02_issue_rate_microbench.txt", "ns_per_plain_valu_per_simd": plain_ns,
                    "ns_per_valu_with_mix": round(mixed_ns, 4), "issue_utilisation_all_plain": round(lo, 3),
                    "issue_utilisation_with_mix": round(hi, 3), "wide_instruction_fraction": None if not m else m["wide_fraction"],
                    "mix_source": None if not m else f"profiles/{ISA_MIX_TXT} (loop {m.get('loop')}: {int(m['plain'])} plain, {int(m['pk'])} v_pk_*, "
                                                     f"{int(m['dpp'])} DPP / cross-lane, {int(m['trans'])} transcendental of {int(m['valu'])} vector instructions)"}
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
    n_actors = int(scene.actor_quats.shape[1]) if has_actors else 0
    pose = build_actor_pose(scene.actor_quats, scene.actor_trans, scene.actor_valid, frame) if has_actors else None

    def one():
        m, q = means, rots
        if has_actors:       # explicit motion first, as the reference does (rigid.py:478-568), then the projection path
            m, q, _ = torch_ref.motion_transform(means, rots, None, scene.actor_id, pose)
        return torch_ref.reference_projection_cpu(m, scales, q, shs, V, Pm, cp, 3)

    times = []
    with torch.no_grad():
        for _ in range(3):   # warm-ups (thread pool, allocator), BASELINE.md section 2
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
            "sample": f"reference pure-PyTorch per-actor rigid transform ({str(n_actors) + ' actors' if has_actors else 'none'}) + geom_transform_points + "
                      f"get_covariance + eval_sh(deg 3) forward, N={scene.N}, fp32, median of {len(times)} calls after 3 warm-ups, "
                      f"{cores} torch threads of {avail} hardware threads"}


CONFIGS = {
    0: dict(name="BASELINE configs[0]: 10k static Gaussians, 256x256, projection + SH forward (plumbing)", gaussians=10_000, height=256, width=256,
            focal=272.0, actors=0, forward_only=True),
    1: dict(name="BASELINE configs[1]: single static frame, 1M Gaussians, 1066x1600, fwd+bwd", gaussians=1_000_000, height=1066, width=1600,
            focal=1700.0, actors=0, forward_only=False),
    2: dict(name="BASELINE configs[2]: 50-frame dynamic clip, per-actor rigid motion on 2M Gaussians", gaussians=2_000_000, height=1066,
            width=1600, focal=1700.0, actors=32, forward_only=False),
    3: dict(name="BASELINE configs[3]: 4-camera rig, 2M Gaussians + deformation residual (dx for all, dq for actor points) as inputs of the fused "
                 "transform, view-parallel", gaussians=2_000_000, height=1066, width=1600, focal=1700.0, actors=32, forward_only=False, rig=4,
            residual=True),
    4: dict(name="BASELINE configs[4]: 6-camera rig, 3M Gaussians, 48 actors, densification statistics every step + one density-control event",
            gaussians=3_000_000, height=1066, width=1600, focal=1700.0, actors=48, forward_only=False, rig=6, densify=True),
}


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun around it: start the N ranks ourselves, as a FRESH child process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), relay rank 0's JSON line and exit with
    the children's code.  Decided before anything has touched the GPU (no torch.cuda call has run in this process and none will: a process that
    has initialised HIP must never replace or fork itself into GPU work).  Returns None when this process is itself a rank (or N = 1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    import socket
    import subprocess
    share = bool(os.environ.get("EMD_BENCH_SHARE_GPU"))
    have = torch.cuda.device_count()          # (counts devices without initialising the runtime)
    if have < args.gpus and not share:
        print(f"[bench] --gpus {args.gpus} but this node shows {have} GPU(s) (EMD_BENCH_SHARE_GPU=1 EMD_DP_BACKEND=gloo puts all ranks on GPU 0 "
              "for a functional run)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd[1:])}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    lines = 0
    for line in child.stdout:
        if line.startswith("{"):
            lines += 1
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    rc = child.wait()
    if rc == 0 and lines != 1:
        print(f"[bench] the ranks printed {lines} JSON lines (expected one, from rank 0)", file=sys.stderr)
        rc = 3
    return rc


from emd_amd.graphs import select_step_inputs          # noqa: E402  (emd_select_step_inputs: the per-step inputs of a replayed step, one launch)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=(0, 1, 2, 3, 4),
                    help="BASELINE.json configs[k]: 0 = 10k static 256x256 (K1 forward next to the CPU leg), 1 = 1M static fwd+bwd, 2 = the headline (default), "
                         "3 = 4-camera rig, 2M + deformation residual (per-rank workload at --gpus 1), 4 = 6-camera rig, 3M, 48 actors, densification "
                         "statistics in every step and one density-control event after the timed steps (reported separately)")
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--repeats", type=int, default=5, help="extra back-to-back repeats of the timed block after it (min / median / max reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fine-stage", action="store_true", help="skip the `fine_stage` block (the S3G fine-stage step at the headline size, measured after the timed "
                    "region and outside `value`; 1 GPU, headline configuration only; ~10 s)")
    ap.add_argument("--sync-count", action="store_true", help="read the duplicate count back every forward (reference behaviour)")
    ap.add_argument("--no-normal", action="store_true", help="skip the normal image (unused by the training loss)")
    ap.add_argument("--factored-sh", action="store_true", help="use the multi-GPU SH-gradient factor exchange at any world size (1 GPU: measures its local cost)")
    ap.add_argument("--no-track-heads", action="store_true", help="leave the learned per-actor track offsets out of the step")
    ap.add_argument("--densify-stats", action="store_true", help="also accumulate the per-view densification statistics every step (one launch)")
    ap.add_argument("--settle-ms", type=float, default=150.0, help="untimed replays of the step before the warm-up steps until this much wall time has passed: "
                    "the clocks of an idle GPU take tens of milliseconds of load to settle, more than 5 warm-up steps of 1.4 ms provide (0 = off)")
    ap.add_argument("--eager", action="store_true", help="issue every step from Python instead of replaying it from a hipGraph (1 GPU)")
    ap.add_argument("--one-graph", action="store_true", help="multi-GPU / --factored-sh: record the step as ONE graph and issue the whole exchange behind its replay "
                    "(default: two graphs cut between the render backward and the projection backward, the factor gathers issued between their replays)")
    ap.add_argument("--exchange-only", action="store_true",
                    help="time ONLY the gradient exchange of a view-parallel step (GradientExchange.start + finish on a fixed backward's outputs): "
                         "separates communication from compute in the 2/4/8-GPU runs")
    ap.add_argument("--exchange", choices=("auto", "dense", "compact"), default="auto",
                    help="multi-GPU gradient exchange: dense (slab all-reduce + factor all-gather), compact (index + value rows of the visible Gaussians, "
                         "added in rank order) or auto = whichever moves fewer bytes per xGMI link at this world size (dp.compact_pays)")
    ap.add_argument("--densify-grad-threshold", type=float, default=None,
                    help="config 4's density-control event: view-space gradient threshold (default: the value that selects ~5 %% of the seen Gaussians of the synthetic scene)")
    args = ap.parse_args()
    rc = self_launch(args, sys.argv[1:])            # N > 1 without torchrun: the ranks run as a fresh child process; nothing here has touched the GPU
    if rc is not None:
        sys.exit(rc)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:                      # (checked before the first GPU call)
        sys.exit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={env_world}: launch with --nproc-per-node {args.gpus}, or without torchrun")
    cfg = CONFIGS[args.config]
    if cfg.get("densify"):
        args.densify_stats = True
    if args.gaussians is None:
        args.gaussians = cfg["gaussians"]
    if args.height is None:
        args.height = cfg["height"]
    if args.width is None:
        args.width = cfg["width"]

    from emd_amd import dp, scenes, _lib
    from emd_amd import RasterCall, RasterOptions
    from emd_amd.model import StreetGaussians, render, l1_loss, unit_gradient
    from emd_amd.motion import DeviceStep

    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("EMD_BENCH_SHARE_GPU"):        # functional test of the N > 1 path on a 1-GPU box (with EMD_DP_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)              # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local)
    rank, world, local = dp.init_from_env()
    ranks_seen = dp.world_size()              # what the process group reports (RCCL / gloo saw this many ranks)

    N, H, W = args.gaussians, args.height, args.width
    num_frames, num_actors = 50, cfg["actors"]
    focal = cfg["focal"] * (W / cfg["width"])
    # cameras of the rig: configs 3 / 4 name theirs (4 / 6: at --gpus 1 the one rank walks the rig's cameras, the per-rank workload of that
    # configuration); configs 0-2: 1 / 2 / 4 cameras on 1 / 2 / 4 GPUs (rank <-> camera of one timestamp), 6 on 8 GPUs
    num_cams = cfg.get("rig") or dp.rig_size(world)
    scene = scenes.make_static_scene(N, seed=0)
    if num_actors:
        scene = scenes.add_actors(scene, num_actors=num_actors, pts_per_actor=5000, num_frames=num_frames, seed=1)
    model = StreetGaussians(scene, dev, track_heads=bool(num_actors) and not args.no_track_heads)
    params = [p for p in model.parameters()]
    residual = None
    if cfg.get("residual"):
        # the learned per-Gaussian deformation residual as the fused transform consumes it (deformable.py:49-68): activations of a network, not
        # parameters -- a leaf each that collects its gradient (what the network's backward would start from), not part of the exchange
        tg = torch.Generator().manual_seed(11)
        rdx = (0.02 * torch.randn(N, 3, generator=tg)).to(dev).requires_grad_(True)
        rdq = (0.02 * torch.randn(N, 4, generator=tg)).to(dev).requires_grad_(True) if num_actors else None
        residual = (rdx, rdq)
    res_leaves = [t for t in (residual or ()) if t is not None]
    bg = torch.zeros(3)
    g3 = torch.Generator().manual_seed(3)
    target = torch.rand(3, H, W, generator=g3).to(dev)
    unit = unit_gradient(dev)                # loss.backward(unit): the root gradient 1.0 as a resident tensor (no ones_like fill per step)

    factored = world > 1 or args.factored_sh or args.exchange_only
    # options of THIS run's rasterizer calls (an instance, handed to every call: nothing process-wide is written)
    opts = RasterOptions(compute_normal=not args.no_normal, factored_sh_grad=factored, no_sync=not args.sync_count,
                         aux_stream=bool(os.environ.get("EMD_BENCH_AUX")))          # (A/B knob: K1's colour half beside the binning stage)
    cams, campos_dev = {}, {}
    stats = None
    if args.densify_stats:
        stats = [torch.zeros(N, 1, device=dev), torch.zeros(N, 1, device=dev), torch.zeros(N, device=dev)]
    S = {"params": params, "stats": stats, "N": N}          # what a density-control event replaces (config 4)

    # the form of the multi-GPU exchange (set once the visible counts of the run's views are known, below): dense until then
    X = {"cap": None, "cap_by_row": None, "compact": False, "overflow": torch.zeros(1, dtype=torch.int32, device=dev)}

    def xkw(step=None):
        """Options of one step's GradientExchange: the row capacity of the compacted form is per STEP (the visible count moves between 0.6 and 1.5 M over
        the clip; rows are padded to the capacity, so a clip-wide maximum would give the saving away)."""
        cap = X["cap"]
        if step is not None and X["cap_by_row"] is not None:
            per = args.warmup + args.steps
            cap = X["cap_by_row"][step if step < per else args.warmup + (step - args.warmup) % max(args.steps, 1)]
        # slab_pure: this step's loss reaches the four small per-Gaussian tensors through the rasterizer alone (L1 on the image, no regulariser),
        # so the slab of a REPLAYED step may travel as rows of the visible Gaussians too (dp.GradientExchange's precondition)
        return dict(compact=X["compact"], compact_capacity=cap, overflow=X["overflow"], slab_pure=True, timing=X.get("timing", False))

    def cam_for(step):
        # views are ordered timestamp-major and dealt out by dp.view_for: every rank renders a DISTINCT (frame, camera);
        # with 8 ranks on the 6-camera rig two ranks hold cameras of the next timestamp (dp.frame_and_camera).
        # Static configurations (0, 1) render one frame: "single static frame".
        f, c = dp.frame_and_camera(step, rank, world, num_frames, num_cams)
        if not num_actors:
            f = 0
        if (f, c) not in cams:
            cams[(f, c)] = scenes.rig_camera(f, c, H, W, fx=focal, fy=focal)
            campos_dev[(f, c)] = cams[(f, c)].camera_center.to(dev)      # once per camera: no per-step host-to-device copy
        return f, c, cams[(f, c)]

    def one_step(step, options=opts, record=None, backward=True):
        f, c, cam = cam_for(step)
        params, stats = S["params"], S["stats"]
        for p in params + res_leaves:
            p.grad = None
        rec = record if record is not None else RasterCall()
        xchg = None
        if options.factored_sh_grad and backward:
            # SH gradient (81 % of the gradient bytes): rank-one factors, 12 B per Gaussian and rank instead of all-reducing
            # 192 B per Gaussian; the collectives are issued from inside backward(), right behind K8 (emd_amd/dp.py)
            xchg = dp.GradientExchange(campos_dev[(f, c)], actor_ids=model.actor_id if model.has_actors else None,
                                       residual_dx=None if residual is None else residual[0].detach(), **xkw(step))
            rec.on_backward = xchg.start
            rec.on_sh_factor = xchg.start_factors        # (the factor gathers run under K8; needs the actor poses of this step: set below)
        out = render(model, cam, bg, frame=f, iteration=step, options=options, record=rec, residual=residual)
        if not backward:
            return out
        if xchg is not None:
            xchg.actor_pose = None if out["actor_pose"] is None else out["actor_pose"].detach()   # (no reference into the autograd graph)
        loss = l1_loss(out["render"], target)
        loss.backward(unit)
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
    total_steps = args.warmup + args.steps * (1 + max(args.repeats, 0))
    for s_ in range(args.warmup + args.steps):          # the cameras are dataset state: built (and their centres uploaded) before timing
        cam_for(s_)
    sync_opts = opts.replace(no_sync=False)
    out = one_step(0, sync_opts, backward=not cfg["forward_only"])
    st0 = out["raster_call"].last_status()
    dmax, vmax = st0["num_rendered"], st0["num_visible"]
    v_by_row = {}
    every = 1 if (world > 1 or args.exchange == "compact") else 7          # (the compacted exchange wants every view's visible count)
    with torch.no_grad():
        for s_ in sorted(set(list(range(0, args.warmup + args.steps, every)) + [args.warmup + args.steps - 1])):
            f, c, cam = cam_for(s_)
            o = render(model, cam, bg, frame=f, options=sync_opts, residual=residual)
            st_ = o["raster_call"].last_status()
            dmax, vmax = max(dmax, st_["num_rendered"]), max(vmax, st_["num_visible"])
            v_by_row[s_] = st_["num_visible"]
    opts.capacity_hint = int(dmax * 1.3) + 1024          # (an option of this run's calls: nothing process-wide is written)
    # rows per view of the visibility-compacted exchange: the same number on every rank (MAX over the ranks' views + margin); dp.compact_pays
    # then decides per world size whether the rows or the dense slab move fewer bytes per link
    if world > 1 or (args.exchange == "compact" and factored):
        v_all = torch.tensor([vmax], device=dev, dtype=torch.int64)
        if world > 1:
            torch.distributed.all_reduce(v_all, op=torch.distributed.ReduceOp.MAX)
        X["cap"] = dp.visible_capacity(int(v_all))
        # per step: the largest visible count among the ranks' views of THAT step (a table gathered once, here; every rank uses the same numbers)
        rows_n = args.warmup + args.steps
        v_tab = torch.tensor([v_by_row.get(r_, vmax) for r_ in range(rows_n)], device=dev, dtype=torch.int64)
        if world > 1:
            torch.distributed.all_reduce(v_tab, op=torch.distributed.ReduceOp.MAX)
        X["cap_by_row"] = [dp.visible_capacity(int(v_)) for v_ in v_tab.tolist()]
        X["compact"] = {"auto": None, "dense": False, "compact": True}[args.exchange]
        mean_cap = sum(X["cap_by_row"][args.warmup:]) / max(len(X["cap_by_row"][args.warmup:]), 1)
        if X["compact"] is None:          # one decision for the run, by the timed steps' mean capacity (dp.compact_pays: bytes per xGMI link)
            X["compact"] = bool(dp.compact_pays(world, N, mean_cap))
        X["uses_rows"] = bool(X["compact"])
        X["mean_cap"] = mean_cap

    if cfg["forward_only"]:
        return bench_forward_only(args, cfg, scene, model, render, cam_for, bg, opts, _lib, N, H, W, rank)
    if args.exchange_only:
        return bench_exchange_only(args, dp, model, params, one_step, cam_for, campos_dev, N, rank, world, dev, xkw, X)

    # ---- the step as a hipGraph: the launches of a step are captured once and replayed; everything that changes from
    # step to step (camera block, frame index, frame time, coarse-to-fine level of the step) lives at fixed device addresses that
    # ONE launch (emd_select_step_inputs) fills from device-resident tables for the row `sel` names.  The host then spends
    # ~20 us per step instead of ~1-2 ms of Python + launch calls, i.e. the run is GPU-bound whatever the host is doing.
    # (--eager, or a failed capture, issues the same step from Python.)
    period = args.warmup + args.steps             # the repeats replay the timed views: row = warmup + (step - warmup) mod steps
    settle = {"replays": 0}
    out = o = None          # no autograd graph of an eager step may be alive at capture time (its AccumulateGrad nodes are bound to the eager stream)
    import gc
    import types
    gc.collect()            # ... including graphs held only by reference cycles (RasterCall <-> GradientExchange)
    status_log = torch.zeros(max(period, 1), 4, dtype=torch.int32, device=dev)
    views = [cam_for(s_) for s_ in range(period)]
    blocks = torch.stack([torch.cat([bg.reshape(-1).float(), c_.world_view_transform.reshape(-1), c_.full_proj_transform.reshape(-1),
                                     c_.camera_center.reshape(-1)]) for _, _, c_ in views]).to(dev).contiguous()            # [rows, 38]
    frame_of = torch.tensor([f_ for f_, _, _ in views], dtype=torch.int32, device=dev)
    # successor of every row: the launch advances `sel` itself (the repeats replay the timed views: last timed row -> first)
    next_row = torch.tensor([r_ + 1 if r_ + 1 < period else args.warmup for r_ in range(period)], dtype=torch.int64, device=dev)
    cam0 = views[0][2]
    th = model.track_heads
    k_sched = (th.min_embeddings, th.max_embeddings, th.c2f_temporal_iter) if th is not None else None

    def settle_replays(replay):
        """replay the step (untimed) until the clocks have settled: the GPU has idled through set-up, capture and host-side checks (measured: with
        --steps 20 --warmup 5 the first timed block was 2.4 % slower than its back-to-back repeats)"""
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for _ in range(8):
                replay()
                settle["replays"] += 1
            torch.cuda.synchronize()

    def record_step(light=False):
        """Capture the step for the CURRENT parameter tensors (S) as one hipGraph (or two, cut inside backward()); returns the state a replay
        needs, or None when the capture failed (the step is then issued from Python).  Called again after a density-control event: the point
        count, every parameter tensor and the workspaces behind them have changed.  `light` (the re-capture behind an event, which a training
        loop pays every 100 iterations): ONE eager step on the capture stream instead of three (it forms the per-stream workspaces for the new
        point count), no self-check against an eager step and no settle phase -- the caller settles, outside of what it reports as re-record time."""
        G = types.SimpleNamespace(graph=None, graph_b=None, gstate={}, two_graphs=False)
        params = S["params"]
        G.sel = sel = torch.zeros(1, dtype=torch.int64, device=dev)
        G.prev_sel = prev_sel = torch.full((1,), -1, dtype=torch.int64, device=dev)
        blk = torch.zeros(38, device=dev)
        frame_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        t_dev = torch.zeros(1, device=dev)
        kf_dev = torch.ones(1, dtype=torch.int32, device=dev)
        G.status_static = status_static = torch.zeros(4, dtype=torch.int32, device=dev)          # the captured call's status words (graph-static address)
        gstate = G.gstate

        def graph_body(cut=None):
            for p in params + res_leaves:
                p.grad = None
            # (the row index doubles as the training step of the coarse-to-fine schedule, as the eager step passes it)
            select_step_inputs(sel, blocks, blk, frame_of, frame_dev, t_dev, num_frames, k_sched, kf_dev if th is not None else None,
                               status_static, status_log, prev_sel, next_row)
            cam_g = types.SimpleNamespace(image_height=H, image_width=W, tanfovx=cam0.tanfovx, tanfovy=cam0.tanfovy,
                                          world_view_transform=blk[3:19].view(4, 4), full_proj_transform=blk[19:35].view(4, 4),
                                          camera_center=blk[35:38])
            rec_g = RasterCall()
            rec_g.status_buffer = status_static          # the call's status words at a fixed address: the next step's select launch logs them
            rec_g.on_sh_factor = cut                     # (two-graph capture: called between the halves of the rasterizer's backward)
            o = render(model, cam_g, blk[0:3], frame=frame_dev, iteration=DeviceStep(k_fine=kf_dev, t=t_dev), options=opts, record=rec_g,
                       residual=residual)
            l1_loss(o["render"], target).backward(unit)
            if S["stats"] is not None:
                dp.add_densification_stats(o["viewspace_points"].grad, o["radii"], *S["stats"])
            # what the gradient exchange of a multi-GPU step reads after the replay: graph-static tensors
            gstate["rec"], gstate["campos"] = rec_g, blk[35:38]
            gstate["pose"] = None if o["actor_pose"] is None else o["actor_pose"].detach()
        try:
            # ONE capture stream for every recording of this run (a re-capture behind a density-control event reuses it: what the step keeps per
            # stream -- the rasterizer's kept-clean backward workspace, the loss's granule table -- is then formed once)
            side = settle.get("side")
            if side is None:
                side = settle["side"] = torch.cuda.Stream(priority=-1 if os.environ.get("EMD_BENCH_HIPRIO") else 0)      # (A/B knob: the captured chain above a forked branch)
            side.wait_stream(torch.cuda.current_stream())
            t_dbg = [time.perf_counter()]
            with torch.cuda.stream(side):
                for i_ in range(1 if light else 3):
                    sel.fill_(i_)
                    graph_body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            t_dbg.append(time.perf_counter())
            prev_sel.fill_(-1)
            graph = torch.cuda.CUDAGraph()
            graph_b = None
            if opts.factored_sh_grad and not args.one_graph:
                # ---- TWO graphs, cut between the halves of the rasterizer's backward: [forward, loss, render backward K7, SH factor] |
                # [projection backward K8, actor chain backward].  RCCL collectives cannot be captured, so a one-graph step can only start its
                # exchange behind the replay; with the cut the factor gathers are issued between the two replays and travel under K8.
                # The cut is made from RasterCall.on_sh_factor, i.e. from autograd's device thread: capture A is ended and capture B begun
                # there, capture B ended by this thread -- stream capture in "relaxed" mode (the only mode that allows begin / end on
                # different threads; it also keeps the process group's watchdog thread from invalidating the capture).
                graph_b = torch.cuda.CUDAGraph()
                cut_state = {"n": 0}

                def cut_capture(_rec):
                    graph.capture_end()
                    graph_b.capture_begin(pool=graph.pool(), capture_error_mode="relaxed")
                    cut_state["n"] += 1
                gc.collect()
                torch.cuda.synchronize()
                with torch.cuda.stream(side):
                    graph.capture_begin(capture_error_mode="relaxed")
                    try:
                        graph_body(cut=cut_capture)
                    except BaseException:
                        try:
                            (graph_b if cut_state["n"] else graph).capture_end()      # (leave capture mode; the body's error is the one to report)
                        except Exception:
                            pass
                        raise
                    (graph_b if cut_state["n"] else graph).capture_end()
                if cut_state["n"] != 1:
                    raise RuntimeError(f"two-graph capture: the backward was cut {cut_state['n']} times")
                gstate["rec"].on_sh_factor = None
            else:
                # (multi-rank: the process group's watchdog thread polls events while we capture; "thread_local" keeps its calls from
                #  invalidating the capture -- nothing of the exchange is captured)
                with torch.cuda.graph(graph, stream=side, capture_error_mode="global" if world == 1 else "thread_local"):
                    graph_body()
            torch.cuda.synchronize()
            G.graph, G.graph_b, G.two_graphs = graph, graph_b, graph_b is not None

            def replay_compute():
                graph.replay()
                if graph_b is not None:
                    graph_b.replay()
            G.replay_compute = replay_compute
            t_dbg.append(time.perf_counter())
            if os.environ.get("EMD_BENCH_DEBUG_RECORD"):
                print(f"[bench] record_step(light={light}): eager warm-up {1e3 * (t_dbg[1] - t_dbg[0]):.1f} ms, capture {1e3 * (t_dbg[2] - t_dbg[1]):.1f} ms", file=sys.stderr)
            if light:
                sel.fill_(0)
                prev_sel.fill_(-1)
                return G
            # ---- self-check of the captured graph before it is trusted with the timed region: two replays of row 0 must reproduce the
            # eager step's device status words (D, V) and leave finite, identical parameter gradients (a memset node captured on ROCm 7.2
            # replayed with a corrupt fill pattern from the SECOND replay on: that is how the library's zero fills became kernels)
            keep_stats = None if S["stats"] is None else [t.clone() for t in S["stats"]]
            # the `.grad` tensors the capture left on the leaves live in the graph's pool and are what every replay writes: the eager step below
            # replaces them with its own, so they are put back afterwards -- the comparison then reads the REPLAY's gradients, and the exchange
            # of a multi-GPU step finds the leaves' `.grad` inside the captured slab again (dp.slab_holds_leaf_grads: ONE all-reduce of the slab
            # instead of four of stale tensors; found in round 6 through `exchange.slab_ms` coming back empty)
            leaves_ = list(params) + list(res_leaves)
            captured_grads = [p.grad for p in leaves_]
            eager = one_step(0)
            torch.cuda.synchronize()
            want_status = eager["raster_call"].status.clone()
            want_grad = model._xyz.grad.clone()
            del eager
            for p_, g_cap in zip(leaves_, captured_grads):
                p_.grad = g_cap
            gc.collect()
            for rep_ in range(2):
                sel.fill_(0)
                prev_sel.fill_(-1)
                replay_compute()
                torch.cuda.synchronize()
                if not torch.equal(status_static[:3], want_status[:3]):
                    raise RuntimeError(f"replay {rep_}: status words {status_static.tolist()} differ from the eager step's {want_status.tolist()}")
                g_ = model._xyz.grad
                if not bool(torch.isfinite(g_).all()):
                    raise RuntimeError(f"replay {rep_}: non-finite parameter gradients")
                if world == 1 and float((g_ - want_grad).abs().max()) > 1e-4 * float(want_grad.abs().max()) + 1e-12:      # (N > 1: the eager step averaged over ranks)
                    raise RuntimeError(f"replay {rep_}: parameter gradients differ from the eager step's")
            # ---- settle (untimed replays, rows walk on from row 1), so that the warm-up steps and the timed region run at the rate a training run sees
            settle_replays(replay_compute)
            if keep_stats is not None:                        # the checks and the settle phase are not training steps: their statistics do not count
                for t_, k_ in zip(S["stats"], keep_stats):
                    t_.copy_(k_)
            sel.fill_(0)                                      # the first replay renders row 0; every replay leaves the next row in `sel`
            prev_sel.fill_(-1)
            torch.cuda.synchronize()
            return G
        except Exception as e:          # capture is an optimisation of the host side only: fall back to issuing the step from Python
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            return None

    G = record_step() if (not args.eager and opts.no_sync) else None

    def row_of(step):
        return step if step < period else args.warmup + (step - args.warmup) % args.steps

    if G is None and args.settle_ms > 0:                  # the eager step settles the same way
        keep_stats = None if S["stats"] is None else [t.clone() for t in S["stats"]]
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            one_step(0)
            settle["replays"] += 1
            torch.cuda.synchronize()
        if keep_stats is not None:
            for t_, k_ in zip(S["stats"], keep_stats):
                t_.copy_(k_)

    def timed_step(step):
        if G is not None:
            G.graph.replay()                                    # (`sel` was advanced to row_of(step) by the replay before)
            if opts.factored_sh_grad:
                # RCCL collectives are not captured.  Two graphs: the factor gathers are issued between the replays and run under K8, the
                # slab all-reduce behind the second; one graph (--one-graph): the whole exchange behind the replay
                gs = G.gstate
                xchg = dp.GradientExchange(gs["campos"], actor_ids=model.actor_id if model.has_actors else None, actor_pose=gs["pose"],
                                           residual_dx=None if residual is None else residual[0].detach(), **xkw(step))
                if G.graph_b is not None:
                    xchg.start_factors(gs["rec"])
                    G.graph_b.replay()
                xchg.start(gs["rec"])
                xchg.finish(model._features, model._xyz, model.active_sh_degree, other_params=S["params"])
                if X.get("timing"):
                    X["timed"].append(xchg)
            elif world > 1:
                dp.allreduce_gradients(S["params"])
        else:
            o = one_step(row_of(step))
            status_log[row_of(step)].copy_(o["raster_call"].status)

    def timed_block(first, count):
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(count):
            timed_step(first + s)
        t_enq = time.perf_counter() - t0      # host time to enqueue the steps (the GPU runs behind it)
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0      # this rank's own time (before the closing barrier): min / max over ranks are reported
        if world > 1:
            torch.distributed.barrier()
        return time.perf_counter() - t0, t_enq, t_own

    def flush_status_log():
        if G is not None:       # the status row of the last replay
            G.sel.fill_(-1)
            select_step_inputs(G.sel, blocks, torch.zeros(38, device=dev), status=G.status_static, status_log=status_log, prev_sel=G.prev_sel)
            torch.cuda.synchronize()

    for s in range(args.warmup):
        timed_step(s)
    if G is None:
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        _lib.profile_read()
    # N > 1 (or the forced one-rank exchange): HIP events around the phases of every timed step's exchange, read after the timed region
    X["timing"], X["timed"] = bool(G is not None and opts.factored_sh_grad and (world > 1 or dp.force_exchange())), []
    dt, t_enqueue, dt_own = timed_block(args.warmup, args.steps)
    exchange_times = None
    if X["timed"]:
        rows_t = [x_.times_ms() for x_ in X["timed"]]
        mean = lambda k: (lambda v: None if not v else round(sum(v) / len(v), 4))([r_[k] for r_ in rows_t if r_[k] is not None])
        exchange_times = {k: mean(k) for k in ("exposed_ms", "slab_ms", "gather_ms", "rebuild_ms")}
        exchange_times["steps"] = len(rows_t)
        exchange_times["note"] = ("HIP events on the stream the exchange is issued from, means over the timed steps of rank 0: exposed = end of this rank's backward "
                                  "kernels -> end of the exchange (what the step pays for communicating); slab / gather = issue -> completion of the slab all-reduce "
                                  "(behind K8) and of the factor gathers (issued between K7 and K8: they travel under K8); rebuild = the local dL/dshs rebuild")
    X["timing"], X["timed"] = False, []
    prof_timed = None
    if G is None:
        prof_timed = _lib.profile_read()
        _lib.profile_enable(False)
    # ---- the same block again, `repeats` times back to back: spread of the measurement (`value` stays the first block)
    rep = []
    for r in range(max(args.repeats, 0)):
        d_r, _, _ = timed_block(args.warmup + args.steps * (1 + r), args.steps)
        rep.append(d_r / args.steps * 1e3)
    flush_status_log()
    st_all = status_log[args.warmup:args.warmup + args.steps].cpu().numpy().astype("int64") & 0xFFFFFFFF          # (D, flags, V, .) of every timed step
    stage_region = "the timed region"
    keep_stats = None if S["stats"] is None else [t.clone() for t in S["stats"]]       # (the diagnostic steps below are not training steps)
    if G is not None:
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
        prof = _lib.profile_read()
        _lib.profile_enable(False)
    else:
        prof_steps, prof = args.steps, prof_timed
    # ---- pair statistics of the render backward (diagnostic instantiation of K7, a few of the timed views): evaluated vs contributing
    pairs = None
    if world == 1:
        ps = torch.zeros(4, dtype=torch.int64, device=dev)
        n_ps = min(args.steps, 5)
        for s in range(n_ps):
            rec_p = RasterCall()
            rec_p.pair_stats = ps
            one_step(args.warmup + s * max(args.steps // n_ps, 1), record=rec_p)
        ev, hit, rows_k7, atoms_k7 = [int(x) for x in ps.cpu().tolist()]
        pairs = {"evaluated_per_launch": ev / n_ps, "contributing_per_launch": hit / n_ps, "useful_fraction": round(hit / max(ev, 1), 4),
                 "launches": n_ps, "note": "(pixel, list entry) pairs the lanes of k_render_backward_q evaluate / pairs with alpha >= 1/255 in front "
                                           "of the pixel's last contributor; counting instantiation of the kernel (EmdBwdArgs.pair_stats), outside the timed region"}
    if keep_stats is not None:
        for t_, k_ in zip(S["stats"], keep_stats):
            t_.copy_(k_)
    step_issue_graph, two_graphs = G is not None, (G is not None and G.two_graphs)

    def release_graphs(G_):
        if G_ is not None:          # release the captured graph and its memory pool explicitly, in a quiet state
            torch.cuda.synchronize()
            if G_.graph_b is not None:
                G_.graph_b.reset()
            G_.graph.reset()
            G_.gstate.clear()
    release_graphs(G)

    # ---- config 4: density-control events behind the timed steps, timed on their own; the loop goes on with the new point count.  TWO events: the
    # first pays what a process pays once (the first allocations of the grown tensors, the first launch of every kernel of the event), the second
    # is what a training loop pays every `densification_interval` = 100 iterations (S3Gaussian/arguments/gaussian_options.py:112-117) -- that one
    # is `event_ms` / `re_record_ms`.  Between them the timed views run again and accumulate the statistics the second event decides on.
    density_event = None
    if cfg.get("densify"):
        from emd_amd.model import density_control
        events = []
        for ev_i in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dp.reduce_densification_stats(*S["stats"])          # every rank then holds the statistics of all views of all ranks (SUM, SUM, MAX)
            thr = args.densify_grad_threshold
            if thr is None:
                # (the synthetic scene's gradients have nothing of a real scene's scale: the reference's 2e-4 would select nothing or everything.
                #  The threshold is the 95 % quantile of the mean view-space gradient over the Gaussians seen at least once -- computed from the REDUCED
                #  statistics, hence identical on every rank.  A real loop has a constant here: the quantile is timed separately.)
                seen = S["stats"][1].reshape(-1) > 0
                avg = (S["stats"][0].reshape(-1) / S["stats"][1].reshape(-1).clamp_min(1.0))[seen]
                thr = float(torch.quantile(avg[:: max(avg.numel() // 1_000_000, 1)], 0.95)) if avg.numel() else 2e-4
            torch.cuda.synchronize()
            t_thr = time.perf_counter() - t0
            N_before = S["N"]
            ev = density_control(model, *S["stats"], max_grad=thr, min_opacity=0.005, extent=27.5, percent_dense=0.01, seed=0, event=ev_i)
            N2 = ev["n_after"]
            S["params"] = [p for p in model.parameters()]
            S["stats"] = [torch.zeros(N2, 1, device=dev), torch.zeros(N2, 1, device=dev), torch.zeros(N2, device=dev)]
            S["N"] = N2
            torch.cuda.synchronize()
            t_event = time.perf_counter() - t0 - t_thr
            if world > 1:          # replicas must agree on the new point count before the next collective
                n_all = torch.tensor([N2], device=dev, dtype=torch.int64)
                n_max, n_min = n_all.clone(), n_all.clone()
                torch.distributed.all_reduce(n_max, op=torch.distributed.ReduceOp.MAX)
                torch.distributed.all_reduce(n_min, op=torch.distributed.ReduceOp.MIN)
                assert int(n_max) == int(n_min) == N2, f"ranks disagree on the point count after density control: {int(n_min)}..{int(n_max)}"
            opts.capacity_hint = int(opts.capacity_hint * (1.0 + 1.5 * max(N2 - N_before, 0) / N_before)) + 1024
            # the visible counts grow with the point count: so do the row capacities of the compacted exchange (host numbers, like the binning capacity;
            # an overflow would be reported by the assert on X["overflow"] below, never accepted silently)
            # (the new points are clones / samples of Gaussians that WERE seen: every one of them may be visible in every view)
            more = int(1.1 * max(N2 - N_before, 0)) + 1
            if X["cap"] is not None:
                X["cap"] = min(dp.visible_capacity(X["cap"] + more, margin=1.0), dp.visible_capacity(N2, margin=1.0))
            if X["cap_by_row"] is not None:
                X["cap_by_row"] = [min(dp.visible_capacity(c_ + more, margin=1.0), dp.visible_capacity(N2, margin=1.0)) for c_ in X["cap_by_row"]]
            gc.collect()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            G = record_step(light=True) if (not args.eager and opts.no_sync) else None
            torch.cuda.synchronize()
            t_record = time.perf_counter() - t0
            if G is not None:
                keep_stats = [t.clone() for t in S["stats"]]
                settle_replays(G.replay_compute)                 # (untimed, and not part of the re-record time)
                for t_, k_ in zip(S["stats"], keep_stats):
                    t_.copy_(k_)
                G.sel.fill_(0)
                G.prev_sel.fill_(-1)
                torch.cuda.synchronize()
            for s in range(args.warmup):
                timed_step(s)
            dt2, _, _ = timed_block(args.warmup, args.steps)
            flush_status_log()
            st2 = status_log[args.warmup:args.warmup + args.steps].cpu().numpy().astype("int64") & 0xFFFFFFFF
            release_graphs(G)
            if world > 1:
                t2 = torch.tensor([dt2, t_event, t_record], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t2, op=torch.distributed.ReduceOp.MAX)
                dt2, t_event, t_record = [float(v) for v in t2.tolist()]
            events.append(dict(ev, grad_threshold=thr, event_ms=round(t_event * 1e3, 3), threshold_quantile_ms=round(t_thr * 1e3, 3),
                               re_record_ms=round(t_record * 1e3, 2), ms_per_step_after=round(dt2 / args.steps * 1e3, 4),
                               iters_per_s_after=round(world * args.steps / dt2, 2), overflow_after=int((st2[:, 1] & 1).sum()),
                               D_mean_after=round(float(st2[:, 0].mean()), 1)))
        density_event = dict(events[1], first_event=events[0],
                             note="two events behind the timed steps (statistics reduced over ranks -> clone / split / prune of the background Gaussians -> fresh "
                                  "statistics), each followed by ONE re-capture of the step for the new point count (one eager step + the capture; no self-check, "
                                  "the settle replays are not counted) and the same views timed again; the top-level fields are the SECOND event -- what a loop "
                                  "pays every 100 iterations --, `first_event` carries the process's one-off costs (first allocations, first kernel loads); "
                                  "`value` is the block BEFORE the events")
    own = [dt_own / args.steps * 1e3]
    if world > 1:
        tmax = torch.tensor([dt] + [r_ * args.steps / 1e3 for r_ in rep], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        vals = tmax.tolist()
        dt, rep = float(vals[0]), [v / args.steps * 1e3 for v in vals[1:]]
        t_all = torch.zeros(world, device=dev, dtype=torch.float64)
        t_all[rank] = dt_own / args.steps * 1e3
        torch.distributed.all_reduce(t_all, op=torch.distributed.ReduceOp.SUM)
        own = [float(v) for v in t_all.tolist()]
    overflow = int((st_all[:, 1] & 1).sum())
    assert overflow == 0, "binning workspace overflowed during the timed region"
    assert int(X["overflow"]) == 0, f"the compacted exchange's capacity ({X['cap']} rows per view) was exceeded: gradients of a step were incomplete"
    assert int((st_all[:, 1] & 2).sum()) == 0, ("a timed step saw a visible Gaussian beyond 65 536 x the near plane with the three-pass depth sort: "
                                                 "its image was blank (use RasterOptions(wide_depth_sort=True))")

    if rank == 0:
        # V and D of EVERY timed step (device status words of rank 0's views), not of one frame
        Ds, Vs = st_all[:, 0], st_all[:, 2]
        D, V = float(Ds.mean()), float(Vs.mean())
        # the roofline's algorithmic bytes use D and V of the SAME steps the stage durations were measured over (the eager repetitions of the
        # first `prof_steps` timed steps behind a graph run; every timed step otherwise), so achieved / frac are self-consistent for any --steps
        Dp, Vp = float(Ds[:prof_steps].mean()), float(Vs[:prof_steps].mean())
        T = ((W + 15) // 16) * ((H + 15) // 16)
        C = 7 if opts.compute_normal else 4
        passes = (max(T - 1, 1).bit_length() + 7) // 8            # radix passes over the D duplicates (tile bits)
        ab = algorithmic_bytes(N, Vp, Dp, H * W, T, C, passes, residual=residual is not None, C_bwd=4)   # the L1 loss sends no gradient into the normal image
        stages = {}
        for name, (ms, cnt) in prof.items():
            if cnt and name in ab:
                avg = ms / prof_steps                                # per iteration (a stage may open twice per step)
                stages[name] = {"ms": round(avg, 4), "alg_GB": round(ab[name] / 1e9, 4),
                                "GBps": round(ab[name] / 1e9 / (avg * 1e-3), 1)}
        dom = max(stages, key=lambda k: stages[k]["ms"])
        kernel_ms = sum(v["ms"] for v in stages.values())
        total_alg = sum(ab.values())
        full = (N, H, W, args.config) == (2_000_000, 1066, 1600, 2)
        traffic = pmc_traffic(dom, passes) if full else None
        if full:                                             # fabric bytes per stage from the same committed counter file (static)
            for name in stages:
                tb = pmc_traffic(name, passes)
                if tb:
                    stages[name]["fabric_GB_static"] = round(tb / 1e9, 4)
                    stages[name]["fabric_over_algorithmic"] = round(tb / ab[name], 2)
        if pairs is not None:
            pairs["atomic_rows_per_launch"] = rows_k7 / pairs["launches"]
            pairs["float_atomics_per_launch"] = atoms_k7 / pairs["launches"]
            pairs["atomic_row_bytes_per_launch"] = 48.0 * rows_k7 / pairs["launches"]
            pairs["compulsory_row_bytes_per_launch"] = 48.0 * Vp
        roofline = {"bound": "hbm", "kernel": dom, "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(stages[dom]["GBps"] / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_ratio": None if not traffic else round(traffic / ab[dom], 3),
                    "traffic_static": {"static": True, "source": None if _pmc_path(PMC_TRAFFIC_CSV) is None else "profiles/" + os.path.basename(_pmc_path(PMC_TRAFFIC_CSV)),
                                       "note": "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of `bench.py --steps 20 --warmup 5 --eager` (the driver's flags: the same frames "
                                               "as the timed region of a default driver run), bytes per launch; a committed file constant, not measured in this run"},
                    "secondary_bound": "fp32 vector issue rate: the render kernels are issue-bound (DESIGN.md section 6)",
                    "issue": issue_bound(dom, stages[dom]["ms"]) if full else None,
                    "pairs": pairs,
                    "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": stages[dom]["ms"],
                    "stage_durations_measured_over": stage_region,
                    "per_step": {"steps": int(len(Ds)), "D_min": int(Ds.min()), "D_mean": round(D, 1), "D_max": int(Ds.max()),
                                 "V_min": int(Vs.min()), "V_mean": round(V, 1), "V_max": int(Vs.max()),
                                 "D_mean_profiled_steps": round(Dp, 1), "V_mean_profiled_steps": round(Vp, 1),
                                 "note": "list entries D (the (tile, Gaussian) pairs inside the Gaussians' alpha >= 1/255 boxes) and visible Gaussians V of every timed step, read from "
                                         "the device status words after the timed region; the algorithmic bytes use the means over the steps the stage durations were measured on"},
                    "whole_iter": {"algorithmic_GB": round(total_alg / 1e9, 3), "kernel_ms": round(kernel_ms, 3),
                                   "GBps": round(total_alg / 1e9 / (kernel_ms * 1e-3), 1),
                                   "frac": round(total_alg / 1e9 / (kernel_ms * 1e-3) / HBM_PEAK_GBS, 4)},
                    "stages": stages}
        if world == 1:
            mapping = ("1 GPU: camera 0 of frame (step mod 50)" if num_actors else "1 GPU: camera 0 of frame 0 (static scene)") if num_cams == 1 else \
                      f"1 GPU walking the {num_cams}-camera rig: step s renders camera (s mod {num_cams}) of frame (s div {num_cams}) -- the per-rank workload of this configuration"
        elif world == num_cams:
            mapping = f"{world} GPUs: rank r <-> camera r of the {num_cams}-camera rig, all ranks of a step share one timestamp"
        else:
            mapping = (f"{world} GPUs on the {num_cams}-camera rig: view (step * {world} + rank) of the timestamp-major view list, i.e. "
                       f"{world - num_cams} rank(s) per step render cameras of the NEXT timestamp; every rank renders a distinct view")
        heads_on = model.track_heads is not None
        rep_sorted = sorted(rep)
        res = {
            "metric": "train iters/s (fwd+bwd) @1066x1600, 2M Gaussians; 1/2/4/8 GPU",
            "value": world * args.steps / dt, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "repeats_ms_per_step": None if not rep else {"n": len(rep), "min": round(rep_sorted[0], 4), "median": round(rep_sorted[len(rep) // 2], 4),
                                                        "max": round(rep_sorted[-1], 4),
                                                        "note": "the timed block replayed again back to back after it; `value` is the first block only"},
            "settle": {"ms": args.settle_ms, "untimed_steps": settle["replays"], "note": "untimed steps before the warm-up steps: an idle GPU's clocks settle under load"},
            "host_enqueue_ms_per_step": round(t_enqueue / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"] + (f" ({num_actors} actors x 5000" + (", learned per-actor track offsets" if heads_on else "") + ")" if num_actors else "")
                                   + ", SH degree 3, one view per GPU per step, L1 loss, fwd+bwd to all 59 floats/Gaussian"
                                   + (" + actor poses" if num_actors else "") + (" + track heads" if heads_on else ""),
                       "baseline_config_index": args.config,
                       "gaussians": N, "height": H, "width": W, "visible_V": round(V, 1), "duplicates_D": round(D, 1), "tiles_T": T,
                       "radix_passes_depth": "3 x 9 bits above the near plane: first over the N keys (compacting to V), two over the V pairs",
                       "radix_passes_tile_on_D": passes, "blended_channels_C": C, "views_per_step": world,
                       "rig_cameras": num_cams, "rank_view_mapping": mapping, "track_heads": heads_on,
                       "ranks_seen": ranks_seen, "rank_ms_per_step": {"min": round(min(own), 4), "max": round(max(own), 4), "per_rank": [round(v, 4) for v in own],
                                                                      "note": "each rank's own time for the timed block (before the closing barrier)"},
                       "deformation_residual": None if residual is None else "residual_dx [N,3]" + (" + residual_dq [N,4]" if residual[1] is not None else "")
                                               + " as inputs of the fused transform, gradients returned to both (not exchanged: activations of a network)",
                       "densification_stats_in_step": bool(args.densify_stats),
                       "step_issue": ("hipGraph replay (one capture; camera block, frame, frame time and coarse-to-fine level selected on the device by one launch"
                                      + (("; TWO graphs cut between the render backward and the projection backward: the SH-factor gathers are issued "
                                          "between their replays and travel under K8, the slab all-reduce behind the second)") if two_graphs else
                                         ("; the gradient exchange is issued behind each replay)" if (world > 1 or opts.factored_sh_grad) else ")")))
                                     if step_issue_graph else "eager (Python issues every launch)",
                       "parallelism": f"view-parallel dp{world}", "count_readback": bool(args.sync_count),
                       "gradient_exchange": ("none (1 GPU)" if world == 1 else
                                             ("visibility-compacted rows (dp.compact_pays: fewer bytes per xGMI link at this world size): all-gather of (index, 3 floats) "
                                              f"factor rows and (index, 11 floats) slab rows of each view's visible Gaussians, {int(X.get('mean_cap') or 0)} rows per view on average (a capacity per step), added in rank "
                                              "order on every rank; + camera centres, per-view actor pose tables; dense SH average rebuilt locally; actor poses / track "
                                              "heads as one bucket") if X.get("uses_rows") else
                                             "SH gradient as rank-one factors: all-gather of 12 B per Gaussian and rank + camera centres + per-view actor "
                                             "pose tables, dense average rebuilt locally; one RCCL all-reduce (AVG) of the remaining 44 B per Gaussian "
                                             "(one slab, started inside backward()) + actor poses / track heads"),
                       "exchange_rows_per_view": int(X.get("mean_cap") or 0) if X.get("uses_rows") else None},
            "roofline": roofline,
        }
        if exchange_times is not None:
            res["exchange"] = exchange_times
        if density_event is not None:
            res["density_control_event"] = density_event
        if world == 1 and full and not args.no_fine_stage:
            # the step S3Gaussian runs for 50 000 of its 55 000 iterations (deformation network in front of the rasterizer, full loss, sky): measured
            # AFTER the timed region and reported beside it -- `value` stays BASELINE.json's metric (profiles/fine_stage.py)
            try:
                sys.path.insert(0, os.path.join(ROOT, "profiles"))
                import fine_stage
                gc.collect()
                torch.cuda.empty_cache()
                res["fine_stage"] = fine_stage.measure(dev, N, H, W)
            except Exception as e:          # the headline line must not depend on the extra block
                res["fine_stage"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:      # the CPU leg is timed on rank 0 of the 1-GPU run only
            f0, _, cam0 = cam_for(0)
            res["cpu_baseline"] = cpu_baseline(scene, cam0, frame=f0)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()



def bench_forward_only(args, cfg, scene, model, render, cam_for, bg, opts, _lib, N, H, W, rank):
    """BASELINE configs[0]: the projection + SH forward (K1) next to the reference's pure-PyTorch CPU path at the same N
    (BASELINE.md section 2, row 1).  `value` is the K1 kernel time; the whole forward (binning + compositing) is reported beside it."""
    with torch.no_grad():
        for s in range(args.warmup):
            f, c, cam = cam_for(0)
            render(model, cam, bg, frame=f, options=opts)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        _lib.profile_read()
        t0 = time.perf_counter()
        for s in range(args.steps):
            f, c, cam = cam_for(0)
            o = render(model, cam, bg, frame=f, options=opts)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    st = o["raster_call"].last_status()
    V, D = st["num_visible"], st["num_rendered"]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    C = 7 if opts.compute_normal else 4
    ab = algorithmic_bytes(N, V, D, H * W, T, C, 1)
    k1_ms = prof["preprocess"][0] / args.steps
    fwd_ms = sum(ms for name, (ms, cnt) in prof.items() if cnt) / args.steps
    if rank == 0:
        res = {"metric": "projection + covariance + SH colour forward (K1) ms @256x256, 10k static Gaussians", "value": k1_ms, "unit": "ms",
               "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": False,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": cfg["name"], "baseline_config_index": 0, "gaussians": N, "height": H, "width": W, "visible_V": V,
                          "duplicates_D": D, "tiles_T": T, "whole_forward_kernel_ms": round(fwd_ms, 4),
                          "note": "launch-latency bound at this size: ms_per_step is the host-issued forward (Python + ~25 launches)"},
               "roofline": {"bound": "hbm", "kernel": "preprocess", "achieved": round(ab["preprocess"] / 1e9 / (k1_ms * 1e-3), 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(ab["preprocess"] / 1e9 / (k1_ms * 1e-3) / HBM_PEAK_GBS, 5), "traffic": None,
                            "algorithmic_bytes_per_launch": ab["preprocess"], "avg_launch_ms": round(k1_ms, 5)}}
        if not args.no_cpu_baseline:
            f0, _, cam0 = cam_for(0)
            res["cpu_baseline"] = cpu_baseline(scene, cam0, frame=f0)
        print(json.dumps(res), flush=True)


def bench_exchange_only(args, dp, model, params, one_step, cam_for, campos_dev, N, rank, world, dev, xkw, X):
    """Only the gradient exchange of a view-parallel step: one backward produces the factors / slab, then GradientExchange.start +
    finish are timed `steps` times on those buffers (on 1 GPU with a process group: the collectives run at world size 1)."""
    from emd_amd import RasterCall
    rec = RasterCall()
    out = one_step(args.warmup, record=rec)            # (its own exchange ran once: buffers and communicators are warm)
    f, c, cam = cam_for(args.warmup)
    pose = None if out["actor_pose"] is None else out["actor_pose"].detach()

    def exchange():
        x = dp.GradientExchange(campos_dev[(f, c)], actor_ids=model.actor_id if model.has_actors else None, actor_pose=pose, **xkw(args.warmup))
        x.start(rec)
        x.finish(model._features, model._xyz, model.active_sh_degree, other_params=params)
        return x
    for _ in range(max(args.warmup, 1)):
        exchange()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = exchange()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        A = 0 if pose is None else int(pose.shape[0])
        print(json.dumps({"metric": "gradient exchange ms per step (GradientExchange.start + finish only)", "value": dt / args.steps * 1e3, "unit": "ms",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "exchange only: all-reduce(AVG) of the 44 B/Gaussian slab + all-gather of the 12 B/Gaussian SH factors, camera "
                                                 "centres and pose tables + local rebuild of dL/dshs + small all-reduces", "gaussians": N,
                                     "collectives_per_step": x.num_collectives, "forced_at_world_1": bool(dp.force_exchange()),
                                     "form": "visibility-compacted rows" if x._rows_s is not None else "dense slab all-reduce + factor all-gather",
                                     "rows_per_view": X["cap"] if x._rows_s is not None else None,
                                     "bytes_per_rank": {"slab_allreduce": 44 * N, "factor_allgather_sent": 12 * N,
                                                        "factor_allgather_received": 12 * N * world, "pose_tables": 48 * A * world}}}), flush=True)
    if world > 1 or torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
