#!/usr/bin/env python3
"""ONE table for every kernel of the repository: launches per step, average duration, algorithmic bytes / FLOPs with the
formula, fabric bytes, and the fraction of the ceiling that bounds it.  Run on the GPU box:

    python profiles/make_roofline_table.py > profiles/r06_roofline_table.md

Durations are measured HERE, with torch.profiler (roctracer device times) over eagerly issued steps of three workloads:
  A  the headline step of bench.py (BASELINE configs[2]: 2 M Gaussians, 32 actors, 1066 x 1600, L1, fwd + bwd), frames 5..24
  B  the S3Gaussian fine-stage step with the optimiser step (profiles/fine_stage.py: deformation network, sky, full loss, Adam)
  C  one density-control event at 3 M Gaussians (BASELINE configs[4]) and OmniRe's refinement event on the same store
Algorithmic bytes / FLOPs: profiles/fine_stage.py:kernel_models and bench.py:algorithmic_bytes.  Fabric bytes: the
committed rocprofv3 counter summaries (FETCH_SIZE x 2 + WRITE_SIZE per launch, profiles/r0N_pmc_hbm_traffic.csv for A and
profiles/r0N_pmc_fine_traffic.csv for B; "-" where no counter pass covers the kernel).  Ceilings: HBM 8 000 GB/s; MFMA 416.7 TFLOP/s
fp32-equivalent for the six-product split-bf16 formulation (measured ceiling 252: profiles/r04_mfma_split_bf16_microbench.txt); vector issue
for the two render kernels (bench.py:issue_bound, instruction counts of profiles/r0N_pmc_valu.csv, mix of profiles/r06_render_isa_mix.txt)."""
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import bench  # noqa: E402
import fine_stage  # noqa: E402
from emd_amd import RasterCall, RasterOptions, scenes  # noqa: E402
from emd_amd.model import StreetGaussians, l1_loss, render  # noqa: E402

dev = torch.device("cuda", 0)


def fabric_table(csv_name):
    path = None
    for r in ("r06_", "r05_", "r04_"):
        p = os.path.join(ROOT, "profiles", r + csv_name)
        if os.path.exists(p):
            path = p
            break
    out = {}
    if path:
        for line in open(path):
            if line.startswith("#") or line.startswith("kernel,"):
                continue
            f = line.rstrip("\n").split(",")
            out[f[0]] = (float(f[3]) + float(f[4])) * 1e6          # fetch x 2 + write, bytes per launch
    return out, (os.path.basename(path) if path else None)


def base_name(name):
    n = fine_stage.short_name(name)
    return re.split(r"[<(]", n)[0]


def emit(title, prof, models, fabric, fabric_src, issue=None, stage_rows=None):
    print(f"\n### {title}\n")
    print("| kernel | launches / step | avg µs | ms / step | bound | algorithmic work per launch | formula | achieved | fraction of ceiling | fabric bytes per launch |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    total = 0.0
    for name, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        total += ms
        per = ms / n if n > 0 else ms
        e = fine_stage.price(name, per, models)
        fb = fabric.get(base_name(name))
        if fb is None and base_name(name).startswith("k_radix"):
            fb = None
        work, ach, frac = "-", "-", "-"
        if e.get("bound") == "hbm" and "alg_GB" in e:
            work, ach, frac = f"{e['alg_GB'] * 1e3:.1f} MB", f"{e['GBps']:.0f} GB/s", f"{e['frac']:.3f} of HBM"
        elif e.get("bound") == "mfma" and "useful_GFLOP" in e:
            work, ach = f"{e['useful_GFLOP']:.1f} GFLOP", f"{e['TFLOPs']:.0f} TFLOP/s"
            frac = f"{e['frac']:.3f} of 416.7 ({e['frac_of_measured_ceiling']:.2f} of the measured 252)"
        if issue and base_name(name) in issue:
            frac += f"; issue {issue[base_name(name)]}"
        fb_s = "-" if fb is None else (f"{fb / 1e6:.0f} MB" + (f" ({fb / (e['alg_GB'] * 1e9):.2f} x)" if e.get("alg_GB") else ""))
        print(f"| `{e['kernel']}` | {n:.2f} | {per * 1e3:.1f} | {ms:.4f} | {e.get('bound') or '-'} | {work} | {e.get('formula', '') or e['what']} | {ach} | {frac} | {fb_s} |")
    print(f"\nKernel time per step: **{total:.3f} ms** (eager issue under the profiler; the replayed step is shorter by the launch gaps). Fabric source: `{fabric_src}`.")
    if stage_rows:
        print("\nStages whose kernels share one formula (SURVEY §8d):\n")
        print("| stage | kernels | ms / step | algorithmic MB | GB/s | fraction of HBM |")
        print("|---|---|---|---|---|---|")
        for r in stage_rows:
            print("| " + " | ".join(str(x) for x in r) + " |")


# ---- A: the headline step ---------------------------------------------------------------------------------------------------------------------
N, H, W, F = 2_000_000, 1066, 1600, 50
scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=32, pts_per_actor=5000, num_frames=F, seed=1)
model = StreetGaussians(scene, dev, track_heads=True)
params = list(model.parameters())
target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
bg = torch.zeros(3)
cams = {f: scenes.rig_camera(f, 0, H, W, fx=1700.0, fy=1700.0) for f in range(F)}
o = render(model, cams[0], bg, frame=0, iteration=0, options=RasterOptions(no_sync=False))
dmax = o["raster_call"].last_status()["num_rendered"]
opts = RasterOptions(no_sync=True, capacity_hint=int(dmax * 1.5) + 1024)
Vs, Ds = [], []


def step_a(s):
    for p in params:
        p.grad = None
    out = render(model, cams[s % F], bg, frame=s % F, iteration=s, options=opts)
    l1_loss(out["render"], target).backward()


for s in range(5):
    step_a(s)
torch.cuda.synchronize()
prof_a = fine_stage.profile_kernels(step_a, 20, first=5)
with torch.no_grad():
    for f in range(5, 25):
        st = render(model, cams[f], bg, frame=f, iteration=f, options=RasterOptions(no_sync=False))["raster_call"].last_status()
        Vs.append(st["num_visible"])
        Ds.append(st["num_rendered"])
V, D = sum(Vs) / len(Vs), sum(Ds) / len(Ds)
T = ((W + 15) // 16) * ((H + 15) // 16)
ab = bench.algorithmic_bytes(N, V, D, H * W, T, 7, 2, C_bwd=4)
models_a = [(r"k_preprocess<", "projection + SH colour (K1)", "hbm", ab["preprocess"], "SURVEY 8d F1: 68 N + V (216 + 4 C)"),
            (r"k_preprocess_backward", "projection backward (K8)", "hbm", ab["preprocess_backward"], "SURVEY 8d B2: V (24 + 4 C + 48 + 192) + N (236 + 12)"),
            (r"k_render_forward_q", "tile compositing forward (K6)", "hbm", ab["render_forward"], "SURVEY 8d F6: D (28 + 4 C) + HW (4 (C + 1) + 8)"),
            (r"k_render_backward_q", "tile compositing backward (K7)", "hbm", ab["render_backward"], "SURVEY 8d B1: D (28 + 4 C) + HW (4 (C + 1) + 8) + V (24 + 4 C)"),
            (r"k_tile_ranges", "tile ranges", "hbm", ab["tile_ranges"], "4 D + 8 T"),
            (r"k_l1_loss", "L1 loss + gradient", "hbm", 3 * H * W * 12, "image + target read, gradient written"),
            (r"k_duplicate", "tile duplication", "hbm", 16 * V + 8 * D, "rect + count per visible Gaussian read, (tile, id) pairs written"),
            (r"k_sorted_counts", "depth-ordered tile counts + scan", "hbm", 28 * N + 8 * V - 16 * V + 16 * V, "28 N + 8 V (with k_duplicate: SURVEY 8d F2 + F3)"),
            (r"k_radix_scatter", "radix scatter (one pass)", "hbm", None, "see the stage row: 16 B per element and pass"),
            (r"k_radix_hist", "radix histogram (one pass)", "hbm", None, "see the stage row"),
            (r"k_radix_scan_bins", "radix bin scan", "hbm", None, "launch-bound (4 µs)"),
            (r"k_tracked_pose|k_track", "per-actor track heads + pose table", "hbm", None, "launch-bound: 32 actors"),
            (r"k_tile_order", "longest-first tile order", "hbm", None, "launch-bound: 6 700 tiles"),
            (r"k_select_step", "per-step inputs", "hbm", None, "launch-bound")]
fab_a, src_a = fabric_table("pmc_hbm_traffic.csv")
k6 = sum(ms for k, (ms, n) in prof_a.items() if "k_render_forward_q" in k)
k7 = sum(ms for k, (ms, n) in prof_a.items() if "k_render_backward_q" in k)
ib6, ib7 = bench.issue_bound("render_forward", k6), bench.issue_bound("render_backward", k7)
issue = {}
if ib6:
    issue["k_render_forward_q"] = f"{ib6['issue_utilisation_all_plain']:.2f} all-plain / {ib6['issue_utilisation_with_mix']:.2f} with the loop's mix"
if ib7:
    issue["k_render_backward_q"] = f"{ib7['issue_utilisation_all_plain']:.2f} all-plain / {ib7['issue_utilisation_with_mix']:.2f} with the loop's mix"
sort_ms = sum(ms for k, (ms, n) in prof_a.items() if "k_radix" in k)
dup_ms = sum(ms for k, (ms, n) in prof_a.items() if "k_sorted_counts" in k or "k_duplicate" in k)
stage_rows = [("radix_sort", "k_radix_hist / scatter / scan_bins (11 launches)", f"{sort_ms:.4f}", f"{ab['radix_sort'] / 1e6:.1f}", f"{ab['radix_sort'] / 1e9 / (sort_ms * 1e-3):.0f}",
               f"{ab['radix_sort'] / 1e9 / (sort_ms * 1e-3) / 8000:.3f}"),
              ("scan_duplicate", "k_sorted_counts + k_duplicate", f"{dup_ms:.4f}", f"{ab['scan_duplicate'] / 1e6:.1f}", f"{ab['scan_duplicate'] / 1e9 / (dup_ms * 1e-3):.0f}",
               f"{ab['scan_duplicate'] / 1e9 / (dup_ms * 1e-3) / 8000:.3f}")]
print("# Roofline table of every kernel (round 6)\n")
print(f"Generated by `profiles/make_roofline_table.py` on {torch.cuda.get_device_name(0)}; durations: torch.profiler device times of eagerly issued steps.")
print(f"Workload A: V = {V:.0f} visible Gaussians, D = {D:.0f} list entries (means over frames 5..24), T = {T} tiles.")
emit("A. Headline step (bench.py, BASELINE configs[2]: 2 M Gaussians, 32 actors, 1066 × 1600, L1, forward + backward)", prof_a, models_a, fab_a, src_a, issue, stage_rows)
del model, params, scene
torch.cuda.empty_cache()

# ---- B: the fine-stage step with the optimiser ---------------------------------------------------------------------------------------------------
S = fine_stage.build(dev, fine=True, adam="hip")
for s in range(4):
    S.step(s)
torch.cuda.synchronize()
prof_b = fine_stage.profile_kernels(S.step, 10, first=5)
with torch.no_grad():
    st = render(S.model, scenes.rig_camera(5, 0, H, W), bg, frame=5, deformation=S.deform, embeddings=S.embeddings, iteration=12000, time=5 / 49,
                options=RasterOptions(no_sync=False))["raster_call"].last_status()
n_param = sum(p.numel() for p in S.params)
models_b = fine_stage.kernel_models(N, st["num_visible"], st["num_rendered"], H * W)
adam_launches = max(sum(n for k, (ms, n) in prof_b.items() if "k_adam" in k), 1.0)
models_b = [m if not m[0].startswith("k_adam") else (m[0], m[1], m[2], n_param * 28 / adam_launches,
                                                     f"{n_param / 1e6:.1f} M parameter elements x (16 B read + 12 B written) / {adam_launches:.0f} launches of <= 32 tensors")
            for m in models_b]
fab_b, src_b = fabric_table("pmc_fine_traffic.csv")
emit("B. S3Gaussian fine-stage step + Adam (profiles/fine_stage.py: HexPlane, MLP heads, rasterizer, sky, full loss, optimiser)", prof_b, models_b, fab_b, src_b)
del S
torch.cuda.empty_cache()

# ---- C: density control -------------------------------------------------------------------------------------------------------------------------
from emd_amd.model import density_control  # noqa: E402
N3 = 3_000_000
scene = scenes.add_actors(scenes.make_static_scene(N3, seed=0), num_actors=48, pts_per_actor=5000, num_frames=F, seed=1)
model = StreetGaussians(scene, dev, track_heads=True)
g = torch.Generator().manual_seed(0)


def event(i):
    n = model._xyz.shape[0]
    acc, den, mr = torch.rand(n, 1, device=dev) * 1e-3, torch.ones(n, 1, device=dev), torch.zeros(n, device=dev)          # (statistics of the event: set-up, three fills)
    density_control(model, acc, den, mr, max_grad=9.5e-4, min_opacity=0.005, extent=27.5, percent_dense=0.01, seed=0, event=i)


event(0)
torch.cuda.synchronize()
prof_c = fine_stage.profile_kernels(event, 2, first=1)
n3 = model._xyz.shape[0]
models_c = [(r"k_densify_gather", "gather of every parameter / moment / statistic row", "hbm", n3 * (59 + 3 + 1) * 4 * 2, "rows x (59 parameter + 3 statistic + 1 table floats) read and written"),
            (r"k_densify_decide|k_refine_decide", "per-point decision + block counts", "hbm", n3 * (12 + 8 + 4), "scales + two statistics read, code written"),
            (r"k_densify_index|k_refine_index", "output row -> (source, kind)", "hbm", n3 * (4 + 8), "code read, src + kind written"),
            (r"k_densify_scan", "block-count scan", "hbm", None, "launch-bound: 12 k blocks")]
emit("C. One density-control event at 3 M Gaussians (densify + prune of `emd_amd.model.density_control`; per EVENT, not per step; the torch fills / `rand` / reductions are the table's own synthetic statistics and the actor-count checks)",
     prof_c, models_c, {}, None)
