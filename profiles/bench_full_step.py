"""An S3Gaussian-style training step assembled from every HIP piece of this repository at the headline size (2 M Gaussians,
32 actors, 1066 x 1600) -- the command line over profiles/fine_stage.py (the builder `bench.py` reports its `fine_stage` block from).
(The bench.py metric is the L1-only step of BASELINE.json; this is the same step with the reference's full loss and sky model.)  One JSON line.
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
measured SLOWER in round 4 (12.38 against 12.30 ms: the regularised instantiation of the narrow heads' forward took 172 us against 130-140) and FASTER in
round 6 (11.04 / 11.01 against 11.15 / 11.08 ms, alternated in one call: that instantiation runs two waves per SIMD since round 5): bench.py's fine_stage block uses it.
--adam / --torch-adam: also take the optimiser step (train.py:428) with emd_amd.optim.Adam / torch.optim.Adam over the groups of
gaussian_model.py:188-199 (per-group learning rates, eps 1e-15)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fine_stage  # noqa: E402

A = sys.argv
FINE = "--fine" in A
FEAT_SEP = "--feat-separate" in A
FEAT = FINE and ("--feat" in A or FEAT_SEP)
GRAPH = "--graph" in A
adam = "hip" if "--adam" in A else ("torch" if "--torch-adam" in A else None)
dev = torch.device("cuda", 0)
S = fine_stage.build(dev, fine=FINE, feat=FEAT, feat_separate=FEAT_SEP, fused_l1="--fused-l1" in A, adam=adam, capturable=GRAPH)
step, F = S.step, S.F
for s in range(10):
    step(s)
torch.cuda.synchronize()
K_STEPS = 50
graphs = None
if GRAPH:
    sg = fine_stage.record_graphs(S, range(F))
    graphs = [sg.graphs[f] for f in range(F)]
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
if adam:
    op += " -> optimiser step (" + ("emd_amd.optim.Adam" if adam == "hip" else "torch.optim.Adam") + ")"
print(json.dumps({"op": op,
                  "gaussians": S.N, "height": S.H, "width": S.W, "steps": K_STEPS, "ms_per_step": round(dt / K_STEPS * 1e3, 4),
                  "iters_per_s": round(K_STEPS / dt, 1), "host_enqueue_ms_per_step": round(t_host / K_STEPS * 1e3, 4),
                  "host_bound": bool(t_host > 0.9 * dt),
                  "step_issue": "hipGraph replay (one graph per frame of the clip, one memory pool)" if graphs is not None else "eager (Python)"}))
