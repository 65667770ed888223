"""Where do the 105 ms of a density-control event at 3 M go?"""
import sys, time
sys.path.insert(0, ".")
import torch
from emd_amd import scenes, dp
from emd_amd.model import StreetGaussians, density_control
dev = torch.device("cuda", 0)
N = 3_000_000
scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=48, pts_per_actor=5000, num_frames=50, seed=1)
model = StreetGaussians(scene, dev, track_heads=True)
g = torch.Generator().manual_seed(0)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    Nn = model._xyz.shape[0]
    acc = (torch.rand(Nn, 1, generator=g) * 1e-3).to(dev); den = torch.ones(Nn, 1, device=dev); mr = torch.zeros(Nn, device=dev)
    t0 = T()
    seen = den.reshape(-1) > 0
    avg = (acc.reshape(-1) / den.reshape(-1).clamp_min(1.0))[seen]
    thr = float(torch.quantile(avg[:: max(avg.numel() // 1_000_000, 1)], 0.95))
    t1 = T()
    ev = density_control(model, acc, den, mr, max_grad=thr, min_opacity=0.005, extent=27.5, percent_dense=0.01, seed=0, event=rep)
    t2 = T()
    params = list(model.parameters())
    stats = [torch.zeros(ev["n_after"], 1, device=dev), torch.zeros(ev["n_after"], 1, device=dev), torch.zeros(ev["n_after"], device=dev)]
    t3 = T()
    print(f"rep {rep}: threshold {1e3*(t1-t0):.1f} ms, density_control {1e3*(t2-t1):.1f} ms, new stats {1e3*(t3-t2):.1f} ms  {ev}", flush=True)
# inside density_control: time pieces with the profiler of host time
import cProfile, pstats
Nn = model._xyz.shape[0]
acc = (torch.rand(Nn, 1, generator=g) * 1e-3).to(dev); den = torch.ones(Nn, 1, device=dev); mr = torch.zeros(Nn, device=dev)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
density_control(model, acc, den, mr, max_grad=thr, min_opacity=0.005, extent=27.5, percent_dense=0.01, seed=0, event=5)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
