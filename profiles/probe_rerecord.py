"""What does the first eager step after a density-control event cost on the host and on the device?"""
import sys, time
sys.path.insert(0, ".")
import torch
from emd_amd import scenes, RasterOptions
from emd_amd.model import StreetGaussians, density_control, render, l1_loss
dev = torch.device("cuda", 0)
N, H, W = 3_000_000, 1066, 1600
scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=48, pts_per_actor=5000, num_frames=50, seed=1)
model = StreetGaussians(scene, dev, track_heads=True)
cam = scenes.rig_camera(3, 0, H, W, fx=1700.0, fy=1700.0)
target = torch.rand(3, H, W).to(dev)
bg = torch.zeros(3)
o = render(model, cam, bg, frame=3, iteration=3, options=RasterOptions(no_sync=False))
opts = RasterOptions(no_sync=True, capacity_hint=int(o["raster_call"].last_status()["num_rendered"] * 2) + 1024)
g = torch.Generator().manual_seed(0)
def step():
    for p in model.parameters():
        p.grad = None
    out = render(model, cam, bg, frame=3, iteration=3, options=opts)
    l1_loss(out["render"], target).backward()
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for i in range(3):
    t0 = T(); step(); t1 = T()
    print(f"warm step {i}: {1e3*(t1-t0):.2f} ms", flush=True)
for ev in range(3):
    n = model._xyz.shape[0]
    acc, den, mr = (torch.rand(n, 1, generator=g) * 1e-3).to(dev), torch.ones(n, 1, device=dev), torch.zeros(n, device=dev)
    t0 = T()
    density_control(model, acc, den, mr, max_grad=9.5e-4, min_opacity=0.005, extent=27.5, percent_dense=0.01, seed=0, event=ev)
    t1 = T()
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    step()
    torch.cuda.synchronize()
    pr.disable()
    t2 = T(); step(); t3 = T()
    print(f"event {ev}: density_control {1e3*(t1-t0):.2f} ms, first step after {1e3*(t2-t1):.2f} ms, second {1e3*(t3-t2):.2f} ms", flush=True)
    if ev == 1:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
