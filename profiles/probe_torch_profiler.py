"""Does torch.profiler give per-kernel device times on this box?  (fine-stage step, eager, 3 steps)"""
import sys, time, json
sys.argv = [sys.argv[0], "--fine"]
sys.path.insert(0, ".")
import torch
t0 = time.time()
import runpy
# build the fine step from bench_full_step's module-level code (runs 10 warm + 50 timed steps; fine for a probe)
ns = runpy.run_path("profiles/bench_full_step.py", run_name="probe")
step = ns["step"]
torch.cuda.synchronize()
print("setup s", time.time() - t0, flush=True)
from torch.profiler import profile, ProfilerActivity
t0 = time.time()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for s in range(5):
        step(20 + s)
    torch.cuda.synchronize()
print("profiled s", time.time() - t0, flush=True)
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print("device events", len(ev))
agg = {}
for e in ev:
    a = agg.setdefault(e.name, [0.0, 0])
    a[0] += e.device_time if hasattr(e, "device_time") else e.cuda_time
    a[1] += 1
for name, (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{us/5:10.1f} us/step x{n/5:5.1f}  {name[:110]}")
print("total ms/step", sum(v[0] for v in agg.values()) / 5 / 1e3)
