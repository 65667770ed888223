"""Which Python lines issue the small torch kernels (add / fill / copy) of the fine-stage step?"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "profiles")
import torch
from torch.profiler import profile, ProfilerActivity
import fine_stage
S = fine_stage.build(torch.device("cuda", 0))
for s in range(4):
    S.step(s)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    S.step(5)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.name.split("::")[1] in (
            "add", "add_", "fill_", "zero_", "copy_", "mul", "mul_", "zeros", "zeros_like", "ones_like", "sum", "clone", "contiguous", "cat", "sub", "div", "neg", "abs", "mean", "expand", "to", "_to_copy"):
        if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
            continue          # only top-level aten ops
        dev_us = sum(float(getattr(k, "device_time", 0.0)) for k in e.kernels) if hasattr(e, "kernels") else 0.0
        st = [f for f in (e.stack or []) if "/repo/" in f or "emd_amd" in f or "profiles" in f][:3]
        rows.append((e.name, str(e.input_shapes)[:60], dev_us, " <- ".join(s_.split("/")[-1] for s_ in st)))
from collections import Counter
c = Counter()
t = Counter()
for name, shp, us, st in rows:
    c[(name, shp, st)] += 1
    t[(name, shp, st)] += us
for k, n in sorted(c.items(), key=lambda kv: -t[kv[0]])[:60]:
    print(f"{n:3d} x {t[k]:8.1f} us  {k[0]:14s} {k[1]:62s} {k[2]}")
