"""Which Python lines issue the small torch kernels (add / fill / copy ...) of the fine-stage step?  (torch.profiler, with_stack)"""
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
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=8)
rows = []
for e in ka:
    if not e.key.startswith("aten::"):
        continue
    dev_us = float(getattr(e, "self_device_time_total", 0.0) or 0.0)
    if dev_us <= 0:
        continue
    st = [f for f in (e.stack or []) if ("/repo/" in f or "emd_amd" in f or "profiles/" in f)]
    rows.append((dev_us, e.count, e.key, str(e.input_shapes)[:70], " <- ".join(x.split("/")[-1][:60] for x in st[:3])))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten ops with device time: {tot:.1f} us over {sum(r[1] for r in rows)} calls")
for r in rows[:70]:
    print(f"{r[0]:8.1f} us x{r[1]:3d}  {r[2]:22s} {r[3]:72s} {r[4]}")
