import sys, torch
sys.argv = ["bench_full_step.py", "--fine"]
src = open("profiles/bench_full_step.py").read()
head = src.split("dmax = 0")[0]
g = {"__name__": "bfs"}
exec(compile(head, "bfs", "exec"), g)
step = g["step"]
for s in range(3): step(s)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(5)
    torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::fill_") and e.input_shapes and any(len(sh) and sh[0] in (2000000,) for sh in e.input_shapes if isinstance(sh, list) and sh):
        st = [f for f in (e.stack or []) if "emd_amd" in f or "bench_full" in f or "bfs" in f or "autograd" in f][:3]
        print(e.name, e.input_shapes, "|", " <- ".join(s.split("/")[-1] for s in st))
