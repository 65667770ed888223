import sys, runpy, torch
sys.argv = ["bench_full_step.py", "--fine"]
# run the bench module up to its definitions by executing it, then profile 3 extra steps
src = open("profiles/bench_full_step.py").read()
head = src.split("dmax = 0")[0]
g = {"__name__": "bfs"}
exec(compile(head, "bfs", "exec"), g)
step = g["step"]
for s in range(5): step(s)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for s in range(3): step(5 + s)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=60))
# every aten op of the three profiled steps that launched something, by number of calls (the launch-bound glue: each is a 5-8 us launch inside the replayed graph)
rows = [(e.key, e.count, e.self_device_time_total if hasattr(e, "self_device_time_total") else e.self_cuda_time_total) for e in prof.key_averages()]
print("\naten ops with device time, per step:")
for name, cnt, t in sorted(rows, key=lambda x: -x[1]):
    if name.startswith("aten::") and t > 0:
        print(f"  {name:40s} calls/step {cnt / 3:6.1f}   device us/step {t / 3:8.1f}")
print("\nthe same by input shape (add / fill_ / copy_ / mul / add_ / sum):")
for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count):
    t = e.self_device_time_total if hasattr(e, "self_device_time_total") else e.self_cuda_time_total
    if e.key in ("aten::add", "aten::fill_", "aten::copy_", "aten::mul", "aten::add_", "aten::sum") and t > 0:
        print(f"  {e.key:14s} calls/step {e.count / 3:5.1f}  us/call {t / e.count:7.1f}  shapes {e.input_shapes}")
