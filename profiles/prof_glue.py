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
