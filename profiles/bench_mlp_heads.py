"""Head forward kernels with and without the folded L1 sum at N rows (python3 profiles/bench_mlp_heads.py [N]); run under rocprofv3 --kernel-trace --stats
for per-kernel times."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emd_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda", 0)
lib = L.load()
g = torch.Generator().manual_seed(0)
h = torch.randn(N, 64, generator=g).to(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for out_dim in (3, 48):
    w1, b1 = (torch.randn(64, 64, generator=g) / 8).to(dev), torch.zeros(64, device=dev)
    wo, bo = (torch.randn(out_dim, 64, generator=g) / 8).to(dev), torch.zeros(out_dim, device=dev)
    out = torch.empty(N, out_dim, device=dev)
    l1 = torch.zeros(1, device=dev)
    for with_l1 in (False, True):
        b = L.EmdMlpBranch()
        b.num_points, b.depth, b.relu_input, b.out_dim, b.h = N, 1, 1, out_dim, h.data_ptr()
        b.w_hidden[0], b.b_hidden[0], b.w_out, b.b_out, b.out = w1.data_ptr(), b1.data_ptr(), wo.data_ptr(), bo.data_ptr(), out.data_ptr()
        if with_l1:
            b.l1_sum = l1.data_ptr()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            L.check(lib.emd_mlp_branch_forward(C.byref(b), st), "fwd")
        ts = []
        for _ in range(15):
            e0.record(); L.check(lib.emd_mlp_branch_forward(C.byref(b), st), "fwd"); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"out_dim {out_dim:2d} l1 {int(with_l1)}: median {ts[7] * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f} us")
