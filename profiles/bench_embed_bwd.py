"""Times emd_mlp_trunk_backward for a level without HexPlane features (ka = 0, kb = 4) at N rows: python3 profiles/bench_embed_bwd.py [N]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emd_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda", 0)
lib = L.load()
g = torch.Generator().manual_seed(0)
xb = torch.randn(N, 4, generator=g).to(dev)
w = torch.randn(64, 36, generator=g).to(dev)
b = torch.zeros(64, device=dev)
gh = torch.randn(N, 64, generator=g).to(dev)
dxb, dw, db = torch.empty_like(xb), torch.zeros_like(w), torch.zeros_like(b)
t = L.EmdMlpTrunk()
t.num_points, t.ka, t.kb, t.ld_w, t.col_a, t.col_b = N, 0, 4, 36, 0, 32
t.xb, t.w, t.b = xb.data_ptr(), w.data_ptr(), b.data_ptr()
tg = L.EmdMlpTrunkGrads()
tg.num_gh = 1
tg.g_h[0] = gh.data_ptr()
tg.d_xb, tg.d_w, tg.d_b = dxb.data_ptr(), dw.data_ptr(), db.data_ptr()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    L.check(lib.emd_mlp_trunk_backward(C.byref(t), C.byref(tg), st), "trunk_backward")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(30):
    e0.record(); L.check(lib.emd_mlp_trunk_backward(C.byref(t), C.byref(tg), st), "trunk_backward"); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print(json.dumps({"op": "trunk backward, ka 0, kb 4", "N": N, "median_us": round(ts[15] * 1e3, 1), "min_us": round(ts[0] * 1e3, 1), "GBps_of_g_h": round(N * 256 / ts[15] / 1e6, 1)}))
