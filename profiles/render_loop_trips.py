"""K6's instructions split by loop: trip counts of the compositing kernel's loops on the headline scene (counting instantiation,
EmdFwdArgs.loop_stats) x the static instruction counts of each loop's body (profiles/r06_render_isa_mix.txt, LOOP lines).  Run on the GPU:
    python profiles/render_loop_trips.py > profiles/r06_render_loop_trips.txt
The product is checked against the counter total of the same kernel (SQ_INSTS_VALU of profiles/r0N_pmc_valu.csv)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emd_amd import RasterCall, RasterOptions, scenes  # noqa: E402
from emd_amd.model import StreetGaussians, render  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
N, H, W, F = 2_000_000, 1066, 1600, 50
scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=32, pts_per_actor=5000, num_frames=F, seed=1)
model = StreetGaussians(scene, dev, track_heads=True)
frames = list(range(5, 25))          # the timed frames of the driver's run (--warmup 5 --steps 20)
st = torch.zeros(6, dtype=torch.int64, device=dev)
D = V = 0
with torch.no_grad():
    for f in frames:
        rec = RasterCall()
        rec.loop_stats = st
        o = render(model, scenes.rig_camera(f, 0, H, W, fx=1700.0, fy=1700.0), torch.zeros(3), frame=f, iteration=f, options=RasterOptions(no_sync=False), record=rec)
        s_ = rec.last_status()
        D += s_["num_rendered"]
        V += s_["num_visible"]
torch.cuda.synchronize()
scan, cull, drain, useful, rounds, waves = [x / len(frames) for x in st.cpu().tolist()]
loops = {}
for line in open(os.path.join(ROOT, "profiles", "r06_render_isa_mix.txt")):
    if line.startswith("LOOP k_render_forward_q"):
        f_ = line.split()
        loops[f_[2]] = {k: float(v) for k, v in (kv.split("=") for kv in f_[3:])}
trips = {"round": rounds, "scan": scan, "cull": cull, "drain": drain / 2.0, "straight": waves}          # (the drain loop's body holds two iterations)
print(f"# k_render_forward_q<true, 0> on frames {frames[0]}..{frames[-1]} of the headline clip (2 M Gaussians, 1066 x 1600), per launch; D = {D / len(frames):.0f} list entries, V = {V / len(frames):.0f}")
print(f"waves that ran                    {waves:14.0f}")
print(f"scan -> cull -> drain rounds      {rounds:14.0f}   ({rounds / waves:.2f} per wave)")
print(f"scan steps (64 list words each)   {scan:14.0f}   ({scan * 64 / (D / len(frames)):.2f} x the list: each of a tile's four quadrant waves scans the tile's list until its pixels saturate)")
print(f"cull steps (64 queued entries)    {cull:14.0f}")
print(f"drain iterations (1 entry x 4 rows) {drain:12.0f}   entries the rows held {useful:.0f} = {useful / max(4 * drain, 1):.3f} of the 4 x 16-lane row slots")
print(f"(pixel, entry) pairs evaluated    {16 * useful:14.0f}   ({64 * drain:.0f} lane-iterations issued)")
print()
print(f"{'loop':10s} {'trips':>14s} {'valu/trip':>10s} {'valu':>14s} {'share':>7s} {'lds':>12s} {'salu':>12s}")
tot = sum(trips[k] * loops[k]["valu"] for k in loops)
for k in ("straight", "round", "scan", "cull", "drain"):
    v = trips[k] * loops[k]["valu"]
    print(f"{k:10s} {trips[k]:14.0f} {loops[k]['valu']:10.0f} {v:14.3e} {v / tot:7.3f} {trips[k] * loops[k]['lds']:12.3e} {trips[k] * loops[k]['salu']:12.3e}")
print(f"{'total':10s} {'':14s} {'':10s} {tot:14.3e}")
path = bench._pmc_path(bench.PMC_VALU_CSV)
for line in open(path):
    if line.startswith("k_render_forward_q,"):
        n = float(line.split(",")[2])
        print(f"counter SQ_INSTS_VALU per launch ({os.path.basename(path)}): {n:.3e}  -> trips x static = {tot / n:.3f} of it")
comp = trips["drain"] * loops["drain"]["valu"]
print(f"compositing (drain loop) share of the vector instructions: {comp / tot:.3f}; scan + cull + queue + the rest: {1 - comp / tot:.3f}")
print(f"vector-issue time of the compositing alone: {comp * 1.32e-9 / 1024 * 1e3:.4f} ms per launch (1.32 ns per plain instruction and SIMD, 1024 SIMDs)")
