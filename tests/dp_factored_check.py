"""Run under torch.distributed.run with 2 or 3 ranks (EMD_BENCH_SHARE_GPU=1 EMD_DP_BACKEND=gloo on a 1-GPU box): each rank renders
its own view of the same dynamic scene; the SH gradient obtained by exchanging the rank-one factors (dp.GradientExchange) must
equal the all-reduced dense gradient.  EMD_DP_MIXED=1: the ranks hold DIFFERENT timestamps (rank r renders frame 2 + r), the case of
six cameras on eight GPUs -- the exchange then also gathers one actor pose table per view.  Prints OK <max abs diff> from rank 0."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emd_amd import dp, scenes, RasterCall, RasterOptions  # noqa: E402
from emd_amd.model import StreetGaussians, render, l1_loss  # noqa: E402

local = 0 if os.environ.get("EMD_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))   # one rank per GPU over RCCL
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
rank, world, _ = dp.init_from_env()
mixed = bool(os.environ.get("EMD_DP_MIXED"))
compact = bool(os.environ.get("EMD_DP_COMPACT"))          # the visibility-compacted form of the exchange (index + value rows, rank-order adds)
N, H, W = 40000, 96, 128
scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=4, pts_per_actor=2000, num_frames=6, seed=1)
model = StreetGaussians(scene, dev, track_heads=True)
params = list(model.parameters())
frame = 2 + rank if mixed else 3
cam = scenes.rig_camera(frame, rank % len(scenes.RIG_YAWS), H, W)
target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
bg = torch.zeros(3)


cap, ovf = None, torch.zeros(1, dtype=torch.int32, device=dev)


def step(factored, capacity=None):
    for p in params:
        p.grad = None
    rec = RasterCall()
    xchg = None
    if factored:
        xchg = dp.GradientExchange(cam.camera_center, actor_ids=model.actor_id, compact=True if capacity else None, compact_capacity=capacity,
                                   overflow=ovf)
        rec.on_backward = xchg.start                       # collectives start inside backward()
    out = render(model, cam, bg, frame=frame, iteration=100, options=RasterOptions(factored_sh_grad=factored), record=rec)
    if xchg is not None:
        xchg.actor_pose = out["actor_pose"]
    l1_loss(out["render"], target).backward()
    return out, xchg


o0, _ = step(False)
if compact:          # rows per view: the largest visible count over the ranks' views + margin (every rank uses the same number)
    v = torch.tensor([o0["raster_call"].last_status()["num_visible"]], device=dev, dtype=torch.int64)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    cap = dp.visible_capacity(int(v), multiple=64)
    assert cap < N, (cap, N)
del o0
ref = {}
for name, p in model.named_parameters():                              # reference: plain dense all-reduce of every gradient
    g = p.grad.clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
    ref[name] = g / world
out, xchg = step(True, cap)
assert model._features.grad is None
xchg.finish(model._features, model._xyz, model.active_sh_degree, other_params=params)
# gathers: factors + camera centres + pose tables (3); the four small per-Gaussian gradients travel as ONE slab (1); the rest
# (two actor-pose tables, temporal tables, 8 head tensors, point embeddings) one small all-reduce each
n_rest = sum(1 for n_, p in model.named_parameters() if n_ not in ("_xyz", "_scaling", "_rotation", "_opacity", "_features"))
assert xchg.num_collectives == 3 + 1 + 1, (xchg.num_collectives, n_rest)      # gathers, the slab, ONE bucket for the n_rest small tensors
for name, p in model.named_parameters():
    want, got = ref[name], p.grad
    tol = 2e-5 * want.abs().max().item() + 1e-12
    assert (got - want).abs().max().item() <= tol, (name, (got - want).abs().max().item(), tol)
diff = (model._features.grad - ref["_features"]).abs().max().item()
if compact:
    assert xchg._rows_f is not None and xchg._rows_s is not None             # both streams travelled as rows
    assert int(ovf) == 0
    # replicas: the rank-order adds leave the same bits on every rank
    for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features"):
        g = dict(model.named_parameters())[name].grad.contiguous()
        parts = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(parts, g)
        assert all(torch.equal(parts[0], q) for q in parts[1:]), name
    # a capacity below the visible count is reported, not silently accepted
    out, xchg = step(True, 64)
    xchg.finish(model._features, model._xyz, model.active_sh_degree, other_params=params)
    assert int(ovf) == 1
if rank == 0:
    print(f"OK {diff:.3e} of {ref['_features'].abs().max().item():.3e} world {world} mixed {int(mixed)} compact {int(compact)} capacity {cap}")
dist.barrier()
dist.destroy_process_group()
