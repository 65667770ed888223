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
N, H, W = 40000, 96, 128
scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=4, pts_per_actor=2000, num_frames=6, seed=1)
model = StreetGaussians(scene, dev, track_heads=True)
params = list(model.parameters())
frame = 2 + rank if mixed else 3
cam = scenes.rig_camera(frame, rank % len(scenes.RIG_YAWS), H, W)
target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
bg = torch.zeros(3)


def step(factored):
    for p in params:
        p.grad = None
    rec = RasterCall()
    xchg = None
    if factored:
        xchg = dp.GradientExchange(cam.camera_center, actor_ids=model.actor_id)
        rec.on_backward = xchg.start                       # collectives start inside backward()
    out = render(model, cam, bg, frame=frame, iteration=100, options=RasterOptions(factored_sh_grad=factored), record=rec)
    if xchg is not None:
        xchg.actor_pose = out["actor_pose"]
    l1_loss(out["render"], target).backward()
    return out, xchg


step(False)
ref = {}
for name, p in model.named_parameters():                              # reference: plain dense all-reduce of every gradient
    g = p.grad.clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
    ref[name] = g / world
out, xchg = step(True)
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
if rank == 0:
    print(f"OK {diff:.3e} of {ref['_features'].abs().max().item():.3e} world {world} mixed {int(mixed)}")
dist.barrier()
dist.destroy_process_group()
