"""Run under torch.distributed.run with 2 ranks (EMD_BENCH_SHARE_GPU=1 EMD_DP_BACKEND=gloo on a 1-GPU box): each rank renders its own
view of the same dynamic scene; the SH gradient obtained by exchanging the rank-one factors (dp.exchange_sh_gradient) must equal
the all-reduced dense gradient.  Prints OK <max abs diff> from rank 0."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emd_amd import dp, scenes, RasterConfig  # noqa: E402
from emd_amd.model import StreetGaussians, render, l1_loss  # noqa: E402

local = 0 if os.environ.get("EMD_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))   # one rank per GPU over RCCL
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
rank, world, _ = dp.init_from_env()
N, H, W = 40000, 96, 128
scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=4, pts_per_actor=2000, num_frames=6, seed=1)
model = StreetGaussians(scene, dev)
cam = scenes.rig_camera(3, rank % len(scenes.RIG_YAWS), H, W)
target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
bg = torch.zeros(3)


def step(factored):
    RasterConfig.factored_sh_grad = factored
    for p in model.parameters():
        p.grad = None
    out = render(model, cam, bg, frame=3)
    l1_loss(out["render"], target).backward()
    return out


step(False)
dense = model._features.grad.clone()
dist.all_reduce(dense, op=dist.ReduceOp.SUM)
dense /= world
small = {}
for name in ("_xyz", "_scaling", "_rotation", "_opacity"):          # reference for the other gradients: plain dense all-reduce
    gsm = getattr(model, name).grad.clone()
    dist.all_reduce(gsm, op=dist.ReduceOp.SUM)
    small[name] = gsm / world
out = step(True)
assert model._features.grad is None
dp.exchange_sh_gradient(model._features, model._xyz, cam.camera_center, model.active_sh_degree, actor_ids=model.actor_id,
                        actor_pose=out["actor_pose"], also_allreduce=list(model.parameters()))
# the four small per-Gaussian gradients travelled as ONE slab (+ the two actor-pose tables): three collectives, not six
assert dp.exchange_sh_gradient.last_num_allreduce == 3, dp.exchange_sh_gradient.last_num_allreduce
for name, want in small.items():
    got = getattr(model, name).grad
    assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item() + 1e-12, name
diff = (model._features.grad - dense).abs().max().item()
ref = dense.abs().max().item()
assert diff <= 2e-5 * ref + 1e-12, (diff, ref)
if rank == 0:
    print(f"OK {diff:.3e} of {ref:.3e}")
dist.barrier()
dist.destroy_process_group()
