"""View-parallel data parallelism: one process per GPU, camera views of a step sharded over ranks, Gaussian
gradients averaged with an RCCL all-reduce over xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference trains one view per step on one GPU (S3Gaussian/train.py:203; OmniRe/models/trainers/base.py:411);
at world_size 1 this module is a no-op and the step is the reference step.  Design:
  - parameters, actor tables and MLPs are replicated (2 M x 236 B = 472 MB << 288 GB);
  - step s, rank r renders view `views[(s * W + r) % len(views)]`; no collective on the data path until gradients;
  - gradients are reduced per attribute tensor, in place, asynchronously, largest (SH, 192 N bytes) first, then
    awaited together: no flatten / copy passes over the 472 MB slab;
  - densification statistics are computed per view BEFORE reduction (gaussian_model.py:728-730; train.py:406):
    sum of ||d mean2D|| and of counts (SUM), max radii (MAX).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # EMD_DP_BACKEND=gloo: functional test of the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("EMD_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def view_for(step, rank, world, num_views):
    """Which view (index into the step-ordered view list) rank `rank` renders at `step`."""
    return (step * world + rank) % num_views


def allreduce_gradients(params, average=True):
    """Average `.grad` of every parameter over ranks, in place.  Largest tensors are issued first."""
    if world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    grads.sort(key=lambda g: -g.numel())
    gloo = dist.get_backend() == "gloo"
    op = dist.ReduceOp.SUM if (gloo or not average) else dist.ReduceOp.AVG
    works = [dist.all_reduce(g, op=op, async_op=True) for g in grads]
    for w in works:
        w.wait()
    if average and gloo:
        w_ = float(world_size())
        for g in grads:
            g.div_(w_)


def reduce_densification_stats(grad_norm_accum, denom, max_radii2D):
    """Cross-view reduction of the per-view densification statistics (norms taken per view, before reduction)."""
    if world_size() == 1:
        return
    dist.all_reduce(grad_norm_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)


def densification_stats(viewspace_points_grad, radii):
    """Per-view statistics exactly as S3Gaussian/scene/gaussian_model.py:728-730 and train.py:406 compute them."""
    vis = radii > 0
    g = torch.zeros(radii.shape[0], 1, device=radii.device)
    g[vis] = torch.norm(viewspace_points_grad[vis, :2], dim=-1, keepdim=True)
    return g, vis.float()[:, None], radii.float()
