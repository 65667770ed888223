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


def sh_grad_from_factors(means3D, campos, sh_color_grads, sh_degree, sh_coeffs=16, actor_ids=None, actor_pose=None, residual_dx=None,
                         scale=1.0):
    """dL/dshs [N,sh_coeffs,3] = scale * sum_v basis(normalize(world_mean - campos[v])) (x) sh_color_grads[v]   (one HIP launch).
    means3D [N,3]; campos [V,3]; sh_color_grads [V,N,3] (GaussianRasterizer.last_sh_color_grad of every view)."""
    import ctypes as C
    from . import _lib as L
    from .rasterizer import _fill_motion
    if means3D.device.type != "cuda":
        raise L.EmdError("sh_grad_from_factors needs tensors on a ROCm device; there is no CPU path")
    N, V = means3D.shape[0], campos.shape[0]
    means3D, campos, g = means3D.detach().contiguous().float(), campos.detach().contiguous().float(), sh_color_grads.contiguous().float()
    assert g.shape == (V, N, 3)
    mo = L.EmdMotion()
    ids = None if actor_ids is None else actor_ids.to(torch.int32).contiguous()          # (kept alive until the launch is queued)
    pose = None if actor_pose is None else actor_pose.detach().contiguous().float()
    rdx = None if residual_dx is None else residual_dx.detach().contiguous().float()
    _fill_motion(mo, ids, pose, rdx, None)
    out = torch.empty(N, sh_coeffs, 3, device=means3D.device, dtype=torch.float32)
    L.check(L.load().emd_sh_grad_from_factors(N, V, int(sh_degree), int(sh_coeffs), means3D.data_ptr(), C.byref(mo), campos.data_ptr(),
                                              g.data_ptr(), float(scale), out.data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emd_sh_grad_from_factors")
    return out


def exchange_sh_gradient(sh_param, means3D, campos_local, sh_degree, actor_ids=None, actor_pose=None, residual_dx=None, average=True,
                         also_allreduce=None):
    """View-parallel replacement of the all-reduce of the SH gradient (81 % of the gradient bytes): dL/dshs of a view is the
    outer product of the SH basis of its view direction and a per-Gaussian colour gradient, so each rank all-gathers the
    [N,3] factors (GaussianRasterizer.last_sh_color_grad, published by the backward when RasterConfig.factored_sh_grad is set)
    and the camera centres -- 12 bytes per Gaussian and rank instead of 192 -- and rebuilds the same dense, averaged gradient
    locally.  Needs the world-space means to be identical on all ranks (same timestamp), which view-parallel steps have.
    Sets sh_param.grad.  `also_allreduce`: the other parameters; their gradients are all-reduced (averaged) in place, issued
    right behind the gathers so that the local rebuild overlaps them."""
    from .rasterizer import GaussianRasterizer
    g_local = GaussianRasterizer.last_sh_color_grad
    if g_local is None:
        raise RuntimeError("no SH colour-gradient factor: set RasterConfig.factored_sh_grad before the backward pass")
    W = world_size()
    works, others = [], []
    if W == 1:
        g_all, campos = g_local[None], campos_local.reshape(1, 3).to(g_local.device)
    else:
        N = g_local.shape[0]
        g_cat = torch.empty(W * N, 3, device=g_local.device, dtype=g_local.dtype)      # ranks concatenated along dim 0
        campos = torch.empty(W, 3, device=g_local.device, dtype=torch.float32)
        gathers = [dist.all_gather_into_tensor(g_cat, g_local.contiguous(), async_op=True),
                   dist.all_gather_into_tensor(campos, campos_local.reshape(1, 3).to(g_local.device, torch.float32).contiguous(), async_op=True)]
        if also_allreduce is not None:
            others = [p.grad for p in also_allreduce if p is not sh_param and p.grad is not None]
            # the rasterizer hands out its four small gradients as views of one slab (44 B per Gaussian): when autograd kept those
            # views as the parameters' .grad, one collective over the slab replaces four
            slab = GaussianRasterizer.last_grad_slab
            if slab is not None:
                base = slab.untyped_storage().data_ptr()
                inside = [g for g in others if g.untyped_storage().data_ptr() == base]
                if inside and sum(g.numel() for g in inside) == slab.numel():
                    others = [g for g in others if g.untyped_storage().data_ptr() != base] + [slab]
            exchange_sh_gradient.last_num_allreduce = len(others)
            others.sort(key=lambda g: -g.numel())
            gloo = dist.get_backend() == "gloo"
            op = dist.ReduceOp.SUM if (gloo or not average) else dist.ReduceOp.AVG
            works = [dist.all_reduce(g, op=op, async_op=True) for g in others]
        for w in gathers:
            w.wait()
        g_all = g_cat.view(W, N, 3)
    sh_param.grad = sh_grad_from_factors(means3D, campos, g_all, sh_degree, sh_param.shape[1], actor_ids, actor_pose, residual_dx,
                                         scale=(1.0 / W) if average else 1.0).view_as(sh_param)
    for w in works:
        w.wait()
    if works and average and dist.get_backend() == "gloo":
        for g in others:
            g.div_(float(W))


def reduce_densification_stats(grad_norm_accum, denom, max_radii2D):
    """Cross-view reduction of the per-view densification statistics (norms taken per view, before reduction)."""
    if world_size() == 1:
        return
    dist.all_reduce(grad_norm_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)


def add_densification_stats(viewspace_points_grad, radii, xyz_gradient_accum, denom, max_radii2D):
    """In-place update of the three per-Gaussian densification statistics for one view in ONE launch and without the
    boolean-mask indexing of the reference (gaussian_model.py:728-730, train.py:403-406), which costs a host sync per step:
    where radii > 0:  xyz_gradient_accum += |grad.xy|, denom += 1, max_radii2D = max(max_radii2D, radii)."""
    import ctypes as C
    from . import _lib as L
    if radii.device.type != "cuda":
        raise L.EmdError("add_densification_stats needs tensors on a ROCm device; there is no CPU path")
    g = viewspace_points_grad.contiguous().float()
    r = radii.contiguous().to(torch.int32)
    for t in (xyz_gradient_accum, denom, max_radii2D):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32 and t.numel() == r.numel())
    L.check(L.load().emd_densification_stats(r.numel(), r.data_ptr(), g.data_ptr(), L.ptr(xyz_gradient_accum), L.ptr(denom),
                                             L.ptr(max_radii2D), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "emd_densification_stats")


def densification_stats(viewspace_points_grad, radii):
    """Per-view statistics exactly as S3Gaussian/scene/gaussian_model.py:728-730 and train.py:406 compute them."""
    vis = radii > 0
    g = torch.zeros(radii.shape[0], 1, device=radii.device)
    g[vis] = torch.norm(viewspace_points_grad[vis, :2], dim=-1, keepdim=True)
    return g, vis.float()[:, None], radii.float()
