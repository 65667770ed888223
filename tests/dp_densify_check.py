"""Two view-parallel ranks through density control (run under torch.distributed.run by tests/test_bench_multirank_gpu.py; gloo, both ranks
on the one GPU of the test box).

The config-5-style loop of tests/test_gaussian_model_gpu.py::test_training_loop_densifies_and_prunes_unattended, view-parallel: rank r
renders view r of every step from a recorded step (emd_amd.StepGraphs, segmented: the SH-factor gathers are issued between the two graphs),
the gradients are exchanged (factors + slab, emd_amd.dp.GradientExchange with the reference's split SH parameter), every rank runs the same
capturable Adam step; every EVENT_EVERY steps the statistics are reduced over the ranks (dp.reduce_densification_stats: SUM, SUM, MAX) and
every rank densifies, every second event prunes, one opacity reset -- the graphs are released and recorded again around each event (the point
count, every parameter tensor, the Adam moments, the backward workspace and the exchange buffers change size).

Asserted after every event: the SAME point count on both ranks and `torch.equal` on every parameter, Adam moment and statistic across the
ranks (a rank that disagrees on N hangs the next collective; one that disagrees on a value drifts); and, on rank 0, agreement with a
single-process run of the same loop that renders both views per step and averages (same seeds, dense SH gradient, no graphs): the first step's
averaged gradients within the float atomics' tolerance, the point counts of every event within 1 %, the parameters in distribution.
Reference semantics: S3Gaussian/scene/gaussian_model.py:442-556 (densify_and_split's torch.normal at :543 -> a Philox draw keyed by
(seed, event), identical on all ranks), :728-730 (statistics), S3Gaussian/train.py:404-423 (the schedule)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STEPS, EVENT_EVERY, RESET_AT = 130, 40, 120
H, W, N0 = 96, 128, 6000


def train_args():
    import types
    return types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-3, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                 position_lr_max_steps=30000, deformation_lr_init=1.6e-4, deformation_lr_final=1.6e-5, deformation_lr_delay_mult=0.01,
                                 grid_lr_init=1.6e-3, grid_lr_final=1.6e-5, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3,
                                 sky_cube_map_lr_init=0.01, sky_cube_map_lr_final=1e-4, sky_cube_map_max_steps=30000, capturable_optimizer=True)


def make_model(dev):
    from emd_amd import scenes
    from emd_amd.gaussian_model import GaussianModel
    sc = scenes.make_static_scene(N0, seed=3)
    means = sc.means.clone()
    means[:, 0] = means[:, 0] * 0.25 + 1.0
    means[:, 1] *= 0.3
    means[:, 2] = means[:, 2] * 0.3 + 1.0
    m = GaussianModel(device=dev, densify_seed=1)
    m.create_from_tensors(means, torch.rand(N0, 3, generator=torch.Generator().manual_seed(12)), sc.log_scales + 1.0, spatial_lr_scale=1.0)
    with torch.no_grad():
        m._features_rest.copy_(0.05 * torch.randn(m._features_rest.shape, generator=torch.Generator().manual_seed(13)).to(dev))
    m.active_sh_degree = 3
    m.training_setup(train_args())
    return m


def views(dev, n):
    from emd_amd import GaussianRasterizationSettings, scenes
    out = []
    for v in range(n):
        cam = scenes.small_camera(H, W, eye=(0.0, 0.4 * v - 0.2, 1.5), yaw=-8.0 + 16.0 * v)
        rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                           cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
        tgt = (torch.rand(3, H, W, generator=torch.Generator().manual_seed(90 + v)) * 0.5 + 0.25).to(dev)
        out.append((cam, rs, tgt))
    return out


def state_tensors(m, stats=True):
    """Every tensor a replica must agree on: parameters, Adam moments + step counts, and -- right behind an event, where they were reduced
    over the ranks and restarted -- the statistics (between events each rank accumulates its own views')."""
    ts = {n: getattr(m, m._ATTR[n]).detach() for n in m.GROUPS}
    for g in m.optimizer.param_groups:
        st = m.optimizer.state.get(g["params"][0])
        if st:
            ts["adam_m/" + g["name"]], ts["adam_v/" + g["name"]] = st["exp_avg"], st["exp_avg_sq"]
            ts["adam_step/" + g["name"]] = st["step"].reshape(-1).float() if torch.is_tensor(st["step"]) else torch.tensor([float(st["step"])], device=m.device)
    if stats:
        ts["xyz_gradient_accum"], ts["denom"], ts["max_radii2D"] = m.xyz_gradient_accum, m.denom, m.max_radii2D
    return ts


def event(m, it, reduce_stats):
    """The schedule of the loop: densify every EVENT_EVERY steps, prune every second event, one opacity reset."""
    did = []
    with torch.no_grad():
        if it % EVENT_EVERY == 0:
            reduce_stats()
            k = m.densify(2e-4, 0.005, 4.0, None)
            did.append(("densify",) + tuple(k))
            if it % (2 * EVENT_EVERY) == 0:
                k = m.prune(2e-4, 0.005, 4.0, 20)
                did.append(("prune",) + tuple(k))
        if it == RESET_AT:
            m.reset_opacity()
            did.append(("reset",))
    return did


def single_process_run(dev, world, log):
    """The same loop in ONE process: both views rendered per step, gradients averaged, statistics of both views added (eager, dense dL/dshs)."""
    from emd_amd import GaussianRasterizer, RasterOptions
    from emd_amd.model import l1_loss
    m = make_model(dev)
    vs = views(dev, world)
    opts = RasterOptions(compute_normal=False)
    first_grads, counts, snaps = None, [], {}
    for it in range(1, STEPS + 1):
        m.update_learning_rate(it)
        m.optimizer.zero_grad(set_to_none=True)
        for cam, rs, tgt in vs:
            sp = torch.zeros_like(m._xyz, requires_grad=True)
            img, _, _, _, radii, _ = GaussianRasterizer(rs, options=opts)(means3D=m._xyz, means2D=sp, shs=m.get_features, opacities=m._opacity,
                                                                          scales=m._scaling, rotations=m._rotation, raw_params=True)
            (l1_loss(img, tgt) / world).backward()
            with torch.no_grad():
                m.add_densification_stats(sp.grad * world, radii)          # (the statistics use each view's own gradient, not the average)
        if it == 1:
            first_grads = {n: getattr(m, m._ATTR[n]).grad.clone() for n in m.GROUPS if getattr(m, m._ATTR[n]).grad is not None}
        with torch.no_grad():
            m.optimizer.step()
        if event(m, it, lambda: None):
            counts.append((it, m._xyz.shape[0]))
            snaps[it] = {n: getattr(m, m._ATTR[n]).detach().clone() for n in ("xyz", "opacity", "scaling")}
    return first_grads, counts, snaps


def main():
    from emd_amd import GaussianRasterizer, RasterCall, RasterOptions, StepGraphs, dp
    from emd_amd.model import l1_loss
    rank, world, _ = dp.init_from_env()
    assert world >= 2, "run under torch.distributed.run with >= 2 ranks"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = make_model(dev)
    cam, rs, target = views(dev, world)[rank]
    campos = cam.camera_center.to(dev)
    opts = RasterOptions(compute_normal=False, no_sync=True, capacity_hint=3_000_000, factored_sh_grad=True)
    state = {}

    class _Cut:                      # the step names its cut hook before the StepGraphs object exists (the constructor already runs the step)
        target = None

        def cut(self, *a):
            if self.target is not None:
                self.target.cut(*a)
    proxy = _Cut()

    def iteration(_key):
        m.optimizer.zero_grad(set_to_none=True)
        sp = state["sp"]
        sp.grad = None
        rec = RasterCall()
        rec.on_sh_factor = proxy.cut
        img, _, _, _, radii, _ = GaussianRasterizer(rs, options=opts)(means3D=m._xyz, means2D=sp, shs=m.get_features.detach(), opacities=m._opacity,
                                                                      scales=m._scaling, rotations=m._rotation, raw_params=True, record=rec)
        l1_loss(img, target).backward()
        with torch.no_grad():
            m.add_densification_stats(sp.grad, radii)
        state["rec"] = rec

    def record():
        state["sp"] = torch.zeros_like(m._xyz, requires_grad=True)
        keep = [t.clone() for t in (m.xyz_gradient_accum, m.denom, m.max_radii2D)]        # the recorder's eager warm-up is not a training step
        g = StepGraphs.__new__(StepGraphs)
        proxy.target = g
        g.__init__(iteration, [0], optimizers=[m.optimizer], warmup=1, segmented=True)
        assert g.segments() == 2, g.segments()
        for t, k in zip((m.xyz_gradient_accum, m.denom, m.max_radii2D), keep):
            t.copy_(k)
        state["rec"].on_sh_factor = None
        return g

    def exchange_after(graphs):
        x = dp.GradientExchange(campos)
        graphs.replay(0, between=lambda i: x.start_factors(state["rec"]))
        x.start(state["rec"])
        x.finish((m._features_dc, m._features_rest), m._xyz, m.active_sh_degree,
                 other_params=[m._xyz, m._opacity, m._scaling, m._rotation])
        return x

    def check_replicas(tag, stats=True):
        n_all = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(n_all, torch.tensor([m._xyz.shape[0]], dtype=torch.int64, device=dev))
        ns = [int(t) for t in n_all]
        assert len(set(ns)) == 1, f"{tag}: ranks disagree on the point count: {ns}"
        for name, t in state_tensors(m, stats).items():
            t = t.detach().contiguous().float().reshape(-1)
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            for r_, p_ in enumerate(parts[1:], 1):
                assert torch.equal(parts[0], p_), f"{tag}: {name} differs between rank 0 and rank {r_} (max |d| {float((parts[0] - p_).abs().max()):.3e})"
        return ns[0]

    graphs = record()
    counts, snaps, first_grads, collectives = [], {}, None, None
    for it in range(1, STEPS + 1):
        m.update_learning_rate(it)
        x = exchange_after(graphs)
        if it == 1:
            torch.cuda.synchronize()
            collectives = x.num_collectives
            first_grads = {n: getattr(m, m._ATTR[n]).grad.clone() for n in m.GROUPS if getattr(m, m._ATTR[n]).grad is not None}
        with torch.no_grad():
            m.optimizer.step()
        if it % EVENT_EVERY == 0 or it == RESET_AT:
            graphs.release()
            did = event(m, it, lambda: dp.reduce_densification_stats(m.xyz_gradient_accum, m.denom, m.max_radii2D))
            n_now = check_replicas(f"step {it} {did}")
            counts.append((it, n_now))
            snaps[it] = {n: getattr(m, m._ATTR[n]).detach().clone() for n in ("xyz", "opacity", "scaling")}
            graphs = record()
    check_replicas("end of the loop", stats=False)
    graphs.release()
    assert collectives == 3, collectives                      # factor gather + camera gather + the slab (nothing else carries a gradient here)
    assert len({n for _, n in counts}) >= 3, counts            # the point count moved at the events (grew, shrank)
    if rank == 0:
        ref_grads, ref_counts, ref_snaps = single_process_run(dev, world, None)
        # (1) the first step starts from identical parameters: the exchanged average equals the one-process average up to the float atomics' order
        for n, g in first_grads.items():
            r = ref_grads[n]
            err = float((g - r).abs().max())
            assert err <= 2e-5 * float(r.abs().max()) + 1e-12, f"first-step gradient of {n}: {err:.3e} vs max {float(r.abs().max()):.3e}"
        # (2) the events: same schedule; point counts within 1 % (threshold decisions on gradients that differ in their last bits), and
        # the parameters of the first event -- identical point sets unless a Gaussian sat within rounding of a threshold -- in distribution
        assert [i for i, _ in counts] == [i for i, _ in ref_counts]
        for (i, a), (_, b) in zip(counts, ref_counts):
            assert abs(a - b) <= 0.01 * b + 2, f"event at step {i}: {a} points view-parallel, {b} in one process"
        i0 = counts[0][0]
        if counts[0][1] == ref_counts[0][1]:
            for n in ("xyz", "opacity", "scaling"):
                a, b = snaps[i0][n], ref_snaps[i0][n]
                d = (a - b).abs().reshape(-1)
                scale = float(b.abs().max())
                frac = float((d > 1e-2 * scale).float().mean())
                assert frac < 0.02, f"{n} after the first event: {frac:.4f} of the entries differ by more than 1 % of the range"
        print(f"OK ranks={world} events={counts} one_process={ref_counts} collectives_per_step={collectives}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
