"""Host-side mirror of the reference's per-step render glue around the rasterizer (benchmark / training harness).

What one step does, with the reference line each piece restates:
  activations exp / normalize / sigmoid          S3Gaussian/gaussian_renderer/__init__.py:99-101
  per-frame actor pose table (q, t, valid)        OmniRe/models/nodes/rigid.py:478-568 (gather + rigid transform are
                                                  fused into the projection kernel; the table is built here)
  GaussianRasterizationSettings + rasterizer call S3Gaussian/gaussian_renderer/__init__.py:49-62,145-155
  L1 photometric loss                             S3Gaussian/train.py:226
  loss.backward()                                 S3Gaussian/train.py:366
(Parameter store with density control and on-disk formats: emd_amd/gaussian_model.py; optimiser: emd_amd/optim.py.)
"""
import torch
import torch.nn.functional as F

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from .scenes import GaussianScene


class StreetGaussians(torch.nn.Module):
    """Parameter store in the reference's layout (S3Gaussian/scene/gaussian_model.py:54-71)."""

    def __init__(self, scene: GaussianScene, device, track_heads=False, track_seed=7):
        """`track_heads`: attach the learned per-actor track offsets (OmniRe/models/nodes/rigid.py:108-122,203-246: temporal
        tables + four linear heads + a 4-d embedding per actor Gaussian).  The reference zero-initialises the heads; here they
        get small seeded weights so that the offsets (and every gradient path through them) are non-trivial in benchmarks."""
        super().__init__()
        P = lambda t: torch.nn.Parameter(t.to(device).contiguous())
        self._xyz = P(scene.means)
        self._scaling = P(scene.log_scales)
        self._rotation = P(scene.quats)
        self._opacity = P(scene.opacity_logits)
        self._features = P(scene.shs)           # [N,16,3] = cat(features_dc, features_rest)
        self.has_actors = scene.actor_id is not None
        if self.has_actors:
            self.register_buffer("actor_id", scene.actor_id.to(device))
            self.instances_quats = P(scene.actor_quats)    # [F,A,4]
            self.instances_trans = P(scene.actor_trans)    # [F,A,3]
            self.register_buffer("instances_fv", scene.actor_valid.to(device))
        self.track_heads = None
        if self.has_actors and track_heads:
            from .motion import TrackOffsetHeads
            A = scene.actor_quats.shape[1]
            g = torch.Generator().manual_seed(track_seed)
            heads = TrackOffsetHeads(A)
            with torch.no_grad():
                # (the temporal tables too: the module's own initialiser draws from the process-wide generator, which made the actor
                #  poses -- and with them V and D of a benchmark run -- differ by a few counts from process to process)
                heads.weight.copy_(torch.randn(heads.weight.shape, generator=g) * 0.01 / heads.weight.shape[2] ** 0.5)
                for lin, sc in ((heads.track_trans_c, 0.05), (heads.track_trans_f, 0.05), (heads.track_rot_c, 0.01), (heads.track_rot_f, 0.01)):
                    lin.weight.copy_(sc * torch.randn(lin.weight.shape, generator=g))
            self.track_heads = heads.to(device)
            dyn = (scene.actor_id >= 0).nonzero()[:, 0]
            self.register_buffer("dyn_ids", scene.actor_id[dyn].to(device).long())
            self._embeddings = P(0.1 * torch.randn(dyn.numel(), 4, generator=g))       # [n_dyn, 4] (gaussian_embedding_dim)
            self.num_frames = scene.actor_quats.shape[0]
        self.active_sh_degree = 3

    @property
    def num_points(self):
        return self._xyz.shape[0]

    def actor_pose(self, frame, step=0):
        """[A,12] pose rows for one frame, with the learned track offsets (translation and rotation about z) when the model has
        the heads: rigid.py:519-532,547-566.  `frame` / `step` may be device tensors (int32 [1] / any integer [1]): the row and the
        coarse-to-fine level are then selected on the device and the call can be replayed from a hipGraph."""
        from .motion import actor_pose_table
        if self.track_heads is not None:
            return self.track_heads.pose_table(self.instances_quats, self.instances_trans, self.instances_fv, frame, self._embeddings,
                                               self.dyn_ids, step)
        return actor_pose_table(self.instances_quats, self.instances_trans, self.instances_fv, frame, None, None)


def density_control(model: "StreetGaussians", xyz_gradient_accum, denom, max_radii2D, max_grad=2e-4, min_opacity=0.005, extent=27.5,
                    max_screen_size=None, percent_dense=0.01, seed=0, event=0, optimizer=None):
    """One density-control event of the training loop (S3Gaussian/train.py:404-423 -> scene/gaussian_model.py:442-556: clone the small,
    split the large Gaussians whose accumulated view-space gradient exceeds `max_grad`, then prune the transparent / oversized ones) on a
    StreetGaussians parameter store, in place; returns {"n_before", "n_after", "cloned", "split", "pruned"}.

    The BACKGROUND Gaussians (actor id -1) take part; an actor's points stay as they are -- the reference holds every tracked actor at a
    fixed budget (<= 5000 Gaussians, OmniRe/configs/paper_legacy/omnire.yaml:93; the bench scene's actors sit at that cap) and stores an
    actor's points contiguously, which the per-actor kernels rely on.  The engine is emd_amd.gaussian_model.restructure_rows -- the device-side
    decide -> scan -> index -> gather of GaussianModel (csrc/densify.hip), here applied to the background rows of the store's own tensors IN
    PLACE of a copy (round 6: no slice / cat passes around the gather; `_features` travels as one 48-float row): two events (densify, prune),
    two host reads.  The split samples are a Philox draw keyed by (seed, event, source row, replica), so every rank of a view-parallel run that
    calls this with the same (reduced) statistics ends with bit-identical parameters -- and the background ends exactly as a GaussianModel
    holding only the background would.  The three statistics tensors are consumed (the caller allocates fresh zeros for the new point count,
    as densification_postfix does, gaussian_model.py:526-530).

    Every per-Gaussian parameter of the store is REPLACED by a fresh `nn.Parameter` (the point count changes).  `optimizer` (torch.optim.Adam /
    emd_amd.optim.Adam whose groups hold `_xyz`, `_scaling`, `_rotation`, `_opacity`, `_features` as single-parameter groups): its groups are
    pointed at the new parameters and both Adam moments travel with the rows -- survivors keep theirs, new rows start at zero -- as the reference's
    `cat_tensors_to_optimizer` / `_prune_optimizer` do (gaussian_model.py:454-500); the actors' rows keep theirs.  WITHOUT it a caller's optimizer
    still points at the old tensors and must be rebuilt (its state for these five parameters is lost)."""
    from . import _lib as L
    from .gaussian_model import restructure_rows
    dev, N = model._xyz.device, model._xyz.shape[0]
    if dev.type != "cuda":
        raise L.EmdError("density_control needs the store on a ROCm device; there is no CPU path")
    n_dyn = int((model.actor_id >= 0).sum()) if model.has_actors else 0
    if model.has_actors and n_dyn and not bool((model.actor_id[:n_dyn] >= 0).all()):
        raise ValueError("density_control expects the actors' points in front of the background's (the reference's node order)")
    names = ("_xyz", "_scaling", "_rotation", "_opacity", "_features")
    roles = {"_xyz": L.DENSIFY_ROLE_XYZ, "_scaling": L.DENSIFY_ROLE_SCALING, "_rotation": L.DENSIFY_ROLE_COPY, "_opacity": L.DENSIFY_ROLE_COPY,
             "_features": L.DENSIFY_ROLE_COPY}
    groups = {}
    if optimizer is not None:
        for grp in optimizer.param_groups:
            for nm in names:
                if len(grp["params"]) == 1 and grp["params"][0] is getattr(model, nm):
                    groups[nm] = grp
    state = {nm: optimizer.state.get(getattr(model, nm)) for nm in groups}
    # the rows that take part: views of the background rows (a dim-0 slice of a contiguous tensor is contiguous: no copy)
    cur = {nm: getattr(model, nm).detach() for nm in names}
    mom = {(nm, k): state[nm][k].detach() for nm in groups if state[nm] and "exp_avg" in state[nm] for k in ("exp_avg", "exp_avg_sq")}
    stats = [xyz_gradient_accum.reshape(N, 1), denom.reshape(N, 1), max_radii2D.reshape(N)]
    full = {k: v for k, v in cur.items()}
    counts = {"cloned": 0, "split": 0, "pruned": 0}
    seed_word = (int(seed) * 0x9E3779B97F4A7C15 + int(event)) & 0xFFFFFFFFFFFFFFFF
    with torch.no_grad():
        for mode in (L.DENSIFY_MODE_DENSIFY, L.DENSIFY_MODE_PRUNE):
            n_rows = full["_xyz"].shape[0]
            nb = n_rows - n_dyn
            bg = lambda t: t[n_dyn:]
            a = L.EmdDensifyArgs()
            a.scaling = bg(full["_scaling"]).data_ptr()
            if mode == L.DENSIFY_MODE_DENSIFY:
                a.grad_accum, a.denom = bg(stats[0]).data_ptr(), bg(stats[1]).data_ptr()
                a.grad_threshold, a.percent_dense, a.scene_extent = float(max_grad), float(percent_dense), float(extent)
            else:
                a.opacity, a.max_radii2D = bg(full["_opacity"]).data_ptr(), bg(stats[2]).data_ptr()
                a.min_opacity, a.scene_extent = float(min_opacity), float(extent)
                a.max_screen_size = float(max_screen_size) if max_screen_size else 0.0
            jobs = [(bg(full[nm]), roles[nm]) for nm in names]
            keys = [("param", nm) for nm in names]
            for (nm, k), t in mom.items():
                jobs.append((bg(t), L.DENSIFY_ROLE_STATE))
                keys.append(("mom", (nm, k)))
            for i_, t in enumerate(stats):
                jobs.append((bg(t), L.DENSIFY_ROLE_ZERO))
                keys.append(("stat", i_))
            outs, (n_keep, n_clone, n_split) = restructure_rows(mode, a, jobs, nb, seed=seed_word, front_rows=n_dyn, scaling=bg(full["_scaling"]),
                                                                rotation=bg(full["_rotation"]))
            if mode == L.DENSIFY_MODE_DENSIFY:
                counts["cloned"], counts["split"] = n_clone, n_split
                if outs is None:          # nothing selected: the reference's densification_postfix still clears the statistics (gaussian_model.py:526-530)
                    for t in stats:
                        t[n_dyn:].zero_()
            else:
                counts["pruned"] = nb - n_keep
            if outs is None:
                continue
            for (what, key), (t_old, _), out in zip(keys, jobs, outs):
                if what == "param":
                    out[:n_dyn].copy_(full[key][:n_dyn])
                    full[key] = out
                elif what == "mom":
                    out[:n_dyn].copy_(mom[key][:n_dyn])
                    mom[key] = out
                else:
                    out[:n_dyn].copy_(stats[key][:n_dyn])
                    stats[key] = out
        n_new = full["_xyz"].shape[0] - n_dyn
        for nm in names:
            old = getattr(model, nm)
            new = torch.nn.Parameter(full[nm]) if full[nm] is not cur[nm] else old
            if nm in groups and new is not old:
                st = optimizer.state.pop(old, None)
                groups[nm]["params"] = [new]
                if st:
                    if (nm, "exp_avg") in mom:
                        st["exp_avg"], st["exp_avg_sq"] = mom[(nm, "exp_avg")], mom[(nm, "exp_avg_sq")]
                    optimizer.state[new] = st
            setattr(model, nm, new)
        if model.has_actors:
            model.actor_id = torch.cat([model.actor_id[:n_dyn], torch.full((n_new,), -1, dtype=model.actor_id.dtype, device=dev)])
    model._zero_xyz = None
    return {"n_before": N, "n_after": n_dyn + n_new, "cloned": int(counts["cloned"]), "split": int(counts["split"]), "pruned": int(counts["pruned"])}


def mix_dynamic_static(opacity_dynamic, opacity_static, shs_dynamic=None, shs_static=None, colors_dynamic=None, colors_static=None):
    """The `combine_dynamic_static` mixing of the reference's render() (S3Gaussian/gaussian_renderer/__init__.py:118-138; flag default off,
    arguments/gaussian_options.py:195): the deformed ("dynamic") and the undeformed ("static") copy of every Gaussian are drawn as ONE
    Gaussian whose ACTIVATED opacities add up (the sum may exceed 1; the rasterizer clamps alpha at 0.99) and whose colour is their
    opacity-weighted mean -- of the SH coefficients, or of the precomputed colours when those are given.  Plain tensor arithmetic in the
    reference's order of operations -> (opacity, shs or None, colors or None)."""
    total = opacity_dynamic + opacity_static
    dynamic_ratio, static_ratio = opacity_dynamic / total, opacity_static / total
    if colors_dynamic is not None:
        return total, None, colors_dynamic * dynamic_ratio + colors_static * static_ratio
    n, tail = shs_dynamic.shape[0], shs_dynamic.shape[1:]
    shs = shs_dynamic.view(n, -1) * dynamic_ratio + shs_static.view(n, -1) * static_ratio
    return total, shs.view(-1, *tail), None


def pre_compute_colors(shs, xyz, camera_center, degree):
    """SH -> RGB on the host side of the boundary (`convert_SHs_python`, gaussian_renderer/__init__.py:19-25): directions from the
    UNDEFORMED means, emd_sh_forward (HIP), + 0.5, clamp at 0."""
    from .gsplat_api import spherical_harmonics
    dirs = xyz - camera_center.to(xyz.device).reshape(1, 3)
    return torch.clamp_min(spherical_harmonics(degree, dirs, shs) + 0.5, 0.0)


def render(model: StreetGaussians, cam, bg, frame=0, debug=False, fuse_activations=True, deformation=None, embeddings=None,
           iteration=None, time=None, options=None, record=None, render_feat=False, need_feat=True, combine_dynamic_static=False,
           convert_SHs_python=False, residual=None, fused_l1=()):
    """The reference render() restricted to the hot path; returns the dict the training loop consumes.
    `deformation` (an emd_amd.deformation.deform_network) switches on the "fine" stage of gaussian_renderer/__init__.py:86-96:
    the residuals of the self-supervised EMD network are added to the raw parameters before the activations.
    `options` (emd_amd.RasterOptions) configures this call's rasterizer; `record` (emd_amd.RasterCall) receives its per-call state
    (also returned as out["raster_call"]).
    `render_feat` (with a deformation network that has the feature head): the reference's two feature passes
    (`colors_precomp = ddict["coarse"]["feat"]` / `["fine"]["feat"]`, gaussian_renderer/__init__.py:170-201) as extra colour
    sets of the SAME rasterizer call -> out["feat_c"], out["feat_f"]: one projection, one sort, one list walk instead of three.
    `need_feat=False` (with an emd_amd deform_network): nobody will read ddict[...]["feat"], so the feature head is not evaluated.
    `combine_dynamic_static` (fine stage; the reference's args.combine_dynamic_static, :118-138): the deformed and the undeformed copy of
    every Gaussian drawn as one (mix_dynamic_static); `convert_SHs_python` evaluates the colours in front of the boundary (:106-110),
    which is also the only configuration in which the reference's own decomposition passes run with that flag (render_decomposition).
    `residual` = (dx [N,3] or None, dq [N,4] or None): the learned per-Gaussian deformation residual of the supervised branch
    (`means = _means + delta_xyz`, `quats = get_quats + delta_quat` IN FRONT of the rigid transform, OmniRe/models/nodes/deformable.py:49-68)
    as inputs of the fused transform inside the projection kernel; gradients return to both tensors."""
    dev = model._xyz.device
    # the reference's zero "screen-space points" leaf that only collects dL/dmean2D: the zeros are never written, so one
    # cached buffer per model serves every step (a fresh leaf view each time, no 24 MB fill launch)
    z = getattr(model, "_zero_xyz", None)
    if z is None or z.shape != model._xyz.shape or z.device != model._xyz.device:
        z = model._zero_xyz = torch.zeros_like(model._xyz)
    screenspace_points = z.detach().requires_grad_(True)
    rs = raster_settings_for(cam, bg, model.active_sh_degree, 1.0, debug)
    rasterizer = GaussianRasterizer(raster_settings=rs, options=options)
    means3D, scales, rotations, opacity, shs, ddict = model._xyz, model._scaling, model._rotation, model._opacity, model._features, None
    if deformation is not None:
        t = getattr(cam, "time", 0.0) if time is None else time
        if isinstance(t, torch.Tensor) and t.device.type != "cpu":
            # a view's time that lives on the device (emd_amd.StepInputs.camera.time): nothing of it is baked into a recorded step
            times_sel = t.reshape(1, 1).to(torch.float32).expand(means3D.shape[0], 1)       # (a broadcast: one frame per step)
        else:
            times_sel = torch.full((1, 1), float(t), device=dev, dtype=torch.float32).expand(means3D.shape[0], 1)
        from .deformation import deform_network as _dn
        # (an emd_amd network hands the SH residuals over unsummed: `shs + dshs_c + dshs_f` is formed inside the projection kernel)
        extra_kw = {"need_feat": bool(need_feat or render_feat), "fused_shs_residuals": True} if isinstance(deformation, _dn) else {}
        if fused_l1 and isinstance(deformation, _dn):
            extra_kw["fused_l1"] = tuple(fused_l1)      # ddict[level]["<key>_abs_mean"]: the residual regularisers formed by the head kernels (residual_abs_mean)
        means3D, scales, rotations, opacity, shs, ddict = deformation(
            means3D, scales, rotations, opacity, shs, times_sel, embeddings, iteration, int(getattr(cam, "cam_no", 0)),
            getattr(cam, "time_diff", 0.0), True, **extra_kw)
    colors_precomp, combined = None, None

    def summed_shs():
        """shs + the residuals an emd_amd network hands over unsummed (formed inside the projection kernel on the usual path)"""
        t = shs
        for r_ in ((ddict.get("shs_residuals") if ddict is not None else None) or []):
            t = t + r_
        if ddict is not None and ddict.get("shs_residuals"):
            ddict["shs_residuals"] = None
        return t
    if combine_dynamic_static and deformation is not None:
        # the mixing works on ACTIVATED opacities (they are added): the three activations run here, as in the reference (:99-101,114-116)
        fuse_activations = False
        shs_dyn = summed_shs()
        o_dyn, o_sta = torch.sigmoid(opacity), torch.sigmoid(model._opacity)
        sc_sta, rot_sta = torch.exp(model._scaling), F.normalize(model._rotation)
        scales, rotations = torch.exp(scales), F.normalize(rotations)
        col_dyn = col_sta = None
        if convert_SHs_python:
            col_dyn = pre_compute_colors(shs_dyn, model._xyz, cam.camera_center, model.active_sh_degree)
            col_sta = pre_compute_colors(model._features, model._xyz, cam.camera_center, model.active_sh_degree)
        opacity, shs, colors_precomp = mix_dynamic_static(o_dyn, o_sta, shs_dyn, model._features, col_dyn, col_sta)
        combined = dict(dynamic=dict(means3D=means3D, shs=None if convert_SHs_python else shs_dyn, colors_precomp=col_dyn, opacities=o_dyn,
                                     scales=scales, rotations=rotations),
                        static=dict(means3D=model._xyz, shs=None if convert_SHs_python else model._features, colors_precomp=col_sta,
                                    opacities=o_sta, scales=sc_sta, rotations=rot_sta))
    else:
        if convert_SHs_python:
            colors_precomp, shs = pre_compute_colors(summed_shs(), model._xyz, cam.camera_center, model.active_sh_degree), None
        if not fuse_activations:
            # (fused: the three activations of the reference run inside K1 / K8 (raw_params), not as ~25 separate torch launches)
            scales = torch.exp(scales)
            rotations = F.normalize(rotations)
            opacity = torch.sigmoid(opacity)
    kw = {}
    feat_sets = []
    if render_feat and ddict is not None:
        feat_sets = [(lvl, ddict[lvl]["feat"]) for lvl in ("coarse", "fine") if ddict.get(lvl) is not None and ddict[lvl].get("feat") is not None]
        if feat_sets:
            kw["colors_extra"] = [f for _, f in feat_sets]
    shs_res = ddict.get("shs_residuals") if ddict is not None else None
    if shs_res:
        kw["shs_residuals"] = shs_res
    if model.has_actors:
        from .motion import DeviceStep
        it = 0 if iteration is None else (iteration if isinstance(iteration, (torch.Tensor, DeviceStep)) else int(iteration))
        kw.update(actor_ids=model.actor_id, actor_pose=model.actor_pose(frame, it))
    if residual is not None:
        kw.update(residual_dx=residual[0], residual_dq=residual[1])
    image, depth, normal, weight, radii, extra = rasterizer(
        means3D=means3D, means2D=screenspace_points, shs=None if colors_precomp is not None else shs, colors_precomp=colors_precomp,
        opacities=opacity, scales=scales, rotations=rotations, cov3Ds_precomp=None, extra_attrs=None, raw_params=fuse_activations,
        record=record, **kw)
    out = _RenderOutputs({"render": image, "viewspace_points": screenspace_points, "radii": radii,
           "depth": depth, "weight": weight, "normal": normal, "actor_pose": kw.get("actor_pose"), "ddict": ddict,
           "raster_call": rasterizer.last_call, "rasterizer": rasterizer,
           "boundary": dict(means3D=means3D, opacities=opacity, scales=scales, rotations=rotations, shs=None if colors_precomp is not None else shs,
                            colors_precomp=colors_precomp, raw_params=fuse_activations, shs_residuals=shs_res, combined=combined)})
    for (lvl, _), img in zip(feat_sets, extra or []):
        out["feat_c" if lvl == "coarse" else "feat_f"] = img
    return out


class _RenderOutputs(dict):
    """The dict render() returns.  `visibility_filter` (= radii > 0, gaussian_renderer/__init__.py:163) is formed when it is first
    read: the training loop of this package feeds `radii` to the densification statistics directly
    (dp.add_densification_stats), so a step that never looks at the mask does not launch an N-sized compare for it."""

    def __missing__(self, key):
        if key == "visibility_filter":
            v = self["radii"] > 0
            self[key] = v
            return v
        raise KeyError(key)

    def __contains__(self, key):
        return key == "visibility_filter" or dict.__contains__(self, key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


def render_decomposition(out, levels=("coarse", "fine", "coarse_fine"), top_fraction=0.005):
    """The evaluation-time decomposition passes of the reference's render() (`return_decomposition`, stage "fine":
    S3Gaussian/gaussian_renderer/__init__.py:203-294; both branches of `combine_dynamic_static`), from the dict `render(..., deformation=...)` returned:
    per level of the deformation (coarse dx, fine dx, and their difference) (a) the `top_fraction` of the Gaussians that move farthest,
    rendered alone -- a boolean-mask subset of every boundary tensor through the SAME rasterizer object -- and (b) the whole scene coloured
    by |dx| / max |dx| (`colors_precomp`).  Returns {"coarse_render": {render, depth, color, weight, normal, dx}, ...} as the reference's
    `ddict_render`.  No gradients are needed on this path (the reference only visualises it): it runs under no_grad."""
    dd, bd, rast = out["ddict"], out["boundary"], out["rasterizer"]
    if dd is None:
        raise ValueError("render_decomposition needs the deformation residuals: call render(..., deformation=...) first")
    dx = {"coarse": dd["coarse"]["dx"] if dd.get("coarse") else None, "fine": dd["fine"]["dx"] if dd.get("fine") else None}
    if dx["coarse"] is not None and dx["fine"] is not None:
        dx["coarse_fine"] = dx["coarse"] - dx["fine"]
    res = {}
    comb = bd.get("combined")
    with torch.no_grad():
        shs_all = bd["shs"]
        for r_ in (bd.get("shs_residuals") or []):           # (handed to the main pass unsummed)
            shs_all = shs_all + r_
        base = dict(opacities=bd["opacities"], scales=bd["scales"], rotations=bd["rotations"], raw_params=bd["raw_params"], cov3Ds_precomp=None,
                    extra_attrs=None)
        m2d = torch.zeros_like(bd["means3D"])
        for lvl in levels:
            d = dx.get(lvl)
            if d is None:
                continue
            d_abs = d.detach().abs()
            if comb is not None:
                # combine_dynamic_static (:206-231): no top-0.5 % subset -- the coarse level shows the DYNAMIC copies (deformed means, their own
                # opacity and colour), the fine and coarse - fine levels the STATIC ones.  (The reference reaches these passes only with
                # precomputed colours: with SH colours its `colors_precomp_static` is unassigned, :214; here both colour forms work.)
                part = comb["dynamic" if lvl == "coarse" else "static"]
                img_d, depth_d, normal_d, weight_d, _, _ = rast(means3D=part["means3D"], means2D=m2d, shs=part["shs"], colors_precomp=part["colors_precomp"],
                                                               opacities=part["opacities"], scales=part["scales"], rotations=part["rotations"],
                                                               cov3Ds_precomp=None, extra_attrs=None)
            else:
                dist = d_abs.norm(dim=1)
                k = int(dist.shape[0] * top_fraction)
                mask = torch.zeros_like(dist, dtype=torch.bool)
                if k > 0:
                    mask[torch.topk(dist, k)[1]] = True
                sub = {n_: (v[mask] if isinstance(v, torch.Tensor) else v) for n_, v in base.items()}
                cp = bd.get("colors_precomp")
                img_d, depth_d, normal_d, weight_d, _, _ = rast(means3D=bd["means3D"][mask], means2D=m2d[mask], shs=None if cp is not None else shs_all[mask],
                                                               colors_precomp=None if cp is None else cp[mask], **sub)
            col = d_abs / d_abs.max(dim=0, keepdim=True)[0]
            color_dx = rast(means3D=bd["means3D"], means2D=m2d, shs=None, colors_precomp=col, **base)[0]
            res[lvl + "_render"] = {"render": img_d, "depth": depth_d, "color": color_dx, "weight": weight_d, "normal": normal_d, "dx": d}
    return res


def raster_settings_for(cam, bg, sh_degree, scaling_modifier=1.0, debug=False):
    """The 12-field record exactly as S3Gaussian/gaussian_renderer/__init__.py:46-62 builds it from a camera."""
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=bg, scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=debug)


def apply_deform(point, scales, rotations, opacity, shs, ddict_c=None, ddict_f=None):
    """Final adds of the EMD deformation (S3Gaussian/scene/deformation.py:439-481) under the run-script flags
    (`no_ds`, `no_dr`: scales and rotations pass through; dx, do, dshs of the coarse and fine level are added).
    The network that produces the residuals: emd_amd/deformation.py (HexPlane lookup, temporal row and fused MLP kernels)."""
    for dd in (ddict_c, ddict_f):
        if dd is None:
            continue
        point = point + dd["dx"]
        opacity = opacity + dd["do"]
        shs = shs + dd["dshs"]
    return point, scales, rotations, opacity, shs


def boundary_tensors(xyz, scaling, rotation, opacity, features, stage="coarse", ddict=None):
    """What render() hands to the rasterizer (gaussian_renderer/__init__.py:86-101,145-155): deformation in the
    fine stage, then exp / normalize / sigmoid."""
    if "fine" in stage:
        xyz, scaling, rotation, opacity, features = apply_deform(xyz, scaling, rotation, opacity, features,
                                                                 ddict["coarse"], ddict["fine"])
    return dict(means3D=xyz, scales=torch.exp(scaling), rotations=F.normalize(rotation), opacities=torch.sigmoid(opacity),
                shs=features)


# (device index, stream) -> the scratch table of emd_l1_loss_ws (zero between calls).  Tables live for the process: 4 KB each, and a hipGraph
# captured with one has its ADDRESS baked in -- an evicted (freed) table would be written by every later replay.
_l1_scratch = {}


def _l1_call(n, a_ptr, b_ptr, loss, grad_ptr):
    """emd_l1_loss_ws with the per-(device, stream) scratch table: no zero-fill launch in front of the kernel.  The table is created (zeroed)
    outside of stream capture only -- a capture that meets a stream for the first time uses the plain entry point."""
    import ctypes as C
    from . import _lib as L
    st = torch.cuda.current_stream()
    key = (loss.device.index, st.cuda_stream)
    sc = _l1_scratch.get(key)
    if sc is None and not torch.cuda.is_current_stream_capturing():
        sc = _l1_scratch[key] = torch.zeros(1024, dtype=torch.int32, device=loss.device)       # EMD_L1_SCRATCH_WORDS
    L.check(L.load().emd_l1_loss_ws(n, a_ptr, b_ptr, loss.data_ptr(), grad_ptr, L.ptr(sc), C.c_void_p(st.cuda_stream)), "emd_l1_loss")


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        import ctypes as C
        from . import _lib as L
        if a.device.type != "cuda":
            raise L.EmdError("l1_loss needs tensors on a ROCm device; there is no CPU path")
        a, b = a.contiguous(), b.contiguous()
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        grad = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        _l1_call(a.numel(), a.data_ptr(), b.data_ptr(), loss, L.ptr(grad))
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        if g is _UNIT.get(grad.device):          # loss.backward(unit_gradient(device)): the factor is 1 by construction, no multiply launch
            return grad, None
        return grad * g, None


_UNIT = {}


def unit_gradient(device):
    """A resident scalar 1.0 to start backward() from: `loss.backward(unit_gradient(device))` saves autograd's ones_like fill, and the
    fused L1 loss recognises the object and returns its stored gradient image without the 1.0-multiply pass over it (two launch-bound
    kernels per step less; any other root gradient takes the general path)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = _UNIT.get(device)
    if t is None:
        t = _UNIT[device] = torch.ones((), device=device, dtype=torch.float32)
    return t


class _AbsMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        import ctypes as C
        from . import _lib as L
        if x.device.type != "cuda":
            raise L.EmdError("abs_mean needs a tensor on a ROCm device; there is no CPU path")
        xc = x.detach().contiguous().float()
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        _l1_call(xc.numel(), xc.data_ptr(), None, out, None)
        ctx.save_for_backward(xc)
        ctx.shape = x.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        from . import _lib as L
        (xc,) = ctx.saved_tensors
        grad = torch.empty_like(xc)
        gc = g.detach().reshape(1).float().contiguous()
        L.check(L.load().emd_abs_mean_backward(xc.numel(), xc.data_ptr(), gc.data_ptr(), grad.data_ptr(),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emd_abs_mean_backward")
        return grad.view(ctx.shape)


def abs_mean(x):
    """mean |x| -- the residual regularisers of the fine stage, `torch.mean(torch.abs(ddict[level][key]))` of S3Gaussian/train.py:
    242-310 -- in one launch forward (no |x| tensor) and one backward (sign(x) g / n with the upstream gradient read on the device)
    instead of abs / mean / sign / mul / expand launches over tensors of up to 48 floats per Gaussian."""
    return _AbsMean.apply(x)


class _ResidualPairL1(torch.autograd.Function):
    """(x_a, x_b) -> (x_a, x_b, mean |x_a|, mean |x_b|): the identity on two residual tensors with their L1 means as extra outputs.  The
    backward forms  up_a + sign(x_a) g_a / n  and  up_b + sign(x_b) g_b / n  in ONE pass (`emd_residual_l1_backward`); when both residuals
    received the same upstream tensor -- the rasterizer's dL/dshs for `shs_residuals=[x_a, x_b]` -- it is read once."""

    @staticmethod
    def forward(ctx, xa, xb):
        import ctypes as C
        from . import _lib as L
        if xa.device.type != "cuda":
            raise L.EmdError("residual_pair_l1 needs tensors on a ROCm device; there is no CPU path")
        if xa.shape != xb.shape:
            raise ValueError("residual_pair_l1: the two residuals must have one shape")
        a, b = xa.detach().contiguous().float(), xb.detach().contiguous().float()
        out = torch.empty(2, device=xa.device, dtype=torch.float32)
        _l1_call(a.numel(), a.data_ptr(), None, out, None)
        _l1_call(b.numel(), b.data_ptr(), None, out[1:], None)
        ctx.save_for_backward(a, b)
        return a.view(xa.shape), b.view(xb.shape), out[0], out[1]

    @staticmethod
    def backward(ctx, up_a, up_b, g_a, g_b):
        import ctypes as C
        from . import _lib as L
        a, b = ctx.saved_tensors
        same = up_a is not None and up_b is not None and (up_a is up_b or (up_a.data_ptr() == up_b.data_ptr() and up_a.stride() == up_b.stride()))
        ua = None if up_a is None else up_a.contiguous().float()
        ub = ua if same else (None if up_b is None else up_b.contiguous().float())
        ga = None if g_a is None else g_a.detach().reshape(1).float().contiguous()
        gb = None if g_b is None else g_b.detach().reshape(1).float().contiguous()
        grad_a, grad_b = torch.empty_like(a), torch.empty_like(b)
        L.check(L.load().emd_residual_l1_backward(a.numel(), L.ptr(ua), L.ptr(ub), a.data_ptr(), b.data_ptr(), L.ptr(ga), L.ptr(gb),
                                                  grad_a.data_ptr(), grad_b.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                "emd_residual_l1_backward")
        return grad_a, grad_b


def residual_pair_l1(xa, xb):
    """The two residuals unchanged, and mean |xa|, mean |xb| (the regularisers of S3Gaussian/train.py:238-310) -- see _ResidualPairL1."""
    return _ResidualPairL1.apply(xa, xb)


def residual_abs_mean(level_dict, key):
    """mean |level_dict[key]|: the value the deformation network formed beside the residual when it ran with fused SH residuals
    (`<key>_abs_mean`), else `abs_mean(level_dict[key])`."""
    v = level_dict.get(key + "_abs_mean")
    return v if v is not None else abs_mean(level_dict[key])


def l1_loss(network_output, gt):
    """mean |network_output - gt| (S3Gaussian/utils/loss_utils.py:21-22) and its gradient in one HIP launch."""
    return _L1Loss.apply(network_output.float(), gt.float())
