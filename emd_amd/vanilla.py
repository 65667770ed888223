"""`VanillaGaussians` -- OmniRe's per-class Gaussian store and its density control on the HIP path (the
OmniRe anchor; section 8a row a17).

Mirrors, with the reference's attribute / method names and argument meaning (behaviour restated, not copied):
  parameter layout `_means _scales _quats _opacities _features_dc [N,3] _features_rest [N,K-1,3]`, activations,
  `get_gaussian_param_groups` (class-prefixed group names)            OmniRe/models/gaussians/vanilla.py:29-158,193-204
  `after_train` (running |grad| sums, visibility counts, max screen size)    :163-191
  `postprocess_per_train_step`                                            :150-161
  `refinement_after` = split_gaussians + dup_gaussians + cull_gaussians + opacity reset, and the optimiser surgery
  `dup_in_optim` / `remove_from_optim`                                  :206-376, models/gaussians/basics.py:198-242

WHERE the work runs differs.  The reference grows every parameter and both Adam moments with boolean-mask indexing, `repeat` and `torch.cat`
three times per event (split, duplicate, cull: ~90 launches, a host sync per mask, and the Adam state touched three times); here one event is
`emd_refine_decide` -> prefix sums -> `emd_refine_index` -> ONE `emd_densify_gather` over the six parameters and their twelve moments
(csrc/densify.hip), with a single host read (the new point count).  The reference's semantics that the fused decision reproduces are spelled out
at `EmdRefineArgs` in include/emd_raster.h -- the ones that are easy to get wrong: a split ORIGINAL stays (scale reduced in place), the
duplicate test runs AFTER that reduction (a Gaussian just above the size threshold is split AND duplicated), the cull sees the grown arrays with
`max_2Dsize` 0 on the new rows, and the output order is originals, samples replica 0, replica 1, duplicates.  The split samples are a Philox
draw keyed by (seed, event, source index, replica): every rank of a view-parallel job draws the same samples (`samples=` takes a recorded draw
instead: tests).  There is no CPU path."""
import ctypes as C

import torch
from torch.nn import Parameter

from . import _lib as L

SH_C0 = 0.28209479177387814


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class VanillaGaussians(torch.nn.Module):
    def __init__(self, class_name, ctrl, reg=None, networks=None, scene_scale=30.0, scene_origin=None, num_train_images=300, device="cuda",
                 refine_seed=0, **kwargs):
        super().__init__()
        self.class_prefix = class_name + "#"
        self.ctrl_cfg, self.reg_cfg, self.networks_cfg = ctrl, reg, networks
        self.scene_scale = scene_scale
        self.scene_origin = torch.zeros(3) if scene_origin is None else scene_origin
        self.num_train_images = num_train_images
        self.step = 0
        self.device = torch.device(device)
        self.in_test_set = False
        self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None
        self.filter_mask = None
        self.refine_seed, self.refine_events = int(refine_seed), 0
        z = lambda *s: torch.zeros(*s, device=self.device)
        self._means, self._scales, self._quats, self._opacities = z(1, 3), z(1, 3), z(1, 4), z(1, 1)
        self._features_dc, self._features_rest = z(1, 3), z(1, (self.sh_degree + 1) ** 2 - 1, 3)

    sh_degree = property(lambda self: _get(self.ctrl_cfg, "sh_degree", 3))
    num_points = property(lambda self: self._means.shape[0])
    get_scaling = property(lambda self: torch.exp(self._scales))
    get_opacity = property(lambda self: torch.sigmoid(self._opacities))
    get_quats = property(lambda self: self.quat_act(self._quats))
    shs_0 = property(lambda self: self._features_dc)
    shs_rest = property(lambda self: self._features_rest)

    @property
    def colors(self):
        return self._features_dc * SH_C0 + 0.5 if self.sh_degree > 0 else torch.sigmoid(self._features_dc)

    def quat_act(self, x):
        return x / x.norm(dim=-1, keepdim=True)

    def create_from_tensors(self, means, colors, log_scales, quats=None):
        """create_from_pcd (vanilla.py:78-106) with the kNN-derived initial log-scales and the random unit quaternions supplied by the caller
        (the reference takes them from sklearn and the global RNG): SH dc = RGB2SH(colors), opacity logit(0.1)."""
        dev, n = self.device, means.shape[0]
        P = lambda t: Parameter(t.to(dev).float().contiguous())
        self._means = P(means)
        self._scales = P(log_scales.reshape(n, -1).expand(n, 3))
        if quats is None:
            quats = torch.zeros(n, 4)
            quats[:, 0] = 1.0
        self._quats = P(quats)
        self._features_dc = P((colors.float() - 0.5) / SH_C0 if self.sh_degree > 0 else torch.logit(colors.float(), eps=1e-10))
        self._features_rest = P(torch.zeros(n, (self.sh_degree + 1) ** 2 - 1, 3))
        self._opacities = P(torch.logit(0.1 * torch.ones(n, 1)))

    def preprocess_per_train_step(self, step):
        self.step = step

    def postprocess_per_train_step(self, step, optimizer, radii, xys_grad, last_size):
        self.after_train(radii, xys_grad, last_size)
        if step % _get(self.ctrl_cfg, "refine_interval") == 0:
            self.refinement_after(step, optimizer)

    def get_gaussian_param_groups(self):
        p = self.class_prefix
        return {p + "xyz": [self._means], p + "sh_dc": [self._features_dc], p + "sh_rest": [self._features_rest], p + "opacity": [self._opacities],
                p + "scaling": [self._scales], p + "rotation": [self._quats]}

    get_param_groups = get_gaussian_param_groups

    # ---- a17: the running statistics of the refinement --------------------------------------------------------------------------------------
    def after_train(self, radii, xys_grad, last_size):
        """vanilla.py:163-191 on this class's rows of the rasterizer's outputs: `radii` [n] int, `xys_grad` [n,2] (means2d.grad or .absgrad, already
        scaled to pixels by the trainer, base.py:279-286).  The first call after a refinement starts the sums the way the reference does: the
        norm of EVERY row (visible or not) and a count of ONE everywhere (:175-178); later calls are one launch (no mask indexing, no sync)."""
        with torch.no_grad():
            n = self.num_points
            if self.filter_mask is not None and not bool(self.filter_mask.all()):
                raise NotImplementedError("a partial filter_mask (rows hidden from the rasterizer) is not used by the reference's VanillaGaussians")
            if radii.device.type != "cuda" or xys_grad.device.type != "cuda":
                raise L.EmdError("VanillaGaussians.after_train needs tensors on a ROCm device; there is no CPU path")
            r = radii.reshape(-1).to(torch.int32).contiguous()
            g = xys_grad.reshape(n, -1).float()
            if g.stride(1) != 1:
                g = g.contiguous()
            assert r.numel() == n and g.shape[1] >= 2
            first = self.xys_grad_norm is None
            if first:
                self.xys_grad_norm = g[:, :2].norm(dim=-1) if g.shape[1] > 2 else g.norm(dim=-1)
                self.vis_counts = torch.ones_like(self.xys_grad_norm)
            if self.max_2Dsize is None:
                self.max_2Dsize = torch.zeros(n, device=r.device, dtype=torch.float32)
            L.check(L.load().emd_after_train_stats(n, r.data_ptr(), g.data_ptr(), int(g.stride(0)), None if first else self.xys_grad_norm.data_ptr(),
                                                   None if first else self.vis_counts.data_ptr(), self.max_2Dsize.data_ptr(), float(last_size),
                                                   _stream()), "emd_after_train_stats")

    # ---- the refinement event ------------------------------------------------------------------------------------------------------------
    def refinement_after(self, step, optimizer, samples=None):
        """vanilla.py:206-297.  `samples` [n_split_samples, n_split, 3]: standard normals to use instead of the Philox draw (tests: the reference's
        recorded `torch.randn((samps * n_splits, 3))`, viewed that way).  Returns {"n_before", "n_after", "split" (sources), "originals_kept",
        "samples_kept", "dups_kept"} (None when the event does nothing: before `warmup_steps`, or inside the guard behind an opacity reset)."""
        assert step == self.step
        c = self.ctrl_cfg
        if self.step <= _get(c, "warmup_steps"):
            return None
        info = None
        with torch.no_grad():
            reset_interval = _get(c, "reset_alpha_interval")
            past_reset = self.step % reset_interval > max(self.num_train_images, _get(c, "refine_interval"))
            do_densification = self.step < _get(c, "stop_split_at") and past_reset
            if do_densification:
                assert self.xys_grad_norm is not None and self.vis_counts is not None and self.max_2Dsize is not None
            if do_densification or past_reset:
                info = self._refine(optimizer, do_densification, past_reset, samples)
            if self.step % reset_interval == _get(c, "refine_interval"):
                # opacity reset (vanilla.py:286-297): logit(min(sigmoid(o), reset_alpha_value)), Adam moments of the opacity group restart at zero
                value = torch.min(self.get_opacity.data, torch.ones_like(self._opacities.data) * _get(c, "reset_alpha_value"))
                self._opacities.data = torch.logit(value)
                for group in optimizer.param_groups:
                    if group["name"] == self.class_prefix + "opacity":
                        st = optimizer.state[group["params"][0]]
                        st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(st["exp_avg"]), torch.zeros_like(st["exp_avg_sq"])
            self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None
        return info

    def _refine(self, optimizer, do_densify, do_cull, samples):
        lib, dev, c = L.load(), self._means.device, self.ctrl_cfg
        if dev.type != "cuda":
            raise L.EmdError("VanillaGaussians.refinement_after needs tensors on a ROCm device; there is no CPU path")
        N, ns = self.num_points, int(_get(c, "n_split_samples", 2))
        a = L.EmdRefineArgs()
        a.num_points, a.do_densify, a.do_cull = N, int(do_densify), int(do_cull)
        screen_on = self.step < _get(c, "stop_screen_size_at")
        a.use_split_screen, a.cull_big, a.use_cull_screen = int(screen_on), int(self.step > _get(c, "reset_alpha_interval")), int(screen_on)
        keep = [self._scales.detach().contiguous(), self._opacities.detach().reshape(-1).contiguous()]
        a.scaling, a.opacity = keep[0].data_ptr(), keep[1].data_ptr()
        if self.max_2Dsize is not None:
            keep.append(self.max_2Dsize.reshape(-1).float().contiguous())
            a.max_2Dsize = keep[-1].data_ptr()
        if do_densify:
            keep += [self.xys_grad_norm.reshape(-1).float().contiguous(), self.vis_counts.reshape(-1).float().contiguous()]
            a.grad_norm, a.vis_counts = keep[-2].data_ptr(), keep[-1].data_ptr()
        # the host's products, rounded to float once (the reference compares float tensors with Python scalars)
        a.grad_threshold, a.size_threshold = float(_get(c, "densify_grad_thresh")), float(_get(c, "densify_size_thresh") * self.scene_scale)
        a.split_screen, a.cull_alpha = float(_get(c, "split_screen_size")), float(_get(c, "cull_alpha_thresh"))
        a.cull_size, a.cull_screen = float(_get(c, "cull_scale_thresh") * self.scene_scale), float(_get(c, "cull_screen_size"))
        code = torch.empty(N, dtype=torch.int32, device=dev)
        inc = torch.empty(4, (N + 255) // 256, dtype=torch.int32, device=dev)          # per-block counts, then exclusive block offsets
        totals = torch.empty(4, dtype=torch.int32, device=dev)
        L.check(lib.emd_refine_decide(C.byref(a), code.data_ptr(), inc.data_ptr(), _stream()), "emd_refine_decide")
        L.check(lib.emd_densify_scan(N, 4, inc.data_ptr(), totals.data_ptr(), _stream()), "emd_densify_scan")
        n_keep, n_dup, n_samp, n_split = (int(v) for v in totals.tolist())          # the event's single host read
        M = n_keep + ns * n_samp + n_dup
        info = {"n_before": N, "n_after": M, "split": n_split, "originals_kept": n_keep, "samples_kept": ns * n_samp, "dups_kept": n_dup}
        if M == N and n_keep == N and n_split == 0:
            return info                                             # nothing split, duplicated or culled: every tensor stays as it is
        src = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
        kind = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
        rank = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
        L.check(lib.emd_refine_index(N, M, ns, code.data_ptr(), inc.data_ptr(), totals.data_ptr(), src.data_ptr(), kind.data_ptr(), rank.data_ptr(), _stream()),
                "emd_refine_index")
        g = L.EmdDensifyGather()
        g.num_out, g.mode, g.num_split = M, L.DENSIFY_MODE_REFINE, n_split
        g.src, g.kind, g.split_rank = src.data_ptr(), kind.data_ptr(), rank.data_ptr()
        keep.append(self._quats.detach().contiguous())
        g.scaling, g.rotation = keep[0].data_ptr(), keep[-1].data_ptr()
        g.seed = (self.refine_seed * 0x9E3779B97F4A7C15 + self.refine_events) & 0xFFFFFFFFFFFFFFFF
        if samples is not None:
            samples = samples.to(dev).float().contiguous()
            assert samples.shape == (ns, n_split, 3), (tuple(samples.shape), ns, n_split)
            g.samples = samples.data_ptr()
            keep.append(samples)
        attrs = {"xyz": ("_means", L.DENSIFY_ROLE_XYZ), "sh_dc": ("_features_dc", L.DENSIFY_ROLE_COPY), "sh_rest": ("_features_rest", L.DENSIFY_ROLE_COPY),
                 "opacity": ("_opacities", L.DENSIFY_ROLE_COPY), "scaling": ("_scales", L.DENSIFY_ROLE_SCALING), "rotation": ("_quats", L.DENSIFY_ROLE_COPY)}
        groups = {grp["name"]: grp for grp in (optimizer.param_groups if optimizer is not None else [])
                  if grp["name"].startswith(self.class_prefix) and grp["name"][len(self.class_prefix):] in attrs}
        jobs = []
        for short, (attr, role) in attrs.items():
            p = getattr(self, attr)
            jobs.append((p, role, ("param", short)))
            grp = groups.get(self.class_prefix + short)
            st = optimizer.state.get(grp["params"][0]) if grp is not None else None
            if st:
                jobs.append((st["exp_avg"], L.DENSIFY_ROLE_STATE, ("exp_avg", short)))
                jobs.append((st["exp_avg_sq"], L.DENSIFY_ROLE_STATE, ("exp_avg_sq", short)))
        assert len(jobs) <= L.DENSIFY_MAX_TENSORS
        outs = []
        for k, (t, role, _) in enumerate(jobs):
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            out = torch.empty((M,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev)
            g.tensors[k].src, g.tensors[k].dst, g.tensors[k].width, g.tensors[k].role = t.data_ptr(), out.data_ptr(), t.numel() // N, role
            keep.append(t)
            outs.append(out)
        g.num_tensors = len(jobs)
        L.check(lib.emd_densify_gather(C.byref(g), _stream()), "emd_densify_gather")
        new_params, new_state = {}, {}
        for (t, role, (what, short)), out in zip(jobs, outs):
            if what == "param":
                new_params[short] = Parameter(out)
            else:
                new_state.setdefault(short, {})[what] = out
        for short, (attr, _) in attrs.items():
            grp = groups.get(self.class_prefix + short)
            if grp is not None:
                st = optimizer.state.pop(grp["params"][0], None)
                grp["params"] = [new_params[short]]
                if st:
                    st["exp_avg"], st["exp_avg_sq"] = new_state[short]["exp_avg"], new_state[short]["exp_avg_sq"]
                    optimizer.state[new_params[short]] = st
            setattr(self, attr, new_params[short])
        if do_densify:
            self.refine_events += 1
        return info
