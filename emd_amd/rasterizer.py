"""`GaussianRasterizationSettings` / `GaussianRasterizer` -- the reference's operator surface for the hot path.

Drop-in for `from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer`
(S3Gaussian/gaussian_renderer/__init__.py:14): same 12 settings fields (:49-62), same keyword call
(:145-155), same 6-tuple result `(rendered_image[3,H,W], depth[1,H,W], normal[3,H,W], alpha[1,H,W],
radii[N] int32, extra)`, gradients returned for means3D, means2D (grad sink, NDC-scaled pixel units as
consumed at scene/gaussian_model.py:728-730), shs / colors_precomp, opacities, scales, rotations, cov3Ds_precomp.

Extension (EMD explicit motion fused into the projection kernel): the optional keywords
`actor_ids[N] int32, actor_pose[A,12], residual_dx[N,3], residual_dq[N,4]` make the kernel apply
OmniRe/models/nodes/rigid.py:478-568 (+ deformable.py:57-69) per Gaussian before projecting, and return
gradients for `actor_pose` and the residuals.

State and re-entrancy: options live on the rasterizer INSTANCE (`RasterOptions`, taken from the
process-wide defaults `RasterConfig` when the object is built and never written by library code); everything a call
produces beyond the reference's 6-tuple -- device status word, workspaces, and after the backward pass the absgrad sums,
the SH colour-gradient factor and the gradient slab -- lives in a per-call `RasterCall` record (`rasterizer.last_call`, or
the object passed as `record=`).  Nothing is kept in class attributes, so several rasterizer calls per step (the
reference makes up to nine per render()) and calls on different streams do not see each other.

Host synchronisation: with `no_sync` the forward never reads the duplicate count back, and camera settings that arrive
as DEVICE tensors (as the reference passes them: `.cuda()` at gaussian_renderer/__init__.py:54-59) are handed to the
kernels by pointer (`EmdFwdArgs.settings_dev`) instead of being copied to the host: such a call performs no
device-to-host copy and no stream synchronisation at all.

All compute happens in libemd_raster.so (hand-written HIP for gfx950) through the C ABI of
include/emd_raster.h.  There is no CPU or eager fallback; a missing extension raises.
"""
import ctypes as C
import dataclasses
import warnings
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib as L


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


@dataclasses.dataclass
class RasterOptions:
    """Knobs of the binding that are not part of the reference surface (one copy per rasterizer instance)."""
    compute_normal: bool = True     # composite the normal image (reference returns it; only used for visualisation)
    near_plane: float = 0.2         # view-space z cull of the diff_gauss surface
    no_sync: bool = False           # True: never read the duplicate count back (overflow surfaces lazily, see RasterCall)
    capacity_margin: float = 1.25   # head-room applied to the last observed duplicate count
    min_capacity: int = 1 << 16
    capacity_hint: int = 0          # a binning capacity the caller knows to be large enough ((tile, Gaussian) pairs; e.g. measured over the clip with
                                    # synchronising forwards): calls made with these options never start below it.  0: the binding's own
                                    # per-(device, H, W) cache of the last observed count decides alone
    absgrad: bool = False           # also accumulate sum |d/d mean2D| (RasterCall.absgrad and `means2D.absgrad`, as gsplat does)
    clamp_rgb01: bool = False       # OmniRe colour clamp
    keep_render_grads: bool = False  # tests: keep the per-Gaussian accumulator rows of the render backward (RasterCall.render_grads, [N, 12+])
    factored_sh_grad: bool = False  # view-parallel DP: the backward leaves dL/dshs out and publishes the [N,3] factor instead
                                    # (RasterCall.sh_color_grad); emd_amd.dp rebuilds the dense, view-averaged gradient
    aux_stream: bool = False        # run the colour half of the projection kernel on a second stream beside the binning stage (the binding
                                    # keeps one side stream per device; results are identical, see EmdFwdArgs.aux_stream).  Measured at the
                                    # headline size: 684 it/s against 700 fused -- off unless a caller's binning is long
    keep_all_pairs: bool = False    # enumerate every tile of upstream's tile rectangles (EMD_FLAG_KEEP_ALL_PAIRS): the sorted keys are then
                                    # upstream's, entry for entry.  Default: only the tiles a Gaussian's alpha >= 1/255 bounding box reaches
                                    # -- same images and gradients bit for bit, 25 % fewer list entries on street scenes
    wide_depth_sort: bool = False   # always use the four-pass depth sort (depths beyond 65 536 x the near plane).  The three-pass sort
                                    # falls back to it by itself when the status word can be read (a retry in the synchronous mode,
                                    # the next call in no_sync mode); a call captured into a hipGraph reads nothing back and keeps the
                                    # flags it was captured with, so scenes with such depths must set this for captured steps (or
                                    # check bit 1 of RasterCall.status after replays: an affected frame renders as background)

    def replace(self, **kw):
        return dataclasses.replace(self, **{k: v for k, v in kw.items() if v is not None})


# process-wide DEFAULTS: read once when a GaussianRasterizer is constructed, never written by the library
RasterConfig = RasterOptions()


class RasterCall:
    """Everything one forward (+ backward) call leaves behind besides the reference's outputs.
    `on_backward` (optional callable) is invoked with the record by the backward pass right after its kernels have been
    enqueued: view-parallel training starts its gradient collectives there (emd_amd.dp.GradientExchange.start).
    `on_sh_factor` (optional callable; factored_sh_grad only) is invoked BETWEEN the two halves of the backward: the render backward and the
    extraction of the SH colour factor (`sh_color_grad`) are enqueued, the projection backward is not yet -- collectives started there
    (GradientExchange.start_factors) run under the projection backward instead of behind it.
    `status_buffer` (optional, set by the caller before the call): an int32[4] device tensor that receives the status words instead
    of a fresh allocation -- a fixed address for callers that replay the call from a hipGraph and read the words on the device.
    `pair_stats` (diagnostic; an int64[4] device tensor set by the caller before backward()): the render backward adds the number of
    (pixel, list entry) pairs it evaluated, the number that contributed, the accumulator rows it sent to memory as float atomics and the
    float atomics issued (EmdBwdArgs.pair_stats).
    `loop_stats` (diagnostic; an int64[6] device tensor set by the caller before the call): the compositing kernel adds the trip counts of its scan /
    cull / drain loops (EmdFwdArgs.loop_stats; profiles/render_loop_trips.py)."""
    __slots__ = ("status", "num_rendered", "num_visible", "geom_ws", "bin_ws", "img_ws", "sizes", "capacity", "N", "H", "W",
                 "flags", "settings_dev", "absgrad", "sh_color_grad", "grad_slab", "on_backward", "render_grads", "pair_stats",
                 "status_buffer", "slab_inputs", "on_sh_factor", "radii", "loop_stats")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)

    def last_status(self):
        """(num_rendered D, overflow, num_visible V) of this call's forward; synchronises.  D is the length of the sorted (tile, Gaussian)
        list: upstream's duplicate count with keep_all_pairs, otherwise without the pairs outside the Gaussians' alpha >= 1/255 boxes."""
        st = self.status.cpu().tolist()
        return dict(num_rendered=st[0] & 0xFFFFFFFF, overflow=st[1], num_visible=st[2] & 0xFFFFFFFF)

    def export_binning(self, with_masks=False):
        """Sorted keys (uint64 as int64 bit pattern), Gaussian ids and tile ranges of this call's forward (+ the quadrant mask of every
        list entry with `with_masks`)."""
        lib = L.load()
        st = self.last_status()
        D = 0 if st["overflow"] else st["num_rendered"]
        dev = self.status.device
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        keys = torch.empty(max(D, 1), device=dev, dtype=torch.int64)
        ids = torch.empty(max(D, 1), device=dev, dtype=torch.int32)
        masks = torch.empty(max(D, 1), device=dev, dtype=torch.int32) if with_masks else None
        ranges = torch.empty(T, 2, device=dev, dtype=torch.int32)
        d = L.EmdDims(self.N, self.H, self.W, self.capacity, self.flags)
        L.check(lib.emd_raster_export_binning(C.byref(d), self.geom_ws.data_ptr(), self.sizes[0], self.bin_ws.data_ptr(), self.sizes[1], D,
                                              keys.data_ptr(), ids.data_ptr(), ranges.data_ptr(), L.ptr(masks), _stream()),
                "emd_raster_export_binning")
        return (keys[:D], ids[:D], ranges, masks[:D]) if with_masks else (keys[:D], ids[:D], ranges)

    def export_geometry(self):
        lib = L.load()
        dev, N = self.status.device, self.N
        e = lambda *s, dt=torch.float32: torch.empty(*s, device=dev, dtype=dt)
        out = dict(means2D=e(N, 2), depths=e(N), conic_opacity=e(N, 4), rgb=e(N, 3),
                   normal=e(N, 3) if self.flags & L.FLAG_NORMAL else None, tiles_touched=e(N, dt=torch.int32))
        d = L.EmdDims(N, self.H, self.W, self.capacity, self.flags)
        L.check(lib.emd_raster_export_geometry(C.byref(d), self.geom_ws.data_ptr(), self.sizes[0], out["means2D"].data_ptr(),
                                               out["depths"].data_ptr(), out["conic_opacity"].data_ptr(), out["rgb"].data_ptr(),
                                               L.ptr(out["normal"]), out["tiles_touched"].data_ptr(), _stream()),
                "emd_raster_export_geometry")
        return out


def prepare_backward_workspace(device, num_gaussians, num_extra=0, stream=None):
    """Seed the kept-clean accumulator rows of the render backward for `stream` (default: the current one) with an eagerly zero-filled
    buffer.  Steps recorded into hipGraphs on that stream then share it (every backward hands it back clean) instead of clearing a
    buffer of their own inside each graph.  emd_amd.StepGraphs calls this for its capture stream."""
    dev = torch.device(device)
    st = torch.cuda.current_stream(dev) if stream is None else stream
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), st.cuda_stream, max(int(num_gaussians), 1) * (L.BWD_STRIDE + 4 * int(num_extra)))
    if key not in _clean_ws:
        with torch.cuda.stream(st):
            _clean_ws[key] = torch.zeros(key[2], device=dev, dtype=torch.float32)
    return key


# ---- binning-capacity hints ------------------------------------------------------------------------------------------------
# A cache, not state: (device, H, W) -> a capacity that was large enough last time.  In no_sync mode the duplicate count of a
# forward is never awaited; its status word is copied to pinned host memory asynchronously and looked at by a LATER forward
# (when the copy's event has completed), which grows the hint and warns if a past call overflowed.
_capacity_hint = {}
_aux_streams = {}             # device index -> the side stream the forward forks the colour half of K1 onto
_clean_ws = {}                # (device, stream, floats) -> accumulator rows of the render backward, zero between backward passes
_wide_depth = set()           # keys whose depth range needs the four-pass sort (EMD_ERR_DEPTH_RANGE seen once)
_watch = {}                   # key -> ring of pinned status copies in flight
_WATCH_SLOTS = 32


class _WatchRing:
    """A fixed ring of pinned 16-byte buffers + events, allocated ONCE per (device, H, W): the steady state of a no_sync training
    loop allocates nothing (a fresh pinned allocation per call costs a hipHostMalloc whenever the host runs ahead of the GPU)."""

    def __init__(self):
        self.buf = torch.empty(_WATCH_SLOTS, 4, dtype=torch.int32, pin_memory=True)
        self.events = [torch.cuda.Event() for _ in range(_WATCH_SLOTS)]
        self.pending = [False] * _WATCH_SLOTS
        self.caps = [0] * _WATCH_SLOTS
        self.next = 0


def _poll_pending(key, opts):
    ring = _watch.get(key)
    if ring is None:
        return
    for i in range(_WATCH_SLOTS):
        if not ring.pending[i] or not ring.events[i].query():
            continue
        ring.pending[i] = False
        d, overflow, cap = int(ring.buf[i, 0]) & 0xFFFFFFFF, int(ring.buf[i, 1]), ring.caps[i]
        need = int(d * opts.capacity_margin) + 1024
        if overflow & 2:
            warnings.warn("emd_amd: an earlier no_sync rasterizer call saw a visible Gaussian beyond 65 536 x the near plane (three-pass depth "
                          "sort): that image was blank.  This camera size now uses the four-pass sort.", RuntimeWarning, stacklevel=3)
            _wide_depth.add(key)
            overflow &= ~2
        if overflow:
            warnings.warn(f"emd_amd: an earlier no_sync rasterizer call overflowed its binning workspace ({d} (tile, Gaussian) pairs, "
                          f"capacity {cap}): that image was blank and its gradients zero.  The capacity hint has been raised to {need}.",
                          RuntimeWarning, stacklevel=3)
        if overflow or need > _capacity_hint.get(key, 0):
            _capacity_hint[key] = max(need, _capacity_hint.get(key, 0))


def _watch_status(key, status, capacity):
    """Copy this call's status word to pinned memory asynchronously; a LATER forward looks at it (never waited for).  When all
    ring slots are still in flight (host far ahead of the GPU) the call is simply not watched: the next one will be."""
    ring = _watch.get(key)
    if ring is None:
        ring = _watch[key] = _WatchRing()
    i = ring.next
    if ring.pending[i]:
        return
    ring.buf[i].copy_(status, non_blocking=True)
    ring.events[i].record()
    ring.pending[i], ring.caps[i] = True, capacity
    ring.next = (i + 1) % _WATCH_SLOTS


def _settings_values(rs):
    """-> (bg[3], viewmatrix[16], projmatrix[16], campos[3] as python floats or None, device block or None).

    CPU tensors / sequences are passed to the library by value.  If any of the four is a device tensor (the reference's call
    site), all four are packed into ONE 38-float device block with a single small launch and passed by pointer: no
    device-to-host copy, no synchronisation (EmdFwdArgs.settings_dev)."""
    parts, dev = [], None
    for t, n in ((rs.bg, 3), (rs.viewmatrix, 16), (rs.projmatrix, 16), (rs.campos, 3)):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t, dtype=torch.float32)
        t = t.detach().reshape(-1).to(torch.float32)
        if t.numel() != n:
            raise ValueError(f"expected {n} values, got {t.numel()}")
        if t.device.type != "cpu":
            dev = t.device
        parts.append(t)
    tan_dev = isinstance(rs.tanfovx, torch.Tensor) and rs.tanfovx.device.type != "cpu"
    if dev is not None and all(p.device == dev and p.is_contiguous() for p in parts):
        # the four (and, when they are device tensors too, tanfovx / tanfovy right behind them) already sit back to back in ONE device buffer
        # in the block's order (a caller that keeps its cameras packed on the device: emd_amd.StepInputs, bench.py): hand that memory over
        # as it is -- no concat launch
        seq = list(parts)
        if tan_dev:
            seq += [rs.tanfovx.detach().reshape(-1), rs.tanfovy.detach().reshape(-1)]
        base, off, packed = parts[0].data_ptr(), 0, all(q.dtype == torch.float32 and q.device == dev and q.is_contiguous() for q in seq)
        for q in seq:
            packed = packed and q.data_ptr() == base + 4 * off
            off += q.numel()
        if packed and parts[0].untyped_storage().nbytes() - 4 * parts[0].storage_offset() >= 4 * off:
            return None, parts[0].as_strided((off,), (1,))
    if dev is None and not tan_dev:
        flat = torch.cat(parts).tolist()
        return (flat[0:3], flat[3:19], flat[19:35], flat[35:38]), None
    if tan_dev:          # cameras given as device-resident intrinsics (emd_amd.gsplat_api): tan(fov / 2) stays on the device as well
        dev = rs.tanfovx.device
        parts += [rs.tanfovx.detach().reshape(1).float(), rs.tanfovy.detach().reshape(1).float()]
    return None, torch.cat([p.to(dev, non_blocking=True) for p in parts]).contiguous()


def make_c_settings(rs: GaussianRasterizationSettings, near_plane=0.2):
    """-> (EmdSettings by value, device block or None)."""
    s = L.EmdSettings()
    host, dev_block = _settings_values(rs)
    s.image_height, s.image_width = int(rs.image_height), int(rs.image_width)
    if dev_block is None or dev_block.numel() == L.SETTINGS_DEV_FLOATS:
        s.tanfovx, s.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
    s.scale_modifier = float(rs.scale_modifier)
    s.sh_degree = int(rs.sh_degree)
    if host is not None:
        s.bg[:], s.viewmatrix[:], s.projmatrix[:], s.campos[:] = host
    s.prefiltered = int(bool(rs.prefiltered))
    s.debug = int(bool(rs.debug))
    s.near_plane = float(near_plane)
    return s, dev_block


def _f32c(t, name, shape_tail=None):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected [N, {', '.join(map(str, shape_tail))}]")
    return t.contiguous()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _fill_motion(m: L.EmdMotion, actor_ids, actor_pose, residual_dx, residual_dq):
    m.actor_id = L.ptr(actor_ids)
    m.actor_pose = L.ptr(actor_pose)
    m.num_actors = 0 if actor_pose is None else int(actor_pose.shape[0])
    m.residual_dx = L.ptr(residual_dx)
    m.residual_dq = L.ptr(residual_dq)


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, actor_pose,
                residual_dx, residual_dq, actor_ids, raster_settings, flags, opts, rec, extra0=None, extra1=None, shs_res0=None, shs_res1=None):
        lib = L.load()
        dev = means3D.device
        if dev.type != "cuda":
            raise L.EmdError("GaussianRasterizer needs tensors on a ROCm device (cuda:N); there is no CPU path")
        N = means3D.shape[0]
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        cs, sdev = make_c_settings(raster_settings, opts.near_plane)
        if sdev is not None and sdev.numel() == L.SETTINGS_DEV_FLOATS + 2:
            flags |= L.FLAG_SDEV_TANFOV
        M = 0 if shs is None else int(shs.shape[1])

        out_color = torch.empty(3, H, W, device=dev, dtype=torch.float32)
        out_depth = torch.empty(1, H, W, device=dev, dtype=torch.float32)
        out_alpha = torch.empty(1, H, W, device=dev, dtype=torch.float32)
        out_normal = torch.empty(3, H, W, device=dev, dtype=torch.float32) if flags & L.FLAG_NORMAL else \
            torch.zeros(3, H, W, device=dev, dtype=torch.float32)
        radii = torch.empty(N, device=dev, dtype=torch.int32)
        status = rec.status_buffer if rec.status_buffer is not None else torch.empty(4, device=dev, dtype=torch.int32)
        if status.dtype != torch.int32 or status.numel() != 4 or status.device != dev or not status.is_contiguous():
            raise ValueError("RasterCall.status_buffer must be a contiguous int32[4] tensor on the call's device")
        extras = [e for e in (extra0, extra1) if e is not None]
        out_extra = [torch.empty(3, H, W, device=dev, dtype=torch.float32) for _ in extras]

        key = (dev.index, H, W)
        capturing = torch.cuda.is_current_stream_capturing()       # hipGraph capture: nothing host-side may depend on this call's results
        if capturing and not opts.no_sync:
            raise L.EmdError("a rasterizer call captured into a hipGraph must be built with no_sync=True (the duplicate count cannot be read back)")
        if not capturing:
            _poll_pending(key, opts)
        if key in _wide_depth or opts.wide_depth_sort:
            flags |= L.FLAG_WIDE_DEPTH_SORT
        capacity = max(int(_capacity_hint.get(key, 0)), opts.min_capacity, int(opts.capacity_hint),
                       4 * N if (key not in _capacity_hint and not opts.capacity_hint) else 0)
        a = L.EmdFwdArgs()
        while True:
            gb, bb, ib, _ = L.workspace_sizes(N, H, W, capacity, flags, len(extras))
            geom_ws = torch.empty(gb, device=dev, dtype=torch.uint8)
            bin_ws = torch.empty(bb, device=dev, dtype=torch.uint8)
            img_ws = torch.empty(ib, device=dev, dtype=torch.uint8)
            a.s = cs
            a.settings_dev = L.ptr(sdev)
            a.num_gaussians, a.sh_coeffs, a.flags, a.bin_capacity = N, M, flags, capacity
            a.means3D, a.shs, a.colors_precomp = L.ptr(means3D), L.ptr(shs), L.ptr(colors_precomp)
            a.shs_residual[0], a.shs_residual[1] = L.ptr(shs_res0), L.ptr(shs_res1)
            a.loop_stats = L.ptr(rec.loop_stats)         # diagnostic: an int64[6] device tensor set on the record before the call, or None
            a.opacities, a.scales, a.rotations = L.ptr(opacities), L.ptr(scales), L.ptr(rotations)
            a.cov3D_precomp = L.ptr(cov3Ds_precomp)
            _fill_motion(a.motion, actor_ids, actor_pose, residual_dx, residual_dq)
            a.out_color, a.out_depth, a.out_alpha = out_color.data_ptr(), out_depth.data_ptr(), out_alpha.data_ptr()
            a.out_normal = out_normal.data_ptr()
            a.radii = radii.data_ptr()
            a.geom_ws, a.geom_bytes = geom_ws.data_ptr(), gb
            a.bin_ws, a.bin_bytes = bin_ws.data_ptr(), bb
            a.img_ws, a.img_bytes = img_ws.data_ptr(), ib
            a.status = status.data_ptr()
            a.num_extra = len(extras)
            for k, (e, o) in enumerate(zip(extras, out_extra)):
                a.colors_extra[k], a.out_extra[k] = e.data_ptr(), o.data_ptr()
            a.aux_stream = None
            if opts.aux_stream:
                aux = _aux_streams.get(dev.index)
                if aux is None:
                    aux = _aux_streams[dev.index] = torch.cuda.Stream(device=dev)
                if aux.cuda_stream != torch.cuda.current_stream().cuda_stream:
                    a.aux_stream = aux.cuda_stream
            rc = lib.emd_raster_forward(C.byref(a), _stream())
            if rc == L.EMD_ERR_DEPTH_RANGE and not (flags & L.FLAG_WIDE_DEPTH_SORT):
                _wide_depth.add(key)
                flags |= L.FLAG_WIDE_DEPTH_SORT
                continue
            if rc == L.EMD_ERR_CAPACITY:
                capacity = int(a.num_rendered * opts.capacity_margin) + 1024
                continue
            L.check(rc, "emd_raster_forward")
            break
        if a.num_rendered >= 0:
            _capacity_hint[key] = max(int(a.num_rendered * opts.capacity_margin) + 1024, opts.min_capacity)
        else:
            _capacity_hint[key] = max(capacity, _capacity_hint.get(key, 0))
            if not capturing:
                _watch_status(key, status, capacity)       # looked at by a later forward; never waited for

        ctx.cs, ctx.flags, ctx.capacity, ctx.N, ctx.M = cs, flags, capacity, N, M
        ctx.num_rendered = int(a.num_rendered)
        ctx.sizes = (gb, bb, ib)
        ctx.opts, ctx.rec = opts, rec
        ctx.means2D_ref = means2D if (flags & L.FLAG_ABSGRAD) else None      # gsplat convention: `.absgrad` is set on this tensor
        ctx.has_shs_res = (shs_res0 is not None, shs_res1 is not None)
        ctx.has = (shs is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None,
                   actor_pose is not None, residual_dx is not None, residual_dq is not None)
        ctx.num_extra = len(extras)
        ctx.save_for_backward(means3D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, actor_pose,
                              residual_dx, residual_dq, actor_ids, radii, geom_ws, bin_ws, img_ws, status, out_color,
                              out_depth, out_normal, sdev, *extras, *out_extra)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # unused outputs (normal, depth, alpha) arrive as None, not as zero images
        rec.status, rec.num_rendered, rec.num_visible = status, int(a.num_rendered), int(a.num_visible)
        rec.radii = radii                       # (int32 [N]: > 0 = visible in this view; the compacted gradient exchange packs by it)
        rec.geom_ws, rec.bin_ws, rec.img_ws, rec.sizes, rec.capacity = geom_ws, bin_ws, img_ws, (gb, bb, ib), capacity
        rec.N, rec.H, rec.W, rec.flags, rec.settings_dev = N, H, W, flags, sdev
        return (out_color, out_depth, out_normal, out_alpha, radii, *out_extra)

    @staticmethod
    def backward(ctx, g_color, g_depth, g_normal, g_alpha, _g_radii, *g_extra):
        lib = L.load()
        nx = ctx.num_extra
        saved = ctx.saved_tensors
        (means3D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, actor_pose, residual_dx,
         residual_dq, actor_ids, radii, geom_ws, bin_ws, img_ws, status, out_color, out_depth,
         out_normal, sdev) = saved[:20]
        extras, out_extra = saved[20:20 + nx], saved[20 + nx:20 + 2 * nx]
        dev = means3D.device
        N, M, flags, opts, rec = ctx.N, ctx.M, ctx.flags, ctx.opts, ctx.rec
        has_shs, has_col, has_sr, has_cov, has_pose, has_rdx, has_rdq = ctx.has
        z = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        g_color = None if g_color is None else g_color.contiguous().float()
        g_depth = None if g_depth is None else g_depth.contiguous().float()
        g_alpha = None if g_alpha is None else g_alpha.contiguous().float()
        g_normal = None if (g_normal is None or not (flags & L.FLAG_NORMAL)) else g_normal.contiguous().float()
        d_means2D = z(N, 3)
        # the four small per-Gaussian gradients of the usual call (means, scales, rotations, opacities: 44 B per Gaussian) are carved
        # from ONE buffer: view-parallel training all-reduces that slab with a single collective (dp.exchange_sh_gradient)
        slab = None
        if has_sr and opacities.numel() == N:
            slab = z(N * 11)
            d_means3D, d_sc_, d_rot_, d_op_ = slab[:3 * N].view(N, 3), slab[3 * N:6 * N].view(N, 3), slab[6 * N:10 * N].view(N, 4), \
                slab[10 * N:].view(opacities.shape)
        else:
            d_means3D = z(N, 3)
        factored = has_shs and opts.factored_sh_grad
        d_shs = z(N, M, 3) if (has_shs and not factored) else None
        d_shc = z(N, 3) if factored else None
        d_col = z(N, 3) if has_col else None
        d_op = d_op_ if slab is not None else z(*opacities.shape)
        d_sc = (d_sc_ if slab is not None else z(N, 3)) if has_sr else None
        d_rot = (d_rot_ if slab is not None else z(N, 4)) if has_sr else None
        d_cov = z(N, 6) if has_cov else None
        d_pose = z(*actor_pose.shape) if has_pose else None
        d_rdx = z(N, 3) if has_rdx else None
        d_rdq = z(N, 4) if has_rdq else None
        d_abs = z(N, 2) if flags & L.FLAG_ABSGRAD else None
        # accumulator rows of the render backward.  Kept across calls per (device, stream, size): the projection backward hands every
        # row it reads back zeroed (EMD_FLAG_BWD_WS_CLEAN), so only the first use pays the 48 N-byte zero fill (20 us per step at 2 M).
        # keep_render_grads (tests read the rows afterwards) takes a fresh buffer the library clears itself.
        ws_key, bflags = None, flags
        capturing = torch.cuda.is_current_stream_capturing()
        if opts.keep_render_grads:
            bwd_ws = torch.empty(max(N, 1) * (L.BWD_STRIDE + 4 * nx), device=dev, dtype=torch.float32)
        else:
            ws_key = (dev.index, torch.cuda.current_stream().cuda_stream, max(N, 1) * (L.BWD_STRIDE + 4 * nx))
            bwd_ws = _clean_ws.pop(ws_key, None)          # (popped: a failed backward must not leave a dirty buffer behind)
            if bwd_ws is not None:
                bflags = flags | L.FLAG_BWD_WS_CLEAN
            elif capturing:
                # No kept buffer for this stream and a hipGraph is being recorded: a zero fill issued now would become a node of THIS graph
                # only (it does not run at capture time), and a buffer from the graph's pool handed to later captures as "clean" would be
                # memory nobody ever cleared.  So: a fresh buffer the library clears itself inside every graph, and nothing is kept.
                # (`prepare_backward_workspace`, or any eager step on the capture stream, seeds a kept buffer before recording.)
                bwd_ws = torch.empty(ws_key[2], device=dev, dtype=torch.float32)
                ws_key = None
            else:
                bwd_ws = torch.zeros(ws_key[2], device=dev, dtype=torch.float32)
                bflags = flags | L.FLAG_BWD_WS_CLEAN
        g_extra = [None if g is None else g.contiguous().float() for g in g_extra]
        d_extra = [z(N, 3) for _ in range(nx)]

        b = L.EmdBwdArgs()
        b.s = ctx.cs
        b.settings_dev = L.ptr(sdev)
        b.num_gaussians, b.sh_coeffs, b.flags, b.bin_capacity, b.num_rendered = N, M, bflags, ctx.capacity, ctx.num_rendered
        b.means3D, b.shs, b.colors_precomp = L.ptr(means3D), L.ptr(shs), L.ptr(colors_precomp)
        b.opacities, b.scales, b.rotations, b.cov3D_precomp = L.ptr(opacities), L.ptr(scales), L.ptr(rotations), L.ptr(cov3Ds_precomp)
        _fill_motion(b.motion, actor_ids, actor_pose, residual_dx, residual_dq)
        b.radii = radii.data_ptr()
        b.geom_ws, b.geom_bytes = geom_ws.data_ptr(), ctx.sizes[0]
        b.bin_ws, b.bin_bytes = bin_ws.data_ptr(), ctx.sizes[1]
        b.img_ws, b.img_bytes = img_ws.data_ptr(), ctx.sizes[2]
        b.status = status.data_ptr()
        b.out_color, b.out_depth = out_color.data_ptr(), out_depth.data_ptr()
        b.out_normal = out_normal.data_ptr() if flags & L.FLAG_NORMAL else None
        b.dL_dcolor, b.dL_ddepth, b.dL_dalpha, b.dL_dnormal = L.ptr(g_color), L.ptr(g_depth), L.ptr(g_alpha), L.ptr(g_normal)
        b.bwd_ws, b.bwd_bytes = bwd_ws.data_ptr(), bwd_ws.numel() * 4
        b.dL_dmeans3D, b.dL_dmeans2D, b.dL_dmeans2D_abs = d_means3D.data_ptr(), d_means2D.data_ptr(), L.ptr(d_abs)
        b.dL_dshs, b.dL_dcolors, b.dL_dopacities = L.ptr(d_shs), L.ptr(d_col), d_op.data_ptr()
        b.dL_dscales, b.dL_drotations, b.dL_dcov3D = L.ptr(d_sc), L.ptr(d_rot), L.ptr(d_cov)
        b.dL_dactor_pose, b.dL_dresidual_dx, b.dL_dresidual_dq = L.ptr(d_pose), L.ptr(d_rdx), L.ptr(d_rdq)
        b.dL_dsh_color = L.ptr(d_shc)
        b.pair_stats = L.ptr(rec.pair_stats)         # diagnostic: an int64[4] device tensor set on the record before backward(), or None
        b.num_extra = nx
        for k in range(nx):
            b.colors_extra[k], b.out_extra[k] = extras[k].data_ptr(), out_extra[k].data_ptr()
            b.dL_dextra[k], b.dL_dcolors_extra[k] = L.ptr(g_extra[k]), d_extra[k].data_ptr()
        if factored and rec.on_sh_factor is not None:
            # two calls: render backward + the SH factor, the caller's hook (collectives of the factors on the communication stream), then the
            # projection backward, under which they run
            b.flags = bflags | L.FLAG_BWD_RENDER_ONLY
            L.check(lib.emd_raster_backward(C.byref(b), _stream()), "emd_raster_backward (render half)")
            rec.sh_color_grad = d_shc
            rec.on_sh_factor(rec)
            b.flags, b.dL_dsh_color = bflags | L.FLAG_BWD_PROJECT_ONLY, None
            L.check(lib.emd_raster_backward(C.byref(b), _stream()), "emd_raster_backward (projection half)")
        else:
            L.check(lib.emd_raster_backward(C.byref(b), _stream()), "emd_raster_backward")
        if ws_key is not None:
            while len(_clean_ws) >= 4:                    # a few sizes at most (the point count changes at densification events)
                _clean_ws.pop(next(iter(_clean_ws)))
            _clean_ws[ws_key] = bwd_ws
        rec.absgrad, rec.sh_color_grad, rec.grad_slab = d_abs, d_shc, slab
        rec.render_grads = bwd_ws.view(max(N, 1), -1) if opts.keep_render_grads else None
        if d_abs is not None and ctx.means2D_ref is not None:
            ctx.means2D_ref.absgrad = d_abs          # what gsplat's backward does with `means2d.absgrad`
        if rec.on_backward is not None:
            rec.on_backward(rec)
        d_x = d_extra + [None] * (2 - nx)
        # (the residuals of the SH coefficients enter as a sum: each receives dL/dshs itself -- the same memory, no copy; as VIEWS, because
        #  autograd's AccumulateGrad clones a gradient whose tensor object is referenced more than once instead of adopting it: handing the
        #  one object to three inputs cost the leaf `shs` a 384 MB copy per step)
        d_r0 = d_r1 = None
        if d_shs is not None and any(ctx.has_shs_res):
            # a second tensor object over the same storage WITHOUT a base relation (a .view() keeps its base referenced): the leaf's
            # gradient object then has a single owner and is adopted, not cloned
            alias = torch.empty(0, device=d_shs.device, dtype=d_shs.dtype).set_(d_shs.untyped_storage(), d_shs.storage_offset(), d_shs.shape, d_shs.stride())
            d_r0, d_r1 = (alias if ctx.has_shs_res[0] else None), (alias if ctx.has_shs_res[1] else None)
        return (d_means3D, d_means2D, d_shs, d_col, d_op, d_sc, d_rot, d_cov, d_pose, d_rdx, d_rdq, None, None, None, None, None, d_x[0], d_x[1],
                d_r0, d_r1)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings, options: Optional[RasterOptions] = None, **overrides):
        """`options` (default: a snapshot of the process-wide `RasterConfig`) and keyword overrides of single fields
        (compute_normal=, no_sync=, absgrad=, factored_sh_grad=, near_plane=, clamp_rgb01=, ...) belong to this instance."""
        super().__init__()
        self.raster_settings = raster_settings
        self.options = dataclasses.replace(options if options is not None else RasterConfig).replace(**overrides)
        self.last_call: Optional[RasterCall] = None

    def markVisible(self, positions):
        """Frustum test of the diff_gauss surface: view-space z > near plane."""
        with torch.no_grad():
            V = torch.as_tensor(self.raster_settings.viewmatrix, dtype=torch.float32, device=positions.device)
            z = positions @ V[:3, 2] + V[3, 2]
            return z > self.options.near_plane

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3Ds_precomp=None, extra_attrs=None, actor_ids: Optional[torch.Tensor] = None,
                actor_pose: Optional[torch.Tensor] = None, residual_dx: Optional[torch.Tensor] = None,
                residual_dq: Optional[torch.Tensor] = None, raw_params: bool = False, record: Optional[RasterCall] = None,
                colors_extra=None, shs_residuals=None):
        """`colors_extra`: up to two more colour sets [N,3] composited by the SAME call -- the reference's feature passes
        (`colors_precomp = ddict["coarse"/"fine"]["feat"]`, gaussian_renderer/__init__.py:170-201) without their second and third
        projection, sort and list walk.  Their images [3,H,W] are returned as a list in the sixth slot of the result (the
        reference's `extra`); each equals the colour image of a separate call with that colour set, bit for bit.
        `shs_residuals`: up to two tensors shaped like `shs`; the colours are evaluated on (shs + r0) + r1, formed inside the projection
        kernel for the visible Gaussians only -- the fine stage's `shs + dshs_coarse + dshs_fine` (S3Gaussian/scene/deformation.py:468-481)
        without its two passes over [N,16,3]; same image bit for bit, and each term receives dL/dshs."""
        rs, opts = self.raster_settings, self.options
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3Ds_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3Ds_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if extra_attrs is not None:
            raise NotImplementedError("extra_attrs: the reference always passes None "
                                      "(S3Gaussian/gaussian_renderer/__init__.py:154); render them with colors_precomp")
        N = means3D.shape[0]
        means3D = _f32c(means3D, "means3D", (3,))
        shs = _f32c(shs, "shs")
        if shs is not None and (shs.dim() != 3 or shs.shape[2] != 3):
            raise ValueError(f"shs must be [N, K, 3], got {tuple(shs.shape)}")
        colors_precomp = _f32c(colors_precomp, "colors_precomp", (3,))
        opacities = _f32c(opacities, "opacities")
        if opacities.numel() != N:
            raise ValueError(f"opacities must hold N={N} values, got {tuple(opacities.shape)}")
        scales = _f32c(scales, "scales", (3,))
        rotations = _f32c(rotations, "rotations", (4,))
        cov3Ds_precomp = _f32c(cov3Ds_precomp, "cov3Ds_precomp", (6,))
        flags = 0
        if opts.compute_normal:
            flags |= L.FLAG_NORMAL
        if opts.no_sync:
            flags |= L.FLAG_NO_SYNC
        if opts.absgrad:
            flags |= L.FLAG_ABSGRAD
        if opts.clamp_rgb01:
            flags |= L.FLAG_CLAMP_RGB01
        if opts.keep_all_pairs:
            flags |= L.FLAG_KEEP_ALL_PAIRS
        if raw_params:
            # scales / rotations / opacities are the raw parameters; exp / normalize / sigmoid
            # (S3Gaussian/gaussian_renderer/__init__.py:99-101) run inside the projection kernel, forward and backward
            if cov3Ds_precomp is not None:
                raise ValueError("raw_params needs scales / rotations, not cov3Ds_precomp")
            flags |= L.FLAG_RAW_PARAMS
        if actor_ids is not None or residual_dx is not None or residual_dq is not None:
            flags |= L.FLAG_MOTION
            if actor_ids is not None:
                if actor_pose is None or actor_pose.dim() != 2 or actor_pose.shape[1] != L.ACTOR_STRIDE:
                    raise ValueError("actor_pose must be [A, 12] = (q_mean[4], trans[3], valid, q_rot[4])")
                actor_ids = actor_ids.to(torch.int32).contiguous()
                actor_pose = _f32c(actor_pose, "actor_pose")
            residual_dx = _f32c(residual_dx, "residual_dx", (3,))
            residual_dq = _f32c(residual_dq, "residual_dq", (4,))
        rec = record if record is not None else RasterCall()
        if opts.factored_sh_grad and shs is not None and shs.requires_grad and not shs.is_leaf:
            # the factored backward returns no dL/dshs at all (the dense gradient is rebuilt from the exchanged factors and assigned
            # to the LEAF parameter by dp.GradientExchange.finish): anything upstream of a non-leaf `shs` would silently get no gradient
            raise ValueError("factored_sh_grad needs `shs` to be the leaf SH parameter; with a network in front of it (shs = features + dshs) "
                             "use the dense gradient (factored_sh_grad=False) and dp.allreduce_gradients")
        # what the gradient slab of the backward will be carved for (dp.GradientExchange decides from these whether it may reduce the
        # slab in place while autograd is still running)
        rec.slab_inputs = (means3D, scales, rotations, opacities)
        extras = [] if colors_extra is None else [_f32c(e, "colors_extra", (3,)) for e in colors_extra]
        if len(extras) > L.MAX_EXTRA:
            raise ValueError(f"at most {L.MAX_EXTRA} extra colour sets per call")
        if any(e.shape[0] != N for e in extras):
            raise ValueError("colors_extra must hold one colour per Gaussian")
        extras += [None] * (2 - len(extras))
        res = [] if shs_residuals is None else [_f32c(r, "shs_residuals") for r in shs_residuals if r is not None]
        if res and shs is None:
            raise ValueError("shs_residuals needs shs")
        if len(res) > 2 or any(r.shape != shs.shape for r in res):
            raise ValueError("shs_residuals: at most two tensors of the shape of shs")
        if res and opts.factored_sh_grad:
            raise ValueError("shs_residuals need the dense dL/dshs (factored_sh_grad=False)")
        res += [None] * (2 - len(res))
        color, depth, normal, alpha, radii, *x_imgs = _Rasterize.apply(
            means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, actor_pose, residual_dx,
            residual_dq, actor_ids, rs, flags, opts, rec, extras[0], extras[1], res[0], res[1])
        self.last_call = rec
        return color, depth, normal, alpha, radii, (list(x_imgs) if x_imgs else None)

    # ---- introspection of THIS object's most recent call (tests / bench; not part of the reference surface) -------------------
    def last_status(self):
        return self.last_call.last_status()

    def export_binning(self, with_masks=False):
        return self.last_call.export_binning(with_masks=with_masks)

    def export_geometry(self):
        return self.last_call.export_geometry()
