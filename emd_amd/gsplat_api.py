"""The OmniRe call sites of the hot path, served by the same HIP rasterizer.

  rasterization(...)          replaces `gsplat.rendering.rasterization`   OmniRe/models/trainers/base.py:393-408
  spherical_harmonics(...)    replaces `gsplat.cuda._wrapper.spherical_harmonics`  OmniRe/models/nodes/rigid.py:584,
                              deformable.py:83, gaussians/vanilla.py:388

Same keywords, shapes and result structure as the reference uses them:
  renders [C,H,W,4] (RGB + expected depth for render_mode="RGB+ED"), alphas [C,H,W,1],
  info{"means2d" [C,N,2] (retain_grad-able; `.grad` in pixel units, `.absgrad` when absgrad=True),
       "radii" [C,N] int32, "width", "height"}
so that `postprocess_per_train_step` (base.py:279-297) works unchanged.

gsplat itself is an un-vendored, un-pinned dependency of the reference; this adapter gives its
interface the semantics of this repository's rasterizer (3-sigma radius, 16x16 tiles, 0.3 px dilation, alpha clamp
0.99, 1/255 cut-off, T < 1e-4 stop) with gsplat's near-plane cull and intrinsics-matrix cameras.  Known, documented
differences from current gsplat releases: tile/radius bookkeeping (not observable in the image beyond the cut-off
rule) and the pixel-centre convention (centres at integer coordinates, i.e. cx is the reference's cx - 0.5 shifted
inside `projection_from_K`).
"""
import ctypes as C
import math

import torch
import torch.nn.functional as F

from . import _lib as L
from .rasterizer import GaussianRasterizationSettings, RasterCall, RasterConfig, _Rasterize


class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, dirs, coeffs):
        lib = L.load()
        if dirs.device.type != "cuda":
            raise L.EmdError("spherical_harmonics needs tensors on a ROCm device; there is no CPU path")
        n, K = coeffs.shape[0], coeffs.shape[1]
        rgb = torch.empty(n, 3, device=dirs.device, dtype=torch.float32)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.emd_sh_forward(n, int(degree), K, dirs.data_ptr(), coeffs.data_ptr(), rgb.data_ptr(), st), "emd_sh_forward")
        ctx.degree = int(degree)
        ctx.save_for_backward(dirs, coeffs)
        return rgb

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        dirs, coeffs = ctx.saved_tensors
        n, K = coeffs.shape[0], coeffs.shape[1]
        g = g.contiguous().float()
        d_coeffs = torch.empty_like(coeffs)
        d_dirs = torch.empty_like(dirs) if ctx.needs_input_grad[1] else None
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.emd_sh_backward(n, ctx.degree, K, dirs.data_ptr(), coeffs.data_ptr(), g.data_ptr(), d_coeffs.data_ptr(),
                                    L.ptr(d_dirs), st), "emd_sh_backward")
        return None, d_dirs, d_coeffs


def spherical_harmonics(degrees_to_use, dirs, coeffs):
    """rgb[N,3] = SH_degree(dirs / |dirs|) . coeffs[N,K,3]   (no +0.5, no clamp; the caller adds them, rigid.py:585)."""
    if dirs.shape[:-1] != coeffs.shape[:-2] or coeffs.shape[-1] != 3:
        raise ValueError(f"dirs {tuple(dirs.shape)} / coeffs {tuple(coeffs.shape)} mismatch")
    if (degrees_to_use + 1) ** 2 > coeffs.shape[-2]:
        raise ValueError("coeffs hold fewer than (degree + 1)^2 bases")
    shp = dirs.shape[:-1]
    out = _SphericalHarmonics.apply(int(degrees_to_use), dirs.reshape(-1, 3).contiguous().float(),
                                    coeffs.reshape(-1, coeffs.shape[-2], 3).contiguous().float())
    return out.reshape(*shp, 3)


_K_CONST = {}


def _device_camera(vm, K, width, height, znear=0.01, zfar=100.0):
    """viewmatrix (= W2C^T), full projection, camera centre and tan(fov / 2) of one gsplat camera, computed ON THE DEVICE from
    the device-resident `viewmats[c]` / `Ks[c]` (OmniRe/models/trainers/base.py:399-400): no device-to-host copy.  Same
    formulas and the same fp32 roundings as emd_amd.camera.projection_from_K / from_c2w_K."""
    dev = vm.device
    ck = (dev, int(width), int(height), float(znear), float(zfar))
    if ck not in _K_CONST:
        base = torch.zeros(4, 4)
        base[2, 3] = 1.0                                   # P[3, 2] in column-vector form
        base[2, 2] = zfar / (zfar - znear)
        base[3, 2] = -(zfar * znear) / (zfar - znear)
        _K_CONST[ck] = (base.to(dev), torch.tensor([0, 5, 8, 9], device=dev),            # flat indices of Pt[0,0], [1,1], [2,0], [2,1]
                        torch.tensor([0.0, 0.0, float(width), float(height)], device=dev),
                        torch.tensor([float(width), float(height), float(width), float(height)], device=dev))
    base, idx, sub, size = _K_CONST[ck]
    k4 = torch.stack([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])                              # fx, fy, cx, cy
    vals = (2.0 * k4 - sub) / size                                                       # 2 fx / W, 2 fy / H, (2 cx - W) / W, (2 cy - H) / H
    Pt = base.clone()
    Pt.view(-1)[idx] = vals
    wvt = vm.t().contiguous()
    full = wvt @ Pt
    campos = -(vm[:3, :3].t() @ vm[:3, 3])                                               # inverse of a rigid W2C
    tanfov = size[:2] / (2.0 * k4[:2])
    return wvt, full, campos, tanfov


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, near_plane=0.01, far_plane=1e10,
                  radius_clip=0.0, eps2d=0.3, sh_degree=None, packed=False, tile_size=16, backgrounds=None,
                  render_mode="RGB", sparse_grad=False, absgrad=False, rasterize_mode="classic", channel_chunk=32,
                  distributed=False, **kwargs):
    if tile_size != 16 or eps2d != 0.3:
        raise NotImplementedError("tile_size 16 and eps2d 0.3 are part of the sort-key / footprint contract")
    if rasterize_mode != "classic" or packed or sparse_grad or distributed or radius_clip != 0.0:
        raise NotImplementedError("only the options the reference uses are served (omnire.yaml:11-18): classic, "
                                  "packed=False, sparse_grad=False, radius_clip=0")
    if far_plane is not None and float(far_plane) < 1e10:
        # gsplat culls Gaussians beyond far_plane; the reference never sets it (base.py:393-408 leaves the 1e10 default) and the
        # projection kernel has no far cull -- refusing is better than silently rendering what gsplat would have dropped
        raise NotImplementedError("far_plane < 1e10 is not served: the reference leaves gsplat's default (no far culling)")
    if render_mode not in ("RGB", "D", "ED", "RGB+D", "RGB+ED"):
        raise ValueError(render_mode)
    Cn = viewmats.shape[0]
    N = means.shape[0]
    dev = means.device
    quats_n = F.normalize(quats, dim=-1)
    opac = opacities.reshape(-1)
    renders, alphas, radii_all, records = [], [], [], []
    # leaf grad sink: after backward `.grad` holds d L / d (pixel-space mean); `.absgrad` the sum of |.| (absgrad=True)
    means2d = torch.zeros(Cn, N, 2, device=dev, requires_grad=True)
    scale = torch.tensor([2.0 / width, 2.0 / height], device=dev)
    if absgrad:
        means2d.absgrad = torch.zeros(Cn, N, 2, device=dev)
    # options of THESE calls (nothing process-wide is touched): no normal image, gsplat's near plane, absgrad as asked
    opts = RasterConfig.replace(compute_normal=False, absgrad=bool(absgrad), near_plane=float(near_plane))
    flags = (L.FLAG_ABSGRAD if absgrad else 0) | (L.FLAG_NO_SYNC if opts.no_sync else 0)
    for c in range(Cn):
        vm = viewmats[c].detach().float()
        K = Ks[c].detach().float()
        if vm.device.type == "cpu" or K.device.type == "cpu":
            vm, K = vm.to(dev), K.to(dev)
        wvt, full, campos, tanfov = _device_camera(vm, K, width, height)
        bg = torch.zeros(3, device=dev) if backgrounds is None else backgrounds[c].detach().float()
        rs = GaussianRasterizationSettings(int(height), int(width), tanfov[0], tanfov[1], bg, 1.0, wvt, full,
                                           0 if sh_degree is None else int(sh_degree), campos, False, False)
        # this rasterizer returns mean2D gradients scaled by (W/2, H/2) (diff_gauss convention): undo it here
        sink = torch.cat([means2d[c] * scale, torch.zeros(N, 1, device=dev)], dim=1)
        shs = colors.contiguous().float() if sh_degree is not None else None
        col = None if sh_degree is not None else colors.contiguous().float()
        rec = RasterCall()
        color, depth, _n, alpha, radii = _Rasterize.apply(means.contiguous().float(), sink, shs, col, opac.contiguous().float(),
                                                          scales.contiguous().float(), quats_n.contiguous(), None, None, None,
                                                          None, None, rs, flags, opts, rec)[:5]
        if absgrad:
            def _hook(g, c=c, rec=rec):          # runs after this call's backward: its record holds this call's |grad| sums
                means2d.absgrad[c] = rec.absgrad * scale
                return g
            sink.register_hook(_hook)
        chans = []
        if "RGB" in render_mode:
            chans.append(color.permute(1, 2, 0))
        if render_mode.endswith("ED"):
            chans.append((depth / alpha.clamp(min=1e-10)).permute(1, 2, 0))
        elif render_mode.endswith("D"):
            chans.append(depth.permute(1, 2, 0))
        renders.append(torch.cat(chans, dim=-1))
        alphas.append(alpha.permute(1, 2, 0))
        radii_all.append(radii)
        records.append(rec)
    info = {"means2d": means2d, "radii": torch.stack(radii_all), "width": int(width), "height": int(height), "raster_calls": records}
    return torch.stack(renders), torch.stack(alphas), info
