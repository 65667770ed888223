"""Sky cube map + final blend on the HIP path.

Host-side mirror of the two reference modules, same names and argument meaning:
  * `SkyCubeMap`   S3Gaussian/scene/sky_cubemap.py:13-87   forward(camera, acc=None, is_train=False) -> [3,H,W]
  * `EnvLight`     OmniRe/models/modules.py:174-208         forward(image_infos) -> [..., 3]
plus the fused composites the callers build right after (gaussian_renderer/__init__.py:299-301 and
models/trainers/base.py:491-497): `composite_s3g`, `composite_add`.  One HIP launch per direction of the pass instead of
get_rays_torch (7 launches) + mask gather + nvdiffrast dr.texture (CUDA-only) + scatter + permute + clamp + blend.
There is no CPU path: the ops raise when the extension is missing or the tensors are not on a ROCm device.
"""
import ctypes as C

import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t, what):
    if t.device.type != "cuda":
        raise L.EmdError(f"{what} needs tensors on a ROCm device; there is no CPU path")


def _fill_camera(a, vals):
    for i in range(9):
        a.Kinv[i] = vals[i]
        a.R[i] = vals[9 + i]
    for i in range(3):
        a.T[i] = vals[18 + i]


class _SkyOp(torch.autograd.Function):
    """sky colour (and optionally the blended image) from the cube map; gradients to cube, fg and acc (blend only)."""

    @staticmethod
    def forward(ctx, cube, fg, acc, dirs, cam, jitter, H, W, flags, threshold, fill, want_blend):
        _need_gpu(cube, "sky lookup")
        lib = L.load()
        cube = cube.contiguous().float()
        res = cube.shape[1]
        if cube.shape != (6, res, res, 3):
            raise L.EmdError(f"cube map must be [6,res,res,3], got {tuple(cube.shape)}")
        dev = cube.device
        il = bool(flags & L.SKY_INTERLEAVED)
        shape = (H * W, 3) if il else (3, H, W)
        a = L.EmdSkyArgs()
        a.height, a.width, a.resolution, a.flags = H, W, res, flags
        a.cube = cube.data_ptr()
        keep = [cube]
        if dirs is not None:
            dirs = dirs.contiguous().float()
            keep.append(dirs)
            a.dirs = dirs.data_ptr()
        elif isinstance(cam, torch.Tensor):
            # the 21 floats (Kinv, R, T) on the device, e.g. a row emd_amd.StepInputs selected: nothing of the camera is baked into a capture
            if cam.device != dev or cam.dtype != torch.float32 or cam.numel() != 21 or not cam.is_contiguous():
                raise L.EmdError("sky: a device-side camera is 21 contiguous float32 values (Kinv[9], R[9], T[3]) on the cube map's device")
            keep.append(cam)
            a.camera_dev = cam.data_ptr()
        else:
            _fill_camera(a, cam)
        if jitter is not None:
            jitter = jitter.contiguous().float()
            keep.append(jitter)
            a.jitter = jitter.data_ptr()
        if acc is not None:
            acc = acc.contiguous().float()
            a.acc = acc.data_ptr()
        a.mask_threshold, a.fill = threshold, fill
        sky = torch.empty(shape, device=dev, dtype=torch.float32)
        a.sky = sky.data_ptr()
        out = None
        if want_blend:
            fg = fg.contiguous().float()
            a.fg = fg.data_ptr()
            out = torch.empty(shape, device=dev, dtype=torch.float32)
            a.out = out.data_ptr()
        L.check(lib.emd_sky_forward(C.byref(a), _stream()), "emd_sky_forward")
        ctx.args, ctx.keep, ctx.want_blend = a, keep, want_blend
        ctx.save_for_backward(cube, fg if want_blend else None, acc)
        return (sky, out) if want_blend else (sky, sky.new_empty(0))

    @staticmethod
    def backward(ctx, g_sky, g_out):
        lib = L.load()
        cube, fg, acc = ctx.saved_tensors
        b = L.EmdSkyBwdArgs()
        C.memmove(C.byref(b.f), C.byref(ctx.args), C.sizeof(L.EmdSkyArgs))
        b.f.sky = None
        b.f.out = None
        keep = []
        if g_sky is not None:
            g_sky = g_sky.contiguous().float(); keep.append(g_sky); b.dL_dsky = g_sky.data_ptr()
        if ctx.want_blend and g_out is not None:
            g_out = g_out.contiguous().float(); keep.append(g_out); b.dL_dout = g_out.data_ptr()
        if b.dL_dsky is None and b.dL_dout is None:
            return (None,) * 12
        d_cube = torch.empty_like(cube) if ctx.needs_input_grad[0] else None
        d_fg = torch.empty_like(fg) if (ctx.want_blend and ctx.needs_input_grad[1]) else None
        d_acc = torch.empty_like(acc) if (ctx.want_blend and acc is not None and ctx.needs_input_grad[2]) else None
        b.dL_dcube, b.dL_dfg, b.dL_dacc = L.ptr(d_cube), L.ptr(d_fg), L.ptr(d_acc)
        L.check(lib.emd_sky_backward(C.byref(b), _stream()), "emd_sky_backward")
        return (d_cube, d_fg, d_acc) + (None,) * 9


def _camera_rays_params(camera):
    """Host copy of (Kinv, R, T) exactly as SkyCubeMap.forward derives them (sky_cubemap.py:52-54): 21 floats that go
    into the kernel arguments.  A camera's matrices are fixed, so the inverse + device-to-host copy happens once per camera
    (cached on the object, keyed by the tensors' storage and version) instead of stalling the stream every step."""
    dev_rays = getattr(camera, "sky_rays", None)
    if dev_rays is not None:                 # a camera whose ray constants live on the device (emd_amd.StepInputs.camera)
        return dev_rays
    K, wvt = camera.intrinsic, camera.world_view_transform
    key = (K.data_ptr(), K._version, wvt.data_ptr(), wvt._version)
    hit = getattr(camera, "_emd_sky_params", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    w2c = wvt.transpose(0, 1)
    vals = sky_ray_constants(K, wvt).tolist()
    try:
        camera._emd_sky_params = (key, vals)
    except AttributeError:
        pass
    return vals


def sky_ray_constants(K, world_view_transform):
    """The 21 floats (Kinv, R, T) of a camera as SkyCubeMap.forward derives them (sky_cubemap.py:52-54), on the CPU."""
    w2c = world_view_transform.transpose(0, 1)
    return torch.cat([torch.inverse(K.float()).reshape(-1), w2c[:3, :3].reshape(-1), w2c[:3, 3].reshape(-1)]).detach().to("cpu", torch.float32)


class SkyCubeMap(torch.nn.Module):
    """S3Gaussian/scene/sky_cubemap.py:13-87.  `cfg` needs sky_resolution, sky_white_background, white_background."""

    def __init__(self, cfg, device="cuda"):
        super().__init__()
        self.cfg = cfg
        self.sky_resolution = cfg.sky_resolution
        eps = 1e-3
        r = self.sky_resolution
        if cfg.sky_white_background:
            base = torch.ones(6, r, r, 3, device=device) * (1.0 - eps)
        else:
            base = torch.zeros(6, r, r, 3, device=device) + eps
        self.sky_cube_map = torch.nn.Parameter(base)

    def forward(self, camera, acc=None, is_train=False, jitter=None):
        """[3,H,W] sky colour.  is_train: sub-pixel jitter U[0,1) (get_rays_torch perturb=True; pass `jitter` [H,W,2] to
        reproduce a given draw) and the camera's `sky_mask` when it has one (rows < 50 forced on), as the reference."""
        H, W = int(camera.image_height), int(camera.image_width)
        dev = self.sky_cube_map.device
        if is_train and jitter is None:
            jitter = torch.rand(H, W, 2, device=dev)
        threshold, a = -1.0, None
        if is_train and hasattr(camera, "sky_mask"):
            m = camera.sky_mask.to(dev)[0].bool().clone()
            m[:50, :] = True
            a, threshold = 1.0 - m.float(), 0.5            # (1 - a) > 0.5  <=>  mask
        elif acc is not None:
            a, threshold = acc[0].detach(), 1e-3
        fill = 1.0 if self.cfg.sky_white_background else 0.0
        sky, _ = _SkyOp.apply(self.sky_cube_map, None, a, None, _camera_rays_params(camera), jitter, H, W, L.SKY_CLAMP01, threshold,
                              fill, False)
        return sky


def composite_s3g(sky_model: SkyCubeMap, camera, render, weight, is_train=False, jitter=None):
    """render * weight + sky * (1 - weight) in the same launch as the lookup (gaussian_renderer/__init__.py:299-301).
    Returns (blended [3,H,W], sky_color [3,H,W]); gradients reach the cube map, `render` and `weight` (blend only)."""
    H, W = int(camera.image_height), int(camera.image_width)
    dev = sky_model.sky_cube_map.device
    if is_train and jitter is None:
        jitter = torch.rand(H, W, 2, device=dev)
    if is_train and hasattr(camera, "sky_mask"):      # the mask comes from the camera, the blend from `weight`: two launches
        sky = sky_model(camera, acc=weight, is_train=True, jitter=jitter)
        return render * weight + sky * (1 - weight), sky
    fill = 1.0 if sky_model.cfg.sky_white_background else 0.0
    sky, out = _SkyOp.apply(sky_model.sky_cube_map, render, weight.reshape(H, W), None, _camera_rays_params(camera), jitter, H, W,
                            L.SKY_CLAMP01 | L.SKY_BLEND_S3G, 1e-3, fill, True)
    return out, sky


class EnvLight(torch.nn.Module):
    """OmniRe/models/modules.py:174-208: cube map looked up along `image_infos["viewdirs"]` (rotated to OpenGL axes)."""

    def __init__(self, class_name="Sky", resolution=1024, device="cuda", **kwargs):
        super().__init__()
        self.class_prefix = class_name + "#"
        self.to_opengl = torch.tensor([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=torch.float32, device=device)
        self.base = torch.nn.Parameter(0.5 * torch.ones(6, resolution, resolution, 3, device=device))

    def forward(self, image_infos):
        l = image_infos["viewdirs"]
        prefix = l.shape[:-1]
        d = (l.reshape(-1, 3) @ self.to_opengl.T).contiguous()
        sky, _ = _SkyOp.apply(self.base, None, None, d, None, None, 1, d.shape[0], L.SKY_INTERLEAVED, -1.0, 0.0, False)
        return sky.view(*prefix, -1)

    def get_param_groups(self):
        return {self.class_prefix + "all": self.parameters()}


def composite_add(env: EnvLight, image_infos, rgb_gaussians, opacity):
    """rgb_gaussians + rgb_sky * (1 - opacity) fused with the lookup (OmniRe/models/trainers/base.py:491-497).
    rgb_gaussians [...,3], opacity [...,1]; returns (rgb, rgb_sky)."""
    l = image_infos["viewdirs"]
    prefix = l.shape[:-1]
    d = (l.reshape(-1, 3) @ env.to_opengl.T).contiguous()
    P = d.shape[0]
    sky, out = _SkyOp.apply(env.base, rgb_gaussians.reshape(P, 3), opacity.reshape(P), d, None, None, 1, P,
                            L.SKY_INTERLEAVED | L.SKY_BLEND_ADD, -1.0, 0.0, True)
    return out.view(*prefix, 3), sky.view(*prefix, 3)
