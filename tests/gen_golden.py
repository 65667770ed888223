#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's Python from /root/reference (build container only).

The reference never travels to the GPU box: only the small input/expected-output vectors written here do.
Recipe: stub the absent third-party modules with a meta_path finder, rewrite
device='cuda' to CPU with a TorchFunctionMode, make .cuda() the identity, then call the reference functions.

    python tests/gen_golden.py          # rewrites tests/golden/*.npz  (seeds fixed below)

Fixtures (all fp32):
  s3g_sh.npz        eval_sh deg 0..3 + clamp_min(sh + 0.5, 0)     S3Gaussian/utils/sh_utils.py:57-112, gaussian_renderer/__init__.py:19-25
  s3g_cov.npz       build_covariance_from_scaling_rotation          S3Gaussian/scene/gaussian_model.py:34-38, utils/general_utils.py:231-277
  s3g_proj.npz      geom_transform_points                           S3Gaussian/utils/graphics_utils.py:42-49
  s3g_camera.npz    getWorld2View2 / getProjectionMatrix / Camera   S3Gaussian/utils/graphics_utils.py:58-92, scene/cameras.py:55-66
  s3g_quat.npz      batch_quaternion_multiply                       S3Gaussian/utils/graphics_utils.py:172-195
  or_quat.npz       quat_to_rotmat, quat_mult, interpolate_quats    OmniRe/models/gaussians/basics.py:30-110
  s3g_render.npz    render() executed with a recording stand-in for diff_gauss: the 12 settings fields and the tensors at
                    the rasterizer boundary (coarse and fine stage, run-script flags no_ds / no_dr), plus the deformation
                    network's residuals that produced them                S3Gaussian/gaussian_renderer/__init__.py:27-168
  s3g_render_combined.npz   the same render() with combine_dynamic_static and return_decomposition: the boundary tensors of its seven rasterizer
                    calls (opacity-mixed main pass; dynamic / static sets and |dx| colour passes)   S3Gaussian/gaussian_renderer/__init__.py:118-138,203-294
  s3g_sky.npz       SkyCubeMap.forward (rays, mask, clamp, layout) and the sky blend of render(), executed by the reference
                    with oracle/sky_oracle.cube_lookup standing in for the absent nvdiffrast dr.texture (its arguments are
                    recorded too)           S3Gaussian/scene/sky_cubemap.py:41-87, gaussian_renderer/__init__.py:299-301
  s3g_loss.npz      l1_loss, ssim, compute_depth("l2"), the sky BCE and the total of train.py:226-363 with their
                    gradients w.r.t. image / depth / weight      S3Gaussian/utils/loss_utils.py:21-98, train.py:226-363
  s3g_hexplane.npz  HexPlaneField.forward (multi-scale product of six bilinear plane lookups) with gradients w.r.t. every
                    plane and the points           S3Gaussian/scene/hexplane.py:18-183
  s3g_densify.npz   add_densification_stats + the max_radii2D update of the training loop over three views
                                                   S3Gaussian/scene/gaussian_model.py:728-730, train.py:403-406
  s3g_deform.npz    deform_network.forward (HexPlane + coarse-to-fine temporal embedding + heads + apply_deform), its state_dict,
                    outputs and gradients, under the run-script flags and with every head on     S3Gaussian/scene/deformation.py:187-527
  s3g_surgery.npz   GaussianModel.densify (clone + split, recorded normal draw), prune, reset_opacity: parameters, Adam moments,
                    statistics after each call; construct_list_of_attributes; capture() layout
  s3g_adam.npz      the optimiser GaussianModel.training_setup builds (torch.optim.Adam, eps 1e-15, ten named groups) stepped over six
                    iterations with seeded gradients and update_learning_rate      S3Gaussian/scene/gaussian_model.py:181-243, train.py:195,428
  or_deform.npz     DeformableNodes.get_deformation through ConditionalDeformNetwork (+ gradients)
                                                   OmniRe/models/nodes/deformable.py:35-47, models/modules.py:318-366,411-457
  or_nodes.npz      RigidNodes.get_gaussians and DeformableNodes.get_gaussians (+ gradients), with a recording stand-in for the absent gsplat
                    spherical_harmonics             OmniRe/models/nodes/rigid.py:570-615, models/nodes/deformable.py:49-114
  or_envlight.npz   EnvLight.forward with the same stand-in                OmniRe/models/modules.py:174-208
  or_refine.npz     VanillaGaussians.after_train (running refinement statistics over several views) and refinement_after (split + duplicate + cull
                    with the torch.randn draw recorded, cull only, opacity reset) with a torch.optim.Adam whose moments are non-trivial: parameters,
                    both moments and the statistics after every call      OmniRe/models/gaussians/vanilla.py:150-376, models/gaussians/basics.py:198-242
  or_rigid.npz      RigidNodes.transform_means / transform_quats / opacity mask (+ gradients), train and
                    test-interpolation branches, non-zero track heads  OmniRe/models/nodes/rigid.py:42-46,150-246,478-615
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("EMD_GOLDEN_OUT") or os.path.join(HERE, "golden")     # (the regeneration test writes elsewhere)
REF = "/root/reference"

ABSENT = {"diff_gauss", "plyfile", "simple_knn", "open3d", "nvdiffrast", "cv2", "imageio", "tkinter", "tinycudann",
          "skimage", "torchvision", "mmcv", "pytorch3d", "gsplat", "omegaconf", "trimesh", "smplx", "kornia", "viser",
          "nerfview", "pytorch_msssim", "torchmetrics", "wandb", "chumpy", "third_party", "lpips", "tensorboard"}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {"__init__": lambda self, *a, **k: None})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _CpuMode(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = kwargs or {}
        if "device" in kwargs and str(kwargs["device"]).startswith("cuda"):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def install_shims():
    sys.meta_path.insert(0, _Finder())
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def unload(prefixes):
    for k in list(sys.modules):
        if any(k == p or k.startswith(p + ".") for p in prefixes):
            del sys.modules[k]


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    clean = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.dtype == np.float64:
            v = v.astype(np.float32)
        clean[k] = v
    np.savez_compressed(os.path.join(OUT, name), **clean)
    listing = ", ".join(f"{k}{list(v.shape)}" for k, v in clean.items()) if len(clean) <= 40 else f"{len(clean)} arrays"
    print(f"  wrote {name}: " + listing)


def gen_s3g():
    sys.path.insert(0, os.path.join(REF, "S3Gaussian"))
    from utils.sh_utils import eval_sh
    from utils.graphics_utils import (geom_transform_points, getWorld2View2, getProjectionMatrix,
                                      batch_quaternion_multiply, focal2fov)
    g = torch.Generator().manual_seed(100)
    # SH
    n = 96
    shs = torch.randn(n, 16, 3, generator=g) * 0.3
    shs[:, 0] += 0.5
    xyz = torch.randn(n, 3, generator=g) * 5
    campos = torch.tensor([0.3, -0.2, 1.5])
    d = xyz - campos
    d = d / d.norm(dim=1, keepdim=True)
    out = {}
    for deg in range(4):
        r = eval_sh(deg, shs.transpose(1, 2).view(-1, 3, 16), d)
        out[f"sh_deg{deg}"] = r
        out[f"rgb_deg{deg}"] = torch.clamp_min(r + 0.5, 0.0)
    save("s3g_sh.npz", shs=shs, xyz=xyz, campos=campos, dirs=d, **out)
    # covariance (general_utils hard-codes device='cuda' -> rewritten by _CpuMode)
    with _CpuMode():
        from utils.general_utils import build_scaling_rotation, strip_symmetric
        scales = torch.exp(torch.randn(n, 3, generator=g) * 0.7 - 2.0)
        rots = torch.randn(n, 4, generator=g)
        cov = {}
        for mod in (1.0, 0.5):
            L = build_scaling_rotation(mod * scales, rots)
            cov[f"cov_mod{mod}"] = strip_symmetric(L @ L.transpose(1, 2))
    save("s3g_cov.npz", scales=scales, rots_raw=rots, **cov)
    # camera + projection
    yaw = 0.3
    c2w_R = np.array([[np.sin(yaw), 0, np.cos(yaw)], [-np.cos(yaw), 0, np.sin(yaw)], [0, -1, 0]], np.float64)
    eye = np.array([2.0, -1.0, 1.5])
    T = -c2w_R.T @ eye
    H, W, fx, fy = 64, 96, 110.0, 105.0
    fovx, fovy = focal2fov(fx, W), focal2fov(fy, H)
    w2c = getWorld2View2(c2w_R, T, np.array([0.0, 0.0, 0.0]), 1.0)
    wvt = torch.tensor(w2c).transpose(0, 1)
    proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    save("s3g_camera.npz", R=c2w_R, T=T, H=H, W=W, fx=fx, fy=fy, fovx=fovx, fovy=fovy, w2c=w2c, world_view_transform=wvt,
         projection_matrix=proj, full_proj_transform=full, camera_center=center)
    pts = torch.randn(n, 3, generator=g) * 4 + torch.tensor([8.0, 0.0, 1.0])
    save("s3g_proj.npz", points=pts, full_proj_transform=full, world_view_transform=wvt,
         ndc=geom_transform_points(pts, full), view=geom_transform_points(pts, wvt))
    q1, q2 = torch.randn(n, 4, generator=g), torch.randn(n, 4, generator=g)
    save("s3g_quat.npz", q1=q1, q2=q2, out=batch_quaternion_multiply(q1, q2))
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def _s3g_render_setup():
    """The reference's render() importable on CPU with a recording stand-in for diff_gauss, a 64-point GaussianModel with a visible
    deformation and one camera -> (args, pc, cam, bg, render, rec, base dict).  Caller: sys.path.pop(0) + unload(...) afterwards."""
    from typing import NamedTuple
    sys.path.insert(0, os.path.join(REF, "S3Gaussian"))
    sys.modules["utils.tcnn_modules"] = _Stub("utils.tcnn_modules")   # raises at import without CUDA (tcnn_modules.py:36-39)
    import diff_gauss

    class RS(NamedTuple):
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

    rec = []

    class FakeRast(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.rs = raster_settings

        def forward(self, **kw):
            rec.append((self.rs, kw))
            n, H, W = kw["means3D"].shape[0], self.rs.image_height, self.rs.image_width
            z = kw["means3D"].sum() * 0
            return (torch.zeros(3, H, W) + z, torch.zeros(1, H, W) + z, torch.zeros(3, H, W) + z, torch.zeros(1, H, W) + z,
                    torch.ones(n, dtype=torch.int32), None)

    diff_gauss.GaussianRasterizationSettings, diff_gauss.GaussianRasterizer = RS, FakeRast
    with _CpuMode():
        from arguments.gaussian_options import BaseOptions
        args = BaseOptions()
        for k in ("no_ds", "no_dr", "no_fine_hexplane_features"):   # scripts/dynamic/run_dynamic_nvs.sh
            setattr(args, k, True)
        args.feat_head = False
        torch.manual_seed(301)
        from scene.gaussian_model import GaussianModel
        from gaussian_renderer import render
        from scene.cameras import Camera
        from utils.graphics_utils import focal2fov
        pc = GaussianModel(args)
        N = 64
        g = torch.Generator().manual_seed(300)
        P = torch.nn.Parameter
        pc._xyz = P(torch.randn(N, 3, generator=g) * 3 + torch.tensor([8.0, 0, 1]))
        pc._features_dc = P(torch.randn(N, 1, 3, generator=g))
        pc._features_rest = P(torch.randn(N, 15, 3, generator=g) * 0.1)
        pc._scaling = P(torch.randn(N, 3, generator=g) * 0.5 - 2)
        pc._rotation = P(torch.randn(N, 4, generator=g))
        pc._opacity = P(torch.randn(N, 1, generator=g))
        pc._embedding = P(torch.randn(N, 4, generator=g) * 0.1)
        pc._deformation_table = torch.ones(N, dtype=torch.bool)
        pc.active_sh_degree = 2
        pc._deformation.deformation_net.set_aabb([20.0, 10.0, 10.0], [-5.0, -10.0, -5.0])
        for prm in pc._deformation.parameters():     # heads are near-zero at init: make the residuals visible
            if prm.dim() > 1:
                prm.data.normal_(0, 0.05)
        pc._sky_model = lambda cam, acc=None, is_train=False: torch.zeros(3, int(cam.image_height), int(cam.image_width))
        yaw = 0.3
        c2w_R = np.array([[np.sin(yaw), 0, np.cos(yaw)], [-np.cos(yaw), 0, np.sin(yaw)], [0, -1, 0]], np.float64)
        T = -c2w_R.T @ np.array([2.0, -1.0, 1.5])
        cam = Camera(colmap_id=0, R=c2w_R, T=T, FoVx=focal2fov(110.0, 96), FoVy=focal2fov(105.0, 64), image=torch.zeros(3, 64, 96),
                     gt_alpha_mask=None, image_name="x", uid=0, data_device="cpu", intrinsic=torch.eye(3), c2w=torch.eye(4),
                     time=0.3, cam_no=0, time_diff=0.0)
        bg = torch.tensor([0.1, 0.2, 0.3])
        out = dict(R=c2w_R, T=T, fovx=cam.FoVx, fovy=cam.FoVy, H=64, W=96, bg=bg, xyz=pc._xyz.data, scaling=pc._scaling.data,
                   rotation=pc._rotation.data, opacity=pc._opacity.data, features=pc.get_features.data, active_sh_degree=2)
    return args, pc, cam, bg, render, rec, out


def gen_s3g_render():
    """Run the reference render() on CPU with a recording fake rasterizer."""
    args, pc, cam, bg, render, rec, out = _s3g_render_setup()
    with _CpuMode():
        for stage in ("coarse", "fine"):
            rec.clear()
            res = render(args, cam, pc, bg, stage=stage, return_dx=True, iter=3000, is_train=True)
            rs, kw = rec[0]
            assert len(rec) == 1
            out.update({f"{stage}_tanfovx": rs.tanfovx, f"{stage}_tanfovy": rs.tanfovy, f"{stage}_scale_modifier": rs.scale_modifier,
                        f"{stage}_viewmatrix": rs.viewmatrix, f"{stage}_projmatrix": rs.projmatrix, f"{stage}_campos": rs.campos,
                        f"{stage}_sh_degree": rs.sh_degree, f"{stage}_prefiltered": int(rs.prefiltered), f"{stage}_debug": int(rs.debug),
                        f"{stage}_image_height": rs.image_height, f"{stage}_image_width": rs.image_width, f"{stage}_bg": rs.bg})
            for k in ("means3D", "shs", "opacities", "scales", "rotations"):
                out[f"{stage}_{k}"] = kw[k]
            assert kw["colors_precomp"] is None and kw["cov3Ds_precomp"] is None and kw["extra_attrs"] is None
            if stage == "fine":
                dd = res["ddict"]
                for lvl in ("coarse", "fine"):
                    for k in ("dx", "do", "dshs"):
                        out[f"ddict_{lvl}_{k}"] = dd[lvl][k]
                    assert dd[lvl]["ds"] is None and dd[lvl]["dr"] is None
        save("s3g_render.npz", **out)
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def gen_s3g_render_combined():
    """The same render() with `combine_dynamic_static` on (gaussian_renderer/__init__.py:118-138): (A) the SH path -- the boundary tensors
    of the one rasterizer call on the opacity-mixed set; (B) with `convert_SHs_python` and `return_decomposition` (:203-294) the seven
    calls: the main pass, then per decomposition level (coarse: the dynamic set, fine and coarse - fine: the static set) the set itself
    and the |dx| colour pass.  [With SH colours the reference's decomposition branch raises NameError: `colors_precomp_static` is only
    assigned on the precomputed-colour path (:129-133 vs :214).]"""
    args, pc, cam, bg, render, rec, out = _s3g_render_setup()
    with _CpuMode():
        args.combine_dynamic_static = True
        rec.clear()
        res = render(args, cam, pc, bg, stage="fine", return_dx=True, iter=3000, is_train=False)
        assert len(rec) == 1
        kw = rec[0][1]
        assert kw["colors_precomp"] is None and kw["cov3Ds_precomp"] is None and kw["extra_attrs"] is None
        for k in ("means3D", "shs", "opacities", "scales", "rotations"):
            out[f"sh_main_{k}"] = kw[k]
        dd = res["ddict"]
        for lvl in ("coarse", "fine"):
            for k in ("dx", "do", "dshs"):
                out[f"ddict_{lvl}_{k}"] = dd[lvl][k]
        args.convert_SHs_python = True
        rec.clear()
        res = render(args, cam, pc, bg, stage="fine", return_dx=True, return_decomposition=True, iter=3000, is_train=False)
        assert len(rec) == 7, len(rec)
        names = ("main", "coarse_set", "coarse_dx", "fine_set", "fine_dx", "coarse_fine_set", "coarse_fine_dx")
        for nm, (rs, kw) in zip(names, rec):
            assert kw["cov3Ds_precomp"] is None and kw["extra_attrs"] is None and kw["shs"] is None
            for k in ("means3D", "colors_precomp", "opacities", "scales", "rotations"):
                out[f"{nm}_{k}"] = kw[k]
        assert set(res["ddict_render"].keys()) == {"coarse_render", "fine_render", "coarse_fine_render"}
        out["camera_center"] = cam.camera_center
        save("s3g_render_combined.npz", **out)
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def _texture_standin(rec):
    """dr.texture(tex[None], dirs[None...], filter_mode=, boundary_mode=) -> oracle lookup; records its arguments."""
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import sky_oracle

    def texture(tex, uv, filter_mode="auto", boundary_mode="wrap", **kw):
        rec.append(dict(tex_shape=tuple(tex.shape), dirs=uv.detach().clone(), filter_mode=filter_mode, boundary_mode=boundary_mode))
        return sky_oracle.cube_lookup(tex[0], uv[0])[None]
    return texture


def gen_s3g_sky():
    sys.path.insert(0, os.path.join(REF, "S3Gaussian"))
    sys.modules["utils.tcnn_modules"] = _Stub("utils.tcnn_modules")
    import nvdiffrast.torch as dr
    rec = []
    dr.texture = _texture_standin(rec)
    with _CpuMode():
        from arguments.gaussian_options import BaseOptions
        import scene.sky_cubemap as sc
        sc.dr = dr
        from scene.cameras import Camera
        from utils.graphics_utils import focal2fov, get_rays_torch
        cfg = BaseOptions()
        cfg.sky_resolution = 16
        out = {}
        H, W = 40, 56
        K = torch.tensor([[60.0, 0, 27.5], [0, 58.0, 20.5], [0, 0, 1]])
        yaw = 0.4
        c2w_R = np.array([[np.sin(yaw), 0, np.cos(yaw)], [-np.cos(yaw), 0, np.sin(yaw)], [0, -1, 0]], np.float64)
        T = -c2w_R.T @ np.array([2.0, -1.0, 1.5])
        cam = Camera(colmap_id=0, R=c2w_R, T=T, FoVx=focal2fov(60.0, W), FoVy=focal2fov(58.0, H), image=torch.zeros(3, H, W),
                     gt_alpha_mask=None, image_name="x", uid=0, data_device="cpu", intrinsic=K, c2w=torch.eye(4),
                     time=0.3, cam_no=0, time_diff=0.0)
        g = torch.Generator().manual_seed(400)
        for white in (True, False):
            cfg.sky_white_background = white
            model = sc.SkyCubeMap(cfg)
            model.sky_cube_map.data = torch.rand(6, 16, 16, 3, generator=g) * 1.4 - 0.2          # some texels outside [0,1]: clamp
            acc = torch.rand(1, H, W, generator=g)
            acc[:, :10] = 1.0                                                                   # fully covered rows: masked out
            tag = "white" if white else "black"
            rec.clear()
            sky_all = model(cam, acc=None, is_train=False)
            sky_msk = model(cam, acc=acc, is_train=False)
            assert len(rec) == 2 and rec[0]["filter_mode"] == "linear" and rec[0]["boundary_mode"] == "cube"
            w2c = cam.world_view_transform.transpose(0, 1)
            _, rays = get_rays_torch(H, W, K, w2c[:3, :3], w2c[:3, 3], perturb=False)
            render = torch.rand(3, H, W, generator=g)
            blended = render * acc + sky_msk * (1 - acc)            # gaussian_renderer/__init__.py:300, verbatim formula
            out.update({f"{tag}_cube": model.sky_cube_map.data, f"{tag}_acc": acc, f"{tag}_sky_all": sky_all, f"{tag}_sky_masked": sky_msk,
                        f"{tag}_dirs_all": rec[0]["dirs"][0], f"{tag}_n_masked_dirs": rec[1]["dirs"].shape[2], f"{tag}_render": render,
                        f"{tag}_blended": blended})
        out.update(dict(H=H, W=W, K=K, world_view_transform=cam.world_view_transform, rays=rays))
        save("s3g_sky.npz", **out)
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def gen_s3g_loss():
    sys.path.insert(0, os.path.join(REF, "S3Gaussian"))
    with _CpuMode():
        from utils.loss_utils import l1_loss, ssim, compute_depth
        g = torch.Generator().manual_seed(500)
        H, W = 45, 70
        gt = torch.rand(3, H, W, generator=g)
        image = (gt + 0.15 * torch.randn(3, H, W, generator=g)).clamp(0, 1).requires_grad_(True)
        gt_depth = torch.rand(1, H, W, generator=g) * 100.0
        gt_depth[:, ::3] = 0.0                                  # no lidar return
        depth = (gt_depth + 3.0 * torch.randn(1, H, W, generator=g)).abs()
        depth[:, :, :4] = 95.0                                  # beyond max_depth: clamp, zero gradient
        depth.requires_grad_(True)
        sky_mask = torch.rand(1, H, W, generator=g) < 0.25
        weight = torch.rand(1, H, W, generator=g)
        weight[0, 0, :5] = 0.0                                  # outside the clamp range
        weight[0, 1, :5] = 1.0
        weight.requires_grad_(True)
        lam_dssim, lam_depth, lam_sky = 0.2, 0.5, 0.05         # arguments/gaussian_options.py:95-105
        mask = ~sky_mask                                        # train.py:221-222
        Ll1 = l1_loss(image, gt)
        loss = Ll1
        depth_loss = compute_depth("l2", depth * mask, gt_depth * mask) * lam_depth          # train.py:346-349
        loss = loss + depth_loss
        ssim_val = ssim(image, gt)
        loss = loss + lam_dssim * (1.0 - ssim_val)                                           # train.py:351-355
        w = torch.clamp(weight, min=1e-6, max=1. - 1e-6)                                     # train.py:357-361
        sky_loss = torch.where(sky_mask, -torch.log(1 - w), -torch.log(w)).mean()
        loss = loss + lam_sky * sky_loss
        loss.backward()
        save("s3g_loss.npz", image=image.data, gt=gt, depth=depth.data, gt_depth=gt_depth, sky_mask=sky_mask.to(torch.uint8),
             weight=weight.data, l1=Ll1, ssim=ssim_val, depth_l2=depth_loss / lam_depth, sky=sky_loss, total=loss,
             g_image=image.grad, g_depth=depth.grad, g_weight=weight.grad, lambdas=torch.tensor([lam_dssim, lam_depth, lam_sky]))
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def gen_s3g_hexplane():
    sys.path.insert(0, os.path.join(REF, "S3Gaussian"))
    with _CpuMode():
        import importlib.util                       # scene/__init__.py pulls in CUDA-only modules; the file itself needs only torch
        spec = importlib.util.spec_from_file_location("ref_hexplane", os.path.join(REF, "S3Gaussian", "scene", "hexplane.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        HexPlaneField = mod.HexPlaneField
        torch.manual_seed(600)
        cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 8, "resolution": [6, 5, 7, 4]}
        field = HexPlaneField(1.6, cfg, [1, 2])
        field.set_aabb([2.0, 1.5, 1.0], [-1.0, -1.5, -0.5])
        g = torch.Generator().manual_seed(601)
        for gp in field.grids:
            for prm in gp:
                prm.data = torch.rand(prm.shape, generator=g) + 0.25
        N = 300
        pts = (torch.rand(N, 3, generator=g) * torch.tensor([3.6, 3.6, 2.0]) + torch.tensor([-1.3, -1.8, -0.75]))   # some outside the box: border
        pts.requires_grad_(True)
        t = torch.rand(N, 1, generator=g) * 2.4 - 1.2                                                               # some outside [-1, 1]
        feat = field(pts, t)
        gout = torch.randn(feat.shape, generator=g)
        (feat * gout).sum().backward()
        out = dict(aabb=field.aabb.data, pts=pts.data, times=t, feat=feat, gout=gout, g_pts=pts.grad, multires=np.array([1, 2]),
                   resolution=np.array(cfg["resolution"]), channels=8)
        for s, gp in enumerate(field.grids):
            for p, prm in enumerate(gp):
                out[f"plane_{s}_{p}"] = prm.data
                out[f"g_plane_{s}_{p}"] = prm.grad
        save("s3g_hexplane.npz", **out)
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def gen_s3g_deform():
    """deform_network.forward of the reference on CPU (run-script flags, default feat_head), with gradients."""
    sys.path.insert(0, os.path.join(REF, "S3Gaussian"))
    sys.modules["utils.tcnn_modules"] = _Stub("utils.tcnn_modules")
    with _CpuMode():
        from arguments.gaussian_options import BaseOptions
        from scene.deformation import deform_network
        out = {}
        # two option sets: the run script's, and one with every head on and HexPlane features in the fine pass too
        for tag, flags, it, cam_no, t0 in (("run", dict(no_ds=True, no_dr=True, no_fine_hexplane_features=True), 9000, 1, 0.37),
                                           ("full", dict(), 21000, 2, 0.93)):
            args = BaseOptions()
            for k, v in flags.items():
                setattr(args, k, v)
            # small planes and tables keep the fixture small; the layer structure is the reference's
            args.kplanes_config = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 8, "resolution": [6, 5, 7, 9]}
            args.multires = [1, 2]
            args.min_embeddings, args.max_embeddings, args.temporal_embedding_dim, args.c2f_temporal_iter = 6, 20, 8, 25000
            torch.manual_seed(700)
            net = deform_network(args)
            net.deformation_net.set_aabb([3.0, 2.0, 1.5], [-1.0, -2.0, -0.5])
            g = torch.Generator().manual_seed(701)
            for n_, prm in net.named_parameters():
                if "grid" in n_:
                    prm.data = torch.rand(prm.shape, generator=g) + 0.25
                elif "time_offset" in n_:
                    prm.data = torch.tensor([[0.0], [0.031], [-0.96]])      # cam 2 at t0 = 0.93 lands below 0: reflection branch
                elif n_.endswith("deformation_net.weight"):
                    prm.data = torch.randn(prm.shape, generator=g) * 0.3
                else:
                    prm.data = torch.randn(prm.shape, generator=g) * (0.25 if prm.dim() > 1 else 0.05)
            N = 96
            point = (torch.rand(N, 3, generator=g) * torch.tensor([4.4, 4.4, 2.4]) + torch.tensor([-1.2, -2.2, -0.7])).requires_grad_(True)
            scales = torch.randn(N, 3, generator=g).requires_grad_(True)
            rotations = torch.randn(N, 4, generator=g).requires_grad_(True)
            opacity = torch.randn(N, 1, generator=g).requires_grad_(True)
            shs = torch.randn(N, 16, 3, generator=g).requires_grad_(True)
            emb = (torch.randn(N, 4, generator=g) * 0.5).requires_grad_(True)
            times = torch.full((N, 1), t0)
            res = net(point, scales, rotations, opacity, shs, times, emb, it, cam_no, 0.1, True)
            names = ("point", "scales", "rotations", "opacity", "shs")
            gouts = [torch.randn(r.shape, generator=g) for r in res[:5]]
            loss = sum((r * go).sum() for r, go in zip(res[:5], gouts))
            dd = res[5]
            gfeat = {}
            for lvl in ("coarse", "fine"):
                if dd[lvl]["feat"] is not None:
                    gfeat[lvl] = torch.randn(dd[lvl]["feat"].shape, generator=g)
                    loss = loss + (dd[lvl]["feat"] * gfeat[lvl]).sum()
            loss.backward()
            out.update({f"{tag}_in_{n_}": v.data for n_, v in zip(names, (point, scales, rotations, opacity, shs))})
            out.update({f"{tag}_in_emb": emb.data, f"{tag}_in_times": times, f"{tag}_iter": it, f"{tag}_cam_no": cam_no})
            out.update({f"{tag}_out_{n_}": r for n_, r in zip(names, res[:5])})
            out.update({f"{tag}_gout_{n_}": go for n_, go in zip(names, gouts)})
            out.update({f"{tag}_g_{n_}": v.grad for n_, v in zip(names + ("emb",), (point, scales, rotations, opacity, shs, emb))})
            for lvl in ("coarse", "fine"):
                for k, v in dd[lvl].items():
                    if v is not None:
                        out[f"{tag}_ddict_{lvl}_{k}"] = v
                if lvl in gfeat:
                    out[f"{tag}_gfeat_{lvl}"] = gfeat[lvl]
            for k, v in net.state_dict().items():
                out[f"{tag}_sd_{k}"] = v
            for n_, prm in net.named_parameters():
                out[f"{tag}_gsd_{n_}"] = prm.grad if prm.grad is not None else torch.zeros_like(prm)
            for k in ("no_ds", "no_dr", "no_fine_hexplane_features", "feat_head", "min_embeddings", "max_embeddings",
                      "temporal_embedding_dim", "c2f_temporal_iter"):
                out[f"{tag}_opt_{k}"] = int(getattr(args, k))
            out[f"{tag}_opt_multires"] = np.array(args.multires)
            out[f"{tag}_opt_resolution"] = np.array(args.kplanes_config["resolution"])
            out[f"{tag}_opt_channels"] = args.kplanes_config["output_coordinate_dim"]
        save("s3g_deform.npz", **out)
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def gen_s3g_adam():
    """The reference's optimiser (GaussianModel.training_setup) and learning-rate schedule over a few iterations, on CPU."""
    sys.path.insert(0, os.path.join(REF, "S3Gaussian"))
    sys.modules["utils.tcnn_modules"] = _Stub("utils.tcnn_modules")
    with _CpuMode():
        from arguments.gaussian_options import BaseOptions
        from scene.gaussian_model import GaussianModel
        args = BaseOptions()
        for k in ("no_ds", "no_dr", "no_fine_hexplane_features"):
            setattr(args, k, True)
        args.kplanes_config = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 4, "resolution": [4, 4, 4, 3]}
        args.multires = [1]
        args.sky_resolution = 4
        args.net_width, args.feat_head = 8, False          # small layers keep the fixture small; the groups are the reference's
        torch.manual_seed(900)
        pc = GaussianModel(args)
        N = 37
        g = torch.Generator().manual_seed(901)
        P = torch.nn.Parameter
        pc._xyz = P(torch.randn(N, 3, generator=g))
        pc._features_dc = P(torch.randn(N, 1, 3, generator=g))
        pc._features_rest = P(torch.randn(N, 15, 3, generator=g) * 0.1)
        pc._scaling = P(torch.randn(N, 3, generator=g))
        pc._rotation = P(torch.randn(N, 4, generator=g))
        pc._opacity = P(torch.randn(N, 1, generator=g))
        pc._embedding = P(torch.randn(N, 4, generator=g) * 0.1)
        pc.spatial_lr_scale = 5.0
        pc.training_setup(args)
        opt = pc.optimizer
        out = dict(eps=opt.defaults["eps"], beta1=opt.defaults["betas"][0], beta2=opt.defaults["betas"][1],
                   group_names=np.array([gr["name"] for gr in opt.param_groups]))
        for gr in opt.param_groups:
            out[f"init_{gr['name']}"] = torch.cat([p.data.reshape(-1) for p in gr["params"]])
            out[f"count_{gr['name']}"] = len(gr["params"])
        iters = [0, 1, 2, 700, 701, 9000]
        out["iters"] = np.array(iters)
        for k, it in enumerate(iters):
            pc.update_learning_rate(it)
            out[f"lr_{k}"] = np.array([gr["lr"] for gr in opt.param_groups], dtype=np.float64)
            for gr in opt.param_groups:
                gs = []
                for p in gr["params"]:
                    p.grad = torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-4, 1, (1,), generator=g)))
                    gs.append(p.grad.reshape(-1))
                out[f"grad_{k}_{gr['name']}"] = torch.cat(gs)
            opt.step()
        for gr in opt.param_groups:
            out[f"final_{gr['name']}"] = torch.cat([p.data.reshape(-1) for p in gr["params"]])
            out[f"exp_avg_{gr['name']}"] = torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in gr["params"]])
            out[f"exp_avg_sq_{gr['name']}"] = torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) for p in gr["params"]])
        # the schedule arguments training_setup derived (the builder's expon_lr must reproduce lr_k from them)
        for k in ("position_lr_init", "position_lr_final", "position_lr_delay_mult", "position_lr_max_steps", "deformation_lr_init",
                  "deformation_lr_final", "deformation_lr_delay_mult", "grid_lr_init", "grid_lr_final", "feature_lr", "opacity_lr",
                  "scaling_lr", "rotation_lr", "sky_cube_map_lr_init", "sky_cube_map_lr_final", "sky_cube_map_max_steps"):
            out[f"arg_{k}"] = float(getattr(args, k))
        out["spatial_lr_scale"] = 5.0
        save("s3g_adam.npz", **out)
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def gen_s3g_densify():
    sys.path.insert(0, os.path.join(REF, "S3Gaussian"))
    sys.modules["utils.tcnn_modules"] = _Stub("utils.tcnn_modules")
    with _CpuMode():
        from arguments.gaussian_options import BaseOptions
        from scene.gaussian_model import GaussianModel
        pc = GaussianModel(BaseOptions())
        N = 500
        g = torch.Generator().manual_seed(700)
        pc.xyz_gradient_accum = torch.rand(N, 1, generator=g)
        pc.denom = torch.randint(0, 5, (N, 1), generator=g).float()
        pc.max_radii2D = torch.randint(0, 30, (N,), generator=g).float()
        out = dict(accum0=pc.xyz_gradient_accum.clone(), denom0=pc.denom.clone(), maxr0=pc.max_radii2D.clone())
        for v in range(3):
            grad = torch.randn(N, 3, generator=g) * 1e-3
            radii = torch.randint(-1, 40, (N,), generator=g).clamp(min=0).to(torch.int32)
            radii[torch.rand(N, generator=g) < 0.3] = 0
            vis = radii > 0
            pc.max_radii2D[vis] = torch.max(pc.max_radii2D[vis], radii[vis])             # train.py:405
            pc.add_densification_stats(grad, vis)                                         # train.py:406
            out.update({f"grad{v}": grad, f"radii{v}": radii})
        out.update(accum=pc.xyz_gradient_accum, denom=pc.denom, maxr=pc.max_radii2D)
        save("s3g_densify.npz", **out)
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def gen_s3g_surgery():
    """The reference's adaptive density control on CPU: GaussianModel.densify (clone + split, with the torch.normal draw recorded),
    prune and reset_opacity on a seeded model whose optimiser has taken two steps -- every parameter, both Adam moments of the seven
    per-point groups, the statistics and the deformation table after each call -- plus construct_list_of_attributes() and the
    layout of capture()."""
    sys.path.insert(0, os.path.join(REF, "S3Gaussian"))
    sys.modules["utils.tcnn_modules"] = _Stub("utils.tcnn_modules")
    with _CpuMode():
        from arguments.gaussian_options import BaseOptions
        from scene.gaussian_model import GaussianModel
        args = BaseOptions()
        for k in ("no_ds", "no_dr", "no_fine_hexplane_features"):
            setattr(args, k, True)
        args.kplanes_config = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 4, "resolution": [4, 4, 4, 3]}
        args.multires = [1]
        args.sky_resolution = 4
        args.net_width, args.feat_head = 8, False
        torch.manual_seed(1200)
        pc = GaussianModel(args)
        N = 400
        g = torch.Generator().manual_seed(1201)
        P = torch.nn.Parameter
        pc._xyz = P(torch.randn(N, 3, generator=g) * 5)
        pc._features_dc = P(torch.randn(N, 1, 3, generator=g))
        pc._features_rest = P(torch.randn(N, 15, 3, generator=g) * 0.1)
        pc._scaling = P(torch.log(0.02 + 0.3 * torch.rand(N, 3, generator=g)))
        pc._rotation = P(torch.randn(N, 4, generator=g))
        pc._opacity = P(torch.randn(N, 1, generator=g) * 2.5)
        pc._embedding = P(torch.randn(N, 4, generator=g) * 0.1)
        pc._deformation_table = torch.rand(N, generator=g) > 0.3
        pc.max_radii2D = torch.zeros(N)
        pc.spatial_lr_scale = 5.0
        pc.training_setup(args)
        names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "embedding")
        attr = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling", rotation="_rotation",
                    embedding="_embedding")
        out = dict(attributes=np.array(pc.construct_list_of_attributes()), percent_dense=float(args.percent_dense))
        for it in range(2):                                # two optimiser steps: non-trivial Adam moments
            pc.update_learning_rate(it)
            for n_ in names:
                p = getattr(pc, attr[n_])
                p.grad = torch.randn(p.shape, generator=g) * 1e-2
            pc.optimizer.step()
        pc.xyz_gradient_accum = torch.rand(N, 1, generator=g) * 1e-3
        pc.denom = torch.randint(0, 4, (N, 1), generator=g).float()
        pc.max_radii2D = torch.randint(0, 40, (N,), generator=g).float()

        def snap(tag):
            for n_ in names:
                p = getattr(pc, attr[n_])
                out[f"{tag}_{n_}"] = p.detach().clone()
                st = pc.optimizer.state[p]
                out[f"{tag}_m_{n_}"], out[f"{tag}_v_{n_}"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
            out[f"{tag}_accum"], out[f"{tag}_denom"], out[f"{tag}_maxr"] = pc.xyz_gradient_accum.clone(), pc.denom.clone(), pc.max_radii2D.clone()
            out[f"{tag}_table"] = pc._deformation_table.clone()
        snap("in")
        extent, max_grad = 20.0, 2.0e-4                      # percent_dense * extent = 0.2: both the clone and the split branch fire
        out["extent"], out["max_grad"] = extent, max_grad
        rec = {}
        real_normal = torch.normal

        def recording_normal(mean, std, **kw):
            z = torch.randn(std.shape, generator=g)
            rec["z"] = z.clone()
            return mean + std * z
        torch.normal = recording_normal
        try:
            pc.densify(max_grad, 0.005, extent, None, 5, 5)
        finally:
            torch.normal = real_normal
        out["normal_z"] = rec["z"]
        snap("dens")
        # between events the training loop accumulates statistics again (train.py:403-406)
        M = pc._xyz.shape[0]
        pc.max_radii2D = torch.randint(0, 40, (M,), generator=g).float()
        pc.xyz_gradient_accum = torch.rand(M, 1, generator=g) * 1e-3
        pc.denom = torch.randint(0, 4, (M, 1), generator=g).float()
        out["prune_in_maxr"], out["prune_in_accum"], out["prune_in_denom"] = pc.max_radii2D.clone(), pc.xyz_gradient_accum.clone(), pc.denom.clone()
        out["min_opacity"], out["max_screen_size"], out["prune_extent"] = 0.05, 20.0, 2.5     # 0.1 * 2.5: the world-size test fires too
        pc.prune(max_grad, 0.05, 2.5, 20)
        snap("prune")
        pc.reset_opacity()
        snap("reset")
        cap = pc.capture()
        out["capture_len"] = len(cap)
        out["capture_kinds"] = np.array([type(c).__name__ for c in cap])
        save("s3g_surgery.npz", **out)
    sys.path.pop(0)
    unload(["utils", "scene", "arguments", "gaussian_renderer"])


def gen_or_envlight():
    sys.path.insert(0, os.path.join(REF, "OmniRe"))
    import nvdiffrast.torch as dr
    rec = []
    dr.texture = _texture_standin(rec)
    with _CpuMode():
        import models.modules as mm
        mm.dr = dr
        env = mm.EnvLight(class_name="Sky", resolution=8, device=torch.device("cpu"))
        g = torch.Generator().manual_seed(410)
        env.base.data = torch.rand(6, 8, 8, 3, generator=g)
        viewdirs = torch.nn.functional.normalize(torch.randn(12, 20, 3, generator=g), dim=-1)
        light = env({"viewdirs": viewdirs})
        rgb, opacity = torch.rand(12, 20, 3, generator=g), torch.rand(12, 20, 1, generator=g)
        blended = rgb + light * (1.0 - opacity)                      # models/trainers/base.py:497, verbatim formula
        assert rec[0]["filter_mode"] == "linear" and rec[0]["boundary_mode"] == "cube"
        save("or_envlight.npz", base=env.base.data, viewdirs=viewdirs, lookup_dirs=rec[0]["dirs"].reshape(-1, 3), light=light,
             rgb=rgb, opacity=opacity, blended=blended, to_opengl=env.to_opengl)
    sys.path.pop(0)
    unload(["models", "utils", "datasets"])


def gen_or_deform():
    """ConditionalDeformNetwork and DeformableNodes.get_deformation of the reference on CPU, with gradients."""
    sys.path.insert(0, os.path.join(REF, "OmniRe"))
    with _CpuMode():
        from models.modules import ConditionalDeformNetwork
        from models.nodes.deformable import DeformableNodes
        torch.manual_seed(800)
        E, A, per = 16, 3, 30
        net = ConditionalDeformNetwork(D=8, W=32, input_ch=3, embed_dim=E, x_multires=10, t_multires=10, deform_quat=True, deform_scale=False)
        g = torch.Generator().manual_seed(801)
        for prm in net.parameters():
            prm.data = torch.randn(prm.shape, generator=g) * (0.2 if prm.dim() > 1 else 0.05)
        node = types.SimpleNamespace()
        node.point_ids = torch.arange(A).repeat_interleave(per)[:, None]
        node.instances_embedding = torch.rand(A, E, generator=g).requires_grad_(True)
        node.instances_size = torch.tensor([[0.6, 0.5, 1.7], [0.7, 0.6, 1.85], [1.8, 0.6, 1.2]])
        node.normalized_timestamps = torch.linspace(0, 1, 7)
        node.cur_frame = 4
        node.deform_network = net
        means = ((torch.rand(A * per, 3, generator=g) - 0.5) * torch.tensor([0.6, 0.5, 1.7])).requires_grad_(True)
        dxyz, dquat, dscale = DeformableNodes.get_deformation(node, means)
        assert dscale is None
        gx, gq = torch.randn(dxyz.shape, generator=g), torch.randn(dquat.shape, generator=g)
        ((dxyz * gx).sum() + (dquat * gq).sum()).backward()
        assert means.grad is None                                   # local_means.data: detached in the reference
        out = dict(means=means.data, point_ids=node.point_ids[:, 0].to(torch.int32), inst_size=node.instances_size,
                   inst_embed=node.instances_embedding.data, t=node.normalized_timestamps[node.cur_frame], dxyz=dxyz, dquat=dquat, gx=gx, gq=gq,
                   g_inst_embed=node.instances_embedding.grad, D=8, W=32, embed_dim=E, x_multires=10, t_multires=10)
        # the encoder input the reference built (first layer's input), recomputed from its own embedders
        x = means.data / node.instances_size[node.point_ids[:, 0]][:, 2:3] * 2
        tt = node.normalized_timestamps[node.cur_frame].unsqueeze(0).repeat(A * per, 1)
        out["h0"] = torch.cat([net.embed_fn(x), net.embed_time_fn(tt), node.instances_embedding.data[node.point_ids[:, 0]]], -1)
        for k, v in net.state_dict().items():
            out[f"sd_{k}"] = v
        for n_, prm in net.named_parameters():
            out[f"gsd_{n_}"] = prm.grad
        save("or_deform.npz", **out)
    sys.path.pop(0)
    unload(["models", "utils", "datasets"])


def gen_or_nodes():
    """RigidNodes.get_gaussians and DeformableNodes.get_gaussians of the reference on CPU (+ gradients).  The absent gsplat
    `spherical_harmonics` is a recording stand-in that evaluates oracle/torch_ref.eval_sh on the arguments the reference passes."""
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import torch_ref as tr
    sys.path.pop(0)
    sys.path.insert(0, os.path.join(REF, "OmniRe"))
    import pytorch3d.transforms as p3t
    p3t.matrix_to_quaternion = lambda M: _matrix_to_quaternion(M).reshape(*M.shape[:-2], 4)
    sh_calls = []

    def sh_standin(degree, dirs, coeffs):
        sh_calls.append(dict(degree=int(degree), dirs_requires_grad=bool(dirs.requires_grad)))
        return tr.eval_sh(int(degree), coeffs.transpose(1, 2), dirs)

    with _CpuMode():
        from models.gaussians import basics
        basics.matrix_to_quaternion = p3t.matrix_to_quaternion
        import models.nodes.rigid as rigid_mod
        import models.nodes.deformable as deform_mod
        rigid_mod.matrix_to_quaternion = p3t.matrix_to_quaternion
        rigid_mod.spherical_harmonics = deform_mod.spherical_harmonics = sh_standin
        g = torch.Generator().manual_seed(950)
        F_, A, P = 6, 3, 30
        c2w = torch.eye(4)
        c2w[:3, 3] = torch.tensor([2.0, -1.0, 1.6])
        cam = types.SimpleNamespace(camtoworlds=c2w)
        out = dict(camera_center=c2w[:3, 3].clone(), num_frames=F_)
        for cls_name, cls in (("rigid", rigid_mod.RigidNodes), ("deformable", deform_mod.DeformableNodes)):
            ctrl = _Cfg(sh_degree=3, gaussian_embedding_dim=4, temporal_embedding_dim=32, no_gaussian_embedding_dim=False,
                        no_temporal_embedding_dim=False, no_coarse_deform=False, no_fine_deform=False, no_c2f_temporal_embedding=False,
                        min_embeddings=30, max_embeddings=150, c2f_temporal_iter=25000, no_apply_embed_shs=True, no_apply_embed_track=False,
                        sh_degree_interval=1000, use_deformgs_for_nonrigid=True, use_deformgs_after=100, stop_optimizing_canonical_xyz=True)
            ctrl.get = lambda k, d=None, c=ctrl: dict.get(c, k, d)
            nets = _Cfg(D=4, W=32, embed_dim=8, x_multires=6, t_multires=4, deform_quat=True, deform_scale=False)
            torch.manual_seed(951)
            node = cls(class_name=cls_name, ctrl=ctrl, reg=_Cfg(), networks=nets, scene_scale=30.0, scene_origin=torch.zeros(3),
                       num_train_images=10, device=torch.device("cpu"))
            inst = {}
            for a in range(A):
                yaw = torch.linspace(0.2 * (a + 1), 0.2 * (a + 1) + 0.4, F_)
                poses = torch.eye(4).repeat(F_, 1, 1)
                poses[:, 0, 0], poses[:, 0, 1], poses[:, 1, 0], poses[:, 1, 1] = torch.cos(yaw), -torch.sin(yaw), torch.sin(yaw), torch.cos(yaw)
                poses[:, :3, 3] = torch.stack([8.0 + 3 * a + torch.arange(F_) * 0.5, torch.full((F_,), -2.0 + 2 * a), torch.full((F_,), 0.9)], 1)
                fv = torch.ones(F_, dtype=torch.bool)
                if a == 2:
                    fv[3] = False
                size = torch.tensor([0.7 + 0.1 * a, 0.6, 1.7 + 0.05 * a])
                inst[a] = dict(class_name="ped", pts=(torch.rand(P, 3, generator=g) - 0.5) * size, colors=torch.rand(P, 3, generator=g), poses=poses,
                               size=size, frame_info=fv, num_pts=P)
            node.create_from_pcd(inst)
            node.register_normalized_timestamps(torch.linspace(0, 1, F_))
            node._features_rest.data.normal_(0, 0.2, generator=g)
            node._quats.data = torch.randn(node._quats.shape, generator=g)
            node._opacities.data = torch.randn(node._opacities.shape, generator=g)
            node.instances_quats.data += 0.05 * torch.randn(node.instances_quats.shape, generator=g)
            if cls_name == "deformable":
                for prm in node.deform_network.parameters():
                    prm.data = torch.randn(prm.shape, generator=g) * (0.15 if prm.dim() > 1 else 0.03)
            node.step, node.cur_frame, node.in_test_set = 12000, 3, False
            sh_calls.clear()
            gs = node.get_gaussians(cam)
            gouts = {k: torch.randn(v.shape, generator=g) for k, v in gs.items()}
            sum((gs[k] * gouts[k]).sum() for k in gs).backward()
            pre = cls_name + "_"
            out.update({pre + "means": node._means.data, pre + "quats": node._quats.data, pre + "opacity_logits": node._opacities.data,
                        pre + "log_scales": node._scales.data, pre + "features_dc": node._features_dc.data, pre + "features_rest": node._features_rest.data,
                        pre + "point_ids": node.point_ids[:, 0].to(torch.int32), pre + "instances_quats": node.instances_quats.data,
                        pre + "instances_trans": node.instances_trans.data, pre + "instances_fv": node.instances_fv, pre + "frame": node.cur_frame,
                        pre + "step": node.step, pre + "instances_size": node.instances_size, pre + "sh_degree_used": sh_calls[0]["degree"],
                        pre + "sh_dirs_requires_grad": int(sh_calls[0]["dirs_requires_grad"])})
            for k, v in gs.items():
                out[pre + "out" + k] = v
                out[pre + "gout" + k] = gouts[k]
            for name in ("_means", "_quats", "_opacities", "_scales", "_features_dc", "_features_rest", "instances_quats", "instances_trans"):
                gr = getattr(node, name).grad
                out[pre + "grad" + name] = gr if gr is not None else torch.zeros_like(getattr(node, name))
            if cls_name == "deformable":
                out[pre + "instances_embedding"] = node.instances_embedding.data
                out[pre + "grad_instances_embedding"] = node.instances_embedding.grad
                out[pre + "t"] = node.normalized_timestamps[node.cur_frame]
                for k, v in node.deform_network.state_dict().items():
                    out[pre + "sd_" + k] = v
                for n_, prm in node.deform_network.named_parameters():
                    out[pre + "gsd_" + n_] = prm.grad
                for k, v in nets.items():
                    out[pre + "net_" + k] = int(v)
        save("or_nodes.npz", **out)
    sys.path.pop(0)
    unload(["models", "utils", "datasets"])


def _matrix_to_quaternion(M):
    """Stand-in for pytorch3d.transforms.matrix_to_quaternion (only used at RigidNodes init, rigid.py:271)."""
    M = M.reshape(-1, 3, 3)
    out = []
    for m in M:
        m = m.double().numpy()
        t = np.trace(m)
        if t > 0:
            s = np.sqrt(t + 1.0) * 2
            q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
        else:
            i = int(np.argmax(np.diag(m)))
            j, k = (i + 1) % 3, (i + 2) % 3
            s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
            q = [0.0] * 4
            q[0] = (m[k, j] - m[j, k]) / s
            q[1 + i] = 0.25 * s
            q[1 + j] = (m[j, i] + m[i, j]) / s
            q[1 + k] = (m[k, i] + m[i, k]) / s
        out.append(q)
    return torch.tensor(out, dtype=torch.float32)


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def gen_omnire():
    sys.path.insert(0, os.path.join(REF, "OmniRe"))
    import pytorch3d.transforms as p3t  # stub module
    p3t.matrix_to_quaternion = lambda M: _matrix_to_quaternion(M).reshape(*M.shape[:-2], 4)
    with _CpuMode():
        from models.gaussians import basics
        basics.matrix_to_quaternion = p3t.matrix_to_quaternion
        g = torch.Generator().manual_seed(200)
        n = 80
        q = torch.randn(n, 4, generator=g)
        q1, q2 = torch.randn(n, 4, generator=g), torch.randn(n, 4, generator=g)
        qa, qb = torch.randn(n, 4, generator=g), torch.randn(n, 4, generator=g)
        qb[: n // 4] = qa[: n // 4] + 0.001 * qb[: n // 4]     # exercises the "similar" (lerp) branch
        save("or_quat.npz", q=q, rotmat=basics.quat_to_rotmat(q), q1=q1, q2=q2, mult=basics.quat_mult(q1, q2),
             qa=qa, qb=qb, interp=basics.interpolate_quats(qa.clone(), qb.clone()))

        from models.nodes.rigid import RigidNodes
        import models.nodes.rigid as rigid_mod
        rigid_mod.matrix_to_quaternion = p3t.matrix_to_quaternion
        ctrl = _Cfg(sh_degree=3, gaussian_embedding_dim=4, temporal_embedding_dim=32, no_gaussian_embedding_dim=False,
                    no_temporal_embedding_dim=False, no_coarse_deform=False, no_fine_deform=False,
                    no_c2f_temporal_embedding=False, min_embeddings=30, max_embeddings=150, c2f_temporal_iter=25000,
                    no_apply_embed_shs=True, no_apply_embed_track=False, sh_degree_interval=1000)
        ctrl.get = lambda k, d=None: dict.get(ctrl, k, d)
        F_, A, P = 6, 3, 40
        torch.manual_seed(201)
        node = RigidNodes(class_name="RigidNodes", ctrl=ctrl, reg=_Cfg(), networks=_Cfg(), scene_scale=30.0,
                          scene_origin=torch.zeros(3), num_train_images=10, device=torch.device("cpu"))
        inst = {}
        for a in range(A):
            yaw = torch.linspace(0.1 * (a + 1), 0.1 * (a + 1) + 0.5, F_)
            poses = torch.eye(4).repeat(F_, 1, 1)
            poses[:, 0, 0], poses[:, 0, 1], poses[:, 1, 0], poses[:, 1, 1] = torch.cos(yaw), -torch.sin(yaw), torch.sin(yaw), torch.cos(yaw)
            poses[:, :3, 3] = torch.stack([10.0 + 3 * a + torch.arange(F_) * 0.8, torch.full((F_,), -3.0 + 3 * a), torch.full((F_,), 0.8)], 1)
            fv = torch.ones(F_, dtype=torch.bool)
            if a == 1:
                fv[2] = False
            inst[a] = dict(class_name="car", pts=(torch.rand(P, 3) - 0.5) * torch.tensor([4.5, 2.0, 1.6]),
                           colors=torch.rand(P, 3), poses=poses, size=torch.tensor([4.5, 2.0, 1.6]), frame_info=fv, num_pts=P)
        node.create_from_pcd(inst)
        # non-zero track heads, non-trivial embeddings and pose noise so every term is exercised
        for lin in (node.track_rot_c, node.track_rot_f, node.track_trans_c, node.track_trans_f):
            lin.weight.data.normal_(0, 0.05)
            lin.bias.data.normal_(0, 0.02)
        node._embeddings.data.normal_(0, 0.5)
        node.weight.data.normal_(0, 0.3)
        node.instances_quats.data += 0.05 * torch.randn_like(node.instances_quats)   # un-normalised on purpose
        node._quats.data *= 1.7
        node.step = 12000
        rec = dict(means=node._means.data.clone(), quats=node._quats.data.clone(), opacity_logits=node._opacities.data.clone(),
                   point_ids=node.point_ids[:, 0].to(torch.int32), instances_quats=node.instances_quats.data.clone(),
                   instances_trans=node.instances_trans.data.clone(), instances_fv=node.instances_fv.clone(),
                   num_frames=F_, step=node.step, embeddings=node._embeddings.data.clone(), temporal_weight=node.weight.data.clone(),
                   track_rot_c_w=node.track_rot_c.weight.data.clone(), track_rot_c_b=node.track_rot_c.bias.data.clone(),
                   track_rot_f_w=node.track_rot_f.weight.data.clone(), track_rot_f_b=node.track_rot_f.bias.data.clone(),
                   track_trans_c_w=node.track_trans_c.weight.data.clone(), track_trans_c_b=node.track_trans_c.bias.data.clone(),
                   track_trans_f_w=node.track_trans_f.weight.data.clone(), track_trans_f_b=node.track_trans_f.bias.data.clone())
        for tag, frame, test in (("train", 2, False), ("train5", 5, False), ("test", 3, True), ("test_edge", 1, True)):
            node.cur_frame = frame
            node.in_test_set = test
            for p in node.parameters():
                p.grad = None
            with torch.set_grad_enabled(not test):   # the test-time interpolation is in-place / eval-only in the reference
                wm = node.transform_means(node._means)
                wq = node.transform_quats(node._quats)
                valid = node.get_pts_valid_mask()
                opac = torch.sigmoid(node._opacities) * valid.float().unsqueeze(-1)
                wq_act = node.quat_act(wq)
            gm = torch.randn(wm.shape, generator=g)
            gq = torch.randn(wq.shape, generator=g)
            if not test:
                ((wm * gm).sum() + (wq_act * gq).sum()).backward()
            # per-actor offsets the reference produced (inputs of this build's pose-table builder)
            dts, dqs = [], []
            with torch.no_grad():
                for a in range(A):
                    emb = node._embeddings[(node.point_ids == a).squeeze(1), :]
                    dts.append(node.embedding_track_trans_offset(frame=frame, start_frame=0, end_frame=F_ - 1, embeddings=emb, weight=node.weight[a]))
                    dqs.append(node.embedding_track_rot_offset(frame=frame, start_frame=0, end_frame=F_ - 1, embeddings=emb, weight=node.weight[a]))
            rec.update({f"{tag}_frame": frame, f"{tag}_in_test": int(test), f"{tag}_world_means": wm, f"{tag}_world_quats_act": wq_act,
                        f"{tag}_opacity": opac, f"{tag}_track_trans": torch.stack(dts), f"{tag}_track_rot": torch.stack(dqs),
                        f"{tag}_gm": gm, f"{tag}_gq": gq})
            if not test:
                rec.update({f"{tag}_grad_means": node._means.grad.clone(), f"{tag}_grad_quats": node._quats.grad.clone(),
                            f"{tag}_grad_instances_trans": node.instances_trans.grad.clone(),
                            f"{tag}_grad_instances_quats": node.instances_quats.grad.clone()})
        save("or_rigid.npz", **rec)
    sys.path.pop(0)
    unload(["models", "utils", "datasets"])


def gen_or_refine():
    """OmniRe's per-class density control on CPU: VanillaGaussians.after_train over views with partial visibility, then refinement_after at four
    steps that exercise every branch -- 3600: split (size and screen-size tests) + duplicate + cull (alpha, world size, screen size); 700: split +
    duplicate + cull by alpha only (before the first opacity reset); 3100: the opacity reset alone (inside the guard behind a reset); 16000: cull
    only (past stop_split_at, screen tests off).  The optimiser is the one the trainer builds (torch.optim.Adam, eps 1e-15, class-prefixed group
    names) after two steps on seeded gradients, so the moments that dup_in_optim / remove_from_optim move are non-trivial."""
    sys.path.insert(0, os.path.join(REF, "OmniRe"))
    import pytorch3d.transforms as p3t
    p3t.matrix_to_quaternion = lambda M: _matrix_to_quaternion(M).reshape(*M.shape[:-2], 4)
    with _CpuMode():
        from models.gaussians import basics
        basics.matrix_to_quaternion = p3t.matrix_to_quaternion
        from models.gaussians.vanilla import VanillaGaussians
        cfg = dict(sh_degree=1, warmup_steps=500, reset_alpha_interval=3000, refine_interval=100, sh_degree_interval=1000, n_split_samples=2,
                   reset_alpha_value=0.01, densify_grad_thresh=0.0003, densify_size_thresh=0.003, cull_alpha_thresh=0.005, cull_scale_thresh=0.5,
                   cull_screen_size=0.15, split_screen_size=0.05, stop_screen_size_at=4000, stop_split_at=15000)
        ctrl = _Cfg(cfg)
        ctrl.get = lambda k, d=None: dict.get(ctrl, k, d)
        scene_scale, n_images, last_size = 2.0, 10, 1600
        node = VanillaGaussians(class_name="Background", ctrl=ctrl, reg=_Cfg(), networks=_Cfg(), scene_scale=scene_scale, scene_origin=torch.zeros(3),
                                num_train_images=n_images, device=torch.device("cpu"))
        g = torch.Generator().manual_seed(1300)
        N = 160          # (SH degree 1 keeps the fixture small: the row width is generic in the gather, 45-wide rows are pinned by s3g_surgery.npz)
        P = torch.nn.Parameter
        node._means = P(torch.randn(N, 3, generator=g) * 4)
        # log-uniform scales from 1e-3 to 3: below / just above (split AND duplicated) / far above the size threshold 0.006, some above the cull size 1.0
        node._scales = P(torch.log(torch.tensor(1e-3)) + torch.rand(N, 3, generator=g) * (torch.log(torch.tensor(3.0)) - torch.log(torch.tensor(1e-3))))
        node._scales.data[: N // 3] -= 1.5
        node._quats = P(torch.randn(N, 4, generator=g))
        node._opacities = P(torch.randn(N, 1, generator=g) * 3.0 - 1.0)
        node._features_dc = P(torch.randn(N, 3, generator=g))
        node._features_rest = P(torch.randn(N, 3, 3, generator=g) * 0.1)
        names = ("xyz", "sh_dc", "sh_rest", "opacity", "scaling", "rotation")
        attr = dict(xyz="_means", sh_dc="_features_dc", sh_rest="_features_rest", opacity="_opacities", scaling="_scales", rotation="_quats")
        lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scaling=5e-3, rotation=1e-3)
        opt = torch.optim.Adam([{"params": v, "lr": lrs[k.split("#")[1]], "name": k} for k, v in node.get_gaussian_param_groups().items()], lr=0.0, eps=1e-15)
        for it in range(2):
            for n_ in names:
                p = getattr(node, attr[n_])
                p.grad = torch.randn(p.shape, generator=g) * 1e-2
            opt.step()
        out = dict(scene_scale=scene_scale, num_train_images=n_images, last_size=last_size, class_name=np.array("Background"),
                   cfg_keys=np.array(list(cfg)), cfg_values=np.array([float(v) for v in cfg.values()]), group_names=np.array([g_["name"] for g_ in opt.param_groups]))

        def state_of(n_):
            for grp in opt.param_groups:
                if grp["name"] == "Background#" + n_:
                    return opt.state[grp["params"][0]]

        def snap(tag):
            for n_ in names:
                out[f"{tag}_{n_}"] = getattr(node, attr[n_]).detach().clone()
                st = state_of(n_)
                out[f"{tag}_m_{n_}"], out[f"{tag}_v_{n_}"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()

        def views(tag, count):
            """`count` calls of after_train with seeded radii (a third invisible) and gradients; inputs and the statistics after each call"""
            n = node.num_points
            node.filter_mask = torch.ones_like(node._means[:, 0], dtype=torch.bool)          # what get_gaussians leaves behind every step (vanilla.py:379-380)
            for v in range(count):
                radii = torch.randint(0, 400, (n,), generator=g, dtype=torch.int32)
                radii[torch.rand(n, generator=g) < 0.35] = 0
                grad = (torch.rand(n, 2, generator=g) - 0.5) * 0.9e-3
                grad[radii == 0] = 0.0
                node.after_train(radii, grad, last_size)
                out[f"{tag}_radii{v}"], out[f"{tag}_grad{v}"] = radii.clone(), grad.clone()
                out[f"{tag}_norm{v}"], out[f"{tag}_vis{v}"], out[f"{tag}_m2d{v}"] = node.xys_grad_norm.clone(), node.vis_counts.clone(), node.max_2Dsize.clone()

        real_randn = torch.randn

        def refine(tag, step):
            rec = {}

            def recording_randn(size, **kw):
                z = real_randn(tuple(size), generator=g)
                rec["z"] = z.clone()
                return z
            node.step = step
            torch.randn = recording_randn
            try:
                node.refinement_after(step, opt)
            finally:
                torch.randn = real_randn
            out[f"{tag}_step"] = step
            out[f"{tag}_randn"] = rec.get("z", torch.zeros(0, 3))
            snap(tag)

        snap("in")
        views("a", 3)
        refine("A", 3600)

        def drift(tag, faint, huge):
            """what training between two events does to a few rows: some fade below the cull threshold, some grow past the world-size limit"""
            n = node.num_points
            i_f = torch.randperm(n, generator=g)[:faint]
            i_h = torch.randperm(n, generator=g)[:huge]
            node._opacities.data[i_f] = -6.0 - torch.rand(faint, 1, generator=g)
            node._scales.data[i_h] = torch.log(torch.tensor(1.0)) + torch.rand(huge, 3, generator=g)
            out[f"{tag}_faint_rows"], out[f"{tag}_huge_rows"] = i_f.to(torch.int32), i_h.to(torch.int32)
            out[f"{tag}_opacity_in"], out[f"{tag}_scaling_in"] = node._opacities.detach().clone(), node._scales.detach().clone()
        drift("B", 25, 10)
        views("b", 2)
        refine("B", 700)
        views("c", 1)
        refine("C", 3100)
        drift("D", 30, 20)
        views("d", 2)
        refine("D", 16000)
        save("or_refine.npz", **out)
    sys.path.pop(0)
    unload(["models", "utils", "datasets"])


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    install_shims()
    print("S3Gaussian:")
    gen_s3g()
    gen_s3g_render()
    gen_s3g_render_combined()
    gen_s3g_sky()
    gen_s3g_loss()
    gen_s3g_hexplane()
    gen_s3g_densify()
    gen_s3g_deform()
    gen_s3g_adam()
    gen_s3g_surgery()
    print("OmniRe:")
    gen_omnire()
    gen_or_envlight()
    gen_or_deform()
    gen_or_nodes()
    gen_or_refine()
