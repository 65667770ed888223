"""CPU restatement of the sky-cubemap path.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and nothing else; the product path is emd_amd/csrc/sky.hip and fails loudly without it.

What the reference does (call sites):
  * S3Gaussian/scene/sky_cubemap.py:41-87   SkyCubeMap.forward: rays from get_rays_torch (utils/graphics_utils.py:220-241),
    mask = (1 - acc[0]) > 1e-3, `dr.texture(cube[None], rays_d[None], filter_mode='linear', boundary_mode='cube')`,
    clamp(0, 1), planar [3,H,W]; S3Gaussian/gaussian_renderer/__init__.py:299-301 blends
    render * weight + sky_color * (1 - weight).
  * OmniRe/models/modules.py:174-208        EnvLight.forward: viewdirs @ to_opengl.T, same dr.texture call, no clamp;
    OmniRe/models/trainers/base.py:491-497 blends rgb_gaussians + rgb_sky * (1 - opacity).

The texture lookup itself lives in a third-party CUDA package that is absent from /root/reference and cannot run here:
`nvdiffrast` (NVlabs/nvdiffrast, version unpinned by the reference).  PARITY UNPINNED for the lookup: this file restates
its published algorithm -- OpenGL cube-map face selection and orientation (faces +x,-x,+y,-y,+z,-z; major axis z if
|z| > max(|x|,|y|), else y if |y| > |x|, else x), bilinear filtering on texel centres (u * res - 0.5), and seamless
filtering across cube edges (a tap that leaves its face is fetched from the adjacent face; at a cube corner the
fourth tap does not exist and the other three are renormalised).  Everything AROUND the lookup (rays, mask, clamp,
layout, blend) is pinned by tests/golden/s3g_sky.npz, produced by the reference's own SkyCubeMap / render() with this
lookup standing in for dr.texture (tests/gen_golden.py).
"""
import torch


def index_cube(d):
    """direction [...,3] -> (face [...], u [...], v [...]) with u, v in [0,1]  (OpenGL convention)."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    ax, ay, az = x.abs(), y.abs(), z.abs()
    is_z = az > torch.maximum(ax, ay)
    is_y = (~is_z) & (ay > ax)
    is_x = ~(is_z | is_y)
    c = torch.where(is_z, z, torch.where(is_y, y, x))
    face = torch.where(is_z, 4, torch.where(is_y, 2, 0)) + (c < 0).to(torch.int64)
    m = 0.5 / c.abs()
    # sc, tc per face: +x (-z,-y)  -x (z,-y)  +y (x,z)  -y (x,-z)  +z (x,-y)  -z (-x,-y)
    sc = torch.where(is_x, torch.where(c > 0, -z, z), torch.where(is_y, x, torch.where(c > 0, x, -x)))
    tc = torch.where(is_y, torch.where(c > 0, z, -z), -y)
    u = (sc * m + 0.5).clamp(0.0, 1.0)
    v = (tc * m + 0.5).clamp(0.0, 1.0)
    return face, u, v


def face_uv_to_dir(face, u, v):
    """inverse of index_cube on the (extended) face plane"""
    s, t = 2.0 * u - 1.0, 2.0 * v - 1.0
    one = torch.ones_like(s)
    table = [torch.stack([one, -t, -s], -1), torch.stack([-one, -t, s], -1), torch.stack([s, one, t], -1),
             torch.stack([s, -one, -t], -1), torch.stack([s, -t, one], -1), torch.stack([-s, -t, -one], -1)]
    out = torch.zeros(face.shape + (3,), dtype=u.dtype)
    for f in range(6):
        out = torch.where((face == f)[..., None], table[f], out)
    return out


def cube_taps(dirs, res):
    """[P,3] -> flat texel indices [P,4] into [6*res*res], weights [P,4] (corner taps weight 0, rest renormalised)."""
    face, u, v = index_cube(dirs)
    uu, vv = u * res - 0.5, v * res - 0.5
    iu0, iv0 = torch.floor(uu), torch.floor(vv)
    fu, fv = uu - iu0, vv - iv0
    iu0, iv0 = iu0.to(torch.int64), iv0.to(torch.int64)
    idx, wts = [], []
    eps = 0.25 / res
    for du, dv, w in ((0, 0, (1 - fu) * (1 - fv)), (1, 0, fu * (1 - fv)), (0, 1, (1 - fu) * fv), (1, 1, fu * fv)):
        iu, iv = iu0 + du, iv0 + dv
        out_u, out_v = (iu < 0) | (iu >= res), (iv < 0) | (iv >= res)
        # the tap's texel centre, with the coordinate that left the face pushed just beyond the edge
        tu = torch.where(iu < 0, torch.full_like(u, -eps), torch.where(iu >= res, torch.full_like(u, 1 + eps), (iu.to(u.dtype) + 0.5) / res))
        tv = torch.where(iv < 0, torch.full_like(v, -eps), torch.where(iv >= res, torch.full_like(v, 1 + eps), (iv.to(v.dtype) + 0.5) / res))
        f2, u2, v2 = index_cube(face_uv_to_dir(face, tu, tv))
        ju = torch.clamp(torch.floor(u2 * res).to(torch.int64), 0, res - 1)
        jv = torch.clamp(torch.floor(v2 * res).to(torch.int64), 0, res - 1)
        inside = ~(out_u | out_v)
        ff = torch.where(inside, face, f2)
        ju = torch.where(inside, iu, ju)
        jv = torch.where(inside, iv, jv)
        corner = out_u & out_v
        idx.append(torch.where(corner, torch.zeros_like(ju), (ff * res + jv) * res + ju))
        wts.append(torch.where(corner, torch.zeros_like(w), w))
    idx, wts = torch.stack(idx, -1), torch.stack(wts, -1)
    return idx, wts / wts.sum(-1, keepdim=True)


def cube_lookup(cube, dirs):
    """dr.texture(cube[None], dirs[None], filter_mode='linear', boundary_mode='cube') for cube [6,res,res,C], dirs [...,3]."""
    res, C = cube.shape[1], cube.shape[-1]
    flat = dirs.reshape(-1, 3)
    idx, w = cube_taps(flat, res)
    tex = cube.reshape(-1, C)
    out = (tex[idx] * w[..., None]).sum(1)
    return out.reshape(dirs.shape[:-1] + (C,))


def rays(H, W, K, R, T, jitter=None):
    """get_rays_torch (S3Gaussian/utils/graphics_utils.py:220-241): unit ray directions [H,W,3]."""
    rays_o = -torch.matmul(R.T, T).squeeze()
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy")
    if jitter is None:
        xy1 = torch.stack([i + 0.5, j + 0.5, torch.ones_like(i)], dim=2)
    else:
        xy1 = torch.stack([i + jitter[..., 0], j + jitter[..., 1], torch.ones_like(i)], dim=2)
    pixel_camera = torch.matmul(xy1, torch.inverse(K).T)
    pixel_world = torch.matmul(pixel_camera - T.squeeze(), R)
    d = pixel_world - rays_o[None, None]
    return d / torch.norm(d, dim=2, keepdim=True)


def sky_s3g(cube, dirs_hw3, acc=None, fill=0.0, threshold=1e-3):
    """SkyCubeMap.forward given the ray directions: [3,H,W] sky colour (sky_cubemap.py:41-87)."""
    H, W = dirs_hw3.shape[:2]
    if acc is None:
        return cube_lookup(cube, dirs_hw3).permute(2, 0, 1).clamp(0.0, 1.0)
    mask = (1 - acc[0]) > threshold
    sky = torch.full((H, W, 3), float(fill))
    sky[mask] = cube_lookup(cube, dirs_hw3[mask])
    return sky.permute(2, 0, 1).clamp(0.0, 1.0)


def blend_s3g(render, weight, sky):
    """gaussian_renderer/__init__.py:300"""
    return render * weight + sky * (1 - weight)


def blend_add(rgb, opacity, sky):
    """OmniRe/models/trainers/base.py:497"""
    return rgb + sky * (1.0 - opacity)
