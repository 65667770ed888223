"""Differentiable torch restatement of the hot path -- TEST INFRASTRUCTURE ONLY (see cpu_oracle.py).

Two jobs:
  1. gradient oracle: the forward below is written with plain torch ops (fp64 by default) so that
     torch.autograd yields dL/d{means3D, shs, opacities, scales, rotations, actor poses, residuals};
     the analytic backward of raster_oracle.c and of the HIP kernels is checked against it on small
     scenes.  Discrete structure (visibility, tile lists) is taken from the C oracle.
  2. `reference_projection_cpu`: the reference's pure-PyTorch projection + cov3D + SH forward
     (S3Gaussian/utils/graphics_utils.py:42-49, scene/gaussian_model.py:34-38,
      utils/general_utils.py:231-277, gaussian_renderer/__init__.py:19-25, utils/sh_utils.py:57-112)
     restated op for op; it is the `cpu_baseline` leg of bench.py.

Upstream-behaviour choices mirrored here (and in raster_oracle.c):
  - alpha = min(0.99, o G) is differentiated straight through the clamp;
  - a view-space x/z (y/z) clamped to 1.3 tan(fov/2) is treated as a constant in backward;
  - the +0.3 px dilation, the radius and the tile rectangle carry no gradient.
"""
import math

import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, dirs):
    """sh: [N, C, K], dirs [N,3] unit.  Same polynomial order as S3Gaussian/utils/sh_utils.py:57-112."""
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5]
                      + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + C2[3] * xz * sh[..., 7]
                      + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10]
                          + C3[2] * y * (4 * zz - xx - yy) * sh[..., 11]
                          + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                          + C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14]
                          + C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


def build_rotation(q):
    """Unit quaternion (w,x,y,z) -> R.  general_utils.py:245-266 without the renormalisation."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)
    return R.reshape(-1, 3, 3)


def quat_mult(a, b):
    """basics.py:100-110"""
    w1, x1, y1, z1 = a.unbind(-1)
    w2, x2, y2, z2 = b.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)


def covariance_from_scaling_rotation(scales, mod, rots):
    """gaussian_model.py:34-38: L = R diag(mod s); Sigma = L L^T; [xx,xy,xz,yy,yz,zz]"""
    R = build_rotation(rots)
    L = R * (mod * scales)[:, None, :]
    Sg = L @ L.transpose(1, 2)
    return torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], -1)


def geom_transform_points(points, M):
    """graphics_utils.py:42-49"""
    ones = torch.ones(points.shape[0], 1, dtype=points.dtype, device=points.device)
    hom = torch.cat([points, ones], 1) @ M
    return hom[:, :3] / (hom[:, 3:] + 0.0000001)


def reference_projection_cpu(means3D, scales, rots, shs, viewmatrix, projmatrix, campos, sh_degree):
    """The reference's only in-Python part of the path (BASELINE config 1): projection + cov3D + SH colour.
    rots are normalised inside, exactly as general_utils.build_rotation does."""
    p_ndc = geom_transform_points(means3D, projmatrix)
    p_view = geom_transform_points(means3D, viewmatrix)
    norm = torch.sqrt(rots[:, 0] * rots[:, 0] + rots[:, 1] * rots[:, 1] + rots[:, 2] * rots[:, 2]
                      + rots[:, 3] * rots[:, 3])
    cov = covariance_from_scaling_rotation(scales, 1.0, rots / norm[:, None])
    shs_view = shs.transpose(1, 2)
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(eval_sh(sh_degree, shs_view, d) + 0.5, 0.0)
    return p_ndc, p_view, cov, rgb


def motion_transform(means, quats, opac, actor_id, pose, rdx=None, rdq=None):
    """rigid.py:478-568 with per-point gather of a per-actor pose table (q_mean, trans, valid, q_rot)."""
    m = means if rdx is None else means + rdx
    dyn = actor_id >= 0
    a = actor_id.clamp_min(0).long()
    P = pose[a]
    R = build_rotation(P[:, 0:4])
    wm = torch.where(dyn[:, None], (R @ m[:, :, None])[:, :, 0] + P[:, 4:7], m)
    wq = None
    if quats is not None:
        ql = quats if rdq is None else quats + rdq
        qn = ql / ql.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        p = quat_mult(P[:, 8:12], qn)
        p = p / p.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        wq = torch.where(dyn[:, None], p, quats)
    wo = None
    if opac is not None:
        wo = torch.where(dyn, opac * P[:, 7], opac)
    return wm, wq, wo


class TorchSettings:
    def __init__(self, H, W, tanfovx, tanfovy, bg, viewmatrix, projmatrix, sh_degree, campos, scale_modifier=1.0,
                 near_plane=0.2, dtype=torch.float64):
        self.H, self.W = int(H), int(W)
        self.tanfovx, self.tanfovy = float(tanfovx), float(tanfovy)
        t = lambda a: torch.as_tensor(np.asarray(a, np.float64), dtype=dtype)
        self.bg, self.view, self.proj, self.campos = t(bg), t(viewmatrix).reshape(4, 4), t(projmatrix).reshape(4, 4), t(campos)
        self.sh_degree, self.scale_modifier, self.near_plane = int(sh_degree), float(scale_modifier), float(near_plane)


def project(S, means3D, shs, colors_precomp, opac, scales, rots, cov3D_precomp, clamp01=False):
    """Differentiable K1 for all Gaussians (visibility is applied by the caller)."""
    V, P = S.view, S.proj
    N = means3D.shape[0]
    ones = torch.ones(N, 1, dtype=means3D.dtype)
    hom = torch.cat([means3D, ones], 1)
    t = hom @ V
    tx, ty, tz = t[:, 0], t[:, 1], t[:, 2]
    h = hom @ P
    pw = 1.0 / (h[:, 3] + 0.0000001)
    px, py = h[:, 0] * pw, h[:, 1] * pw
    if cov3D_precomp is None:
        c6 = covariance_from_scaling_rotation(scales, S.scale_modifier, rots)
    else:
        c6 = cov3D_precomp
    Sg = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]],
                     -1).reshape(N, 3, 3)
    fx, fy = S.W / (2.0 * S.tanfovx), S.H / (2.0 * S.tanfovy)
    limx, limy = 1.3 * S.tanfovx, 1.3 * S.tanfovy
    txtz, tytz = tx / tz, ty / tz
    clx = (txtz < -limx) | (txtz > limx)
    cly = (tytz < -limy) | (tytz > limy)
    cx = torch.where(clx, (txtz.clamp(-limx, limx) * tz).detach(), tx)
    cy = torch.where(cly, (tytz.clamp(-limy, limy) * tz).detach(), ty)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * cx) / (tz * tz), zero, fy / tz, -(fy * cy) / (tz * tz)], -1).reshape(N, 2, 3)
    Wv = V[:3, :3].t()
    M = J @ Wv
    cov = M @ Sg @ M.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], -1)
    ix = ((px + 1.0) * S.W - 1.0) * 0.5
    iy = ((py + 1.0) * S.H - 1.0) * 0.5
    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - S.campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = eval_sh(S.sh_degree, shs.transpose(1, 2), d) + 0.5
        rgb = rgb.clamp(0.0, 1.0) if clamp01 else rgb.clamp_min(0.0)
    return torch.stack([ix, iy], -1), tz, conic, rgb


def composite(S, ids, ranges, means2D, depth, conic, opac, rgb, normal=None):
    """Differentiable K6 following the tile lists (ids/ranges from the C oracle)."""
    H, W = S.H, S.W
    gx = (W + 15) // 16
    dt = means2D.dtype
    C = 3 + 1 + (3 if normal is not None else 0)
    feats = torch.cat([rgb, depth[:, None]] + ([normal] if normal is not None else []), 1)
    out = torch.zeros(C, H, W, dtype=dt)
    alpha_img = torch.zeros(H, W, dtype=dt)
    T_img = torch.ones(H, W, dtype=dt)
    ids = torch.as_tensor(ids.astype(np.int64))
    for t in range(ranges.shape[0]):
        s, e = int(ranges[t, 0]), int(ranges[t, 1])
        y0, x0 = (t // gx) * 16, (t % gx) * 16
        y1, x1 = min(y0 + 16, H), min(x0 + 16, W)
        if y1 <= y0 or x1 <= x0:
            continue
        ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing="ij")
        T = torch.ones_like(xs)
        alive = torch.ones_like(xs, dtype=torch.bool)
        acc = torch.zeros(C, *xs.shape, dtype=dt)
        for k in range(s, e):
            g = ids[k]
            dx, dy = means2D[g, 0] - xs, means2D[g, 1] - ys
            power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
            a_raw = opac[g] * torch.exp(power)
            alpha = a_raw - (a_raw - 0.99).clamp_min(0.0).detach()  # straight-through min(0.99, .)
            ok = alive & (power <= 0) & (alpha >= 1.0 / 255.0)
            test_T = T * (1 - alpha)
            stop = ok & (test_T < 0.0001)
            alive = alive & ~stop
            ok = ok & ~stop
            w = torch.where(ok, alpha * T, torch.zeros_like(T))
            acc = acc + feats[g][:, None, None] * w[None]
            T = torch.where(ok, test_T, T)
        out[:, y0:y1, x0:x1] = acc
        T_img[y0:y1, x0:x1] = T
    color = out[:3] + T_img[None] * S.bg[:, None, None]
    res = dict(color=color, depth=out[3:4], alpha=(1 - T_img)[None])
    if normal is not None:
        res["normal"] = out[4:7]
    return res


def render(S, sc_np, oracle_pre, oracle_bin, flags=0, dtype=torch.float64):
    """Full differentiable forward on a cpu_oracle.Scene; returns (outputs dict, leaves dict)."""
    tt = lambda a, rg=True: None if a is None else torch.tensor(np.asarray(a, np.float64), dtype=dtype, requires_grad=rg)
    L = dict(means3D=tt(sc_np.means3D), shs=tt(sc_np.shs), colors=tt(sc_np.colors_precomp),
             opacities=tt(sc_np.opacities), scales=tt(sc_np.scales), rotations=tt(sc_np.rotations),
             cov3D=tt(sc_np.cov3D_precomp), actor_pose=tt(sc_np.actor_pose), residual_dx=tt(sc_np.residual_dx),
             residual_dq=tt(sc_np.residual_dq))
    means, rots, opac = L["means3D"], L["rotations"], L["opacities"]
    if flags & 2:
        aid = torch.as_tensor(sc_np.actor_id.astype(np.int64))
        means, rots, opac = motion_transform(means, rots, opac, aid, L["actor_pose"], L["residual_dx"], L["residual_dq"])
    m2d, depth, conic, rgb = project(S, means, L["shs"], L["colors"], opac, L["scales"], rots, L["cov3D"],
                                     clamp01=bool(flags & 16))
    vis = torch.as_tensor(oracle_pre["radii"] > 0)
    normal = None
    if flags & 1:
        normal = torch.as_tensor(oracle_pre["normal"].astype(np.float64), dtype=dtype)
    # mean2D leaf in pixel units so that its gradient can be read like viewspace_points.grad
    m2d = m2d + 0
    m2d.retain_grad()
    out = composite(S, oracle_bin["ids"], oracle_bin["ranges"], m2d, depth, conic, opac, rgb, normal)
    out["means2D_pix"] = m2d
    out["visible"] = vis
    return out, L


def track_offsets(weight, heads, frame, num_frames, embeddings, point_ids, step, min_embeddings=30, max_embeddings=150,
                  c2f_temporal_iter=25000):
    """CPU restatement of the per-actor learned track offsets (OmniRe/models/nodes/rigid.py:147-246), batched over actors:
    weight [A, max_embeddings, D] temporal tables; heads = dict name -> (W, b) of track_rot_c/f [1, D+E], track_trans_c/f [3, D+E];
    -> (track_trans [A,3], track_rot [A,4]).  TEST INFRASTRUCTURE: the checker of emd_amd.motion.TrackOffsetHeads."""
    import torch.nn.functional as F
    A, _, fdim = weight.shape

    def temporal_embed(t, k):
        emb = F.interpolate(weight[:, None], size=(k, fdim), mode="bilinear", align_corners=True)[:, 0]       # [A,k,D]
        y = float(t) * (k - 1)
        y0 = min(max(int(y // 1), 0), k - 1)
        y1 = min(y0 + 1, k - 1)
        w = y - y0
        return emb[:, y0] * (1 - w) + emb[:, y1] * w

    t = (frame - 0) / (num_frames - 1 - 0)
    ids = point_ids.long()
    cnt = torch.zeros(A).index_add_(0, ids, torch.ones_like(ids, dtype=embeddings.dtype))
    mean_emb = torch.zeros(A, embeddings.shape[1], dtype=embeddings.dtype).index_add_(0, ids, embeddings) / cnt[:, None]
    k_f = int(min_embeddings + (max_embeddings - min_embeddings) * min(max(step, 0), c2f_temporal_iter) / c2f_temporal_iter)
    h_c = torch.cat([temporal_embed(t, min_embeddings), mean_emb], -1)
    h_f = torch.cat([temporal_embed(t, k_f), mean_emb], -1)
    lin = lambda name, h: h @ heads[name][0].t() + heads[name][1]
    trans = lin("track_trans_c", h_c) + lin("track_trans_f", h_f)
    th_c, th_f = lin("track_rot_c", h_c)[:, 0], lin("track_rot_f", h_f)[:, 0]
    z = torch.zeros_like(th_c)
    a = torch.stack([torch.cos(th_c), z, z, torch.sin(th_c)], -1)
    b = torch.stack([torch.cos(th_f), z, z, torch.sin(th_f)], -1)
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    rot = torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                       aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)
    return trans, rot
