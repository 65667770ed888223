"""CPU restatement of the EMD deformation front-ends (rows a3, a15).  TEST INFRASTRUCTURE ONLY.

S3Gaussian residual network (self-supervised EMD), S3Gaussian/scene/deformation.py:
    t'      = times + time_offset[cam_no]                                                    (:325-328)
    TE_k(t) = row t of the [max_embeddings, dim] table resized to k rows                      (:208-221)
              (F.interpolate bilinear align_corners, then F.grid_sample bilinear align_corners reflection)
    coarse  : h = Linear(cat[HexPlane(xyz, t') | TE_30(t') | emb])                             (:223-252, :254-258, :298-309)
    fine    : h = Linear(cat[(HexPlane(xyz + dx_c, t') unless no_fine_hexplane_features) | TE_k(t') | emb]),
              k = int(30 + (max - 30) * clamp(iter, 0, until) / until)                         (:205-206, :243, :511-512)
    heads   : Sequential(ReLU, Linear(W, W), ReLU, Linear(W, out)) for dx[3], ds[3], dr[4], do[1], dshs[16,3], feat (:135-185, :339-386)
    final   = base + coarse + fine (quaternions composed by the normalised Hamilton product)  (:439-481)
OmniRe learned residual, OmniRe/models/modules.py:318-366 (frequency encoder) and :411-457 (ConditionalDeformNetwork), fed by
DeformableNodes.get_deformation (models/nodes/deformable.py:35-47).

The interpolation steps are written out index by index (not through F.interpolate / F.grid_sample) so that they check the HIP
kernel's arithmetic independently; gradients come from torch autograd over these explicit gathers.  Pinned by
tests/golden/s3g_deform.npz and tests/golden/or_deform.npz (the reference's own modules run on CPU)."""
import torch

from .hexplane_oracle import hexplane_features


def _reflect_clip(v, size):
    """grid_sample's reflect_coordinates(0, 2(size-1)) followed by clip_coordinates, with its gradient conventions
    (slope -1 on odd flips, 0 where clipped)."""
    if size <= 1:
        return v * 0.0
    span = float(size - 1)
    a = v.abs()
    flips = torch.floor(a / span)
    extra = torch.fmod(a, span)
    r = torch.where(flips.long() % 2 == 0, extra, span - extra)
    inside = (r > 0) & (r < span)
    return torch.where(inside, r, r.detach().clamp(0.0, span))


def temporal_embed(weight, k, t):
    """weight [rows, dim]; k rows after the resize; t scalar tensor -> [dim]."""
    rows, dim = weight.shape
    scale = (rows - 1) / (k - 1) if k > 1 else 0.0
    src = torch.arange(k, dtype=torch.float32) * torch.tensor(scale, dtype=torch.float32)
    h0 = src.floor().long().clamp(max=rows - 1)
    h1 = h0 + (h0 < rows - 1).long()
    l1 = (src - h0.float()).clamp(0.0, 1.0)
    resized = weight[h0] * (1.0 - l1)[:, None] + weight[h1] * l1[:, None]          # [k, dim]
    iy = _reflect_clip((((t - 0.5) * 2.0) + 1.0) * 0.5 * (k - 1), k)
    y0 = int(torch.floor(iy.detach()))
    wy1 = iy - y0
    row0 = resized[min(y0, k - 1)]
    row1 = resized[y0 + 1] if y0 + 1 <= k - 1 else torch.zeros(dim)
    gx = (torch.arange(dim, dtype=torch.float32) / (dim - 1) - 0.5) * 2.0 if dim > 1 else -torch.ones(1)
    ix = _reflect_clip((gx + 1.0) * 0.5 * (dim - 1), dim)
    x0 = ix.floor().long()
    wx1 = ix - x0.float()
    x1 = x0 + 1
    inx1 = (x1 <= dim - 1).float()
    x1c = x1.clamp(max=dim - 1)

    def along_x(row):
        return row[x0] * (1.0 - wx1) + row[x1c] * inx1 * wx1
    return along_x(row0) * (1.0 - wy1) + along_x(row1) * wy1


def _head(sd, prefix, h):
    h = torch.relu(h) @ sd[prefix + ".1.weight"].t() + sd[prefix + ".1.bias"]
    return torch.relu(h) @ sd[prefix + ".3.weight"].t() + sd[prefix + ".3.bias"]


def quaternion_multiply(q1, q2):
    """Normalised Hamilton product, S3Gaussian/utils/graphics_utils.py:172-195."""
    w1, x1, y1, z1 = q1.unbind(1)
    w2, x2, y2, z2 = q2.unbind(1)
    q = torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2), dim=1)
    return q / q.norm(dim=1, keepdim=True)


def fine_rows(it, min_embeddings, max_embeddings, until):
    return int(min_embeddings + (max_embeddings - min_embeddings) * min(max(it, 0), until) / until)


def s3g_deform(sd, opts, point, scales, rotations, opacity, shs, times_sel, embeddings, it, cam_no):
    """sd: the reference state_dict (tensors, reference key names); opts: dict of the flags used below.
    Returns (point, scales, rotations, opacity, shs, ddict) like deform_network.forward."""
    P = "deformation_net."
    S = len(opts["multires"])
    planes = [[sd[f"{P}grid.grids.{s}.{p}"] for p in range(6)] for s in range(S)]
    aabb = sd[P + "grid.aabb"]
    t = times_sel if opts.get("no_time_offset") else times_sel + sd[P + "time_offset"][cam_no]
    until = opts["c2f_temporal_iter"]
    if it is None:
        it = until
    ddict = {}
    pts = point
    for lvl, suffix in (("coarse", ""), ("fine", "_f")):
        coarse = lvl == "coarse"
        feats = []
        if not (opts.get("no_coarse_hexplane_features") if coarse else opts.get("no_fine_hexplane_features")):
            feats.append(hexplane_features(pts, t, aabb, planes))
        k = opts["min_embeddings"] if coarse else fine_rows(it, opts["min_embeddings"], opts["max_embeddings"], until)
        feats.append(temporal_embed(sd[P + "weight"], k, t[0, 0])[None].expand(point.shape[0], -1))
        feats.append(embeddings)
        h = torch.cat(feats, dim=-1) @ sd[f"{P}feature_out{suffix}.0.weight"].t() + sd[f"{P}feature_out{suffix}.0.bias"]
        d = {"dx": _head(sd, f"{P}pos_deform{suffix}", h),
             "ds": None if opts.get("no_ds") else _head(sd, f"{P}scales_deform{suffix}", h),
             "dr": None if opts.get("no_dr") else _head(sd, f"{P}rotations_deform{suffix}", h),
             "do": _head(sd, f"{P}opacity_deform{suffix}", h),
             "dshs": _head(sd, f"{P}shs_deform{suffix}", h).reshape(-1, 16, 3), "feat": None}
        if opts.get("feat_head"):
            f = h
            for i in (0, 2, 4):
                f = f @ sd[f"{P}dino_head.{i}.weight"].t() + sd[f"{P}dino_head.{i}.bias"]
                if i < 4:
                    f = torch.relu(f)
            d["feat"] = f
        ddict[lvl] = d
        if coarse:
            pts = point + d["dx"]                                   # apply_coarse_dx
    c, f = ddict["coarse"], ddict["fine"]
    point_f = point + c["dx"] + f["dx"]
    scales_f = scales if opts.get("no_ds") else scales + c["ds"] + f["ds"]
    rot_f = rotations if opts.get("no_dr") else quaternion_multiply(quaternion_multiply(rotations, c["dr"]), f["dr"])
    return point_f, scales_f, rot_f, opacity + c["do"] + f["do"], shs + c["dshs"] + f["dshs"], ddict


# ---------------------------------------------------------------------------------------------------- OmniRe (a15)
def frequency_encode(x, num_freqs):
    """[x, sin(x 2^0), cos(x 2^0), sin(x 2^1), ...]  (OmniRe/models/modules.py:318-366, include_input, log sampling)."""
    outs = [x]
    for f in range(num_freqs):
        outs += [torch.sin(x * float(2 ** f)), torch.cos(x * float(2 ** f))]
    return torch.cat(outs, dim=-1)


def deform_input(means, point_ids, inst_size, inst_embed, t, num_freqs_x=10, num_freqs_t=10):
    """The encoder input of DeformableNodes.get_deformation (models/nodes/deformable.py:35-47)."""
    x = means.detach() / inst_size[point_ids][:, 2:3] * 2
    tt = t.reshape(1, 1).expand(means.shape[0], 1)
    return torch.cat([frequency_encode(x, num_freqs_x), frequency_encode(tt, num_freqs_t), inst_embed[point_ids]], dim=-1)


def conditional_deform(sd, h0, D=8, skips=(4,)):
    """ConditionalDeformNetwork.forward on an already encoded input h0 (models/modules.py:439-457)."""
    h = h0
    for i in range(D):
        h = torch.relu(h @ sd[f"linear.{i}.weight"].t() + sd[f"linear.{i}.bias"])
        if i in skips:
            h = torch.cat([h0, h], dim=-1)
    out = [h @ sd["gaussian_warp.weight"].t() + sd["gaussian_warp.bias"]]
    for name in ("gaussian_rotation", "gaussian_scaling"):
        out.append(h @ sd[name + ".weight"].t() + sd[name + ".bias"] if name + ".weight" in sd else None)
    return tuple(out)
