"""CPU restatement of the HexPlane feature lookup of the EMD / S3Gaussian deformation front-end.
TEST INFRASTRUCTURE ONLY.

Follows S3Gaussian/scene/hexplane.py:18-110,150-183:
    pts  -> normalize_aabb(pts, aabb) = (pts - aabb[0]) * (2 / (aabb[1] - aabb[0])) - 1          (:18-19)
    q    =  cat(pts_n, timestamps)                                                                 (:166)
    for every scale s (spatial resolutions multiplied by multires[s], time resolution kept):       (:127-137)
        feat_s = prod over the 6 coordinate pairs (0,1),(0,2),(0,3),(1,2),(1,3),(2,3) of
                 grid_sample(plane[s][pair] as [1,C,res[pair[1]],res[pair[0]]], q[..., pair],
                             align_corners=True, mode='bilinear', padding_mode='border')          (:20-46, :88-100)
    out  =  cat over scales                                                                        (:102-109)
Gradients come from torch autograd (grid_sample backward).  Pinned by tests/golden/s3g_hexplane.npz (the reference's own
HexPlaneField on CPU, values and gradients w.r.t. the planes and the points)."""
import itertools

import torch
import torch.nn.functional as F

PAIRS = list(itertools.combinations(range(4), 2))


def hexplane_features(pts, timestamps, aabb, planes):
    """pts [N,3], timestamps [N,1], aabb [2,3], planes[s][p] = [1,C,res_h,res_w] (reference layout) -> [N, S*C]."""
    q = (pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0
    q = torch.cat((q, timestamps), dim=-1)
    outs = []
    for scale in planes:
        feat = 1.0
        for ci, pair in enumerate(PAIRS):
            grid = scale[ci]
            coords = q[:, list(pair)].view(1, 1, -1, 2)
            interp = F.grid_sample(grid, coords, align_corners=True, mode="bilinear", padding_mode="border")
            feat = feat * interp.view(grid.shape[1], -1).t()
        outs.append(feat)
    return torch.cat(outs, dim=-1)
