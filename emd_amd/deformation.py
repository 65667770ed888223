"""The EMD deformation front-ends on the HIP path (rows a3 and a15).

`deform_network` / `Deformation` mirror S3Gaussian/scene/deformation.py (constructor argument `args` with the reference's
option names, parameter and buffer names -- reference `state_dict`s load strictly --, `forward` signature and return values,
`get_mlp_parameters` / `get_grid_parameters`, `set_aabb`), so `GaussianModel._deformation` can be swapped for it.  What changes
is how a step is computed:
  * the HexPlane lookup is one HIP launch each way (emd_amd.hexplane), and is skipped in the fine pass when
    `no_fine_hexplane_features` discards it (the reference still evaluates the 24 planes there, deformation.py:256);
  * the coarse-to-fine temporal embedding row is one HIP launch (`emd_temporal_embed_*`) instead of F.interpolate + F.grid_sample
    + an [N, dim] repeat; because the row is the same for every Gaussian it enters the first Linear as a bias, and the
    [N, 164] concat of deformation.py:250 is never materialised (addmm on column slices of the weight);
  * `poc_fre(point)` (deformation.py:486,507) is not computed: only its first three columns are ever read (:194);
  * the heads share one ReLU of the hidden feature and one first-layer GEMM;
  * every Linear's weight gradient (an [out, in] result reduced over N ~ 10^6 rows) is computed as a split batched GEMM
    (`_TallLinear`): the stock heuristics run that shape on a handful of workgroups.
The trunk + heads of a level run as ONE autograd node on the fused fp32-MFMA kernels of csrc/mlp.hip (`emd_amd.mlp.level_mlp`) when the
configuration fits them (width 64, defor_depth 1: the reference's); otherwise, or with `fused_mlp=False`, as the GEMMs described above.
Options outside the run script's configuration that
would need the absent tinycudann hash grid or the dense occupancy grid raise NotImplementedError.

`ConditionalDeformNetwork` mirrors OmniRe/models/modules.py:411-457 (same parameters); `nonrigid_deformation` is
DeformableNodes.get_deformation (OmniRe/models/nodes/deformable.py:35-47) with the encoder input written by one HIP launch.
No CPU path: the ops raise when the tensors are not on a ROCm device."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .hexplane import HexPlaneField


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeformOptions:
    """The option fields the deformation network reads, with S3Gaussian/arguments/gaussian_options.py:128-196 defaults and the
    three flags scripts/dynamic/run_dynamic_nvs.sh sets.  Any object with these attributes (the reference's BaseOptions) works."""

    def __init__(self, **kw):
        self.net_width, self.defor_depth, self.timebase_pe, self.posebase_pe = 64, 1, 4, 10
        self.scale_rotation_pe, self.opacity_pe, self.timenet_output, self.bounds, self.grid_pe = 2, 2, 32, 1.6, 0
        self.kplanes_config = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [64, 64, 64, 25]}
        self.multires = [1, 2, 4, 8]
        self.is_use_hash = self.empty_voxel = self.static_mlp = self.aggregate_feature = self.no_grid = False
        self.feat_head = True
        self.min_embeddings, self.max_embeddings, self.temporal_embedding_dim, self.gaussian_embedding_dim = 30, 150, 32, 4
        self.c2f_temporal_iter = 25000
        self.zero_temporal = self.no_c2f_temporal_embedding = self.no_coarse_deform = self.no_fine_deform = False
        self.no_temporal_embedding_dim = self.no_gaussian_embedding_dim = self.no_coarse_hexplane_features = False
        self.no_time_offset = self.no_dx = self.no_do = self.no_dshs = False
        self.apply_coarse_dx = self.apply_final_dx = True
        self.direct_add_dx = self.direct_add_ds = self.direct_add_dr = self.direct_add_do = self.direct_add_dshs = True
        self.no_ds = self.no_dr = self.no_fine_hexplane_features = True          # run_dynamic_nvs.sh
        self.fused_mlp = True        # (not a reference option) trunk + heads on the fused fp32-MFMA kernels of csrc/mlp.hip; False: rocBLAS GEMMs
        for k, v in kw.items():
            setattr(self, k, v)


class _TemporalEmbed(torch.autograd.Function):
    """weight [B, rows, dim] (or [rows, dim]), t: 1-element device tensor -> the embedding row(s) at time t after a resize to k rows."""

    @staticmethod
    def forward(ctx, weight, t, k):
        if weight.device.type != "cuda":
            raise L.EmdError("temporal_embed needs tensors on a ROCm device; there is no CPU path")
        w = weight.detach().contiguous().float()
        tt = t.detach().reshape(-1)[:1].contiguous().float()
        if tt.numel() == 0:
            raise L.EmdError("temporal_embed: empty time tensor (the reference indexes t[0, 0], deformation.py:213)")
        tables = 1 if w.dim() == 2 else w.shape[0]
        rows, dim = w.shape[-2:]
        out = torch.empty(w.shape[:-2] + (dim,), device=w.device, dtype=torch.float32)
        L.check(L.load().emd_temporal_embed_forward(w.data_ptr(), tables, rows, dim, int(k), tt.data_ptr(), out.data_ptr(), _stream()),
                "emd_temporal_embed_forward")
        ctx.save_for_backward(w, tt)
        ctx.k, ctx.t_shape = int(k), tuple(t.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        w, tt = ctx.saved_tensors
        tables = 1 if w.dim() == 2 else w.shape[0]
        rows, dim = w.shape[-2:]
        g_w = torch.zeros_like(w) if ctx.needs_input_grad[0] else None
        g_t = torch.zeros(1, device=w.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        L.check(L.load().emd_temporal_embed_backward(w.data_ptr(), tables, rows, dim, ctx.k, tt.data_ptr(),
                                                     g_out.contiguous().float().data_ptr(), L.ptr(g_w), L.ptr(g_t), _stream()),
                "emd_temporal_embed_backward")
        return g_w, (g_t.reshape(ctx.t_shape) if g_t is not None else None), None


def _first_time(time_emb):
    """`time_emb.reshape(-1)[:1]`: the level's one timestamp (deformation.py:283).  For a timestamp BROADCAST over the points (stride 0, what
    `forward_time_offset` and emd_amd.model.render hand over) the [1, 1] tensor that was expanded is taken instead of a slice of the expansion: the
    slice makes autograd fill an [N] zero column, copy the scalar into row 0 and sum N values again on the way back (three launches and 16 MB per
    level for one float; the HexPlane lookup avoids the same thing, hexplane.py get_density)."""
    if time_emb.dim() == 2 and time_emb.shape[0] > 1 and time_emb.stride(0) == 0 and time_emb.shape[1] == 1:
        base = getattr(time_emb, "_base", None)
        if base is not None and base.numel() == 1 and base.dtype == time_emb.dtype:
            return base.reshape(1)
    return time_emb.reshape(-1)[:1]


def temporal_embed(weight, t, k):
    """Row of the temporal table (resized to k rows) at time t (a 1-element tensor ON THE DEVICE: no host sync)."""
    return _TemporalEmbed.apply(weight, t, k)


def batch_quaternion_multiply(q1, q2):
    """Normalised Hamilton product (S3Gaussian/utils/graphics_utils.py:172-195)."""
    w1, x1, y1, z1 = q1.unbind(1)
    w2, x2, y2, z2 = q2.unbind(1)
    q = torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2), dim=1)
    return q / torch.norm(q, dim=1, keepdim=True)


class _TallLinear(torch.autograd.Function):
    """y = base + x W^T for a tall x [N, in] (N ~ 10^6, in/out <= a few hundred).  The forward and dL/dx are ordinary GEMMs; the
    weight gradient dY^T x is a [out, in] result reduced over N, for which the BLAS heuristics launch a handful of workgroups
    (2.6 ms at N = 2 M on MI355X) -- here the reduction is split into `SPLIT` batched slabs (0.2-0.5 ms), then summed."""
    SPLIT_ROWS = 4096

    @staticmethod
    def forward(ctx, base, x, weight):
        ctx.save_for_backward(x, weight)
        ctx.base_shape = tuple(base.shape)
        return torch.addmm(base, x, weight.t())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        g_base = g_x = g_w = None
        if ctx.needs_input_grad[0]:
            g_base = gy if len(ctx.base_shape) == 2 and ctx.base_shape[0] == gy.shape[0] else gy.sum(0)
        if ctx.needs_input_grad[1]:
            g_x = gy @ weight
        if ctx.needs_input_grad[2]:
            n, R = gy.shape[0], _TallLinear.SPLIT_ROWS
            S = n // R
            if S >= 4:
                m = S * R
                # (column slices of a wider matrix stay strided views: the row stride becomes the GEMM's leading dimension)
                g_w = torch.bmm(gy[:m].unflatten(0, (S, R)).transpose(1, 2), x[:m].unflatten(0, (S, R))).sum(0)
                if m < n:
                    g_w = g_w.addmm_(gy[m:].t(), x[m:])
            else:
                g_w = gy.t() @ x
        return g_base, g_x, g_w


def tall_linear(x, weight, bias):
    """nn.Linear's function on a tall input; `bias` may be a vector or an already accumulated [N, out] partial result."""
    return _TallLinear.apply(bias, x, weight)


class _TallLinearRelu(torch.autograd.Function):
    """relu(bias + x W^T) with the bias add and the ReLU in the GEMM's epilogue (hipBLASLt through torch._addmm_activation:
    0.67 ms instead of 1.25 ms for [2 M, 64] x [64, 192], same bits), one autograd node instead of two; backward as _TallLinear
    behind the ReLU mask taken from the saved output."""

    @staticmethod
    def forward(ctx, bias, x, weight):
        y = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        g = torch.ops.aten.threshold_backward(gy, y, 0)
        g_x = g @ weight if ctx.needs_input_grad[1] else None
        g_w = None
        if ctx.needs_input_grad[2]:
            n, R = g.shape[0], _TallLinear.SPLIT_ROWS
            S = n // R
            if S >= 4:
                m = S * R
                g_w = torch.bmm(g[:m].unflatten(0, (S, R)).transpose(1, 2), x[:m].unflatten(0, (S, R))).sum(0)
                if m < n:
                    g_w = g_w.addmm_(g[m:].t(), x[m:])
            else:
                g_w = g.t() @ x
        return (g.sum(0) if ctx.needs_input_grad[0] else None), g_x, g_w


def tall_linear_relu(x, weight, bias):
    """relu(nn.Linear) on a tall input (vector bias)."""
    return _TallLinearRelu.apply(bias, x, weight)


def _apply(seq, h):
    """nn.Sequential of Linear / ReLU layers through tall_linear; a Linear followed by a ReLU runs as one fused unit."""
    layers, i = list(seq), 0
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, nn.Linear) and i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU):
            h = tall_linear_relu(h, layer.weight, layer.bias)
            i += 2
            continue
        h = tall_linear(h, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else layer(h)
        i += 1
    return h


def _head(width, out):
    return nn.Sequential(nn.ReLU(), nn.Linear(width, width), nn.ReLU(), nn.Linear(width, out))


class Deformation(nn.Module):
    """S3Gaussian/scene/deformation.py:18-401 (parameter names and shapes identical)."""

    def __init__(self, D=8, W=256, input_ch=27, input_ch_time=9, grid_pe=0, skips=(), args=None):
        super().__init__()
        a = self.args = args
        for flag in ("is_use_hash", "empty_voxel", "static_mlp", "aggregate_feature", "no_grid"):
            if getattr(a, flag, False):
                raise NotImplementedError(f"deformation option {flag} is outside the hot-path scope (SURVEY.md section 8)")
        if grid_pe > 1:
            raise NotImplementedError("grid_pe > 1 is outside the hot-path scope")
        self.D, self.W, self.input_ch, self.input_ch_time, self.skips, self.grid_pe = D, W, input_ch, input_ch_time, list(skips), grid_pe
        self.grid = HexPlaneField(a.bounds, a.kplanes_config, a.multires)
        self.min_embeddings, self.max_embeddings = a.min_embeddings, a.max_embeddings
        self.temporal_embedding_dim, self.gaussian_embedding_dim = a.temporal_embedding_dim, a.gaussian_embedding_dim
        self.c2f_temporal_iter = a.c2f_temporal_iter
        if a.zero_temporal:
            self.weight = nn.Parameter(torch.zeros(self.max_embeddings, self.temporal_embedding_dim))
        else:
            self.weight = nn.Parameter(torch.normal(0.0, 0.01 / np.sqrt(self.temporal_embedding_dim),
                                                    size=(self.max_embeddings, self.temporal_embedding_dim)))
        if not a.no_time_offset:
            self.time_offset = nn.Parameter(torch.zeros((3, 1)))
        self.register_buffer("pos_poc", torch.FloatTensor([(2 ** i) for i in range(a.posebase_pe)]))
        self._create_net()

    # ---- construction (deformation.py:100-185)
    def _in_dim(self, with_hexplane):
        a = self.args
        return ((self.grid.feat_dim if with_hexplane else 0) + (0 if a.no_temporal_embedding_dim else self.temporal_embedding_dim) +
                (0 if a.no_gaussian_embedding_dim else self.gaussian_embedding_dim))

    def _trunk(self, in_dim):
        layers = [nn.Linear(in_dim, self.W)]
        for _ in range(self.D - 1):
            layers += [nn.ReLU(), nn.Linear(self.W, self.W)]
        return nn.Sequential(*layers)

    def _create_net(self):
        a, W = self.args, self.W
        self.feature_out = self._trunk(self._in_dim(not a.no_coarse_hexplane_features))
        self.pos_deform, self.scales_deform, self.rotations_deform = _head(W, 3), _head(W, 3), _head(W, 4)
        self.opacity_deform, self.shs_deform = _head(W, 1), _head(W, 16 * 3)
        if not a.no_fine_deform:
            self.feature_out_f = self._trunk(self._in_dim(not a.no_fine_hexplane_features))
            self.pos_deform_f, self.scales_deform_f, self.rotations_deform_f = _head(W, 3), _head(W, 3), _head(W, 4)
            self.opacity_deform_f, self.shs_deform_f = _head(W, 1), _head(W, 16 * 3)
        if a.feat_head:
            self.dino_head = nn.Sequential(nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 3))

    @property
    def get_aabb(self):
        return self.grid.get_aabb

    def set_aabb(self, xyz_max, xyz_min):
        self.grid.set_aabb(xyz_min, xyz_max)          # argument order as the reference passes it on (deformation.py:95)

    def int_lininterp(self, t, init_val, final_val, until):
        return int(init_val + (final_val - init_val) * min(max(t, 0), until) / until)

    def forward_time_offset(self, time_emb, cam_no):
        if self.args.no_time_offset:
            return time_emb
        if time_emb.dim() == 2 and time_emb.shape[0] > 1 and time_emb.stride(0) == 0:
            # one time broadcast over the points stays a broadcast (the HexPlane lookup recognises it: 1-D time tables); same values
            return (time_emb[:1] + self.time_offset[cam_no]).expand(time_emb.shape[0], -1)
        return time_emb + self.time_offset[cam_no]

    # ---- one level (deformation.py:187-296)
    def _num_rows(self, coarse, it, num_down_emb):
        if coarse:
            return num_down_emb
        if self.args.no_c2f_temporal_embedding:
            return self.max_embeddings
        if it is None:
            it = self.c2f_temporal_iter
        return self.int_lininterp(it, num_down_emb, self.max_embeddings, self.c2f_temporal_iter)

    def _feature(self, pts, time_emb, embeddings, coarse, it, num_down_emb):
        a = self.args
        trunk = self.feature_out if coarse else self.feature_out_f
        lin = trunk[0]
        Wm, bias = lin.weight, lin.bias
        use_hex = not (a.no_coarse_hexplane_features if coarse else a.no_fine_hexplane_features)
        col = self.grid.feat_dim if use_hex else 0
        if not a.no_temporal_embedding_dim:
            T = self.temporal_embedding_dim
            te = temporal_embed(self.weight, _first_time(time_emb), self._num_rows(coarse, it, num_down_emb))
            bias = torch.addmv(bias, Wm[:, col:col + T], te)           # the same row for every Gaussian: a bias, not N copies
            col += T
        if not a.no_gaussian_embedding_dim and embeddings is not None:
            h = tall_linear(embeddings, Wm[:, col:col + embeddings.shape[1]], bias)
        else:
            h = bias.expand(pts.shape[0], -1)
        if use_hex:
            h = tall_linear(self.grid(pts[:, :3], time_emb[:, :1]), Wm[:, :self.grid.feat_dim], h)
        return _apply(trunk[1:], h)

    def _heads(self, hidden, suffix, need_feat=True):
        a, W = self.args, self.W
        names = [n for n, off in (("pos_deform", a.no_dx), ("scales_deform", a.no_ds), ("rotations_deform", a.no_dr),
                                  ("opacity_deform", a.no_do), ("shs_deform", a.no_dshs)) if not off]
        heads = [getattr(self, n + suffix) for n in names]
        out = dict(dx=None, ds=None, dr=None, do=None, dshs=None, feat=None)
        if heads:
            hr = torch.relu(hidden)                                                           # once, not per head
            mid = tall_linear_relu(hr, torch.cat([h[1].weight for h in heads]), torch.cat([h[1].bias for h in heads]))
            # second layers as ONE GEMM with a block-diagonal weight [sum(out_i), heads * W]: slicing `mid` per head would make
            # autograd zero-fill, copy into and add three [N, heads * W] buffers on the way back; the zero blocks cost 0.4 ms of FLOPs
            outs = tall_linear(mid, torch.block_diag(*[h[3].weight for h in heads]), torch.cat([h[3].bias for h in heads]))
            col = 0
            for n, h in zip(names, heads):
                width = h[3].weight.shape[0]
                out[{"pos_deform": "dx", "scales_deform": "ds", "rotations_deform": "dr", "opacity_deform": "do",
                     "shs_deform": "dshs"}[n]] = outs[:, col:col + width]
                col += width
            if out["dshs"] is not None:
                out["dshs"] = out["dshs"].reshape(hidden.shape[0], 16, 3)
        if a.feat_head and need_feat:
            out["feat"] = _apply(self.dino_head, hidden)
        return out

    # ---- one level on the fused fp32-MFMA kernels (csrc/mlp.hip): trunk + every head, one autograd node
    _HEADS = (("pos_deform", "no_dx", "dx"), ("scales_deform", "no_ds", "ds"), ("rotations_deform", "no_dr", "dr"),
              ("opacity_deform", "no_do", "do"), ("shs_deform", "no_dshs", "dshs"))

    def _level_fused(self, pts, time_emb, embeddings, coarse, it, num_down_emb, need_feat=True, l1_dshs=False, l1_keys=()):
        """The same level as `_feature` + `_heads`, or None when the configuration is outside what the fused kernels serve (width 64,
        defor_depth 1, at most 128 HexPlane features, an embedding of at most 8 values, at most six heads): the GEMM path then runs."""
        from . import mlp
        a = self.args
        if not getattr(a, "fused_mlp", True) or self.D != 1 or self.W != mlp.WIDTH or pts.device.type != "cuda":
            return None
        suffix = "" if coarse else "_f"
        lin = (self.feature_out if coarse else self.feature_out_f)[0]
        use_hex = not (a.no_coarse_hexplane_features if coarse else a.no_fine_hexplane_features)
        use_emb = not a.no_gaussian_embedding_dim and embeddings is not None
        ka, kb = (self.grid.feat_dim if use_hex else 0), (embeddings.shape[1] if use_emb else 0)
        heads = [(getattr(self, n + suffix), key) for n, off, key in self._HEADS if not getattr(a, off)]
        branches = [(True, [(h[1].weight, h[1].bias)], (h[3].weight, h[3].bias)) for h, _ in heads]
        feat = a.feat_head and need_feat
        if feat:
            d = self.dino_head
            branches.append((False, [(d[0].weight, d[0].bias), (d[2].weight, d[2].bias)], (d[4].weight, d[4].bias)))
        hidden_shapes = [w.shape for _, hid, _ in branches for w, _ in hid]
        if not branches or not mlp.eligible(ka, kb, hidden_shapes, [wo.shape[0] for _, _, (wo, _) in branches]):
            return None
        Wm, bias, col = lin.weight, lin.bias, ka
        if not a.no_temporal_embedding_dim:
            T = self.temporal_embedding_dim
            te = temporal_embed(self.weight, _first_time(time_emb), self._num_rows(coarse, it, num_down_emb))
            bias = torch.addmv(bias, Wm[:, col:col + T], te)           # the same row for every Gaussian: a bias, not N copies
            col += T
        xa = self.grid(pts[:, :3], time_emb[:, :1]) if use_hex else None
        keys = [key for _, key in heads]
        l1_names = [k for k in (tuple(l1_keys) + (("dshs",) if l1_dshs else ())) if k in keys]
        l1_names = list(dict.fromkeys(l1_names))                   # (unique, in the order asked for)
        l1_heads = [keys.index(k) for k in l1_names]
        outs = mlp.level_mlp(xa, embeddings if use_emb else None, Wm, bias, 0, col, branches, l1_heads=l1_heads)
        out = dict(dx=None, ds=None, dr=None, do=None, dshs=None, feat=None)
        for (_, key), o in zip(heads, outs):
            out[key] = o.reshape(o.shape[0], 16, 3) if key == "dshs" else o
        if feat:
            out["feat"] = outs[len(branches) - 1]
        for j, k in enumerate(l1_names):
            out[k + "_abs_mean"] = outs[len(branches) + j]      # mean |residual| from the head's own kernels (regularisers of train.py:238-310)
        return out

    def forward(self, rays_pts_emb, time_emb=None, embeddings=None, is_coarse=True, iter=None, num_down_emb_c=30, num_down_emb_f=30,
                apply_deform=True, time_diff=1.0, is_train=False, need_feat=True, l1_dshs=False, l1_keys=()):
        """`l1_dshs` (not a reference argument): on the fused kernels the level's dict also carries "dshs_abs_mean" = mean |dshs|.
        `l1_keys` (likewise): residual names ("dx", "do", "ds", "dr", "dshs") whose mean |.| is formed by the head's own kernels and returned as
        "<key>_abs_mean" -- the regularisers of train.py:238-310 without an abs-mean launch each way and without the add autograd needs to join the
        regulariser's gradient with the rasterizer's (emd_amd.model.residual_abs_mean picks them up).
        `need_feat=False` (not a reference argument): the caller will not read ddict["feat"], so the feature head is not evaluated
        (its entry is None); the default evaluates it whenever `feat_head` is set, as the reference does."""
        if time_emb is None:
            raise NotImplementedError("forward_static (static_mlp) is outside the hot-path scope")
        if not apply_deform:
            return None
        n_rows = num_down_emb_c if is_coarse else num_down_emb_f
        fused = self._level_fused(rays_pts_emb, time_emb, embeddings, is_coarse, iter, n_rows, need_feat, l1_dshs, l1_keys)
        if fused is not None:
            return fused
        hidden = self._feature(rays_pts_emb, time_emb, embeddings, is_coarse, iter, n_rows)
        return self._heads(hidden, "" if is_coarse else "_f", need_feat)

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" not in n]

    def get_grid_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" in n]


def initialize_weights(m):
    """deformation.py:529-535 (the bias keeps nn.Linear's default initialisation there, too)."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight, gain=1)


class deform_network(nn.Module):
    """S3Gaussian/scene/deformation.py:402-527."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.temporal_embedding_dim, self.gaussian_embedding_dim = args.temporal_embedding_dim, args.gaussian_embedding_dim
        self.c2f_temporal_iter, self.min_embeddings = args.c2f_temporal_iter, args.min_embeddings
        self.no_coarse_deform, self.no_fine_deform = args.no_coarse_deform, args.no_fine_deform
        self.deformation_net = Deformation(W=args.net_width, D=args.defor_depth, input_ch=3 + 3 * args.posebase_pe * 2, grid_pe=args.grid_pe,
                                           input_ch_time=args.timenet_output, args=args)
        self.register_buffer("time_poc", torch.FloatTensor([(2 ** i) for i in range(args.timebase_pe)]))
        self.register_buffer("pos_poc", torch.FloatTensor([(2 ** i) for i in range(args.posebase_pe)]))
        self.register_buffer("rotation_scaling_poc", torch.FloatTensor([(2 ** i) for i in range(args.scale_rotation_pe)]))
        self.register_buffer("opacity_poc", torch.FloatTensor([(2 ** i) for i in range(args.opacity_pe)]))
        self.apply(initialize_weights)

    @property
    def get_aabb(self):
        return self.deformation_net.get_aabb

    def apply_deform(self, point, scales=None, rotations=None, opacity=None, shs=None, ddict_c=None, ddict_f=None, defer_shs=False):
        """deformation.py:439-481; the `.clone()`s of the reference are dropped where an add follows (same values, same graph).
        `defer_shs`: `shs` is returned as it came and the dshs residuals are left to the caller (the rasterizer's `shs_residuals`)."""
        a = self.args
        levels = [d for off, d in ((a.no_coarse_deform, ddict_c), (a.no_fine_deform, ddict_f)) if not off]

        def base(x, direct, off):
            return x if (direct or off) else torch.zeros_like(x)
        point_f, scales_f = base(point, a.direct_add_dx, a.no_dx), base(scales, a.direct_add_ds, a.no_ds)
        rot_f, opac_f, shs_f = base(rotations, a.direct_add_dr, a.no_dr), base(opacity, a.direct_add_do, a.no_do), base(shs, a.direct_add_dshs, a.no_dshs)
        for d in levels:
            if not a.no_dx and a.apply_final_dx:
                point_f = point_f + d["dx"]
            if not a.no_ds:
                scales_f = scales_f + d["ds"]
            if not a.no_dr:
                rot_f = batch_quaternion_multiply(rot_f, d["dr"])
            if not a.no_do:
                opac_f = opac_f + d["do"]
            if not a.no_dshs and not defer_shs:
                shs_f = shs_f + d["dshs"]
        return point_f, scales_f, rot_f, opac_f, shs_f

    def forward(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None, embeddings=None, iter=None, cam_no=None,
                time_diff=None, is_train=None, need_feat=True, fused_shs_residuals=False, fused_l1=()):
        """`fused_shs_residuals` (not in the reference): the SH residuals are NOT added to `shs`; they are returned as
        ddict["shs_residuals"] for `GaussianRasterizer(..., shs_residuals=...)`, which forms shs + dshs_c + dshs_f for the visible Gaussians
        inside its projection kernel, and each level's dict carries "dshs_abs_mean" -- mean |dshs|, the regulariser of train.py:238-310,
        whose gradient is folded into the residual's in one pass (emd_amd.model.residual_pair_l1).  Needs both levels and the default
        direct_add_dshs; otherwise the call behaves as without the flag."""
        net = self.deformation_net
        times_sel = net.forward_time_offset(times_sel, cam_no)
        lk = {"l1_dshs": True} if fused_shs_residuals else {}
        if fused_l1:                                               # (`fused_l1`, not in the reference: see Deformation.forward's l1_keys)
            lk["l1_keys"] = tuple(fused_l1)
        ddict_c = net(point, times_sel, embeddings, is_coarse=True, iter=iter, num_down_emb_c=self.min_embeddings,
                      apply_deform=not self.no_coarse_deform, time_diff=time_diff, is_train=is_train, need_feat=need_feat, **lk)
        pts = point
        # (the fine level reads its points only through the HexPlane lookup: with `no_fine_hexplane_features`, the run script's setting, the sum
        # would be formed and differentiated for nothing -- an [N, 3] add each way)
        if not self.no_coarse_deform and self.args.apply_coarse_dx and not self.args.no_fine_hexplane_features:
            pts = point + ddict_c["dx"]
        ddict_f = net(pts, times_sel, embeddings, is_coarse=False, iter=iter, num_down_emb_f=self.min_embeddings,
                      apply_deform=not self.no_fine_deform, time_diff=time_diff, is_train=is_train, need_feat=need_feat, **lk)
        a = self.args
        fuse = (fused_shs_residuals and shs is not None and not a.no_dshs and a.direct_add_dshs and not self.no_coarse_deform
                and not self.no_fine_deform and ddict_c.get("dshs") is not None and ddict_f.get("dshs") is not None)
        dd = {"coarse": ddict_c, "fine": ddict_f}
        if fuse and ddict_c.get("dshs_abs_mean") is not None and ddict_f.get("dshs_abs_mean") is not None:
            # (the fused MLP kernels formed mean |dshs| themselves and differentiate it inside the heads' backward: nothing to add here)
            dd["shs_residuals"] = [ddict_c["dshs"].reshape(shs.shape), ddict_f["dshs"].reshape(shs.shape)]
        elif fuse:
            from .model import residual_pair_l1
            rc, rf, l1c, l1f = residual_pair_l1(ddict_c["dshs"].reshape(shs.shape), ddict_f["dshs"].reshape(shs.shape))
            ddict_c["dshs"], ddict_f["dshs"] = rc, rf
            ddict_c["dshs_abs_mean"], ddict_f["dshs_abs_mean"] = l1c, l1f
            dd["shs_residuals"] = [rc, rf]
        out = self.apply_deform(point, scales, rotations, opacity, shs, ddict_c, ddict_f, defer_shs=fuse)
        return (*out, dd)

    def get_mlp_parameters(self):
        return self.deformation_net.get_mlp_parameters()

    def get_grid_parameters(self):
        return self.deformation_net.get_grid_parameters()


DeformNetwork = deform_network


# ------------------------------------------------------------------------------------------------------------ OmniRe (a15)
class _DeformInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, point_ids, inst_size, inst_embed, t, num_freqs_x, num_freqs_t):
        if means.device.type != "cuda":
            raise L.EmdError("deform_input needs tensors on a ROCm device; there is no CPU path")
        lib = L.load()
        a = L.EmdDeformInArgs()
        N, E = means.shape[0], inst_embed.shape[1]
        keep = [means.detach().contiguous().float(), None if point_ids is None else point_ids.to(torch.int32).contiguous(),
                None if inst_size is None else inst_size.detach().contiguous().float(), inst_embed.detach().contiguous().float(),
                t.detach().reshape(-1)[:1].contiguous().float()]
        width = lib.emd_deform_input_width(num_freqs_x, num_freqs_t, E)
        out = torch.empty(N, width, device=means.device, dtype=torch.float32)
        a.num_points, a.num_freqs_x, a.num_freqs_t, a.embed_dim, a.ld = N, num_freqs_x, num_freqs_t, E, width
        a.means, a.point_ids, a.inst_size, a.inst_embed, a.t, a.out = (keep[0].data_ptr(), L.ptr(keep[1]), L.ptr(keep[2]), keep[3].data_ptr(),
                                                                       keep[4].data_ptr(), out.data_ptr())
        L.check(lib.emd_deform_input_forward(C.byref(a), _stream()), "emd_deform_input_forward")
        ctx.ids, ctx.embed_shape, ctx.col0 = keep[1], tuple(inst_embed.shape), width - E
        return out

    @staticmethod
    def backward(ctx, g):
        g_embed = None
        if ctx.needs_input_grad[3]:
            A, E = ctx.embed_shape
            if ctx.ids is None:
                g_embed = g[:, ctx.col0:].contiguous()
            else:
                g = g.contiguous().float()
                g_embed = torch.zeros(A, E, device=g.device, dtype=torch.float32)
                L.check(L.load().emd_deform_input_backward(g.shape[0], E, g.shape[1], ctx.col0, ctx.ids.data_ptr(), g.data_ptr(),
                                                           g_embed.data_ptr(), _stream()), "emd_deform_input_backward")
        return None, None, None, g_embed, None, None, None          # positions and time are detached in the reference


class ConditionalDeformNetwork(nn.Module):
    """OmniRe/models/modules.py:411-457 (same parameters: `linear`, `gaussian_warp`, `gaussian_rotation`, `gaussian_scaling`)."""

    def __init__(self, D=8, W=256, input_ch=3, embed_dim=10, x_multires=10, t_multires=10, deform_quat=True, deform_scale=True):
        super().__init__()
        self.D, self.W, self.embed_dim, self.deform_quat, self.deform_scale = D, W, embed_dim, deform_quat, deform_scale
        self.x_multires, self.t_multires = x_multires, t_multires
        self.skips = [D // 2]
        self.input_ch = 3 * (1 + 2 * x_multires) + (1 + 2 * t_multires) + embed_dim
        self.linear = nn.ModuleList([nn.Linear(self.input_ch, W)] +
                                    [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W) for i in range(D - 1)])
        self.gaussian_warp = nn.Linear(W, 3)
        if deform_quat:
            self.gaussian_rotation = nn.Linear(W, 4)
        if deform_scale:
            self.gaussian_scaling = nn.Linear(W, 3)

    def trunk(self, h0):
        """The D layers on an encoded input [N, input_ch]; the skip concat of modules.py:446-447 becomes a second addmm."""
        h, skipped = h0, False
        for i, lin in enumerate(self.linear):
            if skipped:
                h = tall_linear(h, lin.weight[:, self.input_ch:], tall_linear(h0, lin.weight[:, :self.input_ch], lin.bias))
                h = torch.relu(h)
            else:
                h = tall_linear_relu(h, lin.weight, lin.bias)
            skipped = i in self.skips
        if skipped:                                      # (D - 1 in skips: the heads would see the concat)
            h = torch.cat([h0, h], -1)
        out = lambda lin: tall_linear(h, lin.weight, lin.bias)
        return (out(self.gaussian_warp), out(self.gaussian_rotation) if self.deform_quat else None,
                out(self.gaussian_scaling) if self.deform_scale else None)

    def forward(self, x, t, condition):
        """Reference signature: x [N,3] normalised positions, t [N,1] (one frame: every row equal), condition [N, embed_dim]."""
        return self.trunk(_DeformInput.apply(x, None, None, condition, t, self.x_multires, self.t_multires))


def nonrigid_deformation(network, local_means, point_ids, instances_size, instances_embedding, t):
    """DeformableNodes.get_deformation (OmniRe/models/nodes/deformable.py:35-47): gather of the actor's embedding and height,
    x = mean.data / height * 2, both frequency encodings and the concat in one launch, then the network's layers.
    t: 1-element device tensor (normalized_timestamps[cur_frame])."""
    ids = point_ids[..., 0] if point_ids.dim() == 2 else point_ids
    return network.trunk(_DeformInput.apply(local_means, ids, instances_size, instances_embedding, t, network.x_multires, network.t_multires))
