"""HexPlane feature field on the HIP path.

`HexPlaneField` mirrors S3Gaussian/scene/hexplane.py:112-183 (constructor arguments, `grids` ParameterList layout
[scale][plane] = [1, C, res_h, res_w], initialisation, `aabb`, `set_aabb`, `get_density`, `forward`), so checkpoints and the
deformation network that owns it are untouched; the lookup itself -- per scale six grid_samples, five products, a concat, and
their backward -- is one HIP launch each way (`emd_hexplane_forward/backward`).  No CPU path."""
import ctypes as C
import itertools

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L

PAIRS = list(itertools.combinations(range(4), 2))


class _HexLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, times, aabb, res, order, *planes):
        # `order`: None, an int32 permutation, or a VisitingOrders record (the permutation + the per-plane orders of the fine scales)
        plane_orders = order if isinstance(order, VisitingOrders) else None
        if plane_orders is not None:
            order = plane_orders.order
        if pts.device.type != "cuda":
            raise L.EmdError("HexPlane lookup needs tensors on a ROCm device; there is no CPU path")
        lib = L.load()
        S = len(planes) // 6
        Cc = planes[0].shape[1]
        N = pts.shape[0]
        # `times`: [N, 1], or ONE timestamp [1, 1] shared by all points (HexPlaneField.get_density passes the base of a broadcast): the
        # kernels then read times[0] (EmdHexArgs.times_broadcast), the forward takes its 1-D time tables, and the backward returns the
        # gradient of that one timestamp as a sum formed inside the kernel (EmdHexGrads.dL_dtime_sum) -- no [N] column either way
        bcast = times.numel() == 1 and N > 1
        pts_c, times_c = pts.detach().contiguous().float(), times.detach().reshape(-1).contiguous().float()
        # channel-last copies: a tap becomes one contiguous C x 4-byte row
        cl = [p.detach()[0].permute(1, 2, 0).contiguous().float() for p in planes]
        a = L.EmdHexArgs()
        a.num_points, a.channels, a.num_scales, a.times_broadcast = N, Cc, S, int(bcast)
        for s in range(S):
            for k in range(4):
                a.res[s][k] = res[s][k]
            for p in range(6):
                a.planes[s][p] = cl[s * 6 + p].data_ptr()
        a.pts, a.times = pts_c.data_ptr(), times_c.data_ptr()
        a.order = L.ptr(order)                    # visiting order (int32 permutation) or None
        # One timestamp for all points is uniform by construction, no device read needed: the forward blends the time planes' two time
        # rows into 1-D tables once and reads two taps instead of four there (EmdHexArgs.time_tables)
        tables = None
        if bcast and Cc in (16, 32):
            tables = torch.empty(Cc * sum(int(r[0]) + int(r[1]) + int(r[2]) for r in res[:S]), device=pts.device, dtype=torch.float32)
            a.time_tables = tables.data_ptr()
        for k in range(6):
            a.aabb[k] = aabb[k]
        out = torch.empty(N, S * Cc, device=pts.device, dtype=torch.float32)
        a.out = out.data_ptr()
        L.check(lib.emd_hexplane_forward(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emd_hexplane_forward")
        a.time_tables, a.out = None, None         # (forward-only pointers: `tables` and `out` are not kept alive by ctx, the backward reads neither)
        ctx.args, ctx.keep = a, (pts_c, times_c, cl, order, plane_orders)
        ctx.shapes = [tuple(p.shape) for p in planes]
        ctx.times_shape = tuple(times.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = L.load()
        a = ctx.args
        S, Cc = a.num_scales, a.channels
        g_out = g_out.contiguous().float()
        g = L.EmdHexGrads()
        g.dL_dout = g_out.data_ptr()
        need_planes = any(ctx.needs_input_grad[5:])
        gcl = []
        if need_planes:
            flat = torch.zeros(sum(int(np.prod(s)) for s in ctx.shapes), device=g_out.device, dtype=torch.float32)   # one fill
            off = 0
            for i, shp in enumerate(ctx.shapes):
                n = int(np.prod(shp))
                gcl.append(flat[off:off + n].view(shp[2], shp[3], shp[1]))
                g.dL_dplanes[i // 6][i % 6] = gcl[-1].data_ptr()
                off += n
        d_pts = torch.empty(a.num_points, 3, device=g_out.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        g.dL_dpts = L.ptr(d_pts)
        d_times = None
        if ctx.needs_input_grad[1]:
            if a.times_broadcast:
                d_times = torch.zeros(ctx.times_shape, device=g_out.device, dtype=torch.float32)      # one float: the kernel adds the sum
                g.dL_dtime_sum = d_times.data_ptr()
            else:
                d_times = torch.empty(ctx.times_shape, device=g_out.device, dtype=torch.float32)
                g.dL_dtimes = d_times.data_ptr()
        po = ctx.keep[4]
        rows = None
        if po is not None and po.defer_mask and need_planes and a.num_points * Cc * 4 < 2 ** 32:
            for k in range(3):
                g.order2d[k], g.pos2d[k] = po.order2d[k].data_ptr(), po.pos2d[k].data_ptr()
            rows = torch.empty(bin(po.defer_mask).count("1") * 3 * a.num_points * Cc + 6 * a.num_points, device=g_out.device, dtype=torch.float32)
            g.defer_rows, g.defer_mask = rows.data_ptr(), po.defer_mask
        L.check(lib.emd_hexplane_backward(C.byref(a), C.byref(g), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                "emd_hexplane_backward")
        grads = [t.permute(2, 0, 1)[None] for t in gcl] if need_planes else [None] * len(ctx.shapes)   # channel-last, like the planes
        return (d_pts, d_times, None, None, None, *grads)


def init_grid_param(grid_nd, in_dim, out_dim, reso, a=0.1, b=0.5):
    """S3Gaussian/scene/hexplane.py:48-70"""
    assert in_dim == len(reso) and grid_nd == 2 and in_dim == 4, "the HIP lookup implements the 4-D, plane (2-D) configuration"
    grid_coefs = nn.ParameterList()
    for coo_comb in PAIRS:
        coef = torch.empty([1, out_dim] + [reso[cc] for cc in coo_comb[::-1]])
        if 3 in coo_comb:
            nn.init.ones_(coef)            # time planes start at 1
        else:
            nn.init.uniform_(coef, a=a, b=b)
        # reference shape [1, C, res_h, res_w] and values, channel-last MEMORY: the lookup's tap rows are views, not per-step copies
        grid_coefs.append(nn.Parameter(coef.contiguous(memory_format=torch.channels_last)))
    return grid_coefs


def morton_order(pts, aabb, bits=10):
    """int32 permutation that visits the points along a Z-order curve of their box-normalised positions (10 bits per axis)."""
    with torch.no_grad():
        q = ((pts.detach() - aabb[0]) / (aabb[1] - aabb[0])).clamp_(0.0, 1.0).mul_(float(2 ** bits - 1)).to(torch.int64)

        def spread(v):                                  # abcdefghij -> a00b00c00d00e00f00g00h00i00j
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        return key.argsort().to(torch.int32)


def plane_order(pts, aabb, ax, ay, bits=12):
    """int32 permutation that visits the points along a HILBERT curve of TWO of their box-normalised coordinates (12 bits each): runs
    of consecutive points are compact in the plane (ax, ay) whatever their third coordinate.  (A Z-order curve jumps: 256 consecutive
    of 2 M uniform points then leave the per-plane pass's 16 x 16 window with 6.9 % of their taps at resolution 512, 2.4 % at 256, each
    such tap four partly filled atomic instructions; along the Hilbert curve none do -- tests/analysis/plane_order_sim.py.)"""
    with torch.no_grad():
        q = ((pts.detach() - aabb[0]) / (aabb[1] - aabb[0])).clamp_(0.0, 1.0).mul_(float(2 ** bits - 1)).to(torch.int64)
        x, y = q[:, ax].clone(), q[:, ay].clone()
        d = torch.zeros_like(x)
        s = 1 << (bits - 1)
        while s:                                        # the classic xy -> d walk, one level per pass, all points at once
            rx, ry = (x & s) > 0, (y & s) > 0
            d += (s * s) * ((3 * rx.to(torch.int64)) ^ ry.to(torch.int64))
            flip = ~ry & rx
            x = torch.where(flip, s - 1 - x, x)
            y = torch.where(flip, s - 1 - y, y)
            x, y = torch.where(~ry, y, x), torch.where(~ry, x, y)
            s >>= 1
        return d.argsort().to(torch.int32)


class VisitingOrders:
    """What the aggregating backward is handed: the 3-D visiting order, and for the scales in `defer_mask` the visiting orders of the
    three spatial planes with their inverses (EmdHexGrads.order2d / pos2d)."""

    def __init__(self, order, order2d=None, pos2d=None, defer_mask=0):
        self.order, self.order2d, self.pos2d, self.defer_mask = order, order2d, pos2d, defer_mask

    @staticmethod
    def build(pts, aabb, res, run=256, window=7):
        """Scales on whose spatial planes a run of `run` points (compact in 3-D) spreads over more than the 7 x 7 window are deferred to
        the per-plane pass: a run fills run / N of the box, i.e. ~1.7 (run / N)^(1/3) of its side along the curve, times the scale's
        resolution (2 M points, runs of 256: 5.5 cells at resolution 64, 11 at 128 -- where 37 % of the taps already left a 10 x 10 window,
        each four partly filled atomic instructions --, 22 at 256).  Round 4: with the per-plane pass at 0.32 ms per scale the threshold came
        down from 1.5 x to 1 x the window: main kernel 4.08 -> 3.69 ms, per-plane pass 0.63 -> 0.95 ms at 2 M points.  `run` and `window` are
        the main kernel's HEX_AGG_POINTS and HEX_SW (round 5: 256 points and 7 x 7 cells per workgroup, two workgroups per CU; a mismatch
        costs time, never correctness)."""
        n = pts.shape[0]
        side = min(1.0, 1.7 * (run / max(n, 1)) ** (1.0 / 3.0))       # (a run spans no more than the box)
        mask = 0
        for s, r in enumerate(res):
            if side * max(r[:3]) > window:
                mask |= 1 << s
        keys = None
        if pts.is_cuda and n > 0:
            # all four sort keys in one launch (emd_hexplane_order_keys); `morton_order` / `plane_order` are the same curves in torch ops
            # (~600 element-wise launches for the three Hilbert walks: 6 ms per refresh at 2 M points) and serve CPU tensors
            keys = torch.empty(4 * n, device=pts.device, dtype=torch.int32)
            L.check(L.load().emd_hexplane_order_keys(pts.detach().contiguous().float().data_ptr(), aabb.detach().contiguous().float().data_ptr(), n,
                                                     keys.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emd_hexplane_order_keys")
            keys = keys.view(4, n)
        order = keys[0].argsort().to(torch.int32) if keys is not None else morton_order(pts, aabb)
        if not mask:
            return VisitingOrders(order)
        o2, p2 = [], []
        ar = torch.arange(n, device=pts.device, dtype=torch.int32)
        for k, (ax, ay) in enumerate(((0, 1), (0, 2), (1, 2))):
            o = keys[1 + k].argsort().to(torch.int32) if keys is not None else plane_order(pts, aabb, ax, ay)
            inv = torch.empty_like(o)
            inv[o.long()] = ar
            o2.append(o)
            p2.append(inv)
        return VisitingOrders(order, o2, p2, mask)


class HexPlaneField(nn.Module):
    """S3Gaussian/scene/hexplane.py:112-183"""
    reorder_every = 64        # lookups between refreshes of the cached visiting order (positions drift slowly; any order is correct)
    reorder_min_points = 8192

    def __init__(self, bounds, planeconfig, multires):
        super().__init__()
        aabb = torch.tensor([[bounds, bounds, bounds], [-bounds, -bounds, -bounds]], dtype=torch.float32)
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        self.grid_config = [planeconfig]
        self.multiscale_res_multipliers = multires
        self.concat_features = True
        self.grids = nn.ModuleList()
        self.feat_dim = 0
        self._res = []
        for res in self.multiscale_res_multipliers:
            config = self.grid_config[0].copy()
            config["resolution"] = [r * res for r in config["resolution"][:3]] + config["resolution"][3:]
            gp = init_grid_param(config["grid_dimensions"], config["input_coordinate_dim"], config["output_coordinate_dim"],
                                 config["resolution"])
            self.feat_dim += gp[-1].shape[1]
            self.grids.append(gp)
            self._res.append(list(config["resolution"]))

    @property
    def get_aabb(self):
        return self.aabb[0], self.aabb[1]

    def set_aabb(self, xyz_max, xyz_min):
        aabb = torch.from_numpy(np.array([xyz_max, xyz_min], dtype=np.float32)).to(self.aabb.device)
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        self._aabb_key = None          # the host copy is stale (a new Parameter may reuse the id of the freed one)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._aabb_key = None

    def _aabb_host(self):
        # six floats for the args struct; re-read from the device only when the parameter was written: set_aabb and
        # load_state_dict invalidate explicitly, in-place writes bump _version, a replaced Parameter object changes the reference
        key = (self.aabb, self.aabb._version)
        old = getattr(self, "_aabb_key", None)
        if old is None or old[0] is not key[0] or old[1] != key[1]:
            self._aabb_key, self._aabb_list = key, self.aabb.detach().reshape(-1).to("cpu", torch.float32).tolist()
        return self._aabb_list

    def get_density(self, pts, timestamps=None):
        pts = pts.reshape(-1, pts.shape[-1])
        planes = [p for gp in self.grids for p in gp]
        t = timestamps.reshape(-1, 1)
        if t.shape[0] == pts.shape[0] and t.shape[0] > 1 and t.stride(0) == 0:
            # one timestamp broadcast over the points (emd_amd.model.render, Deformation.forward_time_offset).  The lookup takes the [1, 1] tensor
            # that was expanded, not a slice of the expansion: slicing inside autograd makes the backward materialise an [N, 1] zero column, scatter
            # the scalar into row 0 and sum N values again -- the column the in-kernel sum (dL_dtime_sum) exists to avoid
            base = getattr(timestamps, "_base", None)
            t = base if (base is not None and tuple(base.shape) == (1, 1) and base.dtype == t.dtype) else t[:1]
        return _HexLookup.apply(pts, t, self._aabb_host(), self._res, self._visiting_order(pts), *planes)

    def _visiting_order(self, pts):
        """Morton order of the points, cached across steps: with it 256 consecutive points share plane cells, and the backward
        aggregates their gradients in LDS before they reach HBM (hexplane.hip).  Purely a speed matter: every order gives the
        same sums up to float rounding, so a stale order (Gaussians moved, or were replaced at equal count) is harmless."""
        n = pts.shape[0]
        # the aggregating backward exists for 16 and 32 channels; the direct-atomic kernel is SLOWER on ordered points (many CUs
        # queue on the same cache lines), so other widths keep the caller's order
        if n < self.reorder_min_points or pts.device.type != "cuda" or self.grids[0][0].shape[1] not in (16, 32):
            return None
        cache = getattr(self, "_order_cache", None)
        if cache is None or cache[0] != n or cache[1] >= self.reorder_every or cache[2].order.device != pts.device:
            cache = self._order_cache = [n, 0, VisitingOrders.build(pts, self.aabb, self._res)]
        cache[1] += 1
        return cache[2]

    def forward(self, pts, timestamps=None):
        return self.get_density(pts, timestamps)
