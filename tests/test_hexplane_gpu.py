"""-m gpu: the fused HIP HexPlane lookup (emd_hexplane_forward/backward through emd_amd.hexplane.HexPlaneField) against the
golden vectors of the reference's HexPlaneField and against the CPU oracle at larger sizes.  fp32: features within 1e-6
relative; gradients within 1e-4 of the largest oracle entry (float atomics reorder the sums)."""
import os

import numpy as np
import pytest
import torch

from oracle import hexplane_oracle as ho

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _close(a, b, what, rel=1e-4):
    a, b = a.detach().cpu().numpy(), np.asarray(b)
    assert np.abs(a - b).max() <= rel * max(np.abs(b).max(), 1e-30), (what, np.abs(a - b).max(), np.abs(b).max())


def test_hexplane_matches_reference_golden():
    from emd_amd.hexplane import HexPlaneField
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "s3g_hexplane.npz"))
    cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": int(g["channels"]), "resolution": [int(r) for r in g["resolution"]]}
    field = HexPlaneField(1.6, cfg, [int(m) for m in g["multires"]]).to(dev)
    field.set_aabb(g["aabb"][0].tolist(), g["aabb"][1].tolist())
    for s, gp in enumerate(field.grids):
        for p, prm in enumerate(gp):
            assert tuple(prm.shape) == g[f"plane_{s}_{p}"].shape          # same parameter layout as the reference
            prm.data = torch.from_numpy(g[f"plane_{s}_{p}"]).to(dev)
    pts = torch.from_numpy(g["pts"]).to(dev).requires_grad_(True)
    feat = field(pts, torch.from_numpy(g["times"]).to(dev))
    np.testing.assert_allclose(feat.detach().cpu().numpy(), g["feat"], rtol=2e-6, atol=1e-7)
    (feat * torch.from_numpy(g["gout"]).to(dev)).sum().backward()
    _close(pts.grad, g["g_pts"], "pts")
    for s, gp in enumerate(field.grids):
        for p, prm in enumerate(gp):
            _close(prm.grad, g[f"g_plane_{s}_{p}"], f"plane {s} {p}")


@pytest.mark.parametrize("N,C,res,multires", [(20000, 32, [16, 16, 16, 8], [1, 2, 4]), (5000, 16, [9, 7, 5, 3], [1, 2]), (3000, 4, [4, 4, 4, 2], [1])])
def test_hexplane_vs_oracle(N, C, res, multires):
    from emd_amd.hexplane import HexPlaneField
    dev = torch.device("cuda", 0)
    torch.manual_seed(N)
    cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": C, "resolution": res}
    field = HexPlaneField(1.6, cfg, multires).to(dev)
    for gp in field.grids:
        for prm in gp:
            prm.data = torch.rand_like(prm) + 0.3
    pts = (torch.rand(N, 3) * 3.6 - 1.8)            # bounds 1.6: some points outside
    t = torch.rand(N, 1) * 2.2 - 1.1
    gout = torch.randn(N, C * len(multires))
    p0 = pts.clone().requires_grad_(True)
    planes0 = [[prm.detach().cpu().clone().requires_grad_(True) for prm in gp] for gp in field.grids]
    t0 = t.clone().requires_grad_(True)
    f0 = ho.hexplane_features(p0, t0, field.aabb.detach().cpu(), planes0)
    (f0 * gout).sum().backward()
    p1 = pts.to(dev).requires_grad_(True)
    t1 = t.to(dev).requires_grad_(True)
    f1 = field(p1, t1)
    (f1 * gout.to(dev)).sum().backward()
    np.testing.assert_allclose(f1.detach().cpu().numpy(), f0.detach().numpy(), rtol=2e-6, atol=1e-7)
    _close(p1.grad, p0.grad.numpy(), "pts")
    _close(t1.grad, t0.grad.numpy(), "times")          # reaches the reference's time_offset parameter
    # the planes live channel-last in memory (the lookup's tap rows are views), in the reference's [1, C, H, W] shape
    assert all(prm.is_contiguous(memory_format=torch.channels_last) and prm.grad.shape == prm.shape for gp in field.grids for prm in gp)
    for s, gp in enumerate(field.grids):
        for p, prm in enumerate(gp):
            _close(prm.grad, planes0[s][p].grad.numpy(), f"plane {s} {p}")


@pytest.mark.parametrize("C,layout,N", [(32, "morton", 30011), (32, "random_order", 30011), (16, "morton", 30011), (32, "clustered", 30011),
                                        (32, "same_time", 30011), (32, "plane_pass", 30011), (16, "plane_pass", 30011),
                                        (32, "plane_pass_some_scales", 30011), (32, "plane_pass_random_orders", 30011), (32, "plane_pass_clustered", 30011),
                                        # the edges of the main kernel's 256-point workgroups and of the per-plane pass's 256-point runs
                                        (32, "plane_pass", 1), (32, "plane_pass", 255), (32, "plane_pass", 512), (32, "plane_pass", 513), (16, "same_time", 1025)])
def test_hexplane_aggregating_backward(C, layout, N):
    """The LDS-aggregating backward (taken when a visiting order is given) against the oracle: Morton order (windows hit), an
    arbitrary permutation (almost every tap falls back to the direct atomic), points piled into a few cells, one shared time; and
    with the spatial planes of all / some scales deferred to the per-plane pass (EmdHexGrads.defer_mask), under plane-coherent
    orders, arbitrary permutations (every tap of the pass falls back to the direct atomic) and piled-up points."""
    from emd_amd.hexplane import HexPlaneField, VisitingOrders, _HexLookup, morton_order
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(C + len(layout))
    res, multires = [16, 12, 10, 6], [1, 2, 4]
    cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": C, "resolution": res}
    field = HexPlaneField(1.6, cfg, multires).to(dev)
    for gp in field.grids:
        for prm in gp:
            prm.data = torch.rand_like(prm) + 0.3
    pts = torch.rand(N, 3, generator=g) * 3.6 - 1.8
    if layout.endswith("clustered"):
        pts = torch.randn(N, 3, generator=g) * 0.02 + torch.tensor([0.3, -0.7, 1.1])
    t = torch.full((N, 1), 0.21) if layout in ("same_time", "plane_pass") else torch.rand(N, 1, generator=g) * 2.2 - 1.1
    gout = torch.randn(N, C * len(multires), generator=g)
    p0, t0 = pts.clone().requires_grad_(True), t.clone().requires_grad_(True)
    planes0 = [[prm.detach().cpu().clone().requires_grad_(True) for prm in gp] for gp in field.grids]
    f0 = ho.hexplane_features(p0, t0, field.aabb.detach().cpu(), planes0)
    (f0 * gout).sum().backward()
    p1, t1 = pts.to(dev).requires_grad_(True), t.to(dev).requires_grad_(True)
    order = torch.randperm(N, generator=g).to(torch.int32).to(dev) if layout == "random_order" else morton_order(p1, field.aabb)
    assert sorted(order.cpu().tolist()) == list(range(N))
    if layout.startswith("plane_pass"):
        vo = VisitingOrders.build(p1, field.aabb, field._res, window=0)          # window 0: every scale is deferred
        assert vo.defer_mask == 0b111 and all(sorted(o.cpu().tolist()) == list(range(N)) for o in vo.order2d)
        assert all(torch.equal(inv[o.long()].cpu(), torch.arange(N, dtype=torch.int32)) for o, inv in zip(vo.order2d, vo.pos2d))
        if layout == "plane_pass_some_scales":
            vo.defer_mask = 0b101
        if layout == "plane_pass_random_orders":
            vo.order2d = [torch.randperm(N, generator=g).to(torch.int32).to(dev) for _ in range(3)]
            vo.pos2d = [torch.empty_like(o).scatter_(0, o.long(), torch.arange(N, dtype=torch.int32, device=dev)) for o in vo.order2d]
        order = vo
    planes = [p for gp in field.grids for p in gp]
    f1 = _HexLookup.apply(p1, t1, field._aabb_host(), field._res, order, *planes)
    (f1 * gout.to(dev)).sum().backward()
    np.testing.assert_allclose(f1.detach().cpu().numpy(), f0.detach().numpy(), rtol=2e-6, atol=1e-7)
    _close(p1.grad, p0.grad.numpy(), "pts")
    _close(t1.grad, t0.grad.numpy(), "times")
    for s, gp in enumerate(field.grids):
        for p, prm in enumerate(gp):
            _close(prm.grad, planes0[s][p].grad.numpy(), f"plane {s} {p}")


def test_hexplane_field_caches_visiting_order():
    from emd_amd.hexplane import HexPlaneField
    dev = torch.device("cuda", 0)
    cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [8, 8, 8, 4]}
    field = HexPlaneField(1.6, cfg, [1, 2]).to(dev)
    field.reorder_every = 3
    pts = torch.rand(10000, 3, device=dev) * 3 - 1.5
    t = torch.zeros(10000, 1, device=dev)
    field(pts, t)
    first = field._order_cache[2]
    field(pts, t); field(pts, t)
    assert field._order_cache[2] is first                      # reused
    field(pts, t)
    assert field._order_cache[2] is not first                  # refreshed after reorder_every lookups
    field(pts[:9000], t[:9000])
    assert field._order_cache[0] == 9000                       # and whenever the number of points changes
    assert field._visiting_order(pts[:100]) is None            # small inputs use the plain kernels


def test_channel_last_planes_train_and_load_reference_checkpoints():
    """The planes are nn.Parameters of the reference's shape stored channel-last: an optimiser step works on them, a contiguous
    (reference) state_dict loads into them, and what they save loads back into a contiguous copy."""
    from emd_amd.hexplane import HexPlaneField
    dev = torch.device("cuda", 0)
    cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [8, 6, 4, 3]}
    field = HexPlaneField(1.6, cfg, [1, 2]).to(dev)
    ref_sd = {k: torch.rand_like(v).contiguous() for k, v in field.state_dict().items()}      # what a reference checkpoint holds
    field.load_state_dict(ref_sd, strict=True)
    assert all(p.is_contiguous(memory_format=torch.channels_last) for gp in field.grids for p in gp)
    for k, v in field.state_dict().items():
        assert torch.equal(v.contiguous(), ref_sd[k])
    opt = torch.optim.Adam(field.parameters(), lr=1e-2)
    pts = torch.rand(9000, 3, device=dev) * 3 - 1.5
    before = [p.detach().clone() for gp in field.grids for p in gp]
    for _ in range(2):
        opt.zero_grad()
        field(pts, torch.full((9000, 1), 0.3, device=dev)).square().mean().backward()
        opt.step()
    after = [p for gp in field.grids for p in gp]
    assert all((a - b).abs().max() > 0 for a, b in zip(after, before))
    assert all(p.is_contiguous(memory_format=torch.channels_last) for p in after)


def test_hexplane_full_size_partition_of_unity():
    """2 M points, the reference's plane configuration: with every plane constant (value v_p) the features are the product of the
    constants whatever the taps (bilinear weights sum to one), and the gradient of plane p sums to  sum(gout) * prod_{q != p} v_q  --
    no tap lost or counted twice by the aggregating backward, its windows, fallbacks and flushes, at the headline size."""
    from emd_amd.hexplane import HexPlaneField
    from emd_amd.scenes import make_static_scene
    dev = torch.device("cuda", 0)
    cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [64, 64, 64, 25]}
    field = HexPlaneField(1.6, cfg, [1, 2, 4, 8]).to(dev)
    field.set_aabb([120.0, 30.0, 10.0], [0.0, -30.0, -2.0])
    vals = [1.25, 0.5, 2.0, 0.75, 1.5, 0.8]
    for gp in field.grids:
        for prm, v in zip(gp, vals):
            prm.data.fill_(v)
    N = 2_000_000
    pts = make_static_scene(N).means.to(dev)
    feat = field(pts, torch.full((N, 1), 0.37, device=dev))
    want = float(np.prod(vals))
    assert feat.shape == (N, 128) and float((feat.detach() - want).abs().max()) <= 4e-6 * want
    gout = torch.rand(N, 128, generator=torch.Generator().manual_seed(0)).to(dev)
    feat.backward(gout)
    assert field._order_cache is not None                       # the Morton order / aggregating kernel was used
    for s, gp in enumerate(field.grids):
        total = float(gout[:, s * 32:(s + 1) * 32].double().sum())
        for p, prm in enumerate(gp):
            expect = total * want / vals[p]
            got = float(prm.grad.double().sum())
            assert abs(got - expect) <= 2e-5 * abs(expect), (s, p, got, expect)
            assert float(prm.grad.min()) >= 0.0                 # non-negative contributions only


def test_hexplane_tolerates_nonfinite_points():
    """NaN / Inf coordinates (a diverged Gaussian) must not fault or poison the other points: grid_sample's border rule sends a
    NaN coordinate to cell 0 (the comparison `!(v > 0)` is true for NaN), +-Inf to the borders; the other rows are unaffected."""
    from emd_amd.hexplane import HexPlaneField
    dev = torch.device("cuda", 0)
    cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [8, 8, 8, 4]}
    field = HexPlaneField(1.6, cfg, [1, 2]).to(dev)
    for gp in field.grids:
        for prm in gp:
            prm.data = torch.rand_like(prm) + 0.3
    g = torch.Generator().manual_seed(0)
    N = 20000
    pts = (torch.rand(N, 3, generator=g) * 3 - 1.5).to(dev)
    t = torch.full((N, 1), 0.2, device=dev)
    clean = field(pts, t).detach()
    bad = pts.clone()
    bad[5] = float("nan"); bad[77, 1] = float("inf"); bad[9000, 2] = float("-inf"); bad[12345, 0] = float("nan")
    field._order_cache = None
    out = field(bad.requires_grad_(True), t)
    assert torch.isfinite(out).all()
    rows = torch.ones(N, dtype=torch.bool, device=dev)
    rows[[5, 77, 9000, 12345]] = False
    assert torch.equal(out.detach()[rows], clean[rows])
    out.sum().backward()
    assert all(torch.isfinite(p.grad).all() for gp in field.grids for p in gp)


def test_order_keys_kernel_matches_the_host_curves():
    """emd_hexplane_order_keys (one launch: the Z-order key of the position and the Hilbert keys of the three spatial planes) against the same
    curves written in torch ops (emd_amd.hexplane.morton_order / plane_order, which serve CPU tensors): the orders VisitingOrders.build forms from
    the kernel's keys visit the same cells in the same sequence -- points of one cell may swap (argsort ties) --, including points outside the
    box and NaN coordinates."""
    from emd_amd.hexplane import VisitingOrders, morton_order, plane_order
    DEV = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    N = 40_000
    aabb = torch.tensor([[1.6, 1.6, 1.6], [-1.6, -1.6, -1.6]], device=DEV)
    pts = (torch.rand(N, 3, generator=g) * 3.6 - 1.8).to(DEV)          # some outside the box
    pts[7, 1] = float("nan")
    res = [[64, 64, 64, 25], [512, 512, 512, 25]]
    vo = VisitingOrders.build(pts, aabb, res)
    assert vo.defer_mask == 0b11 and sorted(vo.order.tolist()) == list(range(N))
    clean = torch.where(torch.isnan(pts), aabb[0].expand_as(pts), pts)      # (the kernel sends NaN to cell 0 = the aabb[0] corner)
    q = ((clean - aabb[0]) / (aabb[1] - aabb[0])).clamp(0.0, 1.0)

    def cells(order, bits, axes):
        return (q[order.long()][:, axes] * float(2 ** bits - 1)).to(torch.int64)
    # the sequence of visited cells is the host curve's (ties inside a cell aside)
    assert torch.equal(cells(vo.order, 10, [0, 1, 2]), cells(morton_order(clean, aabb), 10, [0, 1, 2]))
    for k, (ax, ay) in enumerate(((0, 1), (0, 2), (1, 2))):
        o = vo.order2d[k]
        assert sorted(o.tolist()) == list(range(N)) and torch.equal(vo.pos2d[k].long()[o.long()], torch.arange(N, device=DEV))
        assert torch.equal(cells(o, 12, [ax, ay]), cells(plane_order(clean, aabb, ax, ay), 12, [ax, ay]))


def test_backward_without_the_time_gradient_equals_the_one_with_it():
    """k_hexplane_bwd_agg<C, DT>: the instantiation without dL/dtimes (the caller's timestamps do not require a gradient) leaves the same plane and
    point gradients as the one with it, and the time gradient of the latter matches a central difference of the lookup; mixed timestamps, three
    deferred scales (the reference's plane configuration at a size where resolutions 128 .. 512 go through the per-plane pass)."""
    from emd_amd.hexplane import HexPlaneField
    dev = torch.device("cuda", 0)
    cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32, "resolution": [64, 64, 64, 25]}
    torch.manual_seed(5)
    field = HexPlaneField(1.6, cfg, [1, 2, 4, 8]).to(dev)
    with torch.no_grad():
        for gp in field.grids:
            for p in gp:
                p.copy_(torch.rand_like(p) * 0.8 + 0.6)
    N = 1_200_000                                                       # (256-point runs against 7 x 7 windows: resolution 64 stays in the main kernel from ~1 M points on)
    pts = (torch.rand(N, 3, device=dev) * 3.2 - 1.6).requires_grad_(True)
    gout = torch.randn(N, 128, device=dev)
    res = {}
    for with_t in (False, True):
        t = torch.full((N, 1), 0.37, device=dev)
        t[::7] = 0.52                                                   # (blocks with mixed times take the two-row windows)
        t.requires_grad_(with_t)
        pts.grad = None
        for p in field.parameters():
            p.grad = None
        field(pts, t).backward(gout)
        res[with_t] = (pts.grad.clone(), [p.grad.clone() for gp in field.grids for p in gp], t.grad.clone() if with_t else None)
    assert field._order_cache is not None and field._order_cache[2].defer_mask == 0b1110
    scale = float(res[True][0].abs().max())
    assert float((res[False][0] - res[True][0]).abs().max()) <= 1e-5 * scale
    for a, b in zip(res[False][1], res[True][1]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12          # (sums of float atomics: equal up to their order)
    # dL/dtimes against a central difference of sum(out * gout) in t (the lookup is piecewise bilinear in t: exact inside a cell)
    with torch.no_grad():
        t0 = torch.full((N, 1), 0.37, device=dev)
        t0[::7] = 0.52
        h = 1e-3
        fd = ((field(pts.detach(), t0 + h) - field(pts.detach(), t0 - h)) * gout).sum(dim=1, keepdim=True) / (2 * h)
    gt = res[True][2]
    err = (gt - fd).abs()
    assert float(err.max()) <= 2e-2 * float(fd.abs().max()), float(err.max())


@pytest.mark.parametrize("channels", [32, 16, 8], ids=["32ch-tables-agg", "16ch-tables-agg", "8ch-lane-per-channel-direct"])
def test_broadcast_timestamp_takes_the_time_tables_and_changes_nothing(channels):
    """One timestamp broadcast over the points (an expanded tensor, stride 0: what emd_amd.model.render hands over) makes the forward blend the
    time planes into 1-D tables (EmdHexArgs.time_tables: two taps instead of four on the planes xt, yt, zt): same features as the lookup with a
    materialised [N, 1] time column up to rounding, same gradients (the backward is the same kernel), and the gradient of the ONE timestamp is
    the sum over the points -- also through Deformation.forward_time_offset, which keeps the broadcast."""
    from emd_amd.hexplane import HexPlaneField
    dev = torch.device("cuda", 0)
    cfg = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": channels, "resolution": [64, 64, 64, 25]}
    torch.manual_seed(11)
    field = HexPlaneField(1.6, cfg, [1, 2, 4, 8]).to(dev)
    with torch.no_grad():
        for gp in field.grids:
            for p in gp:
                p.copy_(torch.rand_like(p) * 0.8 + 0.6)
    N = 120_000
    pts = (torch.rand(N, 3, device=dev) * 3.4 - 1.7).requires_grad_(True)          # (some outside the box: border clamp)
    gout = torch.randn(N, 4 * channels, device=dev)
    res = {}
    for mode in ("column", "broadcast"):
        t1 = torch.tensor([[0.6180339]], device=dev, requires_grad=True)
        t = t1.expand(N, 1) if mode == "broadcast" else t1.expand(N, 1).contiguous()
        assert (t.stride(0) == 0) == (mode == "broadcast")
        pts.grad = None
        for p in field.parameters():
            p.grad = None
        out = field(pts, t)
        out.backward(gout)
        res[mode] = (out.detach().clone(), pts.grad.clone(), [p.grad.clone() for gp in field.grids for p in gp], t1.grad.clone())
    a, b = res["column"], res["broadcast"]
    assert float((a[0] - b[0]).abs().max()) <= 2e-6 * float(a[0].abs().max())                 # features: rounding of the blend order only
    if channels >= 16:
        assert float((a[0] - b[0]).abs().max()) > 0.0                                          # ... and the table path did run
    assert float((a[1] - b[1]).abs().max()) <= 1e-5 * float(a[1].abs().max())
    for x, y in zip(a[2], b[2]):
        assert float((x - y).abs().max()) <= 2e-5 * float(x.abs().max()) + 1e-12
    assert abs(float(a[3]) - float(b[3])) <= 1e-4 * abs(float(a[3])) + 1e-6
