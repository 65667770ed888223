"""-m gpu: the fused fp32-MFMA MLP kernels (csrc/mlp.hip: trunk + heads of S3Gaussian/scene/deformation.py:100-185,254-337) against
the same network written out in float64 torch: every head output, dL/dxa, dL/dxb and every weight / bias gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _net(g, ka, kb, heads):
    """heads: list of (relu_input, depth, out_dim).  Returns the parameters as float64 CPU leaves."""
    P = lambda *s: (torch.randn(*s, generator=g, dtype=torch.float64) / (s[-1] ** 0.5)).requires_grad_(True)
    ld = ka + 32 + kb                                         # [hex | temporal (folded into the bias) | embedding], as deformation.py lays it out
    net = dict(w0=P(64, ld), b=P(64), col_a=0, col_b=ka + 32, branches=[])
    for relu_input, depth, out_dim in heads:
        hidden = [(P(64, 64), (0.1 * torch.randn(64, generator=g, dtype=torch.float64)).requires_grad_(True)) for _ in range(depth)]
        net["branches"].append((relu_input, hidden, (P(out_dim, 64), (0.1 * torch.randn(out_dim, generator=g, dtype=torch.float64)).requires_grad_(True))))
    return net


def _reference(net, xa, xb, kink=None):
    """`kink` (a list) receives the smallest |pre-activation| per row of every ReLU: a value at its kink (1e-8 happens) makes the float32
    and float64 masks differ for that row -- a measure-zero disagreement, not an error; the test sends no gradient into such rows."""
    ka = 0 if xa is None else xa.shape[1]
    h = net["b"]
    if xa is not None:
        h = h + xa @ net["w0"][:, net["col_a"]:net["col_a"] + ka].t()
    if xb is not None:
        h = h + xb @ net["w0"][:, net["col_b"]:net["col_b"] + xb.shape[1]].t()
    outs = []
    for relu_input, hidden, (wo, bo) in net["branches"]:
        x = torch.relu(h) if relu_input else h
        if kink is not None and relu_input:
            kink.append(h.detach().abs().min(dim=1).values)
        for w, b in hidden:
            pre = x @ w.t() + b
            if kink is not None:
                kink.append(pre.detach().abs().min(dim=1).values)
            x = torch.relu(pre)
        outs.append(x @ wo.t() + bo)
    return outs


def _leaves(net):
    out = [net["w0"], net["b"]]
    for _, hidden, (wo, bo) in net["branches"]:
        for w, b in hidden:
            out += [w, b]
        out += [wo, bo]
    return out


@pytest.mark.parametrize("N,ka,kb,heads", [
    (1000, 128, 4, [(True, 1, 3), (True, 1, 1), (True, 1, 48), (False, 2, 3)]),          # the coarse level of the reference configuration
    (1000, 0, 4, [(True, 1, 3), (True, 1, 1), (True, 1, 48), (False, 2, 3)]),            # its fine level (no HexPlane features)
    (37, 128, 8, [(True, 1, 4), (False, 1, 64), (True, 2, 33)]),                          # ragged tile, other widths
    (96, 16, 4, [(True, 1, 3), (True, 1, 3), (True, 1, 4), (True, 1, 1), (True, 1, 48), (False, 2, 3)]),   # the goldens' network: 16 HexPlane features, six heads
    (500, 72, 5, [(True, 1, 7)]),                                                         # xa ends inside its third tile, odd embedding width
    (70001, 128, 4, [(True, 1, 3), (True, 1, 48)]),                                       # many tiles per wave
    # levels without HexPlane features whose heads all have one hidden layer: no trunk launch, every head forms h from the embedding (EmdMlpBranch.xb)
    (1000, 0, 4, [(True, 1, 3), (True, 1, 1), (True, 1, 48)]),                            # the fine level of the run script without the feature head
    (37, 0, 8, [(True, 1, 4), (False, 1, 64)]),                                           # ragged tile, the widest embedding, an un-rectified input
    (70001, 0, 5, [(True, 1, 3), (True, 1, 48)]),                                         # many tiles per wave, odd embedding width
    # every way dL/dout reaches the backward kernels: one output tile of 8 / 16 / 32 floats (16-byte pieces, partial tile), odd widths (guarded path), with h read ...
    (333, 16, 4, [(True, 1, 16), (True, 1, 32), (True, 1, 8), (True, 1, 7), (True, 1, 33)]),
    (333, 0, 4, [(True, 1, 16), (True, 1, 7), (True, 1, 33), (False, 1, 12)]),            # ... and with h recomputed from the embedding
], ids=["coarse", "fine", "ragged", "golden-shape", "ka72", "70k", "fine-recomputed", "ragged-recomputed", "70k-recomputed", "dout-forms", "dout-forms-recomputed"])
def test_level_mlp_matches_float64(N, ka, kb, heads):
    from emd_amd.mlp import level_mlp
    g = torch.Generator().manual_seed(N + ka + kb)
    net = _net(g, ka, kb, heads)
    xa = torch.randn(N, ka, generator=g, dtype=torch.float64).requires_grad_(True) if ka else None
    xb = torch.randn(N, kb, generator=g, dtype=torch.float64).requires_grad_(True)
    gouts = [torch.randn(N, o, generator=g, dtype=torch.float64) for _, _, o in heads]
    kink = []
    ref = _reference(net, xa, xb, kink)
    near = torch.stack(kink).min(dim=0).values < 1e-5          # rows with a ReLU at its kink: no gradient is sent into them
    assert int(near.sum()) <= max(8, N // 20)
    for go in gouts:
        go[near] = 0.0
    sum((r * go).sum() for r, go in zip(ref, gouts)).backward()
    # the HIP path on float32 copies
    c = lambda t: None if t is None else t.detach().to(DEV, torch.float32).requires_grad_(True)
    hx, hb = c(xa), c(xb)
    hnet = dict(w0=c(net["w0"]), b=c(net["b"]), col_a=net["col_a"], col_b=net["col_b"],
                branches=[(ri, [(c(w), c(b)) for w, b in hid], (c(wo), c(bo))) for ri, hid, (wo, bo) in net["branches"]])
    outs = level_mlp(hx, hb, hnet["w0"], hnet["b"], hnet["col_a"], hnet["col_b"], hnet["branches"])
    for k, (o, r) in enumerate(zip(outs, ref)):
        err = float((o.detach().double().cpu() - r.detach()).abs().max())
        assert err <= 2e-5 * max(1.0, float(r.abs().max())), ("output", k, err)
    sum((o * go.to(DEV, torch.float32)).sum() for o, go in zip(outs, gouts)).backward()

    def check(name, got, want):
        scale = max(float(want.abs().max()), 1e-12)
        err = float((got.double().cpu() - want).abs().max())
        assert err <= 1e-4 * scale, (name, err, scale)
    if xa is not None:
        check("d_xa", hx.grad, xa.grad)
    check("d_xb", hb.grad, xb.grad)
    # the temporal columns of w0 are not the kernels' business (they reach the bias through addmv in the caller): compare the two blocks
    dw0, rw0 = hnet["w0"].grad, net["w0"].grad
    if ka:
        check("d_w0[a]", dw0[:, :ka], rw0[:, :ka])
    check("d_w0[b]", dw0[:, ka + 32:], rw0[:, ka + 32:])
    assert float(dw0[:, ka:ka + 32].abs().max()) == 0.0
    for k, (a, b) in enumerate(zip(_leaves(hnet)[1:], _leaves(net)[1:])):
        check(f"param {k}", a.grad, b.grad)


def test_unused_head_gets_no_gradient_and_costs_no_launch():
    from emd_amd.mlp import level_mlp
    g = torch.Generator().manual_seed(5)
    net = _net(g, 128, 4, [(True, 1, 3), (False, 2, 3)])
    c = lambda t: t.detach().to(DEV, torch.float32).requires_grad_(True)
    xa, xb = torch.randn(500, 128, generator=g).to(DEV).requires_grad_(True), torch.randn(500, 4, generator=g).to(DEV).requires_grad_(True)
    br = [(ri, [(c(w), c(b)) for w, b in hid], (c(wo), c(bo))) for ri, hid, (wo, bo) in net["branches"]]
    outs = level_mlp(xa, xb, c(net["w0"]), c(net["b"]), 0, 160, br)
    outs[0].sum().backward()
    assert br[1][2][0].grad is None and br[1][1][0][0].grad is None          # the unused head's parameters: no gradient at all, as autograd leaves them
    assert float(br[0][2][0].grad.abs().max()) > 0.0 and xa.grad is not None


@pytest.mark.parametrize("with_gout", [True, False])
def test_level_mlp_head_regulariser_is_folded_into_the_head_kernels(with_gout):
    """level_mlp(..., l1_heads=[k]): mean |out_k| comes back behind the outputs (formed by the head's forward kernel) and its gradient
    is added to dL/dout_k inside the head's backward kernel (EmdMlpBranch.l1_sum, EmdMlpBranchGrads.l1_grad / out) -- against the same
    level with the regulariser written as torch ops on the outputs, for a 48-wide (dshs) and a 3-wide head, with and without another
    gradient reaching the regularised head."""
    from emd_amd.mlp import level_mlp
    g = torch.Generator().manual_seed(21)
    N, ka, kb, heads = 3001, 128, 4, [(True, 1, 3), (True, 1, 48), (True, 1, 1)]
    net = _net(g, ka, kb, heads)
    xa, xb = torch.randn(N, ka, generator=g), torch.randn(N, kb, generator=g)
    gouts = [torch.randn(N, o, generator=g).to(DEV) for _, _, o in heads]
    lam = [0.7, 0.0, 1.3]

    def run(folded):
        c = lambda t: t.detach().to(DEV, torch.float32).requires_grad_(True)
        hx, hb = c(xa), c(xb)
        hnet = dict(w0=c(net["w0"]), b=c(net["b"]), branches=[(ri, [(c(w), c(b)) for w, b in hid], (c(wo), c(bo))) for ri, hid, (wo, bo) in net["branches"]])
        if folded:
            outs = level_mlp(hx, hb, hnet["w0"], hnet["b"], net["col_a"], net["col_b"], hnet["branches"], l1_heads=[1, 0])
            assert len(outs) == 5 and outs[3].dim() == 0
            l1_dshs, l1_dx = outs[3], outs[4]
            outs = outs[:3]
        else:
            outs = level_mlp(hx, hb, hnet["w0"], hnet["b"], net["col_a"], net["col_b"], hnet["branches"])
            l1_dshs, l1_dx = outs[1].abs().mean(), outs[0].abs().mean()
        loss = lam[0] * l1_dshs + lam[2] * l1_dx + (outs[2] * gouts[2]).sum() + (outs[0] * gouts[0]).sum()
        if with_gout:
            loss = loss + (outs[1] * gouts[1]).sum()
        loss.backward()
        leaves = [hx, hb, hnet["w0"], hnet["b"]] + [t for _, hid, (wo, bo) in hnet["branches"] for t in [x for wb in hid for x in wb] + [wo, bo]]
        return float(l1_dshs.detach()), float(l1_dx.detach()), [t.grad for t in leaves]
    a1, a2, ga = run(True)
    b1, b2, gb = run(False)
    assert abs(a1 - b1) <= 2e-6 * abs(b1) and abs(a2 - b2) <= 2e-6 * abs(b2)
    for k, (x, y) in enumerate(zip(ga, gb)):
        assert (x is None) == (y is None), k
        if x is not None:
            assert float((x - y).abs().max()) <= 2e-5 * max(float(y.abs().max()), 1e-20), (k, float((x - y).abs().max()), float(y.abs().max()))


@pytest.mark.parametrize("l1", [False, True])
def test_heads_that_recompute_h_equal_the_trunk_launch_bit_for_bit(l1):
    """A level without HexPlane features: the heads form h = b + W0[:, emb] emb themselves (EmdMlpBranch.xb, eight MFMAs per 32 rows in the trunk
    kernel's own operation order) instead of reading the [N, 64] tensor a trunk launch wrote.  Same bits in every head output and in dL/dxb;
    the weight gradients (float atomics over the workgroups) to rounding."""
    from emd_amd import mlp
    g = torch.Generator().manual_seed(77)
    N, kb, heads = 5003, 4, [(True, 1, 3), (True, 1, 48), (True, 1, 1)]
    net = _net(g, 0, kb, heads)
    xb = torch.randn(N, kb, generator=g)
    gouts = [torch.randn(N, o, generator=g).to(DEV) for _, _, o in heads]

    def run(recompute):
        c = lambda t: t.detach().to(DEV, torch.float32).requires_grad_(True)
        hb = c(xb)
        hnet = dict(w0=c(net["w0"]), b=c(net["b"]), branches=[(ri, [(c(w), c(b)) for w, b in hid], (c(wo), c(bo))) for ri, hid, (wo, bo) in net["branches"]])
        old, mlp.RECOMPUTE_H = mlp.RECOMPUTE_H, recompute
        try:
            outs = mlp.level_mlp(None, hb, hnet["w0"], hnet["b"], net["col_a"], net["col_b"], hnet["branches"], l1_heads=[1] if l1 else [])
            loss = sum((o * go).sum() for o, go in zip(outs[:3], gouts))
            if l1:
                loss = loss + 0.9 * outs[3]
            loss.backward()
        finally:
            mlp.RECOMPUTE_H = old
        leaves = [hnet["w0"], hnet["b"]] + [t for _, hid, (wo, bo) in hnet["branches"] for t in [x for wb in hid for x in wb] + [wo, bo]]
        return [o.detach() for o in outs], hb.grad, [t.grad for t in leaves]
    oa, da, ga = run(True)
    ob, db, gb = run(False)
    for k, (x, y) in enumerate(zip(oa[:3], ob[:3])):
        assert torch.equal(x, y), ("output", k)
    assert torch.equal(da, db)
    for k, (x, y) in enumerate(zip(ga, gb)):
        assert float((x - y).abs().max()) <= 2e-5 * max(float(y.abs().max()), 1e-20), (k, float((x - y).abs().max()))


def test_recomputed_h_is_refused_for_two_hidden_layers():
    import ctypes as C
    from emd_amd import _lib as L
    lib = L.load()
    t = torch.zeros(64, 64, device=DEV)
    b = L.EmdMlpBranch()
    b.num_points, b.depth, b.relu_input, b.out_dim = 32, 2, 0, 3
    for d in range(2):
        b.w_hidden[d], b.b_hidden[d] = t.data_ptr(), t.data_ptr()
    b.w_out, b.b_out, b.out = t.data_ptr(), t.data_ptr(), t.data_ptr()
    b.xb, b.w_in, b.b_in, b.kb_in, b.ld_w_in, b.col_in = t.data_ptr(), t.data_ptr(), t.data_ptr(), 4, 64, 0
    assert lib.emd_mlp_branch_forward(C.byref(b), None) == L.EMD_ERR_INVALID
    b.depth, b.kb_in = 1, 9
    assert lib.emd_mlp_branch_forward(C.byref(b), None) == L.EMD_ERR_INVALID
