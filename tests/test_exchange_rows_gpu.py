"""-m gpu: emd_compact_rows / emd_scatter_rows (csrc/exchange.hip), the visibility-compacted rows of the view-parallel gradient exchange, on their own:
row contents, header, capacity overflow, empty and all-visible inputs, rank-ordered additions, and the L1 loss's granule table (emd_l1_loss_ws)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _pack(radii, sources, cap):
    from emd_amd import dp
    rows = dp.compact_rows(radii, sources, cap)
    torch.cuda.synchronize()
    return rows


@pytest.mark.parametrize("n,frac", [(0, 0.5), (1, 1.0), (1000, 0.0), (1000, 1.0), (70001, 0.53)], ids=["empty", "one", "none-visible", "all-visible", "ragged"])
def test_compact_rows_hold_exactly_the_visible_gaussians(n, frac):
    g = torch.Generator().manual_seed(n + 1)
    radii = (torch.rand(n, generator=g) < frac).to(torch.int32) * torch.randint(1, 50, (n,), generator=g, dtype=torch.int32)
    a, b, c, d = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g), torch.randn(n, 1, generator=g)
    cap = n + 5
    rows = _pack(radii.to(DEV), [t.to(DEV) for t in (a, b, c, d)], cap)
    assert rows.shape == (1 + cap, 12)
    vis = (radii > 0).nonzero()[:, 0]
    hdr = rows[0].cpu()
    assert int(hdr[0]) == vis.numel() and int(hdr[1]) == 0 and not bool(hdr[2:].any())
    body = rows[1:1 + vis.numel()].cpu()
    order = torch.argsort(body[:, 0])                                   # (rows are unordered: workgroups claim ranges with one atomic each)
    body = body[order]
    assert torch.equal(body[:, 0].long(), vis)
    want = torch.cat([a, b, c, d], 1)[vis]
    assert torch.equal(body[:, 1:].contiguous().view(torch.float32), want)          # values bit for bit


def test_compact_rows_report_an_undersized_capacity_and_scatter_ignores_the_surplus():
    from emd_amd import dp
    n, cap = 5000, 1000
    g = torch.Generator().manual_seed(3)
    radii = torch.ones(n, dtype=torch.int32, device=DEV)
    src = torch.randn(n, 3, generator=g).to(DEV)
    rows = _pack(radii, [src], cap)
    hdr = rows[0].cpu()
    assert int(hdr[0]) == cap and int(hdr[1]) == 1
    dst = torch.zeros(n, 3, device=DEV)
    ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
    dp.scatter_rows(rows, [dst], add=True, overflow=ovf)
    torch.cuda.synchronize()
    assert int(ovf) == 1
    hit = (dst.abs().sum(1) > 0)
    assert int(hit.sum()) == cap and torch.equal(dst[hit], src[hit])                  # the rows that fit arrived intact, nothing else was written


def test_scatter_rows_adds_views_in_the_order_they_are_launched():
    """Two views that see overlapping sets: set / add semantics, the scale factor, and the same bits whichever process does it (a fixed order of adds)."""
    from emd_amd import dp
    n = 30000
    g = torch.Generator().manual_seed(9)
    vals = [torch.randn(n, 11, generator=g).to(DEV) for _ in range(2)]
    radii = [(torch.rand(n, generator=g) < 0.6).to(torch.int32).to(DEV) for _ in range(2)]
    rows = [_pack(radii[v], [vals[v][:, :3].contiguous(), vals[v][:, 3:6].contiguous(), vals[v][:, 6:10].contiguous(), vals[v][:, 10:].contiguous()], n)
            for v in range(2)]

    def run():
        slab = torch.zeros(n * 11, device=DEV)
        dests = [slab[:3 * n].view(n, 3), slab[3 * n:6 * n].view(n, 3), slab[6 * n:10 * n].view(n, 4), slab[10 * n:].view(n, 1)]
        for v in range(2):
            dp.scatter_rows(rows[v], dests, add=True, scale=0.5)
        torch.cuda.synchronize()
        return torch.cat([d.clone() for d in dests], 1)
    got, again = run(), run()
    assert torch.equal(got, again)
    want = sum(0.5 * vals[v] * (radii[v] > 0).float()[:, None] for v in range(2))
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    only = torch.full((n, 3), 7.0, device=DEV)
    dp.scatter_rows(_pack(radii[0], [vals[0][:, :3].contiguous()], n), [only], add=False)         # "set": rows of invisible Gaussians keep what they held
    torch.cuda.synchronize()
    v0 = radii[0] > 0
    assert torch.equal(only[v0], vals[0][:, :3][v0]) and bool((only[~v0] == 7.0).all())


def test_l1_loss_through_the_granule_table_is_deterministic_and_leaves_its_scratch_clean():
    """emd_l1_loss_ws: no zero fill of the scalar, partial sums through one 8-byte granule per workgroup, workgroup 0 adds them in a fixed tree."""
    from emd_amd.model import l1_loss, _l1_scratch
    g = torch.Generator().manual_seed(5)
    for shape in ((3, 7, 5), (3, 200, 304), (3, 1066, 1600)):
        a, b = torch.rand(*shape, generator=g).to(DEV), torch.rand(*shape, generator=g).to(DEV)
        a.requires_grad_(True)
        vals = []
        for _ in range(3):
            a.grad = None
            loss = l1_loss(a, b)
            loss.backward()
            vals.append(loss.detach().clone())
        torch.cuda.synchronize()
        assert torch.equal(vals[0], vals[1]) and torch.equal(vals[1], vals[2])               # the same bits every time
        want = (a.detach().double() - b.double()).abs().mean()
        assert abs(float(vals[0]) - float(want)) <= 2e-6 * float(want)
        assert torch.equal(a.grad, torch.sign(a.detach() - b) / a.numel())
    assert _l1_scratch and all(not bool(t.any()) for t in _l1_scratch.values())             # every call handed its table back zeroed
    s2 = torch.cuda.Stream()                                                                # a second stream gets a table of its own
    with torch.cuda.stream(s2):
        x = l1_loss(torch.ones(3, 64, 64, device=DEV), torch.zeros(3, 64, 64, device=DEV))
    s2.synchronize()
    assert float(x) == 1.0 and len(_l1_scratch) >= 2
