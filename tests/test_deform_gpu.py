"""-m gpu: the deformation front-ends on the HIP path (emd_amd.deformation) against the golden vectors of the reference's own
deform_network / ConditionalDeformNetwork / DeformableNodes.get_deformation, and against the CPU oracle at larger sizes.
fp32; values within 2e-5 (the GEMMs sum in a different order than the reference's single concat matmul), gradients within 1e-4 of
the largest reference entry."""
import os

import numpy as np
import pytest
import torch

from oracle import deform_oracle as do

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ("point", "scales", "rotations", "opacity", "shs")


def _close(a, b, what, rel=1e-4):
    a, b = a.detach().cpu().numpy(), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.abs(a - b).max() <= rel * max(np.abs(b).max(), 1e-6), (what, np.abs(a - b).max(), np.abs(b).max())


@pytest.mark.parametrize("rows,dim,k,t", [(150, 32, 30, 0.37), (150, 32, 150, 0.0), (150, 32, 77, 1.0), (20, 8, 6, -0.23), (20, 8, 13, 1.4),
                                          (20, 8, 1, 0.5), (7, 64, 20, 0.999), (9, 100, 5, 2.6), (5, 1, 3, 0.3)])
def test_temporal_embed_vs_oracle(rows, dim, k, t):
    from emd_amd.deformation import temporal_embed
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(rows * 1000 + k)
    w = torch.randn(rows, dim, generator=g)
    gout = torch.randn(dim, generator=g)
    w0, t0 = w.clone().requires_grad_(True), torch.tensor(t, requires_grad=True)
    e0 = do.temporal_embed(w0, k, t0)
    (e0 * gout).sum().backward()
    w1, t1 = w.to(dev).requires_grad_(True), torch.tensor([t], device=dev, requires_grad=True)
    e1 = temporal_embed(w1, t1, k)
    (e1 * gout.to(dev)).sum().backward()
    # the sample position t * (k - 1) carries ~1e-6 of fp32 rounding; the table entries are O(1)
    np.testing.assert_allclose(e1.detach().cpu().numpy(), e0.detach().numpy(), rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(w1.grad.cpu().numpy(), w0.grad.numpy(), rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(t1.grad.cpu().numpy().reshape(()), t0.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_temporal_embed_batched_tables():
    """One wave per table: OmniRe keeps a table per actor (rigid.py:150-164)."""
    from emd_amd.deformation import temporal_embed
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    w = torch.randn(6, 150, 32, generator=g)
    gout = torch.randn(6, 32, generator=g)
    w1, t1 = w.to(dev).requires_grad_(True), torch.tensor([0.61], device=dev, requires_grad=True)
    e1 = temporal_embed(w1, t1, 44)
    (e1 * gout.to(dev)).sum().backward()
    gt = 0.0
    for a in range(6):
        w0, t0 = w[a].clone().requires_grad_(True), torch.tensor(0.61, requires_grad=True)
        e0 = do.temporal_embed(w0, 44, t0)
        (e0 * gout[a]).sum().backward()
        np.testing.assert_allclose(e1[a].detach().cpu().numpy(), e0.detach().numpy(), rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(w1.grad[a].cpu().numpy(), w0.grad.numpy(), rtol=1e-5, atol=5e-6)
        gt += float(t0.grad)
    assert abs(float(t1.grad) - gt) <= 1e-4 * max(1.0, abs(gt))


def _load_net(g, tag, dev):
    from emd_amd.deformation import DeformOptions, deform_network
    opt = DeformOptions(no_ds=bool(g[f"{tag}_opt_no_ds"]), no_dr=bool(g[f"{tag}_opt_no_dr"]),
                        no_fine_hexplane_features=bool(g[f"{tag}_opt_no_fine_hexplane_features"]), feat_head=bool(g[f"{tag}_opt_feat_head"]),
                        min_embeddings=int(g[f"{tag}_opt_min_embeddings"]), max_embeddings=int(g[f"{tag}_opt_max_embeddings"]),
                        temporal_embedding_dim=int(g[f"{tag}_opt_temporal_embedding_dim"]), c2f_temporal_iter=int(g[f"{tag}_opt_c2f_temporal_iter"]),
                        multires=g[f"{tag}_opt_multires"].tolist(),
                        kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": int(g[f"{tag}_opt_channels"]),
                                        "resolution": g[f"{tag}_opt_resolution"].tolist()})
    net = deform_network(opt)
    pre = f"{tag}_sd_"
    sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
    net.load_state_dict(sd, strict=True)            # the reference's state_dict, key for key
    return net.to(dev)


@pytest.mark.parametrize("fused", [True, False], ids=["fused-mlp", "gemm-path"])
@pytest.mark.parametrize("tag", ["run", "full"])
def test_deform_network_matches_reference_golden(tag, fused, monkeypatch):
    """Both formulations of the trunk + heads against the reference's own outputs and gradients: the fused fp32-MFMA kernels
    (csrc/mlp.hip, the default) and the hipBLASLt GEMM path (DeformOptions.fused_mlp = False)."""
    from emd_amd import mlp
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "s3g_deform.npz"))
    net = _load_net(g, tag, dev)
    net.args.fused_mlp = fused
    calls = []
    real = mlp.level_mlp
    monkeypatch.setattr(mlp, "level_mlp", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    ins = {n: torch.from_numpy(g[f"{tag}_in_{n}"]).to(dev).requires_grad_(True) for n in NAMES + ("emb",)}
    res = net(ins["point"], ins["scales"], ins["rotations"], ins["opacity"], ins["shs"], torch.from_numpy(g[f"{tag}_in_times"]).to(dev),
              ins["emb"], int(g[f"{tag}_iter"]), int(g[f"{tag}_cam_no"]), 0.1, True)
    for n, r in zip(NAMES, res[:5]):
        np.testing.assert_allclose(r.detach().cpu().numpy(), g[f"{tag}_out_{n}"], rtol=2e-5, atol=2e-5, err_msg=n)
    dd = res[5]
    for lvl in ("coarse", "fine"):
        for k, v in dd[lvl].items():
            key = f"{tag}_ddict_{lvl}_{k}"
            assert (v is None) == (key not in g.files), key
            if v is not None:
                np.testing.assert_allclose(v.detach().cpu().numpy(), g[key], rtol=2e-5, atol=2e-5, err_msg=key)
    loss = sum((r * torch.from_numpy(g[f"{tag}_gout_{n}"]).to(dev)).sum() for n, r in zip(NAMES, res[:5]))
    for lvl in ("coarse", "fine"):
        loss = loss + (dd[lvl]["feat"] * torch.from_numpy(g[f"{tag}_gfeat_{lvl}"]).to(dev)).sum()
    loss.backward()
    for n in NAMES + ("emb",):
        _close(ins[n].grad, g[f"{tag}_g_{n}"], f"grad {n}")
    for n, prm in net.named_parameters():
        want = g[f"{tag}_gsd_{n}"]
        got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        if np.abs(want).max() == 0:
            assert float(got.abs().max()) == 0, n
        else:
            _close(got, want, f"grad {n}")
    assert np.abs(g[f"{tag}_gsd_deformation_net.time_offset"]).max() > 0
    assert len(calls) == (2 if fused else 0)            # both levels really ran on the fused kernels / none did


def test_deform_network_vs_oracle_default_sizes():
    """The reference's default layer sizes (164 -> 64 trunk, 150 x 32 table, 32-channel planes) on 20 000 points."""
    from emd_amd.deformation import DeformOptions, deform_network
    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    opt = DeformOptions(multires=[1, 2, 4], kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                                                             "resolution": [32, 32, 32, 25]})
    net = deform_network(opt)
    net.deformation_net.set_aabb([40.0, 15.0, 8.0], [-5.0, -15.0, -3.0])
    for n, prm in net.named_parameters():
        if "grid" in n and "aabb" not in n:
            prm.data = torch.rand_like(prm) + 0.3
        elif "time_offset" in n:
            prm.data = torch.tensor([[0.0], [0.02], [-0.03]])
        elif prm.dim() > 1 and "grid" not in n:
            prm.data = torch.randn_like(prm) * 0.2
    N = 20000
    pt = torch.rand(N, 3) * torch.tensor([50.0, 34.0, 13.0]) + torch.tensor([-7.0, -17.0, -4.0])
    sc, ro, op, sh, em = torch.randn(N, 3), torch.randn(N, 4), torch.randn(N, 1), torch.randn(N, 16, 3), torch.randn(N, 4) * 0.3
    times = torch.full((N, 1), 0.42)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    o = vars(opt)
    want = do.s3g_deform(sd, o, pt, sc, ro, op, sh, times, em, 12000, 1)
    net = net.to(dev)
    got = net(pt.to(dev), sc.to(dev), ro.to(dev), op.to(dev), sh.to(dev), times.to(dev), em.to(dev), 12000, 1, 0.1, True)
    for n, a, b in zip(NAMES, got[:5], want[:5]):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=1e-4, atol=1e-4, err_msg=n)
    assert got[1].data_ptr() == 0 or torch.equal(got[1].cpu(), sc)          # no_ds: scales pass through


def test_deform_options_outside_scope_raise():
    from emd_amd.deformation import DeformOptions, deform_network
    for flag in ("is_use_hash", "empty_voxel", "static_mlp", "aggregate_feature", "no_grid"):
        with pytest.raises(NotImplementedError):
            deform_network(DeformOptions(**{flag: True}))


def test_deform_ops_refuse_cpu_tensors():
    from emd_amd import _lib as L
    from emd_amd.deformation import ConditionalDeformNetwork, temporal_embed
    with pytest.raises(L.EmdError):
        temporal_embed(torch.zeros(10, 4), torch.zeros(1), 5)
    with pytest.raises(L.EmdError):
        ConditionalDeformNetwork(D=2, W=8, embed_dim=4)(torch.zeros(3, 3), torch.zeros(3, 1), torch.zeros(3, 4))


def _or_net(g, dev):
    from emd_amd.deformation import ConditionalDeformNetwork
    net = ConditionalDeformNetwork(D=int(g["D"]), W=int(g["W"]), input_ch=3, embed_dim=int(g["embed_dim"]), x_multires=int(g["x_multires"]),
                                   t_multires=int(g["t_multires"]), deform_quat=True, deform_scale=False)
    net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}, strict=True)
    return net.to(dev)


def test_nonrigid_deformation_matches_reference_golden():
    from emd_amd.deformation import _DeformInput, nonrigid_deformation
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "or_deform.npz"))
    net = _or_net(g, dev)
    emb = torch.from_numpy(g["inst_embed"]).to(dev).requires_grad_(True)
    means = torch.from_numpy(g["means"]).to(dev).requires_grad_(True)
    ids, size, t = torch.from_numpy(g["point_ids"]).to(dev), torch.from_numpy(g["inst_size"]).to(dev), torch.from_numpy(g["t"]).reshape(1).to(dev)
    h0 = _DeformInput.apply(means, ids, size, emb, t, int(g["x_multires"]), int(g["t_multires"]))
    np.testing.assert_allclose(h0.detach().cpu().numpy(), g["h0"], rtol=2e-6, atol=2e-6)      # sin / cos of arguments up to 2^9 x
    dxyz, dquat, dscale = nonrigid_deformation(net, means, ids[:, None], size, emb, t)
    assert dscale is None
    np.testing.assert_allclose(dxyz.detach().cpu().numpy(), g["dxyz"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dquat.detach().cpu().numpy(), g["dquat"], rtol=2e-5, atol=2e-5)
    ((dxyz * torch.from_numpy(g["gx"]).to(dev)).sum() + (dquat * torch.from_numpy(g["gq"]).to(dev)).sum()).backward()
    assert means.grad is None                                     # local_means.data in the reference
    _close(emb.grad, g["g_inst_embed"], "inst_embed")
    for n, prm in net.named_parameters():
        _close(prm.grad, g[f"gsd_{n}"], n)
    # the network called with the reference's own signature (x, t, condition) gives the same residuals
    x = means.detach() / size[ids.long()][:, 2:3] * 2
    cond = emb.detach()[ids.long()].requires_grad_(True)
    d2, q2, _ = net(x, t.reshape(1, 1).repeat(x.shape[0], 1), cond)
    np.testing.assert_allclose(d2.detach().cpu().numpy(), g["dxyz"], rtol=2e-5, atol=2e-5)
    (d2 * torch.from_numpy(g["gx"]).to(dev)).sum().backward()
    assert cond.grad is not None and cond.grad.shape == cond.shape


def test_deform_input_vs_oracle_unsorted_ids():
    from emd_amd.deformation import _DeformInput
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(9)
    N, A, E = 70001, 37, 16
    means = torch.randn(N, 3, generator=g)
    ids = torch.randint(0, A, (N,), generator=g)
    ids[:30000] = ids[:30000].sort().values                      # runs of equal ids and a shuffled tail
    size = torch.rand(A, 3, generator=g) + 0.5
    emb = torch.randn(A, E, generator=g)
    t = torch.tensor([0.35])
    gout = torch.randn(N, 100, generator=g)
    e0 = emb.clone().requires_grad_(True)
    h0 = do.deform_input(means, ids, size, e0, t, 10, 10)
    (h0 * gout).sum().backward()
    e1 = emb.to(dev).requires_grad_(True)
    h1 = _DeformInput.apply(means.to(dev), ids.to(dev), size.to(dev), e1, t.to(dev), 10, 10)
    (h1 * gout.to(dev)).sum().backward()
    # the arguments x * 2^f are identical fp32 numbers on both sides (same operation order); only sinf / cosf rounding differs
    np.testing.assert_allclose(h1.detach().cpu().numpy(), h0.detach().numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(h1.detach().cpu().numpy()[:, :9], h0.detach().numpy()[:, :9], rtol=1e-6, atol=1e-6)
    _close(e1.grad, e0.grad.numpy(), "embed grad")


def test_deform_edge_sizes():
    """One Gaussian works; zero Gaussians fail loudly (the reference indexes t[0, 0]); an empty encoder input is a no-op."""
    from emd_amd import _lib as L
    from emd_amd.deformation import DeformOptions, _DeformInput, deform_network
    dev = torch.device("cuda", 0)
    opt = DeformOptions(multires=[1], kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 8, "resolution": [4, 4, 4, 3]})
    net = deform_network(opt).to(dev)
    one = lambda *shape: torch.randn(*shape, device=dev)
    out = net(one(1, 3), one(1, 3), one(1, 4), one(1, 1), one(1, 16, 3), torch.full((1, 1), 0.5, device=dev), one(1, 4), 100, 0, 0.1, True)
    assert out[0].shape == (1, 3) and out[4].shape == (1, 16, 3) and all(torch.isfinite(o).all() for o in out[:5])
    with pytest.raises(L.EmdError):
        net(one(0, 3), one(0, 3), one(0, 4), one(0, 1), one(0, 16, 3), torch.zeros(0, 1, device=dev), one(0, 4), 100, 0, 0.1, True)
    h0 = _DeformInput.apply(one(0, 3), torch.zeros(0, dtype=torch.int32, device=dev), one(2, 3), one(2, 16), torch.zeros(1, device=dev), 10, 10)
    assert h0.shape == (0, 100)
