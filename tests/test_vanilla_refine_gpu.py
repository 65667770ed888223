"""-m gpu: OmniRe's per-class density control on the device (emd_amd.vanilla.VanillaGaussians: after_train, refinement_after through
emd_refine_decide / emd_refine_index / emd_densify_gather) against the reference's own VanillaGaussians run on CPU
(tests/golden/or_refine.npz, tests/gen_golden.py:gen_or_refine -- every parameter and both Adam moments after each of four events that
exercise split + duplicate + cull, cull by alpha only, the opacity reset alone and cull only, with the reference's torch.randn draw recorded;
the running refinement statistics after every view).  Row counts, row ORDER and every copied value bit-exact; the transformed ones (split
means, reduced log-scales, reset opacities) to 2e-6."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
G = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ("xyz", "sh_dc", "sh_rest", "opacity", "scaling", "rotation")
ATTR = dict(xyz="_means", sh_dc="_features_dc", sh_rest="_features_rest", opacity="_opacities", scaling="_scales", rotation="_quats")


def _setup(z, optimizer_cls):
    from emd_amd.vanilla import VanillaGaussians
    cfg = {k: (int(v) if float(v).is_integer() and k not in ("reset_alpha_value",) else float(v)) for k, v in zip(z["cfg_keys"].tolist(), z["cfg_values"].tolist())}
    node = VanillaGaussians(str(z["class_name"]), types.SimpleNamespace(**cfg), scene_scale=float(z["scene_scale"]), num_train_images=int(z["num_train_images"]),
                            device=DEV)
    for n in NAMES:
        setattr(node, ATTR[n], torch.nn.Parameter(torch.tensor(z[f"in_{n}"]).to(DEV).contiguous()))
    groups = [{"params": v, "lr": 1e-3, "name": k} for k, v in node.get_gaussian_param_groups().items()]
    assert [g["name"] for g in groups] == z["group_names"].tolist()
    opt = optimizer_cls(groups, lr=0.0, eps=1e-15)
    for n in NAMES:
        opt.state[getattr(node, ATTR[n])] = {"step": torch.tensor(2.0), "exp_avg": torch.tensor(z[f"in_m_{n}"]).to(DEV).contiguous(),
                                             "exp_avg_sq": torch.tensor(z[f"in_v_{n}"]).to(DEV).contiguous()}
    return node, opt


def _check(node, opt, z, tag):
    for n in NAMES:
        p = getattr(node, ATTR[n])
        got, ref = p.detach().cpu().numpy(), z[f"{tag}_{n}"]
        assert got.shape == ref.shape, (tag, n, got.shape, ref.shape)
        if n in ("xyz", "scaling", "opacity"):
            np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6, err_msg=f"{tag} {n}")
        else:
            np.testing.assert_array_equal(got, ref, err_msg=f"{tag} {n}")
        grp = [g for g in opt.param_groups if g["name"] == node.class_prefix + n][0]
        assert len(grp["params"]) == 1 and grp["params"][0] is p, (tag, n)                 # the optimiser holds the NEW leaf
        st = opt.state[p]
        np.testing.assert_array_equal(st["exp_avg"].cpu().numpy(), z[f"{tag}_m_{n}"], err_msg=f"{tag} exp_avg {n}")
        np.testing.assert_array_equal(st["exp_avg_sq"].cpu().numpy(), z[f"{tag}_v_{n}"], err_msg=f"{tag} exp_avg_sq {n}")
    assert node.xys_grad_norm is None and node.vis_counts is None and node.max_2Dsize is None       # every event restarts the statistics


def _views(node, z, tag, count):
    last = float(z["last_size"])
    for v in range(count):
        radii, grad = torch.tensor(z[f"{tag}_radii{v}"]).to(DEV), torch.tensor(z[f"{tag}_grad{v}"]).to(DEV)
        node.after_train(radii, grad, last)
        np.testing.assert_allclose(node.xys_grad_norm.cpu().numpy(), z[f"{tag}_norm{v}"], rtol=1e-6, atol=1e-12, err_msg=f"{tag} norm view {v}")
        np.testing.assert_array_equal(node.vis_counts.cpu().numpy(), z[f"{tag}_vis{v}"], err_msg=f"{tag} vis view {v}")
        np.testing.assert_array_equal(node.max_2Dsize.cpu().numpy(), z[f"{tag}_m2d{v}"], err_msg=f"{tag} max_2Dsize view {v}")


def _drift(node, z, tag):
    node._opacities.data.copy_(torch.tensor(z[f"{tag}_opacity_in"]).to(DEV))
    node._scales.data.copy_(torch.tensor(z[f"{tag}_scaling_in"]).to(DEV))


@pytest.mark.parametrize("optimizer", ["hip_adam", "torch_adam"])
def test_refinement_matches_the_reference_vanilla_gaussians(optimizer):
    from emd_amd.optim import Adam
    z = np.load(os.path.join(G, "or_refine.npz"))
    node, opt = _setup(z, Adam if optimizer == "hip_adam" else torch.optim.Adam)
    ns = 2

    def refine(tag):
        step = int(z[f"{tag}_step"])
        node.preprocess_per_train_step(step)
        zr = z[f"{tag}_randn"]
        samples = torch.tensor(zr).view(ns, -1, 3) if zr.shape[0] else None
        info = node.refinement_after(step, opt, samples=samples)
        _check(node, opt, z, tag)
        return info
    _views(node, z, "a", 3)
    info = refine("A")
    assert info["n_before"] == 160 and info["n_after"] == z["A_xyz"].shape[0] and info["split"] == z["A_randn"].shape[0] // ns
    assert info["split"] > 0 and info["dups_kept"] > 0 and info["originals_kept"] < 160          # all three branches fired
    _drift(node, z, "B")
    _views(node, z, "b", 2)
    info = refine("B")
    assert info["split"] == z["B_randn"].shape[0] // ns and info["n_after"] == z["B_xyz"].shape[0]
    _views(node, z, "c", 1)
    assert refine("C") is None                                  # inside the guard behind a reset: only the opacity reset runs
    _drift(node, z, "D")
    _views(node, z, "d", 2)
    info = refine("D")
    assert info["split"] == 0 and info["dups_kept"] == 0 and info["n_after"] == z["D_xyz"].shape[0] < info["n_before"]


def test_refinement_philox_samples_are_rank_independent_and_plausible():
    """Without a recorded draw the split samples come from Philox keyed by (seed, event, source row, replica): two stores with the same seed
    produce identical bits (what keeps view-parallel replicas identical), another seed differs, and the samples have the spread of the
    reference's draw (standard normals through R(q) diag(exp(scale)))."""
    from emd_amd.optim import Adam
    z = np.load(os.path.join(G, "or_refine.npz"))
    outs = []
    for seed in (0, 0, 1):
        node, opt = _setup(z, Adam)
        node.refine_seed = seed
        _views(node, z, "a", 3)
        node.preprocess_per_train_step(3600)
        node.refinement_after(3600, opt)
        outs.append(node._means.detach().clone())
    assert torch.equal(outs[0], outs[1]) and outs[0].shape == outs[2].shape and not torch.equal(outs[0], outs[2])
    ref = torch.tensor(z["A_xyz"]).to(DEV)
    keep = int((outs[0] == ref).all(dim=1).sum())                # originals and duplicates are copies; only the samples differ from the recorded draw
    n_samples = int(z["A_randn"].shape[0])
    assert keep >= ref.shape[0] - n_samples
    d0, dr = (outs[0] - ref).abs().max().item(), ref.abs().max().item()
    assert 0 < d0 < 20 * dr


def test_after_train_rejects_cpu_tensors():
    from emd_amd import _lib as L
    from emd_amd.vanilla import VanillaGaussians
    node = VanillaGaussians("Background", dict(sh_degree=1, refine_interval=100), device="cpu")
    with pytest.raises(Exception):
        node.after_train(torch.ones(1, dtype=torch.int32), torch.ones(1, 2), 100.0)


def test_refinement_edge_cases_nothing_to_do_and_everything_culled():
    """(a) no Gaussian above the gradient threshold and none to cull: the event leaves every tensor object (and the optimiser) as it is; (b) every Gaussian
    transparent: the class ends with zero rows and the optimiser with zero-row moments (the reference's mask indexing gives the same); (c) a step inside
    the warm-up returns None and keeps the statistics."""
    from emd_amd.optim import Adam
    z = np.load(os.path.join(G, "or_refine.npz"))
    node, opt = _setup(z, Adam)
    n = node.num_points
    node.preprocess_per_train_step(300)                       # (c) warm-up: refinement_after returns before touching anything
    node.after_train(torch.ones(n, dtype=torch.int32, device=DEV), torch.zeros(n, 2, device=DEV), 1600.0)
    assert node.refinement_after(300, opt) is None and node.xys_grad_norm is not None
    # (a) gradients far below the threshold, opacities well above the cull threshold, step before the first reset (alpha-only cull)
    node._opacities.data.fill_(2.0)
    before = {k: getattr(node, v) for k, v in ATTR.items()}
    node.preprocess_per_train_step(700)
    info = node.refinement_after(700, opt)
    assert info["n_after"] == n and info["split"] == 0 and info["dups_kept"] == 0 and info["originals_kept"] == n
    assert all(getattr(node, v) is before[k] for k, v in ATTR.items())
    assert node.xys_grad_norm is None                          # the statistics restart after every event
    # (b) everything transparent
    node._opacities.data.fill_(-12.0)
    node.after_train(torch.ones(n, dtype=torch.int32, device=DEV), torch.zeros(n, 2, device=DEV), 1600.0)
    node.preprocess_per_train_step(800)
    info = node.refinement_after(800, opt)
    assert info["n_after"] == 0 and node.num_points == 0
    for k, v in ATTR.items():
        p = getattr(node, v)
        assert p.shape[0] == 0 and opt.state[p]["exp_avg"].shape[0] == 0


def test_after_train_takes_three_column_gradients_by_their_first_two():
    """`xys_grad` may arrive as the [n, 3] screen-space gradient of the diff_gauss surface (means2D.grad): the statistics use columns 0, 1 with the
    row stride of the tensor handed over, first call (norm of every row) and later calls (masked adds) alike."""
    from emd_amd.vanilla import VanillaGaussians
    n = 1000
    g = torch.Generator().manual_seed(3)
    a, b = VanillaGaussians("A", dict(sh_degree=1, refine_interval=100), device=DEV), VanillaGaussians("B", dict(sh_degree=1, refine_interval=100), device=DEV)
    for node in (a, b):
        node._means = torch.nn.Parameter(torch.zeros(n, 3, device=DEV))
    for v in range(3):
        radii = torch.randint(0, 50, (n,), generator=g, dtype=torch.int32)
        radii[torch.rand(n, generator=g) < 0.4] = 0
        g3 = torch.randn(n, 3, generator=g) * 1e-3
        a.after_train(radii.to(DEV), g3.to(DEV), 800.0)
        b.after_train(radii.to(DEV), g3[:, :2].contiguous().to(DEV), 800.0)
    assert torch.equal(a.xys_grad_norm, b.xys_grad_norm) and torch.equal(a.vis_counts, b.vis_counts) and torch.equal(a.max_2Dsize, b.max_2Dsize)
    assert float(a.vis_counts.max()) == 3.0 and float(a.max_2Dsize.max()) > 0


def test_refinement_at_scale_matches_a_torch_restatement_of_the_reference():
    """300 000 rows (1 172 blocks of 256: the block-count scan runs more than one 1024-block chunk per column): the fused event against
    vanilla.py:206-376 written out with torch mask indexing / repeat / cat on the GPU (split with the original kept and reduced in place, duplicate
    decided AFTER the reduction, cull of the grown arrays with max_2Dsize 0 on new rows) -- row count, row order and every value."""
    from emd_amd.optim import Adam
    from emd_amd.vanilla import VanillaGaussians
    N, ns, scene_scale = 300_000, 2, 2.0
    g = torch.Generator().manual_seed(11)
    cfg = dict(sh_degree=1, warmup_steps=500, reset_alpha_interval=3000, refine_interval=100, n_split_samples=ns, reset_alpha_value=0.01, densify_grad_thresh=0.0003,
               densify_size_thresh=0.003, cull_alpha_thresh=0.005, cull_scale_thresh=0.5, cull_screen_size=0.15, split_screen_size=0.05, stop_screen_size_at=4000,
               stop_split_at=15000)
    node = VanillaGaussians("Background", cfg, scene_scale=scene_scale, num_train_images=10, device=DEV)
    P = lambda t: torch.nn.Parameter(t.to(DEV).contiguous())
    node._means = P(torch.randn(N, 3, generator=g) * 4)
    node._scales = P(torch.log(torch.tensor(1e-3)) + torch.rand(N, 3, generator=g) * 8.0 - 1.0)
    node._quats = P(torch.randn(N, 4, generator=g))
    node._opacities = P(torch.randn(N, 1, generator=g) * 3.0 - 1.0)
    node._features_dc = P(torch.randn(N, 3, generator=g))
    node._features_rest = P(torch.randn(N, 3, 3, generator=g) * 0.1)
    opt = Adam([{"params": v, "lr": 1e-3, "name": k} for k, v in node.get_gaussian_param_groups().items()], lr=0.0, eps=1e-15)
    for v in node.get_gaussian_param_groups().values():
        opt.state[v[0]] = {"step": torch.tensor(1.0), "exp_avg": torch.randn(v[0].shape, generator=g).to(DEV), "exp_avg_sq": torch.rand(v[0].shape, generator=g).to(DEV)}
    node.xys_grad_norm = (torch.rand(N, generator=g) * 1.2e-3).to(DEV)
    node.vis_counts = torch.randint(1, 4, (N,), generator=g).float().to(DEV)
    node.max_2Dsize = (torch.rand(N, generator=g) * 0.2).to(DEV)
    step = 3600
    # ---- the reference's sequence with torch ops (vanilla.py:218-326), on clones
    m, s, q, o = node._means.detach().clone(), node._scales.detach().clone(), node._quats.detach().clone(), node._opacities.detach().clone()
    dc, mom = node._features_dc.detach().clone(), opt.state[node._scales]["exp_avg"].clone()
    high = (node.xys_grad_norm / node.vis_counts) > cfg["densify_grad_thresh"]
    splits = (torch.exp(s).max(dim=-1).values > cfg["densify_size_thresh"] * scene_scale) | (node.max_2Dsize > cfg["split_screen_size"])
    splits &= high
    n_split = int(splits.sum())
    samples = torch.randn(ns * n_split, 3, generator=g).to(DEV)
    qn = q[splits] / q[splits].norm(dim=-1, keepdim=True)
    w_, x_, y_, z_ = qn.unbind(-1)
    R = torch.stack([1 - 2 * (y_ ** 2 + z_ ** 2), 2 * (x_ * y_ - w_ * z_), 2 * (x_ * z_ + w_ * y_), 2 * (x_ * y_ + w_ * z_), 1 - 2 * (x_ ** 2 + z_ ** 2), 2 * (y_ * z_ - w_ * x_),
                     2 * (x_ * z_ - w_ * y_), 2 * (y_ * z_ + w_ * x_), 1 - 2 * (x_ ** 2 + y_ ** 2)], -1).reshape(-1, 3, 3)
    new_means = torch.bmm(R.repeat(ns, 1, 1), (torch.exp(s[splits]).repeat(ns, 1) * samples)[..., None]).squeeze(-1) + m[splits].repeat(ns, 1)
    new_scales = torch.log(torch.exp(s[splits]) / 1.6).repeat(ns, 1)
    s[splits] = torch.log(torch.exp(s[splits]) / 1.6)
    dups = (torch.exp(s).max(dim=-1).values <= cfg["densify_size_thresh"] * scene_scale) & high
    cat = lambda base, a, b: torch.cat([base, a, b], 0)
    M_ = cat(m, new_means, m[dups]); S_ = cat(s, new_scales, s[dups]); O_ = cat(o, o[splits].repeat(ns, 1), o[dups]); D_ = cat(dc, dc[splits].repeat(ns, 1), dc[dups])
    mom_ = cat(mom, torch.zeros_like(new_scales), torch.zeros_like(s[dups]))
    m2d = torch.cat([node.max_2Dsize, torch.zeros(ns * n_split + int(dups.sum()), device=DEV)])
    culls = (torch.sigmoid(O_).squeeze(-1) < cfg["cull_alpha_thresh"]) | (torch.exp(S_).max(dim=-1).values > cfg["cull_scale_thresh"] * scene_scale) | (m2d > cfg["cull_screen_size"])
    keep = ~culls
    # ---- the fused event
    node.preprocess_per_train_step(step)
    info = node.refinement_after(step, opt, samples=samples.view(ns, n_split, 3))
    assert info["split"] == n_split and info["n_after"] == int(keep.sum()) and n_split > 1000 and int(dups.sum()) > 100 and int(culls.sum()) > 1000
    assert torch.equal(node._features_dc.detach(), D_[keep]) and torch.equal(node._opacities.detach(), O_[keep])
    torch.testing.assert_close(node._scales.detach(), S_[keep], rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(node._means.detach(), M_[keep], rtol=2e-6, atol=2e-5)
    assert torch.equal(opt.state[node._scales]["exp_avg"], mom_[keep])
