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
