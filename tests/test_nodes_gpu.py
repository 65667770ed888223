"""-m gpu: the per-class Gaussian dictionaries of the OmniRe flow (emd_amd.nodes) against RigidNodes.get_gaussians and
DeformableNodes.get_gaussians of the reference run on CPU (tests/golden/or_nodes.npz; the absent gsplat spherical_harmonics was a
recording stand-in evaluating the oracle's SH on the reference's own arguments).  Values 1e-5, gradients 1e-4 of the largest entry."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("_means", "_opacities", "_rgbs", "_scales", "_quats")


def _close(a, b, what, rel=1e-4):
    a, b = a.detach().cpu().numpy(), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.abs(a - b).max() <= rel * max(np.abs(b).max(), 1e-6), (what, np.abs(a - b).max(), np.abs(b).max())


def _inputs(g, pre, dev):
    t = lambda k, grad=True: torch.from_numpy(g[pre + k]).to(dev).requires_grad_(grad)
    d = dict(means=t("means"), quats=t("quats"), opacity_logits=t("opacity_logits"), log_scales=t("log_scales"), features_dc=t("features_dc"),
             features_rest=t("features_rest"), instances_quats=t("instances_quats"), instances_trans=t("instances_trans"))
    d["point_ids"] = torch.from_numpy(g[pre + "point_ids"]).to(dev)
    d["instances_fv"] = torch.from_numpy(g[pre + "instances_fv"]).to(dev)
    return d


def _check(g, pre, gs, d, extra=()):
    for k in KEYS:
        np.testing.assert_allclose(gs[k].detach().cpu().numpy(), g[pre + "out" + k], rtol=1e-5, atol=1e-5, err_msg=k)
    sum((gs[k] * torch.from_numpy(g[pre + "gout" + k]).to(gs[k].device)).sum() for k in KEYS).backward()
    for mine, ref in (("means", "_means"), ("quats", "_quats"), ("opacity_logits", "_opacities"), ("log_scales", "_scales"),
                      ("features_dc", "_features_dc"), ("features_rest", "_features_rest"), ("instances_quats", "instances_quats"),
                      ("instances_trans", "instances_trans")):
        want = g[pre + "grad" + ref]
        got = d[mine].grad if d[mine].grad is not None else torch.zeros_like(d[mine])
        if np.abs(want).max() == 0:
            assert float(got.abs().max()) == 0, mine
        else:
            _close(got, want, "grad " + mine)


def test_rigid_gaussians_match_reference_get_gaussians():
    from emd_amd.motion import actor_pose_table
    from emd_amd.nodes import rigid_gaussians
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "or_nodes.npz"))
    assert int(g["rigid_sh_degree_used"]) == 3 and int(g["rigid_sh_dirs_requires_grad"]) == 0      # SH on detached view directions
    d = _inputs(g, "rigid_", dev)
    pose = actor_pose_table(d["instances_quats"], d["instances_trans"], d["instances_fv"], int(g["rigid_frame"]))
    gs = rigid_gaussians(d["means"], d["quats"], d["opacity_logits"], d["log_scales"], d["features_dc"], d["features_rest"], d["point_ids"][:, None],
                         pose, torch.from_numpy(g["camera_center"]).to(dev), sh_degree=3, step=int(g["rigid_step"]))
    _check(g, "rigid_", gs, d)
    assert (g["rigid_out_opacities"] == 0).any()          # the fixture has an actor that is not visible in this frame


def test_deformable_gaussians_match_reference_get_gaussians():
    from emd_amd.deformation import ConditionalDeformNetwork
    from emd_amd.motion import actor_pose_table
    from emd_amd.nodes import deformable_gaussians
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "or_nodes.npz"))
    pre = "deformable_"
    d = _inputs(g, pre, dev)
    net = ConditionalDeformNetwork(D=int(g[pre + "net_D"]), W=int(g[pre + "net_W"]), input_ch=3, embed_dim=int(g[pre + "net_embed_dim"]),
                                   x_multires=int(g[pre + "net_x_multires"]), t_multires=int(g[pre + "net_t_multires"]),
                                   deform_quat=bool(g[pre + "net_deform_quat"]), deform_scale=bool(g[pre + "net_deform_scale"]))
    net.load_state_dict({k[len(pre) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + "sd_")}, strict=True)
    net = net.to(dev)
    emb = torch.from_numpy(g[pre + "instances_embedding"]).to(dev).requires_grad_(True)
    pose = actor_pose_table(d["instances_quats"], d["instances_trans"], d["instances_fv"], int(g[pre + "frame"]))
    gs = deformable_gaussians(net, d["means"], d["quats"], d["opacity_logits"], d["log_scales"], d["features_dc"], d["features_rest"],
                              d["point_ids"], pose, torch.from_numpy(g["camera_center"]).to(dev), torch.from_numpy(g[pre + "instances_size"]).to(dev),
                              emb, torch.from_numpy(g[pre + "t"]).reshape(1).to(dev), sh_degree=3, step=int(g[pre + "step"]))
    _check(g, pre, gs, d)
    _close(emb.grad, g[pre + "grad_instances_embedding"], "instances_embedding")
    for n, prm in net.named_parameters():
        _close(prm.grad, g[pre + "gsd_" + n], n)
    assert np.abs(g[pre + "grad_means"]).max() == 0       # stop_optimizing_canonical_xyz: the canonical means take no gradient


def test_nonfinite_gaussians_raise_like_the_reference():
    from emd_amd.motion import actor_pose_table
    from emd_amd.nodes import rigid_gaussians
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "or_nodes.npz"))
    d = _inputs(g, "rigid_", dev)
    with torch.no_grad():
        d["log_scales"][3, 1] = 200.0                          # exp overflows
    pose = actor_pose_table(d["instances_quats"], d["instances_trans"], d["instances_fv"], int(g["rigid_frame"]))
    with pytest.raises(ValueError, match="Inf detected in gaussian _scales"):
        rigid_gaussians(d["means"], d["quats"], d["opacity_logits"], d["log_scales"], d["features_dc"], d["features_rest"], d["point_ids"], pose,
                        torch.from_numpy(g["camera_center"]).to(dev), step=12000)
