"""-m "not gpu": the CPU oracle and the host-side mirrors pinned against golden vectors generated from the
imported reference (tests/gen_golden.py; fixtures in tests/golden/).  Tolerances are fp32 round-off of a
different but equivalent operation order unless stated."""
import os

import numpy as np
import pytest
import torch

from emd_amd import camera, motion
from oracle import cpu_oracle as co
from oracle import torch_ref as tr

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ld = lambda n: np.load(os.path.join(G, n))


def test_sh_colour_matches_reference_eval_sh():
    z = ld("s3g_sh.npz")
    dirs_raw = z["xyz"] - z["campos"]
    for deg in range(4):
        got = co.sh_forward(deg, dirs_raw, z["shs"])                      # C oracle (normalises dirs itself)
        np.testing.assert_allclose(got, z[f"sh_deg{deg}"], rtol=0, atol=3e-6)
        t = tr.eval_sh(deg, torch.tensor(z["shs"]).transpose(1, 2), torch.tensor(z["dirs"]))
        np.testing.assert_allclose(t.numpy(), z[f"sh_deg{deg}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(np.maximum(got + 0.5, 0), z[f"rgb_deg{deg}"], rtol=0, atol=3e-6)


def test_cov3d_matches_reference_build_covariance():
    z = ld("s3g_cov.npz")
    rn = z["rots_raw"] / np.linalg.norm(z["rots_raw"], axis=1, keepdims=True)   # reference normalises inside build_rotation
    for mod in (1.0, 0.5):
        got = co.cov3d(z["scales"], mod, rn)
        ref = z[f"cov_mod{mod}"]
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-7 * max(1.0, float(np.abs(ref).max())))
        t = tr.covariance_from_scaling_rotation(torch.tensor(z["scales"]), mod, torch.tensor(rn.astype(np.float32)))
        np.testing.assert_allclose(t.numpy(), ref, rtol=2e-5, atol=1e-7)


def test_projection_and_camera_match_reference():
    c = ld("s3g_camera.npz")
    cam = camera.make_camera(c["R"].astype(np.float64), c["T"].astype(np.float64), float(c["fovx"]), float(c["fovy"]),
                             int(c["H"]), int(c["W"]))
    np.testing.assert_allclose(cam.world_view_transform.numpy(), c["world_view_transform"], atol=1e-6)
    np.testing.assert_allclose(cam.projection_matrix.numpy(), c["projection_matrix"], atol=1e-6)
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), c["full_proj_transform"], atol=2e-6)
    np.testing.assert_allclose(cam.camera_center.numpy(), c["camera_center"], atol=2e-6)
    assert abs(camera.focal2fov(float(c["fx"]), int(c["W"])) - float(c["fovx"])) < 1e-7
    # K-based projection (OmniRe cameras) reproduces the FoV-based one for a centred principal point
    K = torch.tensor([[float(c["fx"]), 0, int(c["W"]) / 2], [0, float(c["fy"]), int(c["H"]) / 2], [0, 0, 1]])
    np.testing.assert_allclose(camera.projection_from_K(K, int(c["W"]), int(c["H"])).numpy(), c["projection_matrix"], atol=1e-6)
    p = ld("s3g_proj.npz")
    ndc = tr.geom_transform_points(torch.tensor(p["points"]), torch.tensor(p["full_proj_transform"])).numpy()
    np.testing.assert_allclose(ndc, p["ndc"], rtol=1e-6, atol=1e-6)
    # the oracle's K1 projects to pixel centres ((ndc + 1) * size - 1) / 2 of the same NDC point
    H, W = int(c["H"]), int(c["W"])
    S = co.make_settings(H, W, np.tan(float(c["fovx"]) / 2), np.tan(float(c["fovy"]) / 2), [0, 0, 0], p["world_view_transform"],
                         p["full_proj_transform"], 0, c["camera_center"])
    n = p["points"].shape[0]
    sc = co.Scene(p["points"], np.full(n, 0.5, np.float32), colors_precomp=np.zeros((n, 3), np.float32),
                  scales=np.full((n, 3), 0.05, np.float32), rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1)))
    pre = co.preprocess(S, sc)
    vis = pre["radii"] > 0
    assert vis.sum() > 10
    np.testing.assert_allclose(pre["means2D"][vis, 0], ((p["ndc"][vis, 0] + 1) * W - 1) / 2, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(pre["means2D"][vis, 1], ((p["ndc"][vis, 1] + 1) * H - 1) / 2, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(pre["depths"][vis], p["view"][vis, 2] * (p["points"][vis] @ p["world_view_transform"][:3, 3] + p["world_view_transform"][3, 3]),
                               rtol=1e-5)  # view / w with w = 1 for an affine W2C


def test_quaternion_algebra_matches_reference():
    z = ld("s3g_quat.npz")
    out = tr.quat_mult(torch.tensor(z["q1"]), torch.tensor(z["q2"]))
    out = out / out.norm(dim=1, keepdim=True)
    np.testing.assert_allclose(out.numpy(), z["out"], atol=1e-6)
    o = ld("or_quat.npz")
    q = torch.tensor(o["q"])
    np.testing.assert_allclose(tr.build_rotation(q / q.norm(dim=1, keepdim=True)).numpy(), o["rotmat"], atol=1e-6)
    np.testing.assert_allclose(motion.quat_mult(torch.tensor(o["q1"]), torch.tensor(o["q2"])).numpy(), o["mult"], atol=1e-6)
    np.testing.assert_allclose(motion.interpolate_quats(torch.tensor(o["qa"]), torch.tensor(o["qb"])).numpy(), o["interp"], atol=2e-6)


def _heads_from_golden(z, requires_grad=False):
    """The checker's restatement of the track-offset heads (oracle/torch_ref.track_offsets); the product's TrackOffsetHeads runs
    on the GPU only and is checked against the same golden file in tests/test_motion_sh_gpu.py."""
    weight = torch.tensor(z["temporal_weight"])
    heads = {name: (torch.tensor(z[name + "_w"]), torch.tensor(z[name + "_b"])) for name in ("track_rot_c", "track_rot_f", "track_trans_c", "track_trans_f")}
    return lambda frame, num_frames, emb, ids, step: tr.track_offsets(weight, heads, frame, num_frames, emb, ids, step)


@pytest.mark.parametrize("tag", ["train", "train5", "test", "test_edge"])
def test_rigid_motion_matches_reference_rigidnodes(tag):
    """Pose table (host mirror) + per-point transform (C oracle) == RigidNodes.transform_means/quats + opacity mask."""
    z = ld("or_rigid.npz")
    frame, in_test = int(z[f"{tag}_frame"]), bool(z[f"{tag}_in_test"])
    heads = _heads_from_golden(z)
    ids = torch.tensor(z["point_ids"].astype(np.int64))
    with torch.no_grad():
        tt, trq = heads(frame, int(z["num_frames"]), torch.tensor(z["embeddings"]), ids, int(z["step"]))
    np.testing.assert_allclose(tt.numpy(), z[f"{tag}_track_trans"], atol=2e-6)
    np.testing.assert_allclose(trq.numpy(), z[f"{tag}_track_rot"], atol=2e-6)
    pose = motion.build_actor_pose(torch.tensor(z["instances_quats"]), torch.tensor(z["instances_trans"]),
                                   torch.tensor(z["instances_fv"]), frame, tt, trq, in_test_set=in_test)
    opac = 1 / (1 + np.exp(-z["opacity_logits"][:, 0]))
    wm, wq, wo = co.motion_forward(z["means"], z["quats"], opac, z["point_ids"], pose.numpy())
    np.testing.assert_allclose(wm, z[f"{tag}_world_means"], rtol=1e-6, atol=5e-6)
    np.testing.assert_allclose(wq, z[f"{tag}_world_quats_act"], atol=2e-6)
    np.testing.assert_allclose(wo, z[f"{tag}_opacity"][:, 0], atol=1e-6)
    if in_test and tag == "test":
        assert not np.allclose(pose[:, :4].numpy(), motion.quat_act(torch.tensor(z["instances_quats"][frame])).numpy()), \
            "test-time pose interpolation must have been exercised"


@pytest.mark.parametrize("tag", ["train", "train5"])
def test_rigid_motion_gradients_match_reference(tag):
    """Autograd through (track heads -> pose table -> torch restatement of the per-point transform) reproduces
    the reference's gradients on local means / quats and the per-frame actor poses."""
    z = ld("or_rigid.npz")
    frame = int(z[f"{tag}_frame"])
    heads = _heads_from_golden(z)
    t = lambda k: torch.tensor(z[k], requires_grad=True)
    means, quats, iq, it = t("means"), t("quats"), t("instances_quats"), t("instances_trans")
    ids = torch.tensor(z["point_ids"].astype(np.int64))
    tt, trq = heads(frame, int(z["num_frames"]), torch.tensor(z["embeddings"]), ids, int(z["step"]))
    pose = motion.build_actor_pose(iq, it, torch.tensor(z["instances_fv"]), frame, tt, trq)
    wm, wq, _ = tr.motion_transform(means, quats, None, ids, pose)
    ((wm * torch.tensor(z[f"{tag}_gm"])).sum() + (wq * torch.tensor(z[f"{tag}_gq"])).sum()).backward()
    for got, name in ((means.grad, "grad_means"), (quats.grad, "grad_quats"), (it.grad, "grad_instances_trans"),
                      (iq.grad, "grad_instances_quats")):
        ref = z[f"{tag}_{name}"]
        assert np.abs(got.numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), name


@pytest.mark.parametrize("stage", ["coarse", "fine"])
def test_render_glue_matches_reference_render(stage):
    """Settings record and boundary tensors of the reference render() (captured with a recording stand-in for
    diff_gauss) are reproduced by the host-side glue: camera -> settings, deformation adds, activations."""
    from emd_amd import model
    z = ld("s3g_render.npz")
    cam = camera.make_camera(z["R"].astype(np.float64), z["T"].astype(np.float64), float(z["fovx"]), float(z["fovy"]), int(z["H"]), int(z["W"]))
    rs = model.raster_settings_for(cam, torch.tensor(z["bg"]), int(z["active_sh_degree"]))
    assert (rs.image_height, rs.image_width) == (int(z[f"{stage}_image_height"]), int(z[f"{stage}_image_width"]))
    assert abs(rs.tanfovx - float(z[f"{stage}_tanfovx"])) < 1e-7 and abs(rs.tanfovy - float(z[f"{stage}_tanfovy"])) < 1e-7
    assert rs.scale_modifier == float(z[f"{stage}_scale_modifier"]) and rs.sh_degree == int(z[f"{stage}_sh_degree"])
    assert int(rs.prefiltered) == int(z[f"{stage}_prefiltered"]) and int(rs.debug) == int(z[f"{stage}_debug"])
    np.testing.assert_allclose(rs.viewmatrix.numpy(), z[f"{stage}_viewmatrix"], atol=1e-6)
    np.testing.assert_allclose(rs.projmatrix.numpy(), z[f"{stage}_projmatrix"], atol=2e-6)
    np.testing.assert_allclose(rs.campos.numpy(), z[f"{stage}_campos"], atol=2e-6)
    np.testing.assert_allclose(rs.bg.numpy(), z[f"{stage}_bg"], atol=0)
    t = lambda k: torch.tensor(z[k])
    dd = None
    if stage == "fine":
        dd = {lvl: {k: t(f"ddict_{lvl}_{k}") for k in ("dx", "do", "dshs")} for lvl in ("coarse", "fine")}
    b = model.boundary_tensors(t("xyz"), t("scaling"), t("rotation"), t("opacity"), t("features"), stage, dd)
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        np.testing.assert_allclose(b[k].numpy(), z[f"{stage}_{k}"], rtol=1e-6, atol=1e-6, err_msg=k)
    if stage == "fine":
        assert np.abs(z["fine_means3D"] - z["coarse_means3D"]).max() > 1e-4, "the deformation must have been exercised"


def test_densification_stats_formula_matches_reference():
    """dp.densification_stats (the per-view statistics reduced across ranks) against the reference's add_densification_stats +
    max_radii2D update accumulated over three views (tests/golden/s3g_densify.npz)."""
    import os
    import numpy as np
    import torch
    from emd_amd import dp
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "s3g_densify.npz"))
    accum, denom, maxr = torch.from_numpy(g["accum0"]).clone(), torch.from_numpy(g["denom0"]).clone(), torch.from_numpy(g["maxr0"]).clone()
    for v in range(3):
        gn, dn, mr = dp.densification_stats(torch.from_numpy(g[f"grad{v}"]), torch.from_numpy(g[f"radii{v}"]))
        accum += gn
        denom += dn
        maxr = torch.maximum(maxr, mr)
    np.testing.assert_allclose(accum.numpy(), g["accum"], rtol=1e-6)
    np.testing.assert_array_equal(denom.numpy(), g["denom"])
    np.testing.assert_array_equal(maxr.numpy(), g["maxr"])


def test_combine_dynamic_static_mixing_matches_reference_render():
    """model.mix_dynamic_static against the reference render() run with `combine_dynamic_static` (tests/golden/s3g_render_combined.npz:
    S3Gaussian/gaussian_renderer/__init__.py:118-138): the SH path (one call) and the precomputed-colour path with the dynamic / static
    sets of the decomposition passes; colours through the oracle's SH evaluation (pre_compute_colors, :19-25)."""
    from emd_amd import model
    from oracle import cpu_oracle as co
    z = ld("s3g_render_combined.npz")
    t = lambda k: torch.tensor(z[k])
    dd = {lvl: {k: t(f"ddict_{lvl}_{k}") for k in ("dx", "do", "dshs")} for lvl in ("coarse", "fine")}
    xyz, scaling, rotation, opacity, feats = t("xyz"), t("scaling"), t("rotation"), t("opacity"), t("features")
    fin = model.apply_deform(xyz, scaling, rotation, opacity, feats, dd["coarse"], dd["fine"])
    o_dyn, o_sta = torch.sigmoid(fin[3]), torch.sigmoid(opacity)
    # (A) SH path
    op, shs, col = model.mix_dynamic_static(o_dyn, o_sta, fin[4], feats)
    assert col is None
    np.testing.assert_allclose(op.numpy(), z["sh_main_opacities"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(shs.numpy(), z["sh_main_shs"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fin[0].numpy(), z["sh_main_means3D"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(torch.exp(fin[1]).numpy(), z["sh_main_scales"], rtol=1e-6)
    assert float(op.max()) > 1.0                                   # the summed opacities do leave [0, 1]: the rasterizer's alpha clamp handles it
    # (B) precomputed colours: directions from the undeformed means
    deg = int(z["active_sh_degree"])
    dirs = (xyz - t("camera_center").reshape(1, 3)).numpy()
    c = lambda s_: np.maximum(co.sh_forward(deg, dirs, s_.numpy()) + 0.5, 0.0)
    col_dyn, col_sta = torch.tensor(c(fin[4])), torch.tensor(c(feats))
    np.testing.assert_allclose(col_dyn.numpy(), z["coarse_set_colors_precomp"], rtol=1e-5, atol=2e-6)       # the dynamic set (coarse level)
    np.testing.assert_allclose(col_sta.numpy(), z["fine_set_colors_precomp"], rtol=1e-5, atol=2e-6)         # the static set (fine level)
    op2, shs2, col2 = model.mix_dynamic_static(o_dyn, o_sta, None, None, col_dyn, col_sta)
    assert shs2 is None
    np.testing.assert_allclose(col2.numpy(), z["main_colors_precomp"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(op2.numpy(), z["main_opacities"], rtol=1e-6, atol=1e-7)
    # the sets of the decomposition passes: dynamic = deformed means with their own opacity, static = the undeformed copy
    np.testing.assert_allclose(z["coarse_set_means3D"], fin[0].numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(z["coarse_set_opacities"], o_dyn.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(z["fine_set_means3D"], xyz.numpy(), atol=0)
    np.testing.assert_allclose(z["fine_set_opacities"], o_sta.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(z["coarse_fine_set_means3D"], xyz.numpy(), atol=0)
    for lvl, d in (("coarse", dd["coarse"]["dx"]), ("fine", dd["fine"]["dx"]), ("coarse_fine", dd["coarse"]["dx"] - dd["fine"]["dx"])):
        a = d.abs()
        np.testing.assert_allclose((a / a.max(dim=0, keepdim=True)[0]).numpy(), z[f"{lvl}_dx_colors_precomp"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(z[f"{lvl}_dx_opacities"], z["main_opacities"], atol=0)
