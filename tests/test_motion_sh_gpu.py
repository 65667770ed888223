"""-m gpu: the stand-alone HIP motion / SH entry points (through the C ABI) against the golden vectors of the
imported reference (RigidNodes, eval_sh) and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from emd_amd import gsplat_api, motion
from oracle import cpu_oracle as co
from tests.test_golden_cpu import ld


def _heads_from_golden(z):
    """The product's track-offset heads (HIP temporal-embedding rows, batched over actors) loaded with the golden file's weights."""
    A = z["instances_quats"].shape[1]
    h = motion.TrackOffsetHeads(A)
    h.weight.data = torch.tensor(z["temporal_weight"])
    for name in ("track_rot_c", "track_rot_f", "track_trans_c", "track_trans_f"):
        getattr(h, name).weight.data = torch.tensor(z[name + "_w"])
        getattr(h, name).bias.data = torch.tensor(z[name + "_b"])
    return h

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag", ["train", "train5", "test"])
def test_hip_motion_forward_matches_reference(tag):
    z = ld("or_rigid.npz")
    frame, in_test = int(z[f"{tag}_frame"]), bool(z[f"{tag}_in_test"])
    heads = _heads_from_golden(z).to(DEV)
    g = lambda k: torch.tensor(z[k]).to(DEV)
    ids = g("point_ids")
    with torch.no_grad():
        tt, trq = heads(frame, int(z["num_frames"]), g("embeddings"), ids, int(z["step"]))
        # the heads themselves (one HIP launch per temporal row for all actors) against the reference's embedding_track_*_offset
        np.testing.assert_allclose(tt.cpu().numpy(), z[f"{tag}_track_trans"], atol=2e-6)
        np.testing.assert_allclose(trq.cpu().numpy(), z[f"{tag}_track_rot"], atol=2e-6)
        pose = motion.build_actor_pose(g("instances_quats"), g("instances_trans"), g("instances_fv"), frame, tt, trq, in_test)
        wm, wq, wo = motion.transform_gaussians(g("means"), g("quats"), torch.sigmoid(g("opacity_logits")), ids, pose)
    np.testing.assert_allclose(wm.cpu().numpy(), z[f"{tag}_world_means"], rtol=1e-6, atol=5e-6)
    np.testing.assert_allclose(wq.cpu().numpy(), z[f"{tag}_world_quats_act"], atol=2e-6)
    np.testing.assert_allclose(wo.cpu().numpy(), z[f"{tag}_opacity"][:, 0], atol=1e-6)


@pytest.mark.parametrize("tag", ["train", "train5"])
def test_hip_motion_backward_matches_reference(tag):
    z = ld("or_rigid.npz")
    frame = int(z[f"{tag}_frame"])
    heads = _heads_from_golden(z).to(DEV)
    t = lambda k: torch.tensor(z[k]).to(DEV).requires_grad_(True)
    means, quats, iq, it = t("means"), t("quats"), t("instances_quats"), t("instances_trans")
    ids = torch.tensor(z["point_ids"]).to(DEV)
    tt, trq = heads(frame, int(z["num_frames"]), torch.tensor(z["embeddings"]).to(DEV), ids, int(z["step"]))
    pose = motion.build_actor_pose(iq, it, torch.tensor(z["instances_fv"]).to(DEV), frame, tt, trq)
    wm, wq, _ = motion.transform_gaussians(means, quats, None, ids, pose)
    ((wm * torch.tensor(z[f"{tag}_gm"]).to(DEV)).sum() + (wq * torch.tensor(z[f"{tag}_gq"]).to(DEV)).sum()).backward()
    for got, name in ((means.grad, "grad_means"), (quats.grad, "grad_quats"), (it.grad, "grad_instances_trans"),
                      (iq.grad, "grad_instances_quats")):
        ref = z[f"{tag}_{name}"]
        assert np.abs(got.cpu().numpy() - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), name
    # the learned track heads receive gradients through the pose table as well
    assert heads.track_trans_c.weight.grad is not None and heads.track_trans_c.weight.grad.abs().sum() > 0
    assert heads.weight.grad.abs().sum() > 0


def test_hip_motion_mixed_static_and_residual_matches_oracle():
    rng = np.random.default_rng(0)
    n, A = 5000, 7
    means = rng.standard_normal((n, 3)).astype(np.float32)
    quats = rng.standard_normal((n, 4)).astype(np.float32)
    opac = rng.random(n).astype(np.float32)
    ids = rng.integers(-1, A, n).astype(np.int32)
    pose = rng.standard_normal((A, 12)).astype(np.float32)
    pose[:, 0:4] /= np.linalg.norm(pose[:, 0:4], axis=1, keepdims=True)
    pose[:, 8:12] /= np.linalg.norm(pose[:, 8:12], axis=1, keepdims=True)
    pose[:, 7] = (rng.random(A) > 0.3)
    rdx = (0.05 * rng.standard_normal((n, 3))).astype(np.float32)
    rdq = (0.05 * rng.standard_normal((n, 4))).astype(np.float32)
    wm, wq, wo = co.motion_forward(means, quats, opac, ids, pose, rdx, rdq)
    g = lambda a: torch.tensor(a).to(DEV)
    hm, hq, ho = motion.transform_gaussians(g(means), g(quats), g(opac), g(ids), g(pose), g(rdx), g(rdq))
    np.testing.assert_array_equal(hm.cpu().numpy().view(np.uint32), wm.view(np.uint32))   # same pinned fp32 order: bit-exact
    np.testing.assert_array_equal(hq.cpu().numpy().view(np.uint32), wq.view(np.uint32))
    np.testing.assert_array_equal(ho.cpu().numpy().view(np.uint32), wo.view(np.uint32))


def test_hip_spherical_harmonics_matches_reference_and_autograd():
    z = ld("s3g_sh.npz")
    dirs = torch.tensor(z["xyz"] - z["campos"]).to(DEV)
    for deg in range(4):
        out = gsplat_api.spherical_harmonics(deg, dirs, torch.tensor(z["shs"]).to(DEV))
        np.testing.assert_allclose(out.cpu().numpy(), z[f"sh_deg{deg}"], atol=3e-6)
    # backward vs fp64 autograd of the reference polynomial
    from oracle import torch_ref as tr
    d64 = torch.tensor(z["xyz"] - z["campos"], dtype=torch.float64, requires_grad=True)
    c64 = torch.tensor(z["shs"], dtype=torch.float64, requires_grad=True)
    w = torch.randn(z["shs"].shape[0], 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    (tr.eval_sh(3, c64.transpose(1, 2), d64 / d64.norm(dim=1, keepdim=True)) * w).sum().backward()
    dg = dirs.clone().requires_grad_(True)
    cg = torch.tensor(z["shs"]).to(DEV).requires_grad_(True)
    (gsplat_api.spherical_harmonics(3, dg, cg) * w.float().to(DEV)).sum().backward()
    np.testing.assert_allclose(cg.grad.cpu().numpy(), c64.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(dg.grad.cpu().numpy(), d64.grad.numpy(), atol=1e-5)


@pytest.mark.parametrize("with_offsets", [False, True])
def test_fused_actor_pose_table_matches_host_mirror(with_offsets):
    """emd_actor_pose_forward/backward (one launch each) == build_actor_pose (torch ops, itself pinned by the
    reference golden or_rigid.npz), values and gradients; NaN track offsets are skipped as rigid.py:528,559 do."""
    g = torch.Generator().manual_seed(9)
    F_, A = 7, 37
    iq = (torch.randn(F_, A, 4, generator=g) * 1.3).to(DEV).requires_grad_(True)
    it = torch.randn(F_, A, 3, generator=g).to(DEV).requires_grad_(True)
    fv = (torch.rand(F_, A, generator=g) > 0.2).to(DEV)
    tt = tr_ = None
    if with_offsets:
        tt = (0.1 * torch.randn(A, 3, generator=g)).to(DEV)
        tr_ = torch.randn(A, 4, generator=g).to(DEV)
        tt[3] = float("nan")
        tr_[5, 1] = float("nan")
        tt.requires_grad_(True)
        tr_.requires_grad_(True)
    frame = 4
    ref = motion.build_actor_pose(iq, it, fv, frame, tt, tr_)
    w = torch.randn(A, 12, generator=g).to(DEV)
    (ref * w).sum().backward()
    ref_grads = [iq.grad.clone(), it.grad.clone()] + ([torch.nan_to_num(tt.grad.clone()), torch.nan_to_num(tr_.grad.clone())] if with_offsets else [])
    for t in (iq, it, tt, tr_):
        if t is not None:
            t.grad = None
    got = motion.actor_pose_table(iq, it, fv, frame, tt, tr_)
    torch.testing.assert_close(got, ref.detach(), rtol=1e-6, atol=1e-6)
    (got * w).sum().backward()
    got_grads = [iq.grad, it.grad] + ([tt.grad, tr_.grad] if with_offsets else [])
    for a, b in zip(got_grads, ref_grads):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


def test_fused_l1_loss_matches_torch():
    from emd_amd.model import l1_loss
    g = torch.Generator().manual_seed(4)
    for shape in ((3, 67, 101), (3, 1066, 1600), (5,)):
        a = torch.randn(*shape, generator=g).to(DEV).requires_grad_(True)
        b = torch.randn(*shape, generator=g).to(DEV)
        b.view(-1)[0] = a.detach().view(-1)[0]          # exact tie -> zero gradient like torch.sign
        ref = torch.abs(a - b).mean()
        (ref * 2.5).backward()
        gref = a.grad.clone()
        a.grad = None
        out = l1_loss(a, b)
        (out * 2.5).backward()
        torch.testing.assert_close(out, ref.detach(), rtol=2e-5, atol=1e-7)
        torch.testing.assert_close(a.grad, gref, rtol=1e-6, atol=0)


def test_factored_sh_gradient_equals_dense_average():
    """View-parallel SH-gradient exchange (emd_amd.dp): the dense dL/dshs of two views, averaged, equals the gradient rebuilt
    from the two [N,3] colour-gradient factors and the two camera centres -- with the fused explicit motion in front."""
    import numpy as np
    import torch
    from emd_amd import dp
    from tests.helpers import make_case, run_hip
    dev = torch.device("cuda", 0)
    dense, factors, campos, case0 = [], [], [], None
    for yaw in (0.0, 25.0):
        case = make_case(n=3000, H=64, W=96, seed=77, motion=True, yaw=yaw)
        case0 = case0 or case
        dense.append(run_hip(case, backward=True)["grads"]["shs"])
        out = run_hip(case, backward=True, factored_sh_grad=True)
        assert out["grads"]["shs"] is None                       # the dense tensor is never written in this mode
        factors.append(out["call"].sh_color_grad.clone())        # published in THIS call's record
        campos.append(torch.as_tensor(case["cam"].camera_center, dtype=torch.float32).reshape(3))
    expect = 0.5 * (dense[0] + dense[1])
    got = dp.sh_grad_from_factors(case0["means3D"].to(dev), torch.stack(campos).to(dev), torch.stack(factors), case0["sh_degree"], 16,
                                  actor_ids=case0["actor_ids"].to(dev), actor_pose=case0["actor_pose"].to(dev), scale=0.5).cpu().numpy()
    assert np.count_nonzero(expect) > 1000
    np.testing.assert_allclose(got, expect, rtol=2e-5, atol=1e-7 * float(np.abs(expect).max()) + 1e-12)


def test_fused_densification_stats_match_reference_golden():
    """emd_densification_stats (one launch, no boolean-mask indexing) over three views against the statistics the reference's
    add_densification_stats / max_radii2D lines produced (tests/golden/s3g_densify.npz)."""
    import os
    import numpy as np
    import torch
    from emd_amd import dp
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "s3g_densify.npz"))
    accum = torch.from_numpy(g["accum0"]).to(dev).contiguous()
    denom = torch.from_numpy(g["denom0"]).to(dev).contiguous()
    maxr = torch.from_numpy(g["maxr0"]).to(dev).contiguous()
    for v in range(3):
        dp.add_densification_stats(torch.from_numpy(g[f"grad{v}"]).to(dev), torch.from_numpy(g[f"radii{v}"]).to(dev), accum, denom, maxr)
    np.testing.assert_allclose(accum.cpu().numpy(), g["accum"], rtol=1e-6)
    np.testing.assert_array_equal(denom.cpu().numpy(), g["denom"])
    np.testing.assert_array_equal(maxr.cpu().numpy(), g["maxr"])


@pytest.mark.parametrize("frame,step", [(0, 0), (3, 12000), (4, 40000)])
def test_fused_track_heads_match_the_checker_forward_and_backward(frame, step):
    """emd_track_heads_forward/backward (all actors, both levels, one launch each way) against autograd through the checker's
    restatement of rigid.py:150-246 (oracle/torch_ref.track_offsets, itself pinned by or_rigid.npz): offsets and the gradients of
    the temporal tables, the point embeddings and all eight head tensors."""
    from oracle import torch_ref as tr
    z = ld("or_rigid.npz")
    F_ = int(z["num_frames"])
    heads = _heads_from_golden(z)
    g = torch.Generator().manual_seed(frame + 1)
    with torch.no_grad():                                   # larger head weights than the golden file's: well-conditioned gradients
        for lin in (heads.track_trans_c, heads.track_trans_f, heads.track_rot_c, heads.track_rot_f):
            lin.weight.copy_(0.3 * torch.randn(lin.weight.shape, generator=g))
            lin.bias.copy_(0.1 * torch.randn(lin.bias.shape, generator=g))
        heads.weight.copy_(0.5 * torch.randn(heads.weight.shape, generator=g))
    emb = torch.tensor(z["embeddings"])
    ids = torch.tensor(z["point_ids"].astype(np.int64))
    A = heads.weight.shape[0]
    gt, gr = torch.randn(A, 3, generator=g), torch.randn(A, 4, generator=g)
    # checker (CPU autograd)
    w0 = heads.weight.detach().clone().requires_grad_(True)
    e0 = emb.clone().requires_grad_(True)
    hp = {n: (getattr(heads, n).weight.detach().clone().requires_grad_(True), getattr(heads, n).bias.detach().clone().requires_grad_(True))
          for n in ("track_rot_c", "track_rot_f", "track_trans_c", "track_trans_f")}
    t0, r0 = tr.track_offsets(w0, hp, frame, F_, e0, ids, step)
    ((t0 * gt).sum() + (r0 * gr).sum()).backward()
    # product (GPU)
    hd = heads.to(DEV)
    e1 = emb.to(DEV).requires_grad_(True)
    t1, r1 = hd(frame, F_, e1, ids.to(DEV), step)
    ((t1 * gt.to(DEV)).sum() + (r1 * gr.to(DEV)).sum()).backward()
    np.testing.assert_allclose(t1.detach().cpu().numpy(), t0.detach().numpy(), rtol=1e-5, atol=1e-5)     # 36-term fp32 dot products of O(1) terms
    np.testing.assert_allclose(r1.detach().cpu().numpy(), r0.detach().numpy(), rtol=1e-5, atol=1e-5)

    def close(a, b, what):
        a, b = a.detach().cpu().numpy(), b.detach().numpy()
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-7, (what, np.abs(a - b).max(), np.abs(b).max())
    close(hd.weight.grad, w0.grad, "temporal tables")
    close(e1.grad, e0.grad, "embeddings")
    for n in hp:
        close(getattr(hd, n).weight.grad, hp[n][0].grad, n + ".weight")
        close(getattr(hd, n).bias.grad, hp[n][1].grad, n + ".bias")


@pytest.mark.parametrize("frame,step,on_device", [(0, 0, False), (3, 12000, False), (4, 40000, True)])
def test_one_launch_actor_chain_equals_the_three_launch_path(frame, step, on_device):
    """emd_tracked_pose_forward/backward (embedding sums -> track heads -> pose row in ONE launch each way, round 3) against the
    pinned three-launch path (emd_track_heads_* + emd_actor_pose_*): the pose table bit for bit (same arithmetic, no contraction in
    either), every gradient -- temporal tables, embeddings, the eight head tensors, the dense [F, A, .] pose tables -- to rounding,
    twice in a row, with host and with device-resident frame / step."""
    z = ld("or_rigid.npz")
    F_ = int(z["num_frames"])
    heads = _heads_from_golden(z)
    g = torch.Generator().manual_seed(frame + 11)
    with torch.no_grad():
        for lin in (heads.track_trans_c, heads.track_trans_f, heads.track_rot_c, heads.track_rot_f):
            lin.weight.copy_(0.3 * torch.randn(lin.weight.shape, generator=g))
            lin.bias.copy_(0.1 * torch.randn(lin.bias.shape, generator=g))
        heads.weight.copy_(0.5 * torch.randn(heads.weight.shape, generator=g))
    hd = heads.to(DEV)
    A = hd.weight.shape[0]
    ids = torch.tensor(z["point_ids"].astype(np.int64)).to(DEV)
    assert bool((ids[1:] >= ids[:-1]).all()), "the golden stores an actor's points contiguously"
    emb0 = torch.tensor(z["embeddings"])
    iq0 = torch.randn(F_, A, 4, generator=g) * 1.3
    it0 = torch.randn(F_, A, 3, generator=g)
    fv = (torch.rand(F_, A, generator=g) > 0.2).to(DEV)
    w = torch.randn(A, 12, generator=g).to(DEV)
    params = [hd.weight] + [p for n in ("track_trans_c", "track_trans_f", "track_rot_c", "track_rot_f") for p in (getattr(hd, n).weight, getattr(hd, n).bias)]

    def run(fused):
        for p in params:
            p.grad = None
        e = emb0.to(DEV).requires_grad_(True)
        iq, it = iq0.to(DEV).requires_grad_(True), it0.to(DEV).requires_grad_(True)
        fr = torch.tensor([frame], dtype=torch.int32, device=DEV) if on_device else frame
        st = torch.tensor([step], dtype=torch.int64, device=DEV) if on_device else step
        if fused:
            pose = hd.pose_table(iq, it, fv, fr, e, ids, st)
        else:
            tt, trq = hd(fr, F_, e, ids, st)
            pose = motion.actor_pose_table(iq, it, fv, fr, tt, trq)
        (pose * w).sum().backward()
        return pose.detach().clone(), [e.grad.clone(), iq.grad.clone(), it.grad.clone()] + [p.grad.clone() for p in params]

    ref_pose, ref_g = run(False)
    for rep in range(2):
        pose, got = run(True)
        np.testing.assert_array_equal(pose.cpu().numpy(), ref_pose.cpu().numpy())
        for a, b in zip(got, ref_g):
            a, b = a.cpu().numpy(), b.cpu().numpy()
            assert a.shape == b.shape
            assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max() + 1e-7, (rep, np.abs(a - b).max(), np.abs(b).max())
