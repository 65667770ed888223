"""Shared test plumbing: seeded cases, the HIP path (through the C ABI), the CPU oracle, and the comparisons.

The oracle is the checker only; the product path under test is emd_amd.GaussianRasterizer -> libemd_raster.so.
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from emd_amd import scenes  # noqa: E402
from oracle import cpu_oracle as co  # noqa: E402

IMAGE_TOL = 1e-4   # north_star: rendered-image L_inf <= 1e-4 vs reference
# Gradients (float atomics and wave scans re-associate fp32 sums; the oracle sums the fp32 partials exactly, in double):
#   element-wise   |hip - oracle| <= GRAD_RTOL * |oracle| + GRAD_ATOL_FRAC * max|oracle|     for EVERY element, and
#   whole tensor   ||hip - oracle||_2 <= GRAD_REL_L2 * ||oracle||_2
# This bar is applied per KERNEL: (a) to the render backward's per-Gaussian sums (d mean2D, d conic, d opacity, d colour, d depth)
# against the oracle's, and (b) to the projection backward's outputs against the oracle's projection backward FED WITH THE KERNEL'S
# OWN render gradients.  The chain conic -> cov2D -> covariance -> (scale, rotation) divides by det^2 and cancels terms: it
# amplifies admissible differences of its inputs, so the END-TO-END comparison (c) of scales / rotations carries its own, looser
# absolute floor END2END_ATOL_FRAC (every other end-to-end gradient keeps the strict bar).
GRAD_RTOL = 1e-4
GRAD_ATOL_FRAC = 1e-6
GRAD_REL_L2 = 1e-5
END2END_ATOL_FRAC = 1e-4     # (observed worst over the 40-scene sweep and the full-size scenes: 9.3e-5 of max|ref| on one rotation entry of sweep
                             #  scene s1021 whose render gradients pass (a) and whose projection backward passes (b): pure amplification)
END2END_REL_L2 = 1e-4
# Round 3: that entry belongs to a nearly degenerate footprint (conic 0.1223 / 0.1265 / 0.1331, det 2.8e-4, radius 89 px).  Its
# three conic gradients from K7 equal the oracle's to ONE ulp (-280.30725 vs -280.30728, ...), K8 on them equals the oracle's K8 on
# them bit for bit -- and the rotation gradient still differs from the end-to-end oracle value by 6e-4 relative, because the chain
# divides by det^2: whether that lands at 0.9 or at 1.5 of the floor above depends on the order of K7's float atomics.  No fp32
# implementation can do better than the response of the chain to a rounding of its inputs, so the END-TO-END comparison (c) -- and
# only (c); the per-kernel bars (a) and (b) are what they were -- additionally admits END2END_ULP_RESPONSES times the change of the
# ORACLE's own projection backward when every render gradient it is fed moves by one fp32 ulp (measured per element, per test, on
# the oracle; it applies to every output of that chain -- the same Gaussian's means3D entry sat at 1.17 of the strict bar).
END2END_ULP_RESPONSES = 8.0
# Round 4: the allowance is OBSERVABLE and BOUNDED.  Every comparison that is given one records how many elements exceeded the plain
# bound (i.e. leaned on the allowance) and the largest multiple of the one-ulp response any of them used (ALLOWANCE_LOG, printed by the
# full-size tests); at most ALLOWANCE_MAX_FRACTION of a tensor's elements (but never fewer than ALLOWANCE_MIN_COUNT) may lean on it, and
# none by more than END2END_ULP_RESPONSES responses (16 in round 3, 8 now).  A systematic error of the projection backward confined to
# ill-conditioned footprints would show up as a count, not hide behind the allowance.
ALLOWANCE_MAX_FRACTION = 1e-5
ALLOWANCE_MIN_COUNT = 2
ALLOWANCE_LOG = []
# Per-actor pose gradients are sums over the thousands of Gaussians of an actor, with heavy cancellation (an actor's points pull its
# pose in all directions: the sum can be a thousand times smaller than its terms).  A bound relative to the RESULT is meaningless
# there; theirs is relative to the sum of the magnitudes of the terms (condition-aware): see pose_bound().
POSE_TERM_RTOL = 2e-6


def make_case(n=2000, H=64, W=96, seed=0, sh_degree=3, colors_precomp=False, cov_precomp=False, motion=False,
              residual=False, bg=(0.1, 0.2, 0.3), scale_mult=3.0, actors=3, yaw=0.0):
    """A street-like scene squeezed into the frustum of a small camera so that most Gaussians are visible."""
    sc = scenes.make_static_scene(n, seed=seed)
    cam = scenes.small_camera(H, W, yaw=yaw)
    means = sc.means.clone()
    means[:, 0] = means[:, 0] * 0.25 + 1.0
    means[:, 1] *= 0.3
    means[:, 2] = means[:, 2] * 0.3 + 1.0
    # a few Gaussians behind / very near the camera and far outside the frustum exercise the cull paths
    k = max(n // 50, 1)
    means[:k, 0] = -means[:k, 0]
    means[k:2 * k, 0] = 0.15
    means[2 * k:3 * k, 1] += 40.0
    g = torch.Generator().manual_seed(seed + 1000)
    case = dict(N=n, H=H, W=W, sh_degree=sh_degree, bg=torch.tensor(bg, dtype=torch.float32), cam=cam,
                means3D=means, opacities=torch.sigmoid(sc.opacity_logits),
                scales=torch.exp(sc.log_scales) * scale_mult, rotations=sc.quats.clone(), shs=sc.shs.clone(),
                colors_precomp=None, cov3D_precomp=None, actor_ids=None, actor_pose=None, residual_dx=None,
                residual_dq=None, flags=co.F_NORMAL)
    if colors_precomp:
        case["colors_precomp"] = torch.rand(n, 3, generator=g)
        case["shs"] = None
    if cov_precomp:
        cov = co.cov3d(case["scales"].numpy(), 1.0, case["rotations"].numpy())
        case["cov3D_precomp"] = torch.from_numpy(cov)
        case["scales"] = None
        case["rotations"] = None
    if motion:
        A = actors
        ids = torch.full((n,), -1, dtype=torch.int32)
        n_dyn = n // 2
        ids[:n_dyn] = (torch.arange(n_dyn) * A // n_dyn).to(torch.int32)
        # actor points live in a local box; the pose puts them in front of the camera
        local = (torch.rand(n_dyn, 3, generator=g) - 0.5) * torch.tensor([4.5, 2.0, 1.6])
        means[:n_dyn] = local
        yaws = torch.rand(A, generator=g) * 2 * math.pi
        qm = torch.stack([torch.cos(yaws / 2), torch.zeros(A), torch.zeros(A), torch.sin(yaws / 2)], 1)
        dq = torch.randn(A, 4, generator=g) * 0.05 + torch.tensor([1.0, 0, 0, 0])
        qr = torch.from_numpy(_quat_mul_np(qm.numpy(), (dq / dq.norm(dim=1, keepdim=True)).numpy()))
        qr = qr / qr.norm(dim=1, keepdim=True)
        trans = torch.stack([torch.rand(A, generator=g) * 20 + 6, torch.rand(A, generator=g) * 8 - 4,
                             torch.full((A,), 1.2)], 1)
        valid = torch.ones(A, 1)
        valid[-1] = 0.0 if A > 2 else 1.0   # one invisible actor (instances_fv False, rigid.py:42-46)
        case["actor_pose"] = torch.cat([qm, trans, valid, qr], 1).float().contiguous()
        case["actor_ids"] = ids
        case["means3D"] = means
        # local quaternions are raw (un-normalised) for actor points, like RigidNodes._quats
        rot = case["rotations"].clone()
        rot[:n_dyn] = rot[:n_dyn] * (0.5 + torch.rand(n_dyn, 1, generator=g))
        case["rotations"] = rot
        case["flags"] |= co.F_MOTION
    if residual:
        case["residual_dx"] = 0.02 * torch.randn(n, 3, generator=g)
        if motion:
            case["residual_dq"] = 0.02 * torch.randn(n, 4, generator=g)
        case["flags"] |= co.F_MOTION
    gg = np.random.default_rng(seed + 5)
    case["dL_dcolor"] = gg.standard_normal((3, H, W)).astype(np.float32)
    case["dL_ddepth"] = (0.1 * gg.standard_normal((1, H, W))).astype(np.float32)
    case["dL_dalpha"] = gg.standard_normal((1, H, W)).astype(np.float32)
    return case


def _quat_mul_np(a, b):
    w1, x1, y1, z1 = a.T
    w2, x2, y2, z2 = b.T
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], 1).astype(np.float32)


def oracle_settings(case, near_plane=0.2):
    cam = case["cam"]
    return co.make_settings(case["H"], case["W"], cam.tanfovx, cam.tanfovy, case["bg"].numpy(),
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), case["sh_degree"],
                            cam.camera_center.numpy(), 1.0, near_plane)


def oracle_scene(case):
    n = lambda t: None if t is None else t.numpy()
    return co.Scene(n(case["means3D"]), n(case["opacities"]), shs=n(case["shs"]), colors_precomp=n(case["colors_precomp"]),
                    scales=n(case["scales"]), rotations=n(case["rotations"]), cov3D_precomp=n(case["cov3D_precomp"]),
                    actor_id=n(case["actor_ids"]), actor_pose=n(case["actor_pose"]), residual_dx=n(case["residual_dx"]),
                    residual_dq=n(case["residual_dq"]))


def run_oracle(case, backward=False):
    S = oracle_settings(case)
    sc = oracle_scene(case)
    pre, b, img = co.forward(S, sc, case["flags"])
    out = dict(pre=pre, bin=b, img=img, S=S, scene=sc)
    if backward:
        out["grads"] = co.backward(S, sc, pre, b, img, case["dL_dcolor"], case["dL_ddepth"], case["dL_dalpha"], None,
                                   case["flags"])
    return out


def run_hip(case, backward=False, device="cuda:0", absgrad=False, factored_sh_grad=False, keep_all_pairs=False):
    """The product path: emd_amd.GaussianRasterizer -> C ABI -> HIP kernels."""
    from emd_amd import GaussianRasterizationSettings, GaussianRasterizer
    cam = case["cam"]
    dev = torch.device(device)
    d = lambda t, rg=backward: None if t is None else t.to(dev).clone().requires_grad_(rg and t.is_floating_point())
    rs = GaussianRasterizationSettings(image_height=case["H"], image_width=case["W"], tanfovx=cam.tanfovx,
                                       tanfovy=cam.tanfovy, bg=case["bg"].to(dev), scale_modifier=1.0,
                                       viewmatrix=cam.world_view_transform.to(dev),
                                       projmatrix=cam.full_proj_transform.to(dev), sh_degree=case["sh_degree"],
                                       campos=cam.camera_center.to(dev), prefiltered=False, debug=True)
    T = dict(means3D=d(case["means3D"]), shs=d(case["shs"]), colors_precomp=d(case["colors_precomp"]),
             opacities=d(case["opacities"]), scales=d(case["scales"]), rotations=d(case["rotations"]),
             cov3Ds_precomp=d(case["cov3D_precomp"]), actor_pose=d(case["actor_pose"]),
             residual_dx=d(case["residual_dx"]), residual_dq=d(case["residual_dq"]))
    means2D = torch.zeros(case["N"], 3, device=dev, requires_grad=backward)
    rast = GaussianRasterizer(rs, compute_normal=True, absgrad=absgrad, factored_sh_grad=factored_sh_grad, keep_render_grads=backward,
                              keep_all_pairs=keep_all_pairs)   # options belong to this instance
    kw = {}
    if case["flags"] & co.F_MOTION:
        kw = dict(actor_ids=None if case["actor_ids"] is None else case["actor_ids"].to(dev), actor_pose=T["actor_pose"],
                  residual_dx=T["residual_dx"], residual_dq=T["residual_dq"])
    color, depth, normal, alpha, radii, _ = rast(means3D=T["means3D"], means2D=means2D, shs=T["shs"],
                                                 colors_precomp=T["colors_precomp"], opacities=T["opacities"],
                                                 scales=T["scales"], rotations=T["rotations"],
                                                 cov3Ds_precomp=T["cov3Ds_precomp"], extra_attrs=None, **kw)
    out = dict(color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(),
               normal=normal.detach().cpu().numpy(), alpha=alpha.detach().cpu().numpy(), radii=radii.cpu().numpy())
    keys, ids, ranges, masks = rast.export_binning(with_masks=True)
    out["keys"] = keys.cpu().numpy().view(np.uint64)
    out["ids"] = ids.cpu().numpy().view(np.uint32)
    out["ranges"] = ranges.cpu().numpy().view(np.uint32)
    out["quad_masks"] = masks.cpu().numpy().astype(np.uint8)
    out["keep_all_pairs"] = keep_all_pairs
    out["status"] = rast.last_status()
    geo = rast.export_geometry()
    out["geo"] = {k: (None if v is None else v.cpu().numpy()) for k, v in geo.items()}
    if backward:
        tc = lambda a: torch.from_numpy(a).to(dev)
        loss = (color * tc(case["dL_dcolor"])).sum() + (depth * tc(case["dL_ddepth"])).sum() + \
               (alpha * tc(case["dL_dalpha"])).sum()
        loss.backward()
        g = lambda t: None if t is None or t.grad is None else t.grad.detach().cpu().numpy()
        out["grads"] = dict(means3D=g(T["means3D"]), means2D=g(means2D), shs=g(T["shs"]), colors=g(T["colors_precomp"]),
                            opacities=g(T["opacities"]), scales=g(T["scales"]), rotations=g(T["rotations"]),
                            cov3D=g(T["cov3Ds_precomp"]), actor_pose=g(T["actor_pose"]), residual_dx=g(T["residual_dx"]),
                            residual_dq=g(T["residual_dq"]))
        if absgrad:
            out["grads"]["means2D_abs"] = rast.last_call.absgrad.cpu().numpy()
            assert means2D.absgrad is rast.last_call.absgrad          # gsplat convention: also published on the grad sink
    out["call"], out["flags"] = rast.last_call, case["flags"]
    if backward:
        out["render_grads"] = hip_render_grads(rast.last_call, case["W"], case["H"])
    return out


def locate_in_list(keys, ids, ref_keys, ref_ids):
    """Position of every (key, id) entry in the reference list (sorted by key, then id); -1 where it is absent."""
    pos = np.searchsorted(ref_keys, keys, side="left").astype(np.int64)
    D = ref_keys.shape[0]
    ok = np.zeros(keys.shape[0], bool)
    for _ in range(64):                                  # equal keys = equal tile AND equal depth bits: runs are short
        inb = pos < D
        hit = np.zeros_like(ok)
        hit[inb] = (ref_keys[pos[inb]] == keys[inb]) & (ref_ids[pos[inb]] == ids[inb])
        ok |= hit
        more = ~ok & inb
        more[more] = ref_keys[pos[more]] == keys[more]
        if not more.any():
            break
        pos[more] += 1
    pos[~ok] = -1
    return pos


def compare_binning(hip, orc):
    """The sort-key contract.  With keep_all_pairs the sorted list IS upstream's: keys (tile << 32 | depth bits), ids and tile ranges
    bit for bit.  By default the list is upstream's list WITHOUT the entries whose footprint reaches no pixel of their tile:
      * every entry the product keeps is an entry of the oracle's list, bit for bit, and the entries keep their order;
      * every entry it drops is one the oracle's brute-force check (orc_pair_quadrant_hits) finds no contributing pixel for;
      * the tile ranges are those of the kept list.
    Either way an entry's quadrant mask must contain every quadrant the brute-force check finds a contributing pixel in."""
    pre, b = orc["pre"], orc["bin"]
    true = co.pair_quadrant_hits(orc["S"], pre, b)
    if hip["keep_all_pairs"]:
        assert hip["status"]["num_rendered"] == b["D"]
        np.testing.assert_array_equal(hip["keys"], b["keys"], err_msg="sorted keys (tile<<32 | depth bits)")
        np.testing.assert_array_equal(hip["ids"], b["ids"], err_msg="sorted Gaussian ids")
        np.testing.assert_array_equal(hip["ranges"], b["ranges"], err_msg="tile ranges")
        pos = np.arange(b["D"])
    else:
        assert hip["status"]["num_rendered"] == hip["keys"].shape[0] <= b["D"]
        pos = locate_in_list(hip["keys"], hip["ids"], b["keys"], b["ids"])
        assert (pos >= 0).all(), f"{int((pos < 0).sum())} entries of the sorted list are not entries of the oracle's list"
        assert (np.diff(pos) > 0).all(), "the kept entries are not in the oracle's order"
        kept = np.zeros(b["D"], bool)
        kept[pos] = True
        wrong = int((true[~kept] != 0).sum())
        assert wrong == 0, f"{wrong} dropped (tile, Gaussian) pairs have a contributing pixel in their tile"
        # ranges of the kept list: [first, last + 1) per tile, (0, 0) for tiles without entries
        T = hip["ranges"].shape[0]
        tiles = (hip["keys"] >> np.uint64(32)).astype(np.int64)
        first = np.searchsorted(tiles, np.arange(T), side="left")
        last = np.searchsorted(tiles, np.arange(T), side="right")
        want = np.stack([first, last], 1).astype(np.uint32)
        want[first == last] = 0
        np.testing.assert_array_equal(hip["ranges"], want, err_msg="tile ranges of the kept list")
    missing = int((true[pos] & ~hip["quad_masks"]).astype(bool).sum())
    assert missing == 0, f"{missing} list entries lack the mask bit of a quadrant they contribute to"
    hip["cull_stats"] = dict(D=int(b["D"]), kept=int(hip["keys"].shape[0]), contributing=int((true != 0).sum()),
                             quadrant_bits=int(np.unpackbits(hip["quad_masks"] & 15).sum()), true_quadrant_bits=int(np.unpackbits(true).sum()))


def compare_forward(hip, orc, tol=IMAGE_TOL, exact_images=True):
    pre, b, img = orc["pre"], orc["bin"], orc["img"]
    # integer / key contract: bit-exact
    np.testing.assert_array_equal(hip["radii"], pre["radii"], err_msg="radii")
    np.testing.assert_array_equal(hip["geo"]["tiles_touched"].view(np.uint32), pre["tiles_touched"], err_msg="tiles_touched")
    assert hip["status"]["num_visible"] == int((pre["radii"] > 0).sum())
    compare_binning(hip, orc)
    vis = pre["radii"] > 0
    # pixel means and depths feed the keys / rects: bit-exact; conic, colour: tolerance
    np.testing.assert_array_equal(hip["geo"]["means2D"][vis].view(np.uint32), pre["means2D"][vis].view(np.uint32))
    np.testing.assert_array_equal(hip["geo"]["depths"][vis].view(np.uint32), pre["depths"][vis].view(np.uint32))
    np.testing.assert_allclose(hip["geo"]["conic_opacity"][vis], pre["conic_opacity"][vis], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(hip["geo"]["rgb"][vis], pre["rgb"][vis], rtol=0, atol=2e-6)
    if hip["geo"]["normal"] is not None:
        np.testing.assert_allclose(hip["geo"]["normal"][vis], pre["normal"][vis], rtol=0, atol=2e-6)
    # forward images: the compositing arithmetic (exp, FMA placement) is pinned on both sides, so the contract is
    # bit-exact -- far inside the north_star tolerance L_inf <= 1e-4 (`tol` is kept as the documented bound)
    for k in ("color", "depth", "alpha", "normal"):
        scale = max(1.0, float(np.abs(img[k]).max())) if k == "depth" else 1.0
        err = float(np.abs(hip[k] - img[k]).max())
        assert err <= tol * scale, f"{k}: L_inf {err:.3e} > {tol * scale:.1e}"
        if exact_images:
            nz = int((hip[k].view(np.uint32) != img[k].view(np.uint32)).sum())
            assert nz == 0, f"{k}: {nz} pixels differ from the oracle bit pattern (max abs {err:.3e})"


def grad_err(a, b, atol_frac=None):
    """(worst element-wise excess over the bound, as a multiple of the bound; relative L2 error)"""
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    ref = max(float(np.abs(b).max()), 1e-30)
    bound = GRAD_RTOL * np.abs(b) + (GRAD_ATOL_FRAC if atol_frac is None else atol_frac) * ref
    err = np.abs(a - b)
    worst = float((err / bound).max()) if err.size else 0.0
    l2 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    return worst, l2


def assert_grad_close(got, ref, name, rtol=None, atol_frac=None, rel_l2=None, extra_abs=None):
    """The gradient bar of this repository (see GRAD_RTOL / GRAD_ATOL_FRAC / GRAD_REL_L2 above).  `rtol` scales all three
    bounds together (rtol / GRAD_RTOL) for the few documented cases that need a looser bar.  `extra_abs` (array like ref): an
    element-wise absolute allowance on top of the bound (the conditioned end-to-end comparison, END2END_ULP_RESPONSES)."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    if extra_abs is not None:
        ex = np.asarray(extra_abs, np.float64).reshape(ref.shape)
        d = np.asarray(got, np.float64).reshape(ref.shape) - np.asarray(ref, np.float64)
        # how much of the allowance is actually used: elements over the PLAIN bound, and by how many one-ulp responses
        kk = 1.0 if rtol is None else rtol / GRAD_RTOL
        plain = kk * (GRAD_RTOL * np.abs(np.asarray(ref, np.float64)) + (GRAD_ATOL_FRAC if atol_frac is None else atol_frac) * max(float(np.abs(ref).max()), 1e-30))
        over = np.abs(d) - plain
        leaning = over > 0
        n_lean = int(leaning.sum())
        unit = ex / END2END_ULP_RESPONSES
        with np.errstate(divide="ignore", invalid="ignore"):
            mult = np.where(leaning, over / np.where(unit > 0, unit, np.nan), 0.0)
        max_mult = float(np.nanmax(mult)) if n_lean else 0.0
        ALLOWANCE_LOG.append(dict(name=name, elements=int(ref.size), leaning=n_lean, max_responses=round(max_mult, 3)))
        allowed = max(int(np.ceil(ALLOWANCE_MAX_FRACTION * ref.size)), ALLOWANCE_MIN_COUNT)
        assert n_lean <= allowed, (f"grad {name}: {n_lean} of {ref.size} elements exceed the plain bound and lean on the conditioning allowance "
                                   f"(at most {allowed} may)")
        # move every element towards the reference by its allowance, then apply the ordinary bar
        got = np.asarray(ref, np.float64) + np.sign(d) * np.maximum(np.abs(d) - ex, 0.0)
    if got.size == ref.size:
        got = got.reshape(ref.shape)
    assert got.shape == ref.shape, f"grad {name}: shape {got.shape} vs {ref.shape}"
    assert np.isfinite(got).all(), f"grad {name}: non-finite values"
    if float(np.abs(ref).max()) == 0.0:
        assert float(np.abs(got).max()) <= 1e-6, f"grad {name}: oracle is zero, hip is not"
        return 0.0, 0.0
    k = 1.0 if rtol is None else rtol / GRAD_RTOL
    worst, l2 = grad_err(got, ref, atol_frac)
    assert worst <= k, (f"grad {name}: an element exceeds {k:g} x ({GRAD_RTOL:g} |ref| + {GRAD_ATOL_FRAC if atol_frac is None else atol_frac:g} max|ref|) by a factor "
                        f"{worst / k:.2f} (rel L2 {l2:.2e})")
    l2_bar = k * (GRAD_REL_L2 if rel_l2 is None else rel_l2)
    assert l2 <= l2_bar, f"grad {name}: relative L2 error {l2:.2e} > {l2_bar:.1e}"
    return worst, l2


def pose_bound(go, scene):
    """Per actor and pose component: the sum over the actor's Gaussians of the magnitudes of the terms the pose gradient adds up
    (|dL/dworld mean| for the translation, |dL/dworld mean| |local mean| for the rotation of the means, |dL/dquat| for the
    composed rotation, |dL/dopacity| for the validity) -- the scale fp32 rounding of such a sum is proportional to."""
    A = scene.actor_pose.shape[0]
    ids = scene.actor_id
    gm = np.abs(np.asarray(go["means3D"], np.float64)).sum(1)            # (|R^T g| <= |g|: the local-frame gradient bounds the world one)
    ml = np.abs(np.asarray(scene.means3D, np.float64)).max(1) + 1.0
    gq = np.abs(np.asarray(go["rotations"], np.float64)).sum(1) if go.get("rotations") is not None else np.zeros_like(gm)
    gop = np.abs(np.asarray(go["opacities"], np.float64)).reshape(-1)
    b = np.zeros((A, 12))
    for a in range(A):
        m = ids == a
        b[a, 0:4] = (gm[m] * ml[m]).sum() * 4.0
        b[a, 4:7] = gm[m].sum()
        b[a, 7] = gop[m].sum()
        b[a, 8:12] = gq[m].sum() * 4.0
    return b


def assert_pose_close(got, ref, go, scene):
    got, ref = np.asarray(got, np.float64).reshape(-1, 12), np.asarray(ref, np.float64).reshape(-1, 12)
    bound = GRAD_RTOL * np.abs(ref) + POSE_TERM_RTOL * pose_bound(go, scene) + 1e-12
    worst = float((np.abs(got - ref) / bound).max())
    assert np.isfinite(got).all() and worst <= 1.0, f"grad actor_pose: exceeds {GRAD_RTOL:g} |ref| + {POSE_TERM_RTOL:g} sum|terms| by a factor {worst:.2f}"
    return worst, float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))


CONDITIONED = ("scales", "rotations", "cov3D", "residual_dq", "log_scales", "raw_quats")


def hip_render_grads(call, W, H):
    """The render backward's accumulator rows of a call made with keep_render_grads=True, in the oracle's dict layout."""
    r = call.render_grads.detach().cpu().numpy()
    return dict(mean2D=r[:, 0:2].copy(), depth=r[:, 2].copy(), opacity=r[:, 3].copy(), conic=r[:, 4:7].copy(), rgb=r[:, 7:10].copy(),
                abs=r[:, 10:12].copy(), normal=np.zeros((r.shape[0], 3), np.float32))


def compare_render_grads(hip_g, orc_g, names=("mean2D", "conic", "opacity", "rgb", "depth")):
    """(a) the render backward (K7) against the oracle's, strict bar."""
    for k in names:
        assert_grad_close(hip_g[k], orc_g[k], "render:" + k)


def moved_by_one_ulp(render_grads, seed=99):
    """Every fp32 render gradient moved by one ulp up or down (seeded): the input of the oracle chain's rounding response
    (END2END_ULP_RESPONSES)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k_, v in render_grads.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float32:
            sgn = rng.choice(np.array([-1.0, 1.0], np.float32), size=v.shape)
            out[k_] = (v * (np.float32(1.0) + np.float32(2.0 ** -23) * sgn)).astype(np.float32)
        else:
            out[k_] = v
    return out


def compare_backward(hip, orc, rtol=None, names=None):
    gh, go = hip["grads"], orc["grads"]
    checked = []
    names = names or ("means3D", "means2D", "shs", "colors", "opacities", "scales", "rotations", "cov3D", "actor_pose",
                      "residual_dx", "residual_dq", "means2D_abs")
    k8 = ulp = None
    if hip.get("render_grads") is not None and rtol is None:
        compare_render_grads(hip["render_grads"], go["render_grads"])                                   # (a)
        k8 = co.preprocess_backward(orc["S"], orc["scene"], orc["pre"], hip["render_grads"], hip["flags"])   # (b) reference for K8 alone
        k8m = co.preprocess_backward(orc["S"], orc["scene"], orc["pre"], moved_by_one_ulp(hip["render_grads"]), hip["flags"])
        ulp = {k_: np.abs(np.asarray(k8m[k_], np.float64) - np.asarray(k8[k_], np.float64)) for k_ in k8
               if isinstance(k8.get(k_), np.ndarray) and isinstance(k8m.get(k_), np.ndarray) and k8[k_].shape == k8m[k_].shape}
    for k in names:
        if gh.get(k) is None:
            continue
        if k == "actor_pose":
            assert_pose_close(gh[k], go[k], go, orc["scene"])
        elif k in CONDITIONED and rtol is None:
            if k8 is not None:
                assert_grad_close(gh[k], k8[k], "projection backward on the kernel's own render gradients: " + k)
            extra = END2END_ULP_RESPONSES * ulp[k].reshape(np.asarray(go[k]).shape) if (ulp is not None and k in ulp) else None
            assert_grad_close(gh[k], go[k], k, atol_frac=END2END_ATOL_FRAC, rel_l2=END2END_REL_L2, extra_abs=extra)         # (c)
        else:
            # (means3D runs through the same det^2 division as scales / rotations for a nearly degenerate footprint: the strict bar plus
            #  the oracle chain's rounding response, see END2END_ULP_RESPONSES; zero for everything the projection backward does not amplify)
            extra = END2END_ULP_RESPONSES * ulp[k].reshape(np.asarray(go[k]).shape) if (ulp is not None and k in ulp and rtol is None) else None
            assert_grad_close(gh[k], go[k], k, rtol, extra_abs=extra)
        checked.append(k)
    return checked


def raw_params_parity(case, log_s, raw_q, logit, device="cuda:0", check_images_exact=True, rtol=None):
    """EMD_FLAG_RAW_PARAMS path (the one bench.py times): exp / normalize / sigmoid (gaussian_renderer/__init__.py:99-101) fused
    into K1 / K8, optionally with the fused explicit-motion transform.  `case` as make_case() builds it (its activated
    scales / rotations / opacities are ignored); `log_s [N,3]`, `raw_q [N,4]`, `logit [N,1]` are the raw parameters.
    The oracle is fed the activations exactly as the library computes them (emd_activations_forward), so the integer
    contract and the images stay bit-exact; the oracle's gradients are chained through the activations in float64 numpy.
    Returns a dict of the per-tensor (worst element excess, rel L2) pairs."""
    import ctypes as C
    from emd_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib as L
    dev = torch.device(device)
    N = case["N"]
    motion = case.get("actor_ids") is not None
    log_s = log_s.to(dev).requires_grad_(True)
    raw_q = raw_q.to(dev).requires_grad_(True)
    logit = logit.to(dev).requires_grad_(True)
    s_act, q_act, o_act = torch.empty(N, 3, device=dev), torch.empty(N, 4, device=dev), torch.empty(N, device=dev)
    L.check(L.load().emd_activations_forward(N, log_s.data_ptr(), s_act.data_ptr(), raw_q.data_ptr(), q_act.data_ptr(),
                                             logit.data_ptr(), o_act.data_ptr(), None), "emd_activations_forward")
    torch.cuda.synchronize()
    torch.testing.assert_close(s_act, torch.exp(log_s.detach()), rtol=2e-6, atol=0)
    torch.testing.assert_close(o_act, torch.sigmoid(logit.detach()).reshape(-1), rtol=2e-6, atol=1e-7)
    ocase = dict(case)
    ocase["scales"], ocase["opacities"] = s_act.cpu(), o_act.cpu()[:, None]
    rots = q_act.cpu().clone()
    dyn = None
    if motion:   # actor points are normalised inside the motion transform from the raw quaternion
        dyn = (case["actor_ids"] >= 0)
        rots[dyn] = raw_q.detach().cpu()[dyn]
    ocase["rotations"] = rots
    orc = run_oracle(ocase, backward=True)
    cam = case["cam"]
    rs = GaussianRasterizationSettings(case["H"], case["W"], cam.tanfovx, cam.tanfovy, case["bg"], 1.0, cam.world_view_transform,
                                       cam.full_proj_transform, case["sh_degree"], cam.camera_center, False, False)
    means = case["means3D"].to(dev).requires_grad_(True)
    shs = case["shs"].to(dev).requires_grad_(True)
    m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
    kw = {}
    pose = None
    if motion:
        pose = case["actor_pose"].to(dev).requires_grad_(True)
        kw = dict(actor_ids=case["actor_ids"].to(dev), actor_pose=pose)
    rdx = rdq = None
    if case.get("residual_dx") is not None:          # the learned deformation residual (deformable.py:35-47) as an input of K1
        rdx = kw["residual_dx"] = case["residual_dx"].to(dev).requires_grad_(True)
    if case.get("residual_dq") is not None:
        rdq = kw["residual_dq"] = case["residual_dq"].to(dev).requires_grad_(True)
    rast = GaussianRasterizer(rs, compute_normal=True, keep_render_grads=True)
    color, depth, normal, alpha, radii, _ = rast(means3D=means, means2D=m2, shs=shs, opacities=logit, scales=log_s,
                                                 rotations=raw_q, raw_params=True, **kw)
    keys, ids, ranges, masks = rast.export_binning(with_masks=True)
    st = rast.last_status()
    assert st["num_visible"] == int((orc["pre"]["radii"] > 0).sum())
    np.testing.assert_array_equal(radii.cpu().numpy(), orc["pre"]["radii"], err_msg="radii")
    hb = dict(keys=keys.cpu().numpy().view(np.uint64), ids=ids.cpu().numpy().view(np.uint32), ranges=ranges.cpu().numpy().view(np.uint32),
              quad_masks=masks.cpu().numpy().astype(np.uint8), keep_all_pairs=False, status=st)
    compare_binning(hb, orc)                      # the default list: upstream's without the entries that reach no pixel of their tile
    cull = hb["cull_stats"]
    # ... and upstream's list itself, entry for entry, with keep_all_pairs (forward only; same images bit for bit)
    with torch.no_grad():
        rast_all = GaussianRasterizer(rs, compute_normal=True, keep_all_pairs=True)
        c_all, d_all, n_all, a_all, r_all, _ = rast_all(means3D=means.detach(), means2D=m2.detach(), shs=shs.detach(), opacities=logit.detach(),
                                                        scales=log_s.detach(), rotations=raw_q.detach(), raw_params=True,
                                                        **{k_: (v.detach() if v.is_floating_point() else v) for k_, v in kw.items()})
        k_all, i_all, rg_all, m_all = rast_all.export_binning(with_masks=True)
        ha = dict(keys=k_all.cpu().numpy().view(np.uint64), ids=i_all.cpu().numpy().view(np.uint32), ranges=rg_all.cpu().numpy().view(np.uint32),
                  quad_masks=m_all.cpu().numpy().astype(np.uint8), keep_all_pairs=True, status=rast_all.last_status())
        compare_binning(ha, orc)
        assert torch.equal(c_all, color.detach()) and torch.equal(d_all, depth.detach()) and torch.equal(a_all, alpha.detach()) \
            and torch.equal(n_all, normal.detach()) and torch.equal(r_all, radii), "keep_all_pairs changes the images"
        del c_all, d_all, n_all, a_all, k_all, i_all, rg_all, m_all, rast_all
    for name, t in (("color", color), ("depth", depth), ("alpha", alpha), ("normal", normal)):
        got = t.detach().cpu().numpy()
        scale = max(1.0, float(np.abs(orc["img"][name]).max())) if name == "depth" else 1.0
        assert np.abs(got - orc["img"][name]).max() <= IMAGE_TOL * scale, name
        if check_images_exact:
            nz = int((got.view(np.uint32) != orc["img"][name].view(np.uint32)).sum())
            assert nz == 0, f"{name}: {nz} pixels differ from the oracle bit pattern"
    tc = lambda a: torch.from_numpy(a).to(dev)
    loss = (color * tc(case["dL_dcolor"])).sum()
    if case.get("dL_ddepth") is not None:
        loss = loss + (depth * tc(case["dL_ddepth"])).sum()
    if case.get("dL_dalpha") is not None:
        loss = loss + (alpha * tc(case["dL_dalpha"])).sum()
    loss.backward()
    s_np, o_np = s_act.cpu().numpy().astype(np.float64), o_act.cpu().numpy().astype(np.float64)
    rq = raw_q.detach().cpu().numpy().astype(np.float64)
    nrm = np.linalg.norm(rq, axis=1, keepdims=True)
    qu = rq / nrm

    def through_activations(g):
        """gradients w.r.t. the raw parameters from the oracle's gradients w.r.t. the activated ones (float64 numpy)"""
        e = dict(log_scales=g["scales"] * s_np, opacity_logits=g["opacities"] * o_np * (1 - o_np))
        gq = g["rotations"].astype(np.float64)
        e["raw_quats"] = (gq - qu * (qu * gq).sum(1, keepdims=True)) / nrm
        if motion:
            e["raw_quats"][dyn.numpy()] = g["rotations"][dyn.numpy()]      # the oracle already differentiates the in-transform normalisation
        for k_ in ("residual_dx", "residual_dq"):
            if g.get(k_) is not None:
                e[k_] = g[k_]
        return e
    go = orc["grads"]
    # (a) the render backward alone, (b) the projection backward on the kernel's own render gradients, (c) end to end
    hg = hip_render_grads(rast.last_call, case["W"], case["H"])
    if rtol is None:
        compare_render_grads(hg, go["render_grads"])
    k8 = through_activations(co.preprocess_backward(orc["S"], orc["scene"], orc["pre"], hg, case["flags"]))
    k8_raw = co.preprocess_backward(orc["S"], orc["scene"], orc["pre"], hg, case["flags"])
    k8m_raw = co.preprocess_backward(orc["S"], orc["scene"], orc["pre"], moved_by_one_ulp(hg), case["flags"])
    k8m = through_activations(k8m_raw)
    ulp = {k_: END2END_ULP_RESPONSES * np.abs(np.asarray(k8m[k_], np.float64) - np.asarray(k8[k_], np.float64)) for k_ in k8 if k_ in k8m}
    ulp["means3D"] = END2END_ULP_RESPONSES * np.abs(np.asarray(k8m_raw["means3D"], np.float64) - np.asarray(k8_raw["means3D"], np.float64))
    exp = through_activations(go)
    res = {}
    got = dict(log_scales=log_s.grad.cpu().numpy(), opacity_logits=logit.grad.cpu().numpy().reshape(-1), raw_quats=raw_q.grad.cpu().numpy())
    for name in ("log_scales", "raw_quats"):
        if rtol is None:
            assert_grad_close(got[name], k8[name], "projection backward on the kernel's own render gradients: " + name)
        res[name] = assert_grad_close(got[name], exp[name], name, rtol, atol_frac=END2END_ATOL_FRAC, rel_l2=END2END_REL_L2,
                                      extra_abs=ulp[name].reshape(np.asarray(exp[name]).shape))
    res["opacity_logits"] = assert_grad_close(got["opacity_logits"], exp["opacity_logits"], "opacity_logits", rtol)
    res["means3D"] = assert_grad_close(means.grad.cpu().numpy(), go["means3D"], "means3D", rtol, extra_abs=ulp["means3D"] if rtol is None else None)
    res["means2D"] = assert_grad_close(m2.grad.cpu().numpy(), go["means2D"], "means2D", rtol)
    res["shs"] = assert_grad_close(shs.grad.cpu().numpy(), go["shs"], "shs", rtol)
    if motion:
        res["actor_pose"] = assert_pose_close(pose.grad.cpu().numpy(), go["actor_pose"], go, orc["scene"])
    if rdx is not None:
        res["residual_dx"] = assert_grad_close(rdx.grad.cpu().numpy(), go["residual_dx"], "residual_dx", rtol)
    if rdq is not None:
        if rtol is None:
            assert_grad_close(rdq.grad.cpu().numpy(), k8["residual_dq"], "projection backward on the kernel's own render gradients: residual_dq")
        res["residual_dq"] = assert_grad_close(rdq.grad.cpu().numpy(), go["residual_dq"], "residual_dq", rtol, atol_frac=END2END_ATOL_FRAC,
                                               rel_l2=END2END_REL_L2, extra_abs=ulp["residual_dq"].reshape(np.asarray(go["residual_dq"]).shape))
    res["D"], res["V"], res["D_sorted"], res["D_contributing"] = orc["bin"]["D"], st["num_visible"], cull["kept"], cull["contributing"]
    res["allowance"] = [a for a in ALLOWANCE_LOG[-8:] if a["leaning"]]
    return res
