"""-m "not gpu": the CPU oracle checked against an independent fp64 autograd restatement and its own invariants."""
import numpy as np
import pytest
import torch

from oracle import cpu_oracle as co
from oracle import torch_ref as tr
from tests.helpers import make_case, oracle_scene, oracle_settings, run_oracle


def _torch_settings(case):
    cam = case["cam"]
    return tr.TorchSettings(case["H"], case["W"], cam.tanfovx, cam.tanfovy, case["bg"].numpy(),
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), case["sh_degree"],
                            cam.camera_center.numpy())


@pytest.mark.parametrize("kw", [dict(n=600, H=32, W=48, seed=0), dict(n=500, H=32, W=48, seed=1, motion=True, residual=True),
                                dict(n=500, H=32, W=48, seed=2, cov_precomp=True, colors_precomp=True)],
                         ids=["static-sh", "motion-residual", "cov-colors"])
def test_oracle_backward_matches_fp64_autograd(kw):
    case = make_case(**kw)
    orc = run_oracle(case, backward=True)
    out, leaves = tr.render(_torch_settings(case), orc["scene"], orc["pre"], orc["bin"], flags=case["flags"])
    for k in ("color", "depth", "alpha", "normal"):
        assert np.abs(out[k].detach().numpy() - orc["img"][k]).max() < 5e-5, k
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    loss = (out["color"] * t(case["dL_dcolor"])).sum() + (out["depth"] * t(case["dL_ddepth"])).sum() + \
           (out["alpha"] * t(case["dL_dalpha"])).sum()
    loss.backward()
    g = orc["grads"]
    pairs = dict(means3D=leaves["means3D"], shs=leaves["shs"], colors=leaves["colors"], opacities=leaves["opacities"],
                 scales=leaves["scales"], rotations=leaves["rotations"], cov3D=leaves["cov3D"],
                 actor_pose=leaves["actor_pose"], residual_dx=leaves["residual_dx"], residual_dq=leaves["residual_dq"])
    n_checked = 0
    for name, leaf in pairs.items():
        if leaf is None or leaf.grad is None:
            continue
        ref = leaf.grad.numpy()
        got = np.asarray(g[name], np.float64).reshape(ref.shape)
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(got - ref).max() / scale < 2e-4, name
        n_checked += 1
    assert n_checked >= 3
    m2 = out["means2D_pix"].grad.numpy()
    ref2 = np.stack([m2[:, 0] * 0.5 * case["W"], m2[:, 1] * 0.5 * case["H"]], 1)
    assert np.abs(g["means2D"][:, :2] - ref2).max() / max(np.abs(ref2).max(), 1e-12) < 2e-4


def test_oracle_binning_invariants():
    case = make_case(n=3000, H=70, W=90, seed=3)
    orc = run_oracle(case)
    pre, b = orc["pre"], orc["bin"]
    k = b["keys"]
    assert np.all(k[1:] >= k[:-1])
    gx = (case["W"] + 15) // 16
    # every (tile, id) pair lies inside that Gaussian's rectangle and pairs are unique
    tiles = (k >> np.uint64(32)).astype(np.int64)
    ids = b["ids"].astype(np.int64)
    r = pre["rect"][ids]
    tx, ty = tiles % gx, tiles // gx
    assert np.all((tx >= r[:, 0]) & (tx < r[:, 2]) & (ty >= r[:, 1]) & (ty < r[:, 3]))
    assert len(set(zip(tiles.tolist(), ids.tolist()))) == len(ids)
    # depth bits of the key are the float bits of the view depth
    np.testing.assert_array_equal((k & np.uint64(0xFFFFFFFF)).astype(np.uint32), pre["depths"][ids].view(np.uint32))
    # culled Gaussians: behind the near plane
    assert np.all(pre["radii"][: case["N"] // 50] == 0)


def test_oracle_alpha_and_background():
    case = make_case(n=800, H=32, W=48, seed=4, bg=(1.0, 0.5, 0.25))
    img = run_oracle(case)["img"]
    np.testing.assert_allclose(img["alpha"][0], 1 - img["final_T"], atol=0)
    assert img["alpha"].min() >= 0 and img["alpha"].max() <= 1
    case0 = dict(case)
    case0["bg"] = torch.zeros(3)
    img0 = run_oracle(case0)["img"]
    np.testing.assert_allclose(img["color"] - img0["color"], img["final_T"][None] * case["bg"].numpy()[:, None, None], atol=1e-6)



def test_oracle_pair_quadrant_hits_is_the_render_loops_own_skip_test():
    """orc_pair_quadrant_hits (the checker of the product's footprint culling) against the render forward: rendering only the entries
    it marks as contributing gives the same image bit for bit, the entry that contributed last to a pixel always carries the bit of
    that pixel's quadrant, and on a street-like scene a sizeable part of upstream's (tile, Gaussian) pairs contributes nowhere."""
    case = make_case(n=3000, H=70, W=90, seed=6)
    orc = run_oracle(case)
    pre, b, img = orc["pre"], orc["bin"], orc["img"]
    m = co.pair_quadrant_hits(orc["S"], pre, b)
    assert m.shape[0] == b["D"] and m.max() <= 15
    keep = m != 0
    assert 0.05 < 1.0 - keep.mean() < 0.95
    tiles = (b["keys"][keep] >> np.uint64(32)).astype(np.int64)
    T = b["ranges"].shape[0]
    first, last = np.searchsorted(tiles, np.arange(T), "left"), np.searchsorted(tiles, np.arange(T), "right")
    rg = np.stack([first, last], 1).astype(np.uint32)
    rg[first == last] = 0
    img2 = co.render_forward(orc["S"], pre, dict(D=int(keep.sum()), keys=b["keys"][keep], ids=b["ids"][keep], ranges=rg), case["flags"])
    for k in ("color", "depth", "alpha", "normal", "final_T"):
        np.testing.assert_array_equal(img2[k].view(np.uint32), img[k].view(np.uint32), err_msg=k)
    gx = (case["W"] + 15) // 16
    for py in range(0, case["H"], 5):
        for px in range(0, case["W"], 7):
            n = int(img["n_contrib"][py, px])
            if n:
                t = (py // 16) * gx + px // 16
                e = int(b["ranges"][t, 0]) + n - 1
                assert m[e] & (1 << (((py % 16) >> 3) * 2 + ((px % 16) >> 3)))
