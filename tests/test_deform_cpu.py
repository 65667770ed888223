"""CPU: the deformation oracle against the reference's own deform_network / ConditionalDeformNetwork outputs and gradients
(tests/golden/s3g_deform.npz, tests/golden/or_deform.npz -- the reference's modules imported and run on CPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import deform_oracle as do

G = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ("point", "scales", "rotations", "opacity", "shs")


def load_s3g(tag, req_grad=True):
    g = np.load(os.path.join(G, "s3g_deform.npz"))
    pre = f"{tag}_sd_"
    sd = {k[len(pre):]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith(pre)}
    for k, v in sd.items():
        if v.dtype == torch.float32 and not k.endswith("_poc") and "aabb" not in k:
            v.requires_grad_(req_grad)
    opts = {k: int(g[f"{tag}_opt_{k}"]) for k in ("no_ds", "no_dr", "no_fine_hexplane_features", "feat_head", "min_embeddings",
                                                 "max_embeddings", "temporal_embedding_dim", "c2f_temporal_iter")}
    opts["multires"] = g[f"{tag}_opt_multires"].tolist()
    ins = {n: torch.from_numpy(g[f"{tag}_in_{n}"]).clone().requires_grad_(req_grad) for n in NAMES + ("emb",)}
    return g, sd, opts, ins


@pytest.mark.parametrize("tag", ["run", "full"])
def test_s3g_deform_oracle_matches_reference(tag):
    g, sd, opts, ins = load_s3g(tag)
    res = do.s3g_deform(sd, opts, ins["point"], ins["scales"], ins["rotations"], ins["opacity"], ins["shs"],
                        torch.from_numpy(g[f"{tag}_in_times"]), ins["emb"], int(g[f"{tag}_iter"]), int(g[f"{tag}_cam_no"]))
    for n, r in zip(NAMES, res[:5]):
        np.testing.assert_allclose(r.detach().numpy(), g[f"{tag}_out_{n}"], rtol=2e-5, atol=2e-5, err_msg=n)
    dd = res[5]
    for lvl in ("coarse", "fine"):
        for k, v in dd[lvl].items():
            key = f"{tag}_ddict_{lvl}_{k}"
            assert (v is None) == (key not in g.files), key
            if v is not None:
                np.testing.assert_allclose(v.detach().numpy(), g[key], rtol=2e-5, atol=2e-5, err_msg=key)
    loss = sum((r * torch.from_numpy(g[f"{tag}_gout_{n}"])).sum() for n, r in zip(NAMES, res[:5]))
    for lvl in ("coarse", "fine"):
        loss = loss + (dd[lvl]["feat"] * torch.from_numpy(g[f"{tag}_gfeat_{lvl}"])).sum()
    loss.backward()
    for n in NAMES + ("emb",):
        np.testing.assert_allclose(ins[n].grad.numpy(), g[f"{tag}_g_{n}"], rtol=1e-4, atol=1e-4, err_msg=f"grad {n}")
    for k, v in sd.items():
        gk = f"{tag}_gsd_{k}"
        if gk in g.files and v.requires_grad:
            got = v.grad.numpy() if v.grad is not None else np.zeros_like(g[gk])
            scale = max(1.0, float(np.abs(g[gk]).max()))
            np.testing.assert_allclose(got, g[gk], rtol=1e-4, atol=1e-4 * scale, err_msg=f"grad {k}")
    # the fixture reaches the time gradient (time_offset) through both the planes and the temporal table
    assert np.abs(g[f"{tag}_gsd_deformation_net.time_offset"]).max() > 0


def test_temporal_embed_oracle_edges():
    """k == rows is the identity resize; t = 0 and t = 1 hit the first / last row; t outside [0, 1] reflects."""
    w = torch.randn(12, 5, generator=torch.Generator().manual_seed(3))
    np.testing.assert_allclose(do.temporal_embed(w, 12, torch.tensor(0.0)).numpy(), w[0].numpy(), atol=1e-6)
    np.testing.assert_allclose(do.temporal_embed(w, 12, torch.tensor(1.0)).numpy(), w[11].numpy(), atol=1e-6)
    np.testing.assert_allclose(do.temporal_embed(w, 7, torch.tensor(-0.2)).numpy(), do.temporal_embed(w, 7, torch.tensor(0.2)).numpy(), atol=1e-6)
    np.testing.assert_allclose(do.temporal_embed(w, 7, torch.tensor(1.3)).numpy(), do.temporal_embed(w, 7, torch.tensor(0.7)).numpy(), atol=1e-5)
    import torch.nn.functional as F
    for k, t in ((4, 0.41), (9, 0.77), (12, 0.5), (30, 0.123)):
        emb = F.interpolate(w[None, None], size=(k, 5), mode="bilinear", align_corners=True)
        grid = torch.cat([torch.arange(5).unsqueeze(-1) / 4, torch.ones(5, 1) * t], dim=-1)[None, None]
        want = F.grid_sample(emb, (grid - 0.5) * 2, align_corners=True, mode="bilinear", padding_mode="reflection").reshape(-1)
        np.testing.assert_allclose(do.temporal_embed(w, k, torch.tensor(t)).numpy(), want.numpy(), atol=1e-6)


def test_or_deform_oracle_matches_reference():
    g = np.load(os.path.join(G, "or_deform.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("sd_")}
    emb = torch.from_numpy(g["inst_embed"]).clone().requires_grad_(True)
    h0 = do.deform_input(torch.from_numpy(g["means"]), torch.from_numpy(g["point_ids"]).long(), torch.from_numpy(g["inst_size"]), emb,
                         torch.from_numpy(g["t"]), int(g["x_multires"]), int(g["t_multires"]))
    np.testing.assert_allclose(h0.detach().numpy(), g["h0"], rtol=1e-6, atol=1e-6)
    dxyz, dquat, dscale = do.conditional_deform(sd, h0, D=int(g["D"]), skips=(int(g["D"]) // 2,))
    assert dscale is None
    np.testing.assert_allclose(dxyz.detach().numpy(), g["dxyz"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dquat.detach().numpy(), g["dquat"], rtol=2e-5, atol=2e-5)
    ((dxyz * torch.from_numpy(g["gx"])).sum() + (dquat * torch.from_numpy(g["gq"])).sum()).backward()
    np.testing.assert_allclose(emb.grad.numpy(), g["g_inst_embed"], rtol=1e-4, atol=1e-4)
    for k, v in sd.items():
        np.testing.assert_allclose(v.grad.numpy(), g[f"gsd_{k}"], rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(g[f'gsd_{k}']).max())), err_msg=k)
