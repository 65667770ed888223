"""-m gpu: emd_amd.optim.Adam (one HIP launch per 32 tensors) against the reference's optimiser run on CPU
(tests/golden/s3g_adam.npz), against torch.optim.Adam on the same device, and through the reference's densification surgery on
the optimiser state.  fp32: 2e-6 relative, plus 1e-6 of the tensor's scale for cancellation."""
import os

import numpy as np
import pytest
import torch

from oracle import adam_oracle as ao
from tests.test_adam_cpu import schedule

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _close(a, b, what, rel=2e-6):
    a, b = a.detach().cpu().numpy().reshape(-1), np.asarray(b).reshape(-1)
    np.testing.assert_allclose(a, b, rtol=rel, atol=1e-6 * max(np.abs(b).max(), 1e-30), err_msg=what)


def test_adam_matches_reference_optimizer_golden():
    from emd_amd.optim import Adam, expon_lr
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(G, "s3g_adam.npz"))
    names = [str(n) for n in g["group_names"]]
    # groups as gaussian_model.py:188-199 builds them; multi-tensor groups (deformation, grid) are split like the fixture's count
    groups, params = [], {}
    for n in names:
        flat = torch.from_numpy(g[f"init_{n}"]).to(dev)
        parts = [torch.nn.Parameter(c.clone()) for c in flat.chunk(int(g[f"count_{n}"]))] if int(g[f"count_{n}"]) > 1 else [torch.nn.Parameter(flat.clone())]
        params[n] = parts
        groups.append({"params": parts, "lr": 0.0, "name": n})
    opt = Adam(groups, lr=0.0, eps=1e-15)
    a = lambda k: float(g["arg_" + k])
    s = float(g["spatial_lr_scale"])
    sched = {"xyz": expon_lr(a("position_lr_init") * s, a("position_lr_final") * s, lr_delay_mult=a("position_lr_delay_mult"), max_steps=a("position_lr_max_steps")),
             "deformation": expon_lr(a("deformation_lr_init") * s, a("deformation_lr_final") * s, lr_delay_mult=a("deformation_lr_delay_mult"), max_steps=a("position_lr_max_steps")),
             "grid": expon_lr(a("grid_lr_init") * s, a("grid_lr_final") * s, lr_delay_mult=a("deformation_lr_delay_mult"), max_steps=a("position_lr_max_steps")),
             "sky_cube_map": expon_lr(a("sky_cube_map_lr_init"), a("sky_cube_map_lr_final"), max_steps=a("sky_cube_map_max_steps"))}
    for k, it in enumerate(g["iters"]):
        for j, gr in enumerate(opt.param_groups):
            n = gr["name"]
            gr["lr"] = float(sched[n](int(it))) if n in sched else schedule(g, n, int(it))
            np.testing.assert_allclose(gr["lr"], g[f"lr_{k}"][j], rtol=2e-7)                  # expon_lr == the reference's schedule
            flat = torch.from_numpy(g[f"grad_{k}_{n}"]).to(dev)
            off = 0
            for p in gr["params"]:
                p.grad = flat[off:off + p.numel()].view_as(p).clone()
                off += p.numel()
        opt.step()
    for n in names:
        _close(torch.cat([p.data.reshape(-1) for p in params[n]]), g[f"final_{n}"], f"final {n}")
        _close(torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in params[n]]), g[f"exp_avg_{n}"], f"exp_avg {n}")
        _close(torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) for p in params[n]]), g[f"exp_avg_sq_{n}"], f"exp_avg_sq {n}")


def test_adam_vs_torch_adam_many_tensors_and_layouts():
    """70 tensors (three launches), odd sizes, an unaligned view, a channel-last parameter, a parameter without gradient."""
    from emd_amd.optim import Adam
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(0)
    shapes = [(int(torch.randint(1, 5000, (1,), generator=gen)), int(torch.randint(1, 7, (1,), generator=gen))) for _ in range(66)]
    base = [torch.randn(*s, generator=gen) for s in shapes]
    base.append(torch.randn(300001, generator=gen))
    base.append(torch.randn(1, 8, 6, 5, generator=gen).contiguous(memory_format=torch.channels_last))
    odd = torch.randn(1001, generator=gen)
    mine = [torch.nn.Parameter(b.clone().to(dev)) for b in base] + [torch.nn.Parameter(odd.clone().to(dev)[1:])] + [torch.nn.Parameter(torch.ones(5, device=dev))]
    ref = [torch.nn.Parameter(b.clone().to(dev)) for b in base] + [torch.nn.Parameter(odd.clone().to(dev)[1:])] + [torch.nn.Parameter(torch.ones(5, device=dev))]
    mk = lambda cls, ps: cls([{"params": ps[:30], "lr": 1e-2}, {"params": ps[30:], "lr": 3e-4, "betas": (0.8, 0.99)}], lr=0.0, eps=1e-15)
    o1, o2 = mk(Adam, mine), mk(torch.optim.Adam, ref)
    for step in range(4):
        for a, b in zip(mine[:-1], ref[:-1]):
            gr = torch.randn(a.shape, generator=gen).to(dev) * 10.0 ** (step - 2)
            a.grad, b.grad = gr.clone(), gr.clone()
        o1.step(); o2.step()
    for i, (a, b) in enumerate(zip(mine, ref)):
        _close(a, b.detach().cpu().numpy(), f"param {i}")
        if i < len(mine) - 1:
            _close(o1.state[a]["exp_avg_sq"], o2.state[b]["exp_avg_sq"].cpu().numpy(), f"v {i}")
    assert len(o1.state[mine[-1]]) == 0 and torch.equal(mine[-1].data, torch.ones(5, device=dev))        # no gradient: untouched
    assert mine[-3].is_contiguous(memory_format=torch.channels_last)


def test_adam_state_survives_reference_densification_surgery():
    """cat_tensors_to_optimizer / _prune_optimizer of the reference (gaussian_model.py:470-556) edit exp_avg / exp_avg_sq and swap
    the parameter object; the optimiser must carry on from the edited state, and its state_dict must round-trip."""
    from emd_amd.optim import Adam
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(1)
    p = torch.nn.Parameter(torch.randn(100, 3, generator=gen).to(dev))
    opt = Adam([{"params": [p], "lr": 1e-2, "name": "xyz"}], lr=0.0, eps=1e-15)
    p.grad = torch.randn(100, 3, generator=gen).to(dev)
    opt.step()
    group = opt.param_groups[0]
    st = opt.state.get(group["params"][0])
    ext = torch.randn(20, 3, generator=gen).to(dev)
    mask = torch.rand(120, generator=gen).to(dev) > 0.3
    st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)[mask]
    st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)[mask]
    del opt.state[group["params"][0]]
    group["params"][0] = torch.nn.Parameter(torch.cat((p.data, ext), dim=0)[mask].requires_grad_(True))
    opt.state[group["params"][0]] = st
    q = group["params"][0]
    q.grad = torch.randn(q.shape, generator=gen).to(dev)
    want = ao.adam_step(q.detach().cpu().numpy(), q.grad.cpu().numpy(), st["exp_avg"].cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), 2, 1e-2)
    opt.step()
    _close(q, want[0], "param after surgery")
    _close(opt.state[q]["exp_avg"], want[1], "exp_avg after surgery")
    sd = opt.state_dict()
    opt2 = Adam([{"params": [torch.nn.Parameter(q.detach().clone())], "lr": 1e-2, "name": "xyz"}], lr=0.0, eps=1e-15)
    opt2.load_state_dict(sd)
    assert float(opt2.state[opt2.param_groups[0]["params"][0]]["step"]) == 2.0


def test_adam_refuses_cpu_and_unsupported_options():
    from emd_amd import _lib as L
    from emd_amd.optim import Adam
    with pytest.raises(NotImplementedError):
        Adam([torch.nn.Parameter(torch.zeros(3))], weight_decay=0.1)
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    with pytest.raises(L.EmdError):
        Adam([p]).step()


def test_adam_full_size_against_torch():
    """The largest parameter of the headline configuration (2 M x 16 x 3 SH coefficients, 96 M elements) over three steps."""
    from emd_amd.optim import Adam
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    p0 = torch.randn(2_000_000, 16, 3, device=dev, generator=gen)
    a, b = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    o1, o2 = Adam([{"params": [a], "lr": 2.5e-3}], lr=0.0, eps=1e-15), torch.optim.Adam([{"params": [b], "lr": 2.5e-3}], lr=0.0, eps=1e-15)
    for _ in range(3):
        gr = torch.randn(p0.shape, device=dev, generator=gen)
        a.grad, b.grad = gr, gr.clone()
        o1.step(); o2.step()
    d = (a.detach() - b.detach()).abs().max()
    assert float(d) <= 2e-6 * float(b.detach().abs().max()), float(d)
    assert float((o1.state[a]["exp_avg_sq"] - o2.state[b]["exp_avg_sq"]).abs().max()) <= 2e-6 * float(o2.state[b]["exp_avg_sq"].max())


def test_capturable_adam_matches_the_host_stepped_one_and_replays_from_a_graph():
    """capturable=True (step counts and learning rates on the device, bias corrections formed in the kernel, EmdAdamTensor.step_dev / lr_dev):
    (a) stepped eagerly it follows torch.optim.Adam like the host-stepped optimiser does, through a learning-rate change; (b) ONE captured
    step() replayed n times equals n eager steps of torch.optim.Adam with the same gradients -- the step count advances inside the graph and
    a rate uploaded between replays (push_lrs) takes effect."""
    from emd_amd.optim import Adam
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(4)
    shapes = [(4000, 3), (4000, 16, 3), (77,), (4000, 1)]
    base = [torch.randn(*s, generator=gen) for s in shapes]
    mk_params = lambda: [torch.nn.Parameter(b.clone().to(dev)) for b in base]
    mk = lambda cls, ps, **kw: cls([{"params": ps[:2], "lr": 1e-2}, {"params": ps[2:], "lr": 3e-3, "betas": (0.8, 0.99)}], lr=0.0, eps=1e-15, **kw)
    grads = [[(torch.randn(s, generator=gen) * 10.0 ** (k % 3 - 1)).to(dev) for s in shapes] for k in range(6)]
    # (a) eager
    mine, ref = mk_params(), mk_params()
    o1, o2 = mk(Adam, mine, capturable=True), mk(torch.optim.Adam, ref)
    for k in range(6):
        if k == 3:
            for o in (o1, o2):
                o.param_groups[0]["lr"] = 2e-3
        for a, b, g in zip(mine, ref, grads[k]):
            a.grad, b.grad = g.clone(), g.clone()
        o1.step(); o2.step()
    for i, (a, b) in enumerate(zip(mine, ref)):
        _close(a, b.detach().cpu().numpy(), f"eager param {i}")
    assert float(o1.state[mine[0]]["step"]) == 6.0 and o1.state[mine[0]]["step"].device.type == "cuda"
    # (b) one captured step, replayed
    mine, ref = mk_params(), mk_params()
    o1, o2 = mk(Adam, mine, capturable=True), mk(torch.optim.Adam, ref)
    static_g = [torch.zeros_like(p) for p in mine]
    for a, g in zip(mine, static_g):
        a.grad = g
    for b in ref:
        b.grad = torch.zeros_like(b)
    o1._capturable_state()                 # state tensors are allocated outside the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            o1.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # (the capture itself executed nothing: the parameters and the step counts are untouched)
    assert float(o1.state[mine[0]]["step"]) == 0.0 and torch.equal(mine[0].detach().cpu(), base[0])
    for k in range(6):
        if k == 3:
            o1.param_groups[1]["lr"] = o2.param_groups[1]["lr"] = 1e-3
            o1.push_lrs()
        for sg, b, g in zip(static_g, ref, grads[k]):
            sg.copy_(g)
            b.grad.copy_(g)
        graph.replay()
        o2.step()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(mine, ref)):
        _close(a, b.detach().cpu().numpy(), f"replayed param {i}")
    assert float(o1.state[mine[0]]["step"]) == 6.0
