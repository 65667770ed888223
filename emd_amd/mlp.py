"""One level of the EMD deformation network's MLPs (trunk + heads) on the fused fp32-MFMA kernels of csrc/mlp.hip.

    h     = b_eff + W0[:, col_a : col_a + ka] xa + W0[:, col_b : col_b + kb] xb                 (Deformation.feature_out, defor_depth = 1)
    out_k = W_out act(... act(W_1 in_k + b_1) ...) + b_out,  in_k = relu(h) (deformation heads) or h (dino_head)

S3Gaussian/scene/deformation.py:100-185 (construction), 254-337 (use).  `level_mlp` is ONE autograd node: its backward runs one
kernel per head (each writes its own contribution to dL/dh and accumulates its weight gradients) and one for the trunk (sums the
contributions, writes dL/dxa, dL/dxb, accumulates dW0) -- no intermediate [N, 64..192] tensor and no element-wise launch exists in
either direction.  There is no CPU path (the checker's restatement is oracle/deform_oracle.py)."""
import ctypes as C
import os

import torch

from . import _lib as L

WIDTH = 64
# heads of a level without HexPlane features form h from the embedding themselves (csrc/mlp.hip, EmdMlpBranch.xb); EMD_MLP_RECOMPUTE_H=0: the trunk launch (A/B)
RECOMPUTE_H = os.environ.get("EMD_MLP_RECOMPUTE_H", "1") != "0"


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def eligible(ka, kb, hidden_shapes, out_dims):
    """The shapes csrc/mlp.hip serves: width 64 everywhere, ka a multiple of 4 up to 128, kb <= 8, out_dim <= 64, at most six heads."""
    return (0 <= ka <= 128 and ka % 4 == 0 and 0 <= kb <= 8 and ka + kb > 0 and all(tuple(s) == (WIDTH, WIDTH) for s in hidden_shapes) and
            all(1 <= o <= 64 for o in out_dims) and 1 <= len(out_dims) <= L.MLP_MAX_BRANCHES)


class _LevelMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, xa, xb, w0, b_eff, *params):
        """spec = (col_a, col_b, ((relu_input, depth, out_dim), ...), l1_heads); params = per branch: (w_h, b_h) x depth, w_out, b_out.
        l1_heads: indices of the heads whose regulariser mean |out| is returned behind the outputs (formed by the head's own kernels)."""
        lib = L.load()
        col_a, col_b, branches, l1_heads = spec
        dev = w0.device
        if dev.type != "cuda":
            raise L.EmdError("level_mlp needs tensors on a ROCm device; there is no CPU path")
        f = lambda t: None if t is None else t.detach().contiguous().float()
        xa_c, xb_c, w0_c, b_c = f(xa), f(xb), f(w0), f(b_eff)
        params_c = [f(p) for p in params]
        N = (xa_c if xa_c is not None else xb_c).shape[0]
        # A level without HexPlane features (`no_fine_hexplane_features`, the reference's run script): h = b + W0[:, emb] emb is eight MFMAs per 32
        # rows, so every one-hidden-layer head recomputes it from the [N, kb] embedding (EmdMlpBranch.xb) -- no trunk launch, no [N, 64] tensor
        # written once and read back by every head in both directions.  (RECOMPUTE_H = False keeps the trunk launch: the A/B switch.)
        recompute = RECOMPUTE_H and xa_c is None and xb_c is not None and all(depth == 1 for _, depth, _ in branches)
        h = None if recompute else torch.empty(N, WIDTH, device=dev, dtype=torch.float32)
        t = L.EmdMlpTrunk()
        t.num_points, t.ka, t.kb, t.ld_w = N, 0 if xa_c is None else xa_c.shape[1], 0 if xb_c is None else xb_c.shape[1], w0_c.shape[1]
        t.col_a, t.col_b = int(col_a), int(col_b)
        t.xa, t.xb, t.w, t.b, t.h = L.ptr(xa_c), L.ptr(xb_c), w0_c.data_ptr(), b_c.data_ptr(), L.ptr(h)
        if not recompute:
            L.check(lib.emd_mlp_trunk_forward(C.byref(t), _stream()), "emd_mlp_trunk_forward")
        outs, structs, i = [], [], 0
        l1 = torch.zeros(max(len(l1_heads), 1), device=dev, dtype=torch.float32)
        for k, (relu_input, depth, out_dim) in enumerate(branches):
            b = L.EmdMlpBranch()
            b.num_points, b.depth, b.relu_input, b.out_dim, b.h = N, depth, 1 if relu_input else 0, out_dim, L.ptr(h)
            if recompute:
                b.xb, b.w_in, b.b_in, b.kb_in, b.ld_w_in, b.col_in = xb_c.data_ptr(), w0_c.data_ptr(), b_c.data_ptr(), t.kb, t.ld_w, t.col_b
            for d in range(depth):
                b.w_hidden[d], b.b_hidden[d] = params_c[i].data_ptr(), params_c[i + 1].data_ptr()
                i += 2
            b.w_out, b.b_out = params_c[i].data_ptr(), params_c[i + 1].data_ptr()
            i += 2
            out = torch.empty(N, out_dim, device=dev, dtype=torch.float32)
            b.out = out.data_ptr()
            if k in l1_heads:
                b.l1_sum = l1[l1_heads.index(k):].data_ptr()
            L.check(lib.emd_mlp_branch_forward(C.byref(b), _stream()), "emd_mlp_branch_forward")
            outs.append(out)
            structs.append(b)
        ctx.spec, ctx.trunk, ctx.structs, ctx.N = spec, t, structs, N
        ctx.set_materialize_grads(False)        # a head nobody uses costs no backward launch and leaves its parameters' .grad at None
        ctx.has = (xa is not None, xb is not None)
        ctx.num_params = len(params_c)
        ctx.save_for_backward(xa_c, xb_c, w0_c, b_c, h, *params_c, *[outs[k] for k in l1_heads])
        return tuple(outs) + tuple(l1[j] for j in range(len(l1_heads)))

    @staticmethod
    def backward(ctx, *g_outs):
        lib = L.load()
        xa_c, xb_c, w0_c, b_c, h, *rest = ctx.saved_tensors
        params_c, l1_outs = rest[:ctx.num_params], rest[ctx.num_params:]
        col_a, col_b, branches, l1_heads = ctx.spec
        g_l1 = g_outs[len(branches):]
        g_outs = g_outs[:len(branches)]
        N, dev = ctx.N, w0_c.device
        # every accumulated gradient is carved from ONE zero-filled allocation
        sizes = [w0_c.numel(), b_c.numel()] + [p.numel() for p in params_c]
        flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        parts = torch.split(flat, sizes)
        d_w0, d_b = parts[0].view_as(w0_c), parts[1].view_as(b_c)
        d_params = [p.view_as(q) for p, q in zip(parts[2:], params_c)]
        g_hs, i = [], 0
        unused = set()               # parameters of heads whose output received no gradient: their gradient is None, as autograd would leave it
        keep = []
        for k, ((relu_input, depth, out_dim), b, g_out) in enumerate(zip(branches, ctx.structs, g_outs)):
            n_par = 2 * depth + 2
            gl = g_l1[l1_heads.index(k)] if k in l1_heads else None        # gradient of the head's mean |out|: folded into g_out by the kernel
            if g_out is None and gl is None:
                unused.update(range(i, i + n_par))
            else:
                g = L.EmdMlpBranchGrads()
                # the one-hidden-layer heads of a level chain their dL/dh through ONE buffer (g_h = g_h_in + own, in place): the trunk's
                # backward, which is bound by its reads, gets one tensor instead of one per head (measured at 2 M rows: trunk 0.96 -> 0.81 and
                # 0.46 -> 0.25 ms against +0.02 ms per head).  The two-hidden-layer kernel has no registers left for the extra tile
                # (1.13 -> 1.42 ms with it): it keeps its own buffer.
                chained = bool(g_hs) and depth == 1
                if chained:
                    g_h = g_hs[0]
                    g.g_h_in = g_h.data_ptr()
                else:
                    g_h = torch.empty(N, WIDTH, device=dev, dtype=torch.float32)
                if g_out is not None:
                    g_out = g_out.contiguous().float()
                    g.g_out = g_out.data_ptr()
                if gl is not None:
                    gl = gl.detach().reshape(1).float().contiguous()
                    keep.append(gl)
                    g.l1_grad, g.out = gl.data_ptr(), l1_outs[l1_heads.index(k)].data_ptr()
                g.g_h = g_h.data_ptr()
                for d in range(depth):
                    g.d_w_hidden[d], g.d_b_hidden[d] = d_params[i + 2 * d].data_ptr(), d_params[i + 2 * d + 1].data_ptr()
                g.d_w_out, g.d_b_out = d_params[i + 2 * depth].data_ptr(), d_params[i + 2 * depth + 1].data_ptr()
                L.check(lib.emd_mlp_branch_backward(C.byref(b), C.byref(g), _stream()), "emd_mlp_branch_backward")
                if not chained:
                    g_hs.append(g_h)
            i += n_par
        d_xa = d_xb = None
        if g_hs:
            tg = L.EmdMlpTrunkGrads()
            tg.num_gh = len(g_hs)
            for k, g_h in enumerate(g_hs):
                tg.g_h[k] = g_h.data_ptr()
            if ctx.has[0] and ctx.needs_input_grad[1]:
                d_xa = torch.empty_like(xa_c)
            if ctx.has[1] and ctx.needs_input_grad[2]:
                d_xb = torch.empty_like(xb_c)
            tg.d_xa, tg.d_xb, tg.d_w, tg.d_b = L.ptr(d_xa), L.ptr(d_xb), d_w0.data_ptr(), d_b.data_ptr()
            L.check(lib.emd_mlp_trunk_backward(C.byref(ctx.trunk), C.byref(tg), _stream()), "emd_mlp_trunk_backward")
        if not g_hs:
            d_w0 = d_b = None
        return (None, d_xa, d_xb, d_w0, d_b, *[None if k in unused else p for k, p in enumerate(d_params)])


def level_mlp(xa, xb, w0, b_eff, col_a, col_b, branches, l1_heads=()):
    """`branches`: list of (relu_input, [(w_hidden, b_hidden), ...], (w_out, b_out)); returns the list of head outputs [N, out_dim]
    followed, for every index in `l1_heads`, by that head's mean |out| (a 0-d tensor: the L1 regulariser of a residual head,
    S3Gaussian/train.py:238-310, formed by the head's forward kernel and differentiated inside its backward kernel).
    `xa` [N,128] or None, `xb` [N,kb <= 8] or None; `w0` is the first layer's full weight [64, ld] whose column blocks starting at
    `col_a` / `col_b` multiply xa / xb; `b_eff` [64] is its bias plus whatever is constant over the Gaussians."""
    spec, params = [], []
    for relu_input, hidden, (w_out, b_out) in branches:
        spec.append((bool(relu_input), len(hidden), int(w_out.shape[0])))
        for w, b in hidden:
            params += [w, b]
        params += [w_out, b_out]
    return list(_LevelMLP.apply((int(col_a), int(col_b), tuple(spec), tuple(int(k) for k in l1_heads)), xa, xb, w0, b_eff, *params))
