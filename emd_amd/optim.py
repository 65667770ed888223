"""Adam for the training loop on the HIP path.

`Adam` is a drop-in for the optimiser the reference builds -- `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over named parameter
groups (S3Gaussian/scene/gaussian_model.py:188-201; OmniRe/models/trainers/base.py:213-253): same constructor arguments, same
`param_groups`, same per-parameter state (`step`, `exp_avg`, `exp_avg_sq`), so the reference's learning-rate updates
(`update_learning_rate`, gaussian_model.py:224-243), its densification surgery on the optimiser state (`_prune_optimizer`,
`cat_tensors_to_optimizer`, `replace_tensor_to_optimizer`, :470-556) and its checkpoints (`optimizer.state_dict()`, :74-118)
work unchanged.  `step()` is one HIP launch for every 32 tensors (`emd_adam_step`) instead of ~10 passes per group.
`capturable=True` (torch.optim.Adam's flag of the same name): the step counts and the groups' learning rates live on the device and the
kernel forms the bias corrections itself, so a `step()` recorded into a hipGraph replays correctly (`push_lrs()` uploads changed rates).
`expon_lr` restates the schedule helper (S3Gaussian/utils/general_utils.py:196-229).  No CPU path."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L


def expon_lr(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation from lr_init (step 0) to lr_final (step max_steps) with the optional sine warm-up."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay_rate = 1.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return helper


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, capturable=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("the reference uses plain Adam (no weight decay, no amsgrad)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self.capturable = bool(capturable)
        self._step_buf = None          # capturable: one device float per parameter, incremented by ONE launch per step
        self._lr_dev = {}              # capturable: one device float per group

    def _capturable_state(self):
        """Device-resident step counts (views of one buffer: a single add advances them all) and learning rates."""
        plist = [p for g in self.param_groups for p in g["params"]]
        dev = plist[0].device
        # The step counts live in ONE buffer (state[p]["step"] are views of it).  It is rebuilt from the per-parameter state whenever that
        # state no longer IS the buffer: load_state_dict or the densification surgery replaced `state[p]` (then its step is a plain number,
        # a fresh tensor, or missing: a parameter whose state was dropped starts again at 0), or the parameter list changed.
        def is_view(i, p):
            st = self.state.get(p)
            t = None if not st else st.get("step")
            return (self._step_buf is not None and isinstance(t, torch.Tensor) and t.device == self._step_buf.device and t.numel() == 1
                    and t.data_ptr() == self._step_buf.data_ptr() + 4 * i)
        if self._step_buf is None or self._step_buf.numel() != len(plist) or self._step_buf.device != dev or \
                not all(is_view(i, p) for i, p in enumerate(plist)):
            old = [float(self.state[p]["step"]) if p in self.state and "step" in self.state[p] else 0.0 for p in plist]
            self._step_buf = torch.tensor(old, dtype=torch.float32, device=dev)
        for i, p in enumerate(plist):
            st = self.state[p]
            if len(st) == 0 or "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] = self._step_buf[i]
        for gi, g in enumerate(self.param_groups):
            if gi not in self._lr_dev or self._lr_dev[gi].device != dev:
                self._lr_dev[gi] = torch.full((1,), float(g["lr"]), dtype=torch.float32, device=dev)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._step_buf = None          # the loaded step counts are the truth: the device buffer is rebuilt from them at the next step

    def push_lrs(self):
        """capturable: upload the groups' current learning rates (call after changing `param_groups[i]["lr"]`, outside a capture)."""
        if not self.capturable:
            return
        self._capturable_state()
        for gi, g in enumerate(self.param_groups):
            self._lr_dev[gi].fill_(float(g["lr"]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        batch, keep = [], []
        cap = self.capturable
        if cap:
            self._capturable_state()
            if not torch.cuda.is_current_stream_capturing():
                self.push_lrs()            # (a capture must not bake the rate of the step it was recorded at: replays read the device copy)
            self._step_buf.add_(1.0)       # every parameter's step, one launch (parameters without a gradient this step included, like
                                           # a fused optimiser step; the reference gives every parameter a gradient every step)

        def flush():
            if not batch:
                return
            a = L.EmdAdamArgs()
            a.num_tensors = len(batch)
            for i, d in enumerate(batch):
                t = a.tensors[i]
                (t.param, t.grad, t.exp_avg, t.exp_avg_sq, t.numel, t.step_size, t.bias_correction2_sqrt, t.one_minus_beta1, t.beta2,
                 t.one_minus_beta2, t.eps, t.step_dev, t.lr_dev) = d
            L.check(lib.emd_adam_step(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emd_adam_step")
            batch.clear()

        for gi, group in enumerate(self.param_groups):
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda":
                    raise L.EmdError("emd_amd.optim.Adam needs parameters on a ROCm device; there is no CPU path")
                if p.dtype != torch.float32 or p.grad.is_sparse:
                    raise L.EmdError("emd_amd.optim.Adam handles dense fp32 parameters")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not cap:
                    state["step"] += 1
                    step = float(state["step"])
                else:
                    step = 1.0             # (unused: the kernel reads the device copy)
                m, v, g = state["exp_avg"], state["exp_avg_sq"], p.grad
                # the kernel walks the four tensors as flat arrays: one memory layout for all of them
                for name, t in (("exp_avg", m), ("exp_avg_sq", v)):
                    if t.stride() != p.stride() or t.shape != p.shape:
                        fixed = torch.empty_like(p, memory_format=torch.preserve_format)
                        fixed.copy_(t)
                        state[name] = fixed
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if g.stride() != p.stride():
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                if not _dense(p):
                    raise L.EmdError("emd_amd.optim.Adam needs parameters that cover their storage without gaps")
                keep.append(g)
                bias_correction1 = 1 - beta1 ** step
                bias_correction2 = 1 - beta2 ** step
                batch.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), group["lr"] / bias_correction1,
                              math.sqrt(bias_correction2), 1 - beta1, beta2, 1 - beta2, group["eps"],
                              state["step"].data_ptr() if cap else None, self._lr_dev[gi].data_ptr() if cap else None))
                if len(batch) == L.ADAM_MAX_TENSORS:
                    flush()
        flush()
        return loss


def _dense(t):
    """True when the tensor's elements tile a contiguous block of memory (any permutation of a contiguous layout)."""
    expected = 1
    for size, stride in sorted(((s, st) for s, st in zip(t.shape, t.stride()) if s > 1), key=lambda x: x[1]):
        if stride != expected:
            return False
        expected *= size
    return True
