"""CPU restatement of the optimiser step of the training loop.  TEST INFRASTRUCTURE ONLY.

The reference steps `torch.optim.Adam(l, lr=0.0, eps=1e-15)` (S3Gaussian/scene/gaussian_model.py:188-201, train.py:428) with
per-group learning rates that `update_learning_rate` (:224-243) refreshes from `get_expon_lr_func`
(S3Gaussian/utils/general_utils.py:196-229).  The algorithm lives in PyTorch (torch/optim/adam.py, `_single_tensor_adam`; the
version in this image: see `torch.__version__`); restated here in numpy float32, operation by operation:
    m = m + (1 - beta1) (g - m);   v = v beta2 + (1 - beta2) g g
    p = p - lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
Pinned by tests/golden/s3g_adam.npz: the reference's own `GaussianModel.training_setup` optimiser and `update_learning_rate`
run on CPU for several iterations with seeded gradients."""
import math

import numpy as np


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    delay = 1.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
    t = min(max(step / max_steps, 0), 1)
    return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-15):
    """One step on float32 arrays (returned, not modified in place); `step` counts from 1."""
    f = np.float32
    p, g, m, v = (np.asarray(x, dtype=np.float32) for x in (p, g, m, v))
    m = m + f(1 - beta1) * (g - m)
    v = v * f(beta2) + f(1 - beta2) * g * g
    denom = np.sqrt(v) / f(math.sqrt(1 - beta2 ** step)) + f(eps)
    p = p + f(-(lr / (1 - beta1 ** step))) * (m / denom)
    return p.astype(np.float32), m.astype(np.float32), v.astype(np.float32)
