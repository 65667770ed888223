"""CPU: the Adam / learning-rate-schedule oracle against the reference's own optimiser (tests/golden/s3g_adam.npz:
GaussianModel.training_setup + update_learning_rate + optimizer.step over six iterations on CPU)."""
import os

import numpy as np

from oracle import adam_oracle as ao

G = os.path.join(os.path.dirname(__file__), "golden")


def schedule(g, name, it):
    """update_learning_rate (gaussian_model.py:224-243) from the recorded option values."""
    a = lambda k: float(g["arg_" + k])
    s = float(g["spatial_lr_scale"])
    if name == "xyz":
        return ao.expon_lr(it, a("position_lr_init") * s, a("position_lr_final") * s, 0, a("position_lr_delay_mult"), a("position_lr_max_steps"))
    if "grid" in name:
        return ao.expon_lr(it, a("grid_lr_init") * s, a("grid_lr_final") * s, 0, a("deformation_lr_delay_mult"), a("position_lr_max_steps"))
    if name == "deformation":
        return ao.expon_lr(it, a("deformation_lr_init") * s, a("deformation_lr_final") * s, 0, a("deformation_lr_delay_mult"), a("position_lr_max_steps"))
    if name == "sky_cube_map":
        return ao.expon_lr(it, a("sky_cube_map_lr_init"), a("sky_cube_map_lr_final"), 0, 1.0, a("sky_cube_map_max_steps"))
    return {"f_dc": a("feature_lr"), "f_rest": a("feature_lr") / 20.0, "opacity": a("opacity_lr"), "scaling": a("scaling_lr"),
            "rotation": a("rotation_lr"), "embedding": a("feature_lr")}[name]


def test_lr_schedule_matches_reference():
    g = np.load(os.path.join(G, "s3g_adam.npz"))
    names = [str(n) for n in g["group_names"]]
    assert names == ["xyz", "deformation", "grid", "f_dc", "f_rest", "opacity", "scaling", "rotation", "embedding", "sky_cube_map"]
    for k, it in enumerate(g["iters"]):
        for j, n in enumerate(names):
            np.testing.assert_allclose(schedule(g, n, int(it)), g[f"lr_{k}"][j], rtol=2e-7, err_msg=f"{n} @ {it}")


def test_adam_oracle_matches_reference():
    g = np.load(os.path.join(G, "s3g_adam.npz"))
    assert abs(float(g["eps"]) / 1e-15 - 1) < 1e-6 and abs(float(g["beta1"]) - 0.9) < 1e-7 and abs(float(g["beta2"]) - 0.999) < 1e-7
    for j, n in enumerate(str(x) for x in g["group_names"]):
        p = g[f"init_{n}"]
        m, v = np.zeros_like(p), np.zeros_like(p)
        for k in range(len(g["iters"])):
            p, m, v = ao.adam_step(p, g[f"grad_{k}_{n}"], m, v, k + 1, schedule(g, n, int(g["iters"][k])))
        # fp32 rounding only (cancellation in g - m leaves ~1e-7 of the tensor's scale)
        for got, key in ((m, "exp_avg"), (v, "exp_avg_sq"), (p, "final")):
            want = g[f"{key}_{n}"]
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6 * np.abs(want).max(), err_msg=f"{key} {n}")
