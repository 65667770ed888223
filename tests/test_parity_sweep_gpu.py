"""-m gpu: randomised parity sweep -- 40 seeded scenes of random size (1 .. 70 k Gaussians), ragged image sizes, SH degree,
footprint scale, camera yaw, with and without fused motion / residuals; every one bit-exact on keys, geometry and images and
within tolerance on all gradients against the CPU oracle."""
import random

import pytest

from tests.helpers import IMAGE_TOL, compare_backward, compare_forward, make_case, run_hip, run_oracle

pytestmark = pytest.mark.gpu


def _cases():
    rng = random.Random(1234)
    out = []
    for it in range(40):
        n = rng.choice([1, 7, 300, 2000, 9000, 30000, 70000])
        kw = dict(n=n, H=rng.choice([16, 33, 64, 97, 128, 200]), W=rng.choice([16, 40, 96, 150, 256, 333]), seed=1000 + it,
                  sh_degree=rng.choice([0, 1, 2, 3]), scale_mult=rng.choice([0.5, 1.0, 3.0, 6.0]), yaw=rng.choice([0.0, 17.0, -40.0]))
        if n >= 300 and rng.random() < 0.4:
            kw["motion"] = True
            if rng.random() < 0.5:
                kw["residual"] = True
        out.append(kw)
    return out


@pytest.mark.parametrize("kw", _cases(), ids=lambda k: f"n{k['n']}-{k['H']}x{k['W']}-s{k['seed']}")
def test_random_scene(kw):
    case = make_case(**kw)
    orc = run_oracle(case, backward=True)
    hip = run_hip(case, backward=True)
    compare_forward(hip, orc, tol=IMAGE_TOL)
    compare_backward(hip, orc)
