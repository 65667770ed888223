"""CPU, build container only: tests/gen_golden.py, run against the mounted reference, reproduces every committed fixture bit for bit
(the fixtures are what the generator says they are).  Skipped where /root/reference is absent (the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference is only mounted in the build container")
def test_generator_reproduces_committed_fixtures(tmp_path):
    env = dict(os.environ, EMD_GOLDEN_OUT=str(tmp_path))
    p = subprocess.run([sys.executable, os.path.join(HERE, "gen_golden.py")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    committed = sorted(f for f in os.listdir(os.path.join(HERE, "golden")) if f.endswith(".npz"))
    assert committed == sorted(os.listdir(tmp_path))
    for f in committed:
        a, b = np.load(os.path.join(HERE, "golden", f)), np.load(os.path.join(tmp_path, f))
        assert set(a.files) == set(b.files), f
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (f, k)
