"""-m gpu: the RCCL ("nccl") path of the gradient exchange executed for real on the one GPU of the test box -- a process group of
world size 1 with EMD_DP_FORCE=1 (tests/dp_nccl_world1_check.py): all_reduce(AVG) of the slab, the three all_gather_into_tensor calls,
emd_sh_grad_from_factors, a hipGraph replay with the process group alive, and the deferred path for non-leaf rasterizer inputs; plus
bench.py --exchange-only.  (Two and more ranks over gloo on one GPU: tests/test_bench_multirank_gpu.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_gradient_exchange_over_rccl_with_world_size_one():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_nccl_world1_check.py")], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert any(l.startswith("OK nccl world 1") for l in p.stdout.splitlines()), p.stdout[-1500:]


def test_bench_exchange_only_runs_the_collectives_on_one_gpu():
    """bench.py --exchange-only under a one-rank RCCL process group (RANK / WORLD_SIZE from the environment, EMD_DP_FORCE=1)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), EMD_DP_FORCE="1", EMD_DP_INIT_WORLD1="1", RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--exchange-only", "--steps", "3", "--warmup", "1",
                        "--gaussians", "60000", "--height", "128", "--width", "192"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["unit"] == "ms" and d["value"] > 0 and d["config"]["forced_at_world_1"] is True and d["config"]["collectives_per_step"] >= 4


@pytest.mark.parametrize("one_graph", [False, True], ids=["two-graphs-cut-inside-backward", "one-graph-exchange-behind"])
def test_bench_step_with_the_exchange_over_rccl_on_one_gpu(one_graph):
    """The multi-GPU step of bench.py at world size 1 (EMD_DP_FORCE=1, a live RCCL process group with its watchdog thread): by default the
    step is recorded as TWO graphs cut between the halves of the rasterizer's backward -- capture A ended and capture B begun from autograd's
    device thread (RasterCall.on_sh_factor), relaxed capture mode -- with the SH-factor gathers issued between their replays; bench.py's own
    self-check replays them twice and compares status words and parameter gradients with the step issued from Python before it times anything
    (a failed capture would fall back to the eager step, which `step_issue` would show)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), EMD_DP_FORCE="1", EMD_DP_INIT_WORLD1="1", RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--factored-sh", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
           "--gaussians", "60000", "--height", "128", "--width", "192"] + (["--one-graph"] if one_graph else [])
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    issue = d["config"]["step_issue"]
    assert issue.startswith("hipGraph replay"), issue
    assert ("TWO graphs" in issue) == (not one_graph), issue
    assert d["value"] > 0 and "capture failed" not in p.stderr
    # where the step's exchange went (HIP events around its phases, dp.GradientExchange(timing=True)): present whenever collectives are issued
    ex = d["exchange"]
    assert ex["steps"] == 4 and ex["exposed_ms"] > 0 and ex["slab_ms"] > 0 and ex["gather_ms"] > 0 and ex["rebuild_ms"] > 0, ex
