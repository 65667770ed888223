"""-m gpu: the N > 1 path of bench.py (per-rank cameras, gradient all-reduce, barriers, max-over-ranks timing, one JSON line
from rank 0) run as TWO ranks on the one GPU of the test box: torch.distributed.run + EMD_BENCH_SHARE_GPU=1 + the gloo backend
(RCCL cannot put two ranks on one device).  Functional check only; the numbers mean nothing."""
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


def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ, EMD_BENCH_SHARE_GPU="1", EMD_DP_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--gaussians", "60000", "--height", "128", "--width", "192"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["views_per_step"] == 2 and d["config"]["parallelism"] == "view-parallel dp2" and d["config"]["rig_cameras"] == 2
    assert d["config"]["step_issue"].startswith("hipGraph replay"), d["config"]["step_issue"]     # N > 1 is issued like N = 1


def test_bench_launches_its_own_ranks_without_torchrun():
    """`python bench.py --gpus 2` with NO torchrun in the command and no WORLD_SIZE in the environment -- the driver's command form with N > 1:
    bench.py starts its ranks itself as a fresh child process (decided before any GPU call), relays rank 0's one JSON line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EMD_BENCH_SHARE_GPU="1", EMD_DP_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "1",
           "--gaussians", "60000", "--height", "128", "--width", "192"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["config"]["ranks_seen"] == 2 and len(d["config"]["rank_ms_per_step"]["per_rank"]) == 2
    assert d["config"]["rank_ms_per_step"]["min"] > 0
    assert "cpu_baseline" not in d                       # the CPU leg belongs to the 1-GPU run


@pytest.mark.parametrize("ranks,cams", [(4, 4), (8, 6)], ids=["4-ranks-4-camera-rig", "8-ranks-6-camera-rig"])
def test_bench_at_the_world_sizes_of_the_scale_run(ranks, cams):
    """`python bench.py --gpus 4` / `--gpus 8` (the other two points of the driver's scaling run) end to end on the test box's one GPU (gloo): the
    rig of that world size, rank -> view mapping (8 ranks on 6 cameras: mixed timestamps, per-view pose tables gathered), the dense exchange
    (dp.compact_pays says rows do not pay at 4 and 8 ranks), one line from rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EMD_BENCH_SHARE_GPU="1", EMD_DP_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--repeats", "0",
           "--gaussians", "60000", "--height", "128", "--width", "192"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2500:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c = d["config"]
    assert d["n_gpus"] == ranks and c["ranks_seen"] == ranks and c["rig_cameras"] == cams and len(c["rank_ms_per_step"]["per_rank"]) == ranks
    assert c["step_issue"].startswith("hipGraph replay") and c["exchange_rows_per_view"] is None and d["value"] > 0


def test_bench_refuses_a_world_size_that_contradicts_gpus_before_touching_the_gpu():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


def test_bench_config4_two_ranks_agree_through_the_density_control_event():
    """BASELINE configs[4] at reduced size on two ranks: statistics every step, reduced over the ranks before the event, identical surgery on both
    (bench.py asserts the point counts agree), the step recorded again, more steps."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EMD_BENCH_SHARE_GPU="1", EMD_DP_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "4", "--steps", "4", "--warmup", "1", "--repeats", "0",
           "--gaussians", "300000", "--height", "160", "--width", "240"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2500:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    ev = d["density_control_event"]
    assert d["config"]["rig_cameras"] == 6 and d["config"]["densification_stats_in_step"] is True
    first = ev["first_event"]          # (two events; the top-level fields are the second one)
    assert first["n_before"] == 300000 and ev["n_before"] == first["n_after"]
    for e in (first, ev):
        assert e["n_after"] != e["n_before"] and e["cloned"] + e["split"] > 0 and e["overflow_after"] == 0 and e["iters_per_s_after"] > 0


@pytest.mark.parametrize("ranks,mixed,compact", [(2, False, False), (3, True, False), (2, False, True), (3, True, True)],
                         ids=["2-ranks-one-timestamp", "3-ranks-mixed-timestamps", "2-ranks-compact-rows", "3-ranks-mixed-compact-rows"])
def test_factored_sh_exchange_equals_dense_allreduce(ranks, mixed, compact):
    """The in-backward gradient exchange (SH factors + one slab) against the plain dense all-reduce; with three ranks on three
    different timestamps the rebuild needs one gathered actor pose table per view (6 cameras on 8 GPUs, config 5).  `compact`: the
    visibility-compacted form (index + value rows of the visible Gaussians, added in rank order): same values, bit-identical replicas,
    an undersized capacity reported."""
    env = dict(os.environ, EMD_BENCH_SHARE_GPU="1", EMD_DP_BACKEND="gloo")
    if mixed:
        env["EMD_DP_MIXED"] = "1"
    if compact:
        env["EMD_DP_COMPACT"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_factored_check.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    assert any(l.startswith("OK ") for l in p.stdout.splitlines()), p.stdout[-1500:]


def test_two_ranks_through_density_control_stay_bit_identical():
    """tests/dp_densify_check.py: 130 steps of the view-parallel training loop on two ranks -- recorded steps, factored exchange, reduced
    statistics, densify / prune / opacity reset with re-recording -- replicas bit-identical after every event, and agreement with the
    one-process loop that renders both views per step."""
    env = dict(os.environ, EMD_BENCH_SHARE_GPU="1", EMD_DP_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_densify_check.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert any(l.startswith("OK ") for l in p.stdout.splitlines()), p.stdout[-1500:]
    print([l for l in p.stdout.splitlines() if l.startswith("OK ")][0])


def test_bench_and_factored_exchange_over_rccl_when_two_gpus_are_present():
    """On a node with >= 2 GPUs: the same two checks over the real RCCL backend, one rank per GPU (skipped on 1-GPU boxes)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("EMD_BENCH_SHARE_GPU", "EMD_DP_BACKEND")}
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    p = subprocess.run(launch + ["--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                                 "--no-cpu-baseline", "--gaussians", "60000", "--height", "128", "--width", "192"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    for mixed in (False, True):
        e2 = dict(env, EMD_DP_MIXED="1") if mixed else env
        p = subprocess.run(launch + ["--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_factored_check.py")],
                           cwd=ROOT, env=e2, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
        assert any(l.startswith("OK ") for l in p.stdout.splitlines()), p.stdout[-1500:]
