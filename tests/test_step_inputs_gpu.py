"""-m gpu: emd_select_step_inputs -- the per-step inputs of a replayed training step written by one launch (camera block, frame index,
frame time, coarse-to-fine level, status log, the row of the next launch) -- against the host arithmetic it replaces, and the
bench presets of BASELINE.json configs[0] / configs[1] end to end."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda", 0)


def test_select_step_inputs_matches_the_host_arithmetic():
    sys.path.insert(0, ROOT)
    import bench
    from emd_amd.motion import TrackOffsetHeads
    rows, F = 37, 50
    g = torch.Generator().manual_seed(4)
    table = torch.randn(rows, 38, generator=g).to(DEV)
    frames = torch.randint(0, F, (rows,), generator=g, dtype=torch.int32).to(DEV)
    nxt = torch.tensor([(r * 7 + 3) % rows for r in range(rows)], dtype=torch.int64, device=DEV)
    sel = torch.zeros(1, dtype=torch.int64, device=DEV)
    prev = torch.full((1,), -1, dtype=torch.int64, device=DEV)
    out_row = torch.zeros(38, device=DEV)
    frame_out = torch.zeros(1, dtype=torch.int32, device=DEV)
    t_out = torch.zeros(1, device=DEV)
    kf = torch.zeros(1, dtype=torch.int32, device=DEV)
    status = torch.zeros(4, dtype=torch.int32, device=DEV)
    log = torch.full((rows, 4), -7, dtype=torch.int32, device=DEV)
    heads = TrackOffsetHeads(3)
    sched = (heads.min_embeddings, heads.max_embeddings, 20)          # a short schedule so that the clamp is exercised
    row, seen = 5, []
    sel.fill_(row)
    for step in range(12):
        status.copy_(torch.tensor([100 + step, 0, 200 + step, 300 + step], dtype=torch.int32))
        bench.select_step_inputs(sel, table, out_row, frames, frame_out, t_out, F, sched, kf, status, log, prev, nxt)
        torch.cuda.synchronize()
        assert torch.equal(out_row, table[row])
        assert int(frame_out) == int(frames[row]) and abs(float(t_out) - int(frames[row]) / (F - 1)) < 1e-7
        assert int(kf) == heads.int_lininterp(row, sched[0], sched[1], sched[2])
        assert int(prev) == row
        if seen:                          # the status words of the launch before went to the row that launch selected
            assert log[seen[-1]].tolist() == [100 + step, 0, 200 + step, 300 + step]
        seen.append(row)
        row = int(nxt[row])
        assert int(sel) == row            # the launch advanced its own selector
    sel.fill_(-1)
    status.fill_(9)
    bench.select_step_inputs(sel, table, out_row, status=status, status_log=log, prev_sel=prev)
    torch.cuda.synchronize()
    assert log[seen[-1]].tolist() == [9, 9, 9, 9] and int(prev) == -1
    untouched = [r for r in range(rows) if r not in seen]
    assert all(log[r].tolist() == [-7, -7, -7, -7] for r in untouched)


@pytest.mark.parametrize("config,extra", [(0, []), (1, ["--gaussians", "120000", "--height", "200", "--width", "304"])], ids=["config0", "config1-small"])
def test_bench_presets_run(config, extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--steps", "4", "--warmup", "2", "--repeats", "1",
                        "--no-cpu-baseline"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2500:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["baseline_config_index"] == config and d["value"] > 0
    if config == 0:
        assert d["unit"] == "ms" and d["config"]["gaussians"] == 10_000 and d["config"]["height"] == 256 and d["higher_is_better"] is False
    else:
        assert d["unit"] == "iters/s" and d["config"]["track_heads"] is False and d["config"]["step_issue"].startswith("hipGraph replay")


@pytest.mark.parametrize("config,extra", [(3, ["--gaussians", "200000", "--height", "200", "--width", "304"]),
                                          (4, ["--gaussians", "300000", "--height", "200", "--width", "304"])], ids=["config3-small", "config4-small"])
def test_bench_multi_gpu_presets_run_as_the_per_rank_workload(config, extra):
    """BASELINE configs[3] (4-camera rig, deformation residual as an input of the fused transform) and configs[4] (6-camera rig, 48 actors,
    densification statistics every step + one density-control event behind the timed steps) at --gpus 1 = the per-rank workload."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--steps", "6", "--warmup", "2", "--repeats", "1",
                        "--no-cpu-baseline"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2500:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c = d["config"]
    assert c["baseline_config_index"] == config and d["value"] > 0 and c["step_issue"].startswith("hipGraph replay") and c["ranks_seen"] == 1
    if config == 3:
        assert c["rig_cameras"] == 4 and "residual_dx" in c["deformation_residual"] and "residual_dq" in c["deformation_residual"]
        assert "density_control_event" not in d
    else:
        ev = d["density_control_event"]
        assert c["rig_cameras"] == 6 and c["densification_stats_in_step"] is True
        first = ev["first_event"]          # (two events: the top-level fields are the second one, what a loop pays every 100 iterations)
        assert first["n_before"] == 300000 and ev["n_before"] == first["n_after"]
        for e in (first, ev):
            assert e["cloned"] + e["split"] > 0 and e["n_after"] == e["n_before"] + e["cloned"] + e["split"] - e["pruned"]
            assert e["overflow_after"] == 0 and e["ms_per_step_after"] > 0 and e["re_record_ms"] > 0
