"""-m "not gpu": the N > 1 path of the view-parallel DP layer with world_size-2 gloo processes on CPU
(view sharding, in-place gradient averaging, densification statistics reduced as sum / sum / max)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from emd_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dp.world_size() == world
    g = torch.Generator().manual_seed(0)           # identical replicas on every rank
    params = [torch.nn.Parameter(torch.randn(50, 3, generator=g)), torch.nn.Parameter(torch.randn(50, 16, 3, generator=g)),
              torch.nn.Parameter(torch.randn(50, 1, generator=g))]
    views = list(range(6))
    picked = []
    for step in range(3):
        v = views[dp.view_for(step, rank, world, len(views))]
        picked.append(v)
        for p in params:
            p.grad = None
        # a per-view "loss": gradients differ per view, third parameter untouched by odd views (grad None on some ranks is
        # not allowed in DP; give it zeros like a rasterizer backward does for invisible Gaussians)
        loss = sum(((p * (v + 1)) ** 2).sum() for p in params[:2]) + params[2].sum() * (0.0 if v % 2 else 1.0)
        loss.backward()
        dp.allreduce_gradients(params)
    gn = torch.full((50, 1), float(rank + 1))
    dn = torch.ones(50, 1)
    mr = torch.arange(50, dtype=torch.float32) * (rank + 1)
    dp.reduce_densification_stats(gn, dn, mr)
    torch.save(dict(picked=picked, grads=[p.grad.clone() for p in params], gn=gn, dn=dn, mr=mr), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_view_parallel_gradient_averaging_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    # step s: rank r renders view (2 s + r) % 6 -> disjoint views per step, all views covered over 3 steps
    assert r0["picked"] == [0, 2, 4] and r1["picked"] == [1, 3, 5]
    for a, b in zip(r0["grads"], r1["grads"]):
        torch.testing.assert_close(a, b, rtol=0, atol=0)           # replicas hold identical averaged gradients
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(50, 3, generator=g)
    expect = 0.5 * (2 * p0 * 5 ** 2 + 2 * p0 * 6 ** 2)             # last step: views 4 and 5 -> factors 5 and 6
    torch.testing.assert_close(r0["grads"][0], expect, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(r0["grads"][2], torch.full((50, 1), 0.5))
    torch.testing.assert_close(r0["gn"], torch.full((50, 1), 3.0))    # sum over views
    torch.testing.assert_close(r0["dn"], torch.full((50, 1), 2.0))
    torch.testing.assert_close(r0["mr"], torch.arange(50, dtype=torch.float32) * 2)   # max over views


def test_world_size_one_is_the_reference_step():
    p = torch.nn.Parameter(torch.ones(4, 3))
    (p * 3).sum().backward()
    before = p.grad.clone()
    dp.allreduce_gradients([p])                # no process group: must be a no-op
    assert torch.equal(p.grad, before) and dp.world_size() == 1
    assert dp.view_for(7, 0, 1, 5) == 2
    vg = torch.tensor([[3.0, 4.0, 9.0], [1.0, 0.0, 0.0]])
    g, d, r = dp.densification_stats(vg, torch.tensor([5, 0], dtype=torch.int32))
    assert g[0, 0] == 5.0 and g[1, 0] == 0.0 and d[:, 0].tolist() == [1.0, 0.0] and r.tolist() == [5.0, 0.0]


def test_bench_view_mapping_gives_every_rank_a_distinct_view():
    """bench.py's (frame, camera) per rank = dp.frame_and_camera: distinct views inside every step at 1, 2, 4 and 8 GPUs, one
    timestamp per step when the rig has as many cameras as there are ranks, two ranks on the next timestamp at 8 GPUs / 6 cameras."""
    from emd_amd import dp
    for world in (1, 2, 4, 8):
        cams = dp.rig_size(world)
        assert cams == {1: 1, 2: 2, 4: 4, 8: 6}[world]
        for step in range(60):
            views = [dp.frame_and_camera(step, r, world, 50, cams) for r in range(world)]
            assert len(set(views)) == world, (world, step, views)
            if world == cams:
                assert len({f for f, _ in views}) == 1
    v8 = [dp.frame_and_camera(0, r, 8, 50, 6) for r in range(8)]
    assert v8 == [(0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (1, 0), (1, 1)]


def test_compact_exchange_is_chosen_by_bytes_per_link():
    """dp.compact_pays: bytes ONE xGMI link carries per step, dense (2 x 44 N / W + 12 N) against compact (64 B x capacity); at the bench
    scene's visibility (V / N = 0.53, capacity with its margins ~0.6 N) that means rows at 2 ranks, the dense slab at 4 and 8."""
    from emd_amd import dp
    N = 2_000_000
    cap = dp.visible_capacity(int(0.53 * N * 1.15))
    assert cap % 1024 == 0 and 0.6 * N < cap < 0.7 * N
    assert dp.compact_pays(2, N, cap) and not dp.compact_pays(4, N, cap) and not dp.compact_pays(8, N, cap)
    assert not dp.compact_pays(1, N, cap)
    assert dp.compact_pays(4, N, int(0.3 * N)) and not dp.compact_pays(2, N, N)          # few visible: rows even at 4 ranks; all visible: never


def test_bench_decides_about_ranks_before_touching_a_gpu():
    """`python bench.py --gpus N` (N > 1) without torchrun starts its own ranks as a child process -- or, on a node with fewer GPUs, says so and
    exits; a WORLD_SIZE that contradicts --gpus is refused.  Both decisions are taken before any GPU call: they work on this GPU-less box."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EMD_BENCH_SHARE_GPU")}
    if __import__("torch").cuda.device_count() < 3:
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3"], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 2 and "--gpus 3 but this node shows" in p.stderr and not p.stdout.strip()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr
