"""One process, ONE GPU, backend "nccl" (= RCCL) with world size 1 and EMD_DP_FORCE=1: GradientExchange issues its collectives for real
-- all_reduce(AVG) of the gradient slab, the all_gather_into_tensor calls, the small all-reduces -- and rebuilds dL/dshs from the
factors.  At world size 1 every collective is the identity, so the result must equal the plain dense gradient of the same step:
 (a) the exchange started from inside backward() (eager step),
 (b) a hipGraph-captured step replayed with the process group (and its watchdog thread) alive, the exchange issued behind the replay,
 (c) a NON-leaf means3D / opacity in front of the rasterizer (a network's residual): the slab must not be reduced in place while
 (d) the factor gathers started BETWEEN the halves of the backward (RasterCall.on_sh_factor -> GradientExchange.start_factors): by HIP
     events they have completed before the projection backward has, i.e. they ran under it; gradients as in (a).
Prints OK from rank 0.  Run by tests/test_dp_nccl_gpu.py."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ["EMD_DP_FORCE"] = "1"
from emd_amd import dp, scenes, RasterCall, RasterOptions, GaussianRasterizer  # noqa: E402
from emd_amd.model import StreetGaussians, render, l1_loss, raster_settings_for  # noqa: E402

faulthandler.enable()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dp.force_exchange() and dp.world_size() == 1
N, H, W = 40000, 96, 128
scene = scenes.add_actors(scenes.make_static_scene(N, seed=0), num_actors=4, pts_per_actor=2000, num_frames=6, seed=1)
model = StreetGaussians(scene, dev, track_heads=True)
params = list(model.parameters())
frame = 3
cam = scenes.rig_camera(frame, 0, H, W)
target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
bg = torch.zeros(3)


def grads():
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def check(got, ref, what):
    assert set(got) == set(ref), (what, sorted(set(got) ^ set(ref)))
    for n in ref:
        tol = 2e-5 * ref[n].abs().max().item() + 1e-12
        err = (got[n] - ref[n]).abs().max().item()
        assert err <= tol, (what, n, err, tol)


def step(factored, options=None):
    for p in params:
        p.grad = None
    rec = RasterCall()
    xchg = None
    opts = options or RasterOptions(factored_sh_grad=factored)
    if factored:
        xchg = dp.GradientExchange(cam.camera_center, actor_ids=model.actor_id)
        rec.on_backward = xchg.start
    out = render(model, cam, bg, frame=frame, iteration=100, options=opts, record=rec)
    if xchg is not None:
        xchg.actor_pose = out["actor_pose"].detach()            # values only: no reference into the autograd graph
    l1_loss(out["render"], target).backward()
    return out, xchg, rec


def say(msg):
    print("[nccl-w1]", msg, file=sys.stderr, flush=True)


# ---- reference: the plain step
step(False)
say("plain step done")
ref = grads()
# ---- (a) exchange from inside backward()
out, xchg, rec = step(True)
assert model._features.grad is None
xchg.finish(model._features, model._xyz, model.active_sh_degree, other_params=params)
n_rest = sum(1 for n_, p in model.named_parameters() if n_ not in ("_xyz", "_scaling", "_rotation", "_opacity", "_features"))
assert xchg.num_collectives == 3 + 1 + 1, (xchg.num_collectives, n_rest)      # gathers, the slab, ONE bucket for the n_rest small tensors
assert xchg._slab_work is not None, "leaf parameters: the slab is reduced in place from inside backward()"
check(grads(), ref, "eager")
say("(a) exchange from inside backward: ok")
# ---- (b) captured step, replayed with the process group alive; exchange behind the replay
opts = RasterOptions(factored_sh_grad=True, no_sync=True)
o = step(False, RasterOptions(no_sync=False))[0]
opts.capacity_hint = int(o["raster_call"].last_status()["num_rendered"] * 1.3) + 1024
n_coll = xchg.num_collectives
del o, out, xchg, rec      # no autograd graph of an eager step may be alive at capture time (its AccumulateGrad nodes are bound to the eager stream)
import gc
gc.collect()
state = {}


def body():
    for p in params:
        p.grad = None
    rec_g = RasterCall()
    o_ = render(model, cam, bg, frame=frame, iteration=100, options=opts, record=rec_g)
    l1_loss(o_["render"], target).backward()
    state["rec"], state["pose"] = rec_g, o_["actor_pose"].detach()


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
    body()
torch.cuda.synchronize()
for _ in range(3):
    graph.replay()
    x = dp.GradientExchange(cam.camera_center.to(dev), actor_ids=model.actor_id, actor_pose=state["pose"])
    x.start(state["rec"])
    assert x._slab_work is not None, "after backward(): the leaves' .grad are views of the slab"
    x.finish(model._features, model._xyz, model.active_sh_degree, other_params=params)
    torch.cuda.synchronize()
    check(grads(), ref, "graph replay")
graph.reset()
say("(b) graph replay + exchange: ok")
# ---- (c) non-leaf inputs in front of the rasterizer
for p in params:
    p.grad = None
delta = torch.zeros(N, 3, device=dev, requires_grad=True)
rs = raster_settings_for(cam, bg, model.active_sh_degree)
rast = GaussianRasterizer(rs, RasterOptions(factored_sh_grad=True))
rec = RasterCall()
xc = dp.GradientExchange(cam.camera_center, actor_ids=model.actor_id)
rec.on_backward = xc.start
pose = model.actor_pose(frame, 100)
xc.actor_pose = pose
img = rast(means3D=model._xyz + delta, means2D=torch.zeros(N, 3, device=dev, requires_grad=True), shs=model._features, opacities=model._opacity,
           scales=model._scaling, rotations=model._rotation, raw_params=True, actor_ids=model.actor_id, actor_pose=pose, record=rec)[0]
l1_loss(img, target).backward()
assert xc._slab_work is None, "a non-leaf means3D: the slab must not be reduced while autograd consumes it"
xc.finish(model._features, model._xyz, model.active_sh_degree, other_params=params + [delta])
g = grads()
check(g, ref, "non-leaf means")
assert (delta.grad - ref["_xyz"]).abs().max().item() <= 2e-5 * ref["_xyz"].abs().max().item()
try:        # a non-leaf shs with the factored gradient would train nothing upstream of it: refused
    rast(means3D=model._xyz, means2D=torch.zeros(N, 3, device=dev, requires_grad=True), shs=model._features * 1.0, opacities=model._opacity,
         scales=model._scaling, rotations=model._rotation, raw_params=True, actor_ids=model.actor_id, actor_pose=pose.detach())
    raise SystemExit("factored_sh_grad accepted a non-leaf shs")
except ValueError:
    pass
say("(c) non-leaf inputs: ok")
# ---- (d) the factor gathers run UNDER the projection backward (a scene large enough for K8 to take ~0.1 ms)
del img, rec, xc, rast, delta, g
gc.collect()
N2 = 800000
scene2 = scenes.add_actors(scenes.make_static_scene(N2, seed=2), num_actors=4, pts_per_actor=2000, num_frames=6, seed=1)
model2 = StreetGaussians(scene2, dev, track_heads=True)
params2 = list(model2.parameters())
H2, W2 = 320, 480
cam2 = scenes.rig_camera(frame, 0, H2, W2)
target2 = torch.rand(3, H2, W2, generator=torch.Generator().manual_seed(4)).to(dev)
ev = {}
s2 = torch.cuda.Stream()


def step2(mode):
    for p in params2:
        p.grad = None
    r_ = RasterCall()
    x_ = None
    if mode != "plain":
        x_ = dp.GradientExchange(cam2.camera_center, actor_ids=model2.actor_id)

        def on_factor(rec_):
            ev["mid"] = torch.cuda.Event(enable_timing=True); ev["mid"].record()             # K7 + factor extraction enqueued, K8 not yet
            x_.start_factors(rec_)
            with torch.cuda.stream(s2):
                for w_ in x_._gathers:
                    w_.wait()                                                                 # s2 waits for the collectives
                ev["gathered"] = torch.cuda.Event(enable_timing=True); ev["gathered"].record()

        def on_bwd(rec_):
            ev["k8"] = torch.cuda.Event(enable_timing=True); ev["k8"].record()               # K8 enqueued: this event follows it
            x_.start(rec_)
        r_.on_backward = on_bwd
        if mode == "overlap":
            r_.on_sh_factor = on_factor
    o_ = render(model2, cam2, bg, frame=frame, iteration=100, options=RasterOptions(factored_sh_grad=(mode != "plain")), record=r_)
    if x_ is not None:
        x_.actor_pose = o_["actor_pose"].detach()
    l1_loss(o_["render"], target2).backward()
    if x_ is not None:
        x_.finish(model2._features, model2._xyz, model2.active_sh_degree, other_params=params2)
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model2.named_parameters() if p.grad is not None}, x_


ref2, _ = step2("plain")
for _ in range(2):
    got2, x2 = step2("overlap")
check(got2, ref2, "gathers between the halves of the backward")
assert x2.num_collectives == 3 + 1 + 1
t_g, t_k8 = ev["mid"].elapsed_time(ev["gathered"]), ev["mid"].elapsed_time(ev["k8"])
say(f"(d) factor gathers done {t_g * 1e3:.0f} us after K7, projection backward done after {t_k8 * 1e3:.0f} us")
assert t_g < t_k8, ("the factor gathers did not run under the projection backward", t_g, t_k8)
print("OK nccl world 1:", len(ref), "gradients, collectives per step", n_coll, f"| factor gathers under K8: {t_g * 1e3:.0f} us vs {t_k8 * 1e3:.0f} us")
dist.barrier()
dist.destroy_process_group()
