"""View-parallel data parallelism: one process per GPU, camera views of a step sharded over ranks, Gaussian
gradients averaged with an RCCL all-reduce over xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference trains one view per step on one GPU (S3Gaussian/train.py:203; OmniRe/models/trainers/base.py:411);
at world_size 1 this module is a no-op and the step is the reference step.  Design:
  - parameters, actor tables and MLPs are replicated (2 M x 236 B = 472 MB << 288 GB);
  - step s, rank r renders view (s * W + r) of the timestamp-major view list (`frame_and_camera`: a rig's cameras of one timestamp
    first, surplus ranks take cameras of the next timestamp); every rank a distinct view; no collective on the data path until gradients;
  - gradient exchange per step (`GradientExchange`): the SH gradient travels as rank-one factors (all-gather of 12 B per Gaussian
    and rank + camera centres + per-view actor pose tables, dense gradient rebuilt locally), the four small per-Gaussian
    gradients as ONE in-place all-reduce of the slab the rasterizer's backward carved them from, started from inside
    backward(); `allreduce_gradients` is the plain dense path (no rasterizer record needed);
  - densification statistics are computed per view BEFORE reduction (gaussian_model.py:728-730; train.py:406):
    sum of ||d mean2D|| and of counts (SUM), max radii (MAX).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (EMD_DP_INIT_WORLD1=1: a one-rank process group, so that EMD_DP_FORCE=1 can run the collectives of the exchange on one GPU)
    if (world > 1 or os.environ.get("EMD_DP_INIT_WORLD1")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # EMD_DP_BACKEND=gloo: functional test of the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("EMD_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def force_exchange():
    """EMD_DP_FORCE=1 with an initialised process group of ONE rank: GradientExchange issues its collectives anyway (all-reduce,
    all-gathers, the factor rebuild) instead of short-cutting -- the RCCL path can then be executed, and checked against the plain
    gradient, on a single GPU (tests/test_dp_nccl_gpu.py, bench.py --exchange-only)."""
    return os.environ.get("EMD_DP_FORCE", "") not in ("", "0") and dist.is_initialized()


def _in_backward():
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f is None or f() != -1          # unknown: assume the engine is running


def slab_holds_leaf_grads(rec):
    """After backward(): True when the `.grad` of every leaf among the slab's four tensors IS its view of the slab (autograd kept the
    views), so that one in-place all-reduce of the slab reduces all of them."""
    ins, slab = getattr(rec, "slab_inputs", None), rec.grad_slab
    if slab is None or ins is None:
        return False
    lo, hi = slab.data_ptr(), slab.data_ptr() + slab.numel() * 4
    leaves = [t for t in ins if t is not None and t.is_leaf and t.requires_grad]
    return len(leaves) > 0 and all(t.grad is not None and lo <= t.grad.data_ptr() < hi for t in leaves)


def slab_is_exclusive(rec):
    """True when the gradient slab of `rec`'s backward may be all-reduced IN PLACE while autograd is still running: each of the four
    tensors it was carved for (means3D, scales, rotations, opacities) is a leaf that requires grad and holds no `.grad` yet -- then
    autograd only stores the views as the parameters' `.grad` and nothing reads or accumulates into them before `finish()`.  With a
    network in front of the rasterizer (means3D = xyz + dx, ...) the views are inputs of further backward nodes, and a leaf that
    already holds a gradient gets the view ADDED to it: in both cases the reduction has to wait for backward() to end."""
    ins = getattr(rec, "slab_inputs", None)
    if rec.grad_slab is None or ins is None:
        return False
    return all(t is not None and t.is_leaf and t.requires_grad and t.grad is None for t in ins)


def view_for(step, rank, world, num_views):
    """Which view (index into the step-ordered view list) rank `rank` renders at `step`."""
    return (step * world + rank) % num_views


def allreduce_gradients(params, average=True):
    """Average `.grad` of every parameter over ranks, in place.  Largest tensors are issued first."""
    if world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    grads.sort(key=lambda g: -g.numel())
    gloo = dist.get_backend() == "gloo"
    op = dist.ReduceOp.SUM if (gloo or not average) else dist.ReduceOp.AVG
    works = [dist.all_reduce(g, op=op, async_op=True) for g in grads]
    for w in works:
        w.wait()
    if average and gloo:
        w_ = float(world_size())
        for g in grads:
            g.div_(w_)


def rig_size(world):
    """Cameras of the synthetic rig used at `world` GPUs (BASELINE configs 3-5): 1 camera on 1 GPU, a 2- / 4-camera rig on 2 / 4
    GPUs (config 4: rank r <-> camera r of the same timestamp), the 6-camera rig on 8 GPUs (config 5)."""
    return min(max(int(world), 1), 6)


def frame_and_camera(step, rank, world, num_frames, num_cams):
    """(frame, camera) rendered by `rank` at `step`: views are ordered timestamp-major, `view = frame * num_cams + camera`, and
    dealt out by `view_for`.  With `world == num_cams` (configs 3, 4) the ranks of a step hold the cameras of ONE timestamp; with 8
    ranks on the 6-camera rig (config 5) ranks 6 and 7 take the first two cameras of the NEXT timestamp -- every rank renders a
    distinct view, nobody idles, and the exchange below handles the mixed timestamps."""
    v = view_for(step, rank, world, num_frames * num_cams)
    return v // num_cams, v % num_cams


def sh_grad_from_factors(means3D, campos, sh_color_grads, sh_degree, sh_coeffs=16, actor_ids=None, actor_pose=None, residual_dx=None,
                         scale=1.0):
    """dL/dshs [N,sh_coeffs,3] = scale * sum_v basis(normalize(world_mean_v - campos[v])) (x) sh_color_grads[v]   (one HIP launch).
    means3D [N,3]; campos [V,3]; sh_color_grads [V,N,3] (RasterCall.sh_color_grad of every view); actor_pose [A,12] (all views
    share a timestamp) or [V,A,12] (one pose table per view: mixed timestamps)."""
    import ctypes as C
    from . import _lib as L
    from .rasterizer import _fill_motion
    if means3D.device.type != "cuda":
        raise L.EmdError("sh_grad_from_factors needs tensors on a ROCm device; there is no CPU path")
    N, V = means3D.shape[0], campos.shape[0]
    means3D, campos, g = means3D.detach().contiguous().float(), campos.detach().contiguous().float(), sh_color_grads.contiguous().float()
    assert g.shape == (V, N, 3)
    mo = L.EmdMotion()
    ids = None if actor_ids is None else actor_ids.to(torch.int32).contiguous()          # (kept alive until the launch is queued)
    pose = None if actor_pose is None else actor_pose.detach().contiguous().float()
    per_view = pose is not None and pose.dim() == 3
    if per_view:
        assert pose.shape[0] == V, "one pose table per view"
    rdx = None if residual_dx is None else residual_dx.detach().contiguous().float()
    _fill_motion(mo, ids, pose if not per_view else pose[0], rdx, None)
    if per_view:
        mo.actor_pose = pose.data_ptr()
    out = torch.empty(N, sh_coeffs, 3, device=means3D.device, dtype=torch.float32)
    L.check(L.load().emd_sh_grad_from_factors(N, V, int(sh_degree), int(sh_coeffs), means3D.data_ptr(), C.byref(mo), 1 if per_view else 0,
                                              campos.data_ptr(), g.data_ptr(), float(scale), out.data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emd_sh_grad_from_factors")
    return out


def compact_pays(world, num_gaussians, capacity):
    """Whether the visibility-compacted exchange moves fewer bytes over each xGMI link than the dense one, for `capacity` rows per view.
    xGMI is point to point and every collective here can use all W - 1 links of a rank at once (direct schedule), so what counts is the
    bytes one link carries per step: dense = the slab all-reduce's 2 x 44 N / W + the factor all-gather's 12 N; compact = the all-gathers
    of (index + 3 floats) and (index + 11 floats) rows, 64 B per row.  With V / N = 0.53 on the bench scene and a 10 % capacity margin
    this says compact at 2 ranks (37 MB / M Gaussians against 56), dense at 4 (37 against 34) and at 8 (37 against 23)."""
    if world < 2:
        return False
    return 64.0 * capacity < (88.0 / world + 12.0) * num_gaussians


def _ptr_array(tensors):
    import ctypes as C
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def compact_rows(radii, sources, capacity, out=None, counter=None):
    """emd_compact_rows: uint32 rows [(1 + capacity), 1 + sum(widths)] -- header row (count, overflow, 0...) + one (index, values) row per
    Gaussian with radii > 0, values gathered from the [N, w_k] float tensors `sources` (at most four)."""
    import ctypes as C
    from . import _lib as L
    if radii.device.type != "cuda":
        raise L.EmdError("compact_rows needs tensors on a ROCm device; there is no CPU path")
    N = radii.shape[0]
    width = lambda t: 1 if t.dim() < 2 else int(torch.tensor(t.shape[1:]).prod())
    srcs = [t.detach().reshape(N, width(t)) for t in sources]
    for t in srcs:
        assert t.dtype == torch.float32 and t.is_contiguous(), "compact_rows: contiguous float32 sources"
    widths = [int(t.shape[1]) for t in srcs]
    rw = 1 + sum(widths)
    if out is None:
        out = torch.empty((1 + capacity) * rw, dtype=torch.int32, device=radii.device)
    if counter is None:
        counter = torch.empty(1, dtype=torch.int32, device=radii.device)
    r = radii.contiguous().to(torch.int32)
    L.check(L.load().emd_compact_rows(N, r.data_ptr(), len(srcs), _ptr_array(srcs), (C.c_int32 * len(widths))(*widths), int(capacity), out.data_ptr(),
                                      counter.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emd_compact_rows")
    return out.view(1 + capacity, rw)


def scatter_rows(rows, dests, add, scale=1.0, overflow=None):
    """emd_scatter_rows: one view's rows ([(1 + capacity), row_words] uint32, header first) into the [N, w_k] float tensors `dests`."""
    import ctypes as C
    from . import _lib as L
    N = dests[0].shape[0]
    ds = [t.view(N, 1 if t.dim() < 2 else int(torch.tensor(t.shape[1:]).prod())) for t in dests]
    for t in ds:
        assert t.dtype == torch.float32 and t.is_contiguous(), "scatter_rows: contiguous float32 destinations"
    widths = [int(t.shape[1]) for t in ds]
    assert rows.shape[1] == 1 + sum(widths), (tuple(rows.shape), widths)
    L.check(L.load().emd_scatter_rows(rows.data_ptr(), int(rows.shape[0] - 1), N, len(ds), _ptr_array(ds), (C.c_int32 * len(widths))(*widths), 1 if add else 0,
                                      float(scale), L.ptr(overflow), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emd_scatter_rows")


class GradientExchange:
    """One step's gradient exchange of view-parallel training, driven by the rasterizer call's `RasterCall` record.

    SH coefficients (192 of the 236 gradient bytes per Gaussian) are not all-reduced: dL/dshs of a view is the outer product of the
    SH basis of its view direction and a per-Gaussian colour gradient, so every rank all-gathers the [N,3] factors
    (`RasterCall.sh_color_grad`, published by a backward whose rasterizer was built with `factored_sh_grad=True`), the camera
    centres and -- because the ranks of a step may hold different timestamps -- the per-view actor pose tables (A x 12 floats), and
    rebuilds the same dense, view-averaged gradient locally (emd_sh_grad_from_factors).  Everything else is all-reduced in place:
    the four small per-Gaussian gradients as ONE slab (`RasterCall.grad_slab`, 44 B per Gaussian).

        xchg = GradientExchange(campos, actor_ids, actor_pose)        # this rank's view
        rec = RasterCall(); rec.on_backward = xchg.start               # collectives start INSIDE backward(), right behind K8
        rec.on_sh_factor = xchg.start_factors                          # ... the factor gathers already between K7 and K8
        ... render(record=rec) ... loss.backward()
        xchg.finish(sh_param, means_param, sh_degree, other_params)    # rebuild + wait

    World size 1: `start` does nothing and `finish` only rebuilds (the local cost of the factored path).

    `compact_capacity` (rows per view; `compact=None` then chooses by `compact_pays`, True / False force it): the VISIBILITY-COMPACTED form
    of the same exchange -- a view's gradient is zero for every Gaussian it does not see, so the factors travel as (index, 3 floats) rows
    and the slab as (index, 11 floats) rows of the view's visible Gaussians (all-gathers of `1 + capacity` rows; emd_compact_rows), and every
    rank adds the gathered rows into the zeroed slab in RANK order (emd_scatter_rows: a fixed order of float additions, so the replicas stay
    bit-identical).  At 2 ranks that is 37 MB per million Gaussians and link against 56.  The capacity is the caller's (host) number, like
    the binning capacity: more visible Gaussians than rows raise bit 0 of `overflow` (a device int32: the caller's, or one the exchange
    allocates when none is given -- never a null pointer, so a dropped row is always recorded; `overflowed()` reads it, one device-to-host
    copy, and the step's gradients are then incomplete); `dp.visible_capacity(V_max)` adds the margin.

    PRECONDITION of the compacted SLAB (not of the compacted factors): `finish()` zeroes the slab and re-creates it from the gathered rows of
    the visible Gaussians, so the slab must hold NOTHING but this call's rasterizer gradients -- zero on every Gaussian the view does not
    see.  Inside backward() that is what `slab_is_exclusive(rec)` establishes (four leaves without a `.grad`: K8's output is all there is).
    AFTER backward() (a replayed graph) the leaves' `.grad` are views of the slab and autograd may have accumulated other terms into them in
    place (a scale / opacity regulariser, a gradient the leaf already held): the rows are then used only when the caller states
    `slab_pure=True` (the step's loss reaches the four tensors through the rasterizer alone); otherwise the slab travels as the dense
    all-reduce, which keeps and averages every term.  EMD_DP_DEBUG=1 checks the statement on the device before the rows are packed."""

    def __init__(self, campos_local, actor_ids=None, actor_pose=None, residual_dx=None, average=True, bucket_small=True, bucket_bytes=16 << 20,
                 compact=None, compact_capacity=None, overflow=None, slab_pure=False, timing=False):
        self.campos_local, self.actor_ids, self.residual_dx = campos_local, actor_ids, residual_dx
        self.bucket_small, self.bucket_bytes = bucket_small, bucket_bytes
        self.compact, self.compact_capacity, self.overflow = compact, compact_capacity, overflow
        self.slab_pure = bool(slab_pure)
        # timing=True: HIP events on the stream the exchange is issued from (a collective's completion reaches that stream through work.wait()):
        # `times_ms()` then says where a step's exchange went -- see there.  Events are only READ after the caller's synchronisation.
        self._ev = {} if timing else None
        self._rows_f = self._rows_s = None
        self.actor_pose = None if actor_pose is None else actor_pose.detach()       # values only: no reference into an autograd graph
        self.average = average
        self.rec = None
        self._gathers, self._slab_work, self._g_cat, self._campos, self._poses = [], None, None, None, None
        self.num_collectives = 0

    def start_factors(self, rec):
        """Called by the rasterizer's backward BETWEEN its halves (RasterCall.on_sh_factor): the render backward and the extraction of the
        SH colour factor are enqueued, the projection backward (K8) is not.  The all-gathers of the factors, the camera centres and the
        per-view pose tables are issued here, so they run on the communication stream UNDER K8 (0.2 ms at the headline size) instead of
        behind it.  (Round 4: 0.38 ms of factor gather per step leave the exposed part of the exchange at 8 GPUs.)"""
        self.rec = rec
        g_local = rec.sh_color_grad
        if g_local is None:
            raise RuntimeError("no SH colour-gradient factor: build the rasterizer with factored_sh_grad=True")
        W = world_size()
        if (W == 1 and not force_exchange()) or self._gathers:
            return
        dev, N = g_local.device, g_local.shape[0]
        self._mark("gather_issue")
        self._campos = torch.empty(W, 3, device=dev, dtype=torch.float32)
        if self._use_compact(W, N) and getattr(rec, "radii", None) is None:
            raise RuntimeError("the compacted exchange packs by the call's radii (RasterCall.radii): this record carries none")
        if self._use_compact(W, N):
            cap = int(self.compact_capacity)
            self._overflow_word(dev)
            mine = compact_rows(rec.radii, [g_local], cap)                             # (index, 3 floats) rows of the visible Gaussians
            self._rows_f = torch.empty(W * mine.numel(), dtype=torch.int32, device=dev)
            self._gathers = [dist.all_gather_into_tensor(self._rows_f, mine.view(-1), async_op=True)]
        else:
            self._g_cat = torch.empty(W * N, 3, device=dev, dtype=g_local.dtype)       # ranks concatenated along dim 0
            self._gathers = [dist.all_gather_into_tensor(self._g_cat, g_local.contiguous(), async_op=True)]
        self._gathers.append(dist.all_gather_into_tensor(self._campos, self.campos_local.reshape(1, 3).to(dev, torch.float32).contiguous(),
                                                         async_op=True))
        if self.actor_pose is not None:
            A = self.actor_pose.shape[0]
            self._poses = torch.empty(W * A, self.actor_pose.shape[1], device=dev, dtype=torch.float32)
            self._gathers.append(dist.all_gather_into_tensor(self._poses, self.actor_pose.detach().float().contiguous(), async_op=True))
        self.num_collectives = len(self._gathers)

    def _mark(self, name):
        if self._ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev[name] = e

    def times_ms(self):
        """After the step has been synchronised, with timing=True: {"gather_ms": issue of the factor / camera / pose gathers -> their completion as
        seen by the issuing stream (they travel under K8: mostly hidden), "slab_ms": issue of the slab all-reduce (or its row all-gather) -> its
        completion, "rebuild_ms": the local rebuild of dL/dshs from the gathered factors (+ the row scatters of the compacted form), "exposed_ms":
        end of this rank's own backward kernels (`finish()` is called right behind them) -> end of the exchange: what the step pays for
        communicating}.  Intervals on ONE stream, so they include whatever else that stream did in between."""
        ev = self._ev or {}

        def dt(a, b):
            return ev[a].elapsed_time(ev[b]) if a in ev and b in ev else None
        return {"gather_ms": dt("gather_issue", "gather_done"), "slab_ms": dt("slab_issue", "slab_done"), "rebuild_ms": dt("gather_done", "rebuild_done"),
                "exposed_ms": dt("finish_begin", "finish_end")}

    def _overflow_word(self, dev):
        """The device word the row kernels raise on a dropped row: the caller's, or one of this exchange's own (the kernels never get a null)."""
        if self.overflow is None:
            self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        return self.overflow

    def overflowed(self):
        """True when a view held more visible Gaussians than `compact_capacity` rows (this step's gradients are incomplete on every rank: raise
        the capacity, or run the next step with compact=False).  One device-to-host copy; False for the dense form."""
        return self.overflow is not None and bool(int(self.overflow.reshape(-1)[0].item()) & 1)

    def _slab_rows_allowed(self, rec):
        """Whether the slab may travel as rows of the visible Gaussians (see the class docstring's precondition)."""
        if _in_backward():
            return slab_is_exclusive(rec)
        if not (self.slab_pure and slab_holds_leaf_grads(rec)):
            return False
        if os.environ.get("EMD_DP_DEBUG", "") not in ("", "0"):
            sl, N = rec.grad_slab, rec.grad_slab.numel() // 11
            hidden = (rec.radii.reshape(-1) <= 0)
            rows = torch.cat([sl[:3 * N].view(N, 3), sl[3 * N:6 * N].view(N, 3), sl[6 * N:10 * N].view(N, 4), sl[10 * N:].view(N, 1)], dim=1)
            if bool((rows[hidden] != 0).any()):
                raise RuntimeError("GradientExchange(slab_pure=True): the gradient slab holds non-zero rows on Gaussians this view does not see "
                                   "(another loss term reaches the leaves): the compacted slab would drop them")
        return True

    def _use_compact(self, W, N):
        if self.compact is False or self.compact_capacity is None:
            if self.compact:
                raise ValueError("GradientExchange(compact=True) needs compact_capacity (rows per view)")
            return False
        return True if self.compact else compact_pays(W, N, self.compact_capacity)

    def start(self, rec):
        """Called by the rasterizer's backward (RasterCall.on_backward) as soon as K8 has been enqueued: the gathers (unless
        start_factors issued them between the halves of the backward already) and the in-place all-reduce of the gradient slab --
        or, compacted, the all-gather of the slab's rows of the visible Gaussians."""
        self.start_factors(rec)
        W = world_size()
        if W == 1 and not force_exchange():
            return
        self._mark("slab_issue")
        if self._rows_f is not None and self._slab_rows_allowed(rec):
            N = rec.grad_slab.numel() // 11
            sl = rec.grad_slab
            mine = compact_rows(rec.radii, [sl[:3 * N].view(N, 3), sl[3 * N:6 * N].view(N, 3), sl[6 * N:10 * N].view(N, 4), sl[10 * N:].view(N, 1)],
                                int(self.compact_capacity))
            self._rows_s = torch.empty(W * mine.numel(), dtype=torch.int32, device=sl.device)
            self._slab_work = dist.all_gather_into_tensor(self._rows_s, mine.view(-1), async_op=True)
            self.num_collectives += 1
            return
        # called from inside backward() (RasterCall.on_backward): only an exclusive slab may be reduced while autograd runs; called
        # after backward() (a replayed graph): the slab if the leaves' .grad are its views.  Otherwise the four gradients are
        # reduced as leaf `.grad`s in finish().
        if (slab_is_exclusive(rec) if _in_backward() else slab_holds_leaf_grads(rec)):
            gloo = dist.get_backend() == "gloo"
            op = dist.ReduceOp.SUM if (gloo or not self.average) else dist.ReduceOp.AVG
            self._slab_work = dist.all_reduce(rec.grad_slab, op=op, async_op=True)
            self.num_collectives += 1

    def finish(self, sh_param, means3D, sh_degree, other_params=()):
        """Sets `sh_param.grad` (dense, view-averaged) and averages the gradients of `other_params` over the ranks in place.
        `sh_param`: the [N,K,3] SH parameter, or a tuple of parameters that split its coefficient axis -- the reference's
        (`_features_dc` [N,1,3], `_features_rest` [N,15,3]), S3Gaussian/scene/gaussian_model.py:58-59: each receives its slice of the rebuilt
        gradient (the rasterizer is then fed `torch.cat(...).detach()`: the factored backward returns no dL/dshs to chain through the cat)."""
        rec = self.rec
        if rec is None:
            raise RuntimeError("GradientExchange.finish before the backward pass ran (RasterCall.on_backward was not wired)")
        W = world_size()
        g_local = rec.sh_color_grad
        self._mark("finish_begin")
        works, others, small, bucket, bucket_work = [], [], [], None, None
        if W == 1 and not force_exchange():
            g_all, campos, pose = g_local[None], self.campos_local.reshape(1, 3).to(g_local.device), self.actor_pose
        else:
            N = g_local.shape[0]
            slab = rec.grad_slab
            lo = hi = None
            if self._slab_work is not None:
                lo, hi = slab.data_ptr(), slab.data_ptr() + slab.numel() * 4
            sh_parts = tuple(sh_param) if isinstance(sh_param, (tuple, list)) else (sh_param,)
            for p in other_params:
                if any(p is q for q in sh_parts) or p.grad is None:
                    continue
                g = p.grad
                if lo is not None and lo <= g.data_ptr() < hi:
                    continue                     # a view of the slab that is already being reduced (autograd kept it as .grad)
                others.append(g)
            others.sort(key=lambda g: -g.numel())
            gloo = dist.get_backend() == "gloo"
            op = dist.ReduceOp.SUM if (gloo or not self.average) else dist.ReduceOp.AVG
            # the small rest (actor pose tables, temporal tables, head tensors, point embeddings: a dozen tensors, ~3 MB in all) travels
            # as ONE bucket: a collective costs tens of microseconds of latency whatever its size, twelve of them a third of a step
            bucket = None
            if self.bucket_small and len(others) >= 3:
                small = [g for g in others if g.numel() * 4 <= self.bucket_bytes and g.dtype == torch.float32 and g.is_contiguous()]
                if len(small) >= 3:
                    bucket = torch.cat([g.reshape(-1) for g in small])
                    others = [g for g in others if not any(g is s_ for s_ in small)]
            works = [dist.all_reduce(g, op=op, async_op=True) for g in others]
            bucket_work = dist.all_reduce(bucket, op=op, async_op=True) if bucket is not None else None
            self.num_collectives += len(works) + (1 if bucket is not None else 0)
            for w in self._gathers:
                w.wait()
            self._mark("gather_done")
            if self._rows_f is not None:
                # the gathered factor rows back into dense [W, N, 3] columns (zeros where a view does not see the Gaussian)
                g_all = torch.zeros(W, N, 3, device=g_local.device, dtype=torch.float32)
                rf = self._rows_f.view(W, 1 + int(self.compact_capacity), 4)
                for v in range(W):
                    scatter_rows(rf[v], [g_all[v]], add=False, overflow=self._overflow_word(g_local.device))
            else:
                g_all = self._g_cat.view(W, N, 3)
            campos = self._campos
            pose = None if self._poses is None else self._poses.view(W, -1, self._poses.shape[1])
        parts = tuple(sh_param) if isinstance(sh_param, (tuple, list)) else (sh_param,)
        dense = sh_grad_from_factors(means3D, campos, g_all, sh_degree, sum(int(q.shape[1]) for q in parts), self.actor_ids, pose, self.residual_dx,
                                     scale=(1.0 / W) if self.average else 1.0)
        if len(parts) == 1:
            sh_param.grad = dense.view_as(sh_param)
        else:
            k0 = 0
            for q in parts:
                q.grad = dense[:, k0:k0 + q.shape[1]].contiguous()
                k0 += q.shape[1]
        self._mark("rebuild_done")
        if self._slab_work is not None:
            self._slab_work.wait()
            self._mark("slab_done")
            if self._rows_s is not None:
                # every view's rows added into the zeroed slab, in rank order on every rank: the same float additions everywhere
                sl = rec.grad_slab
                Ns = sl.numel() // 11
                sl.zero_()
                dests = [sl[:3 * Ns].view(Ns, 3), sl[3 * Ns:6 * Ns].view(Ns, 3), sl[6 * Ns:10 * Ns].view(Ns, 4), sl[10 * Ns:].view(Ns, 1)]
                rs = self._rows_s.view(W, 1 + int(self.compact_capacity), 12)
                for v in range(W):
                    scatter_rows(rs[v], dests, add=True, scale=(1.0 / W) if self.average else 1.0, overflow=self._overflow_word(sl.device))
        for w in works:
            w.wait()
        if W > 1 or force_exchange():
            if bucket_work is not None:
                bucket_work.wait()
                if self.average and dist.get_backend() == "gloo":
                    bucket.div_(float(W))
                torch._foreach_copy_([g.reshape(-1) for g in small], list(torch.split(bucket, [g.numel() for g in small])))
        if W > 1 and self.average and dist.get_backend() == "gloo":
            for g in others:
                g.div_(float(W))
            if self._slab_work is not None and self._rows_s is None:
                rec.grad_slab.div_(float(W))
        self._mark("finish_end")


def visible_capacity(v_max, margin=1.1, multiple=1024):
    """Rows per view of the compacted exchange for at most `v_max` visible Gaussians (a host number: the maximum over the views a loop is about to
    render, reduced with MAX over the ranks -- every rank must use the same capacity)."""
    c = int(v_max * margin) + 1
    return (c + multiple - 1) // multiple * multiple


def reduce_densification_stats(grad_norm_accum, denom, max_radii2D):
    """Cross-view reduction of the densification statistics (norms taken per view, BEFORE reduction: gaussian_model.py:728-730).
    SUM / SUM / MAX are associative, so each rank accumulates its own views locally (`add_densification_stats`, one launch per
    step) and this reduction runs once right before a densification event (every `densification_interval` = 100 steps,
    train.py:410), not every step: afterwards all ranks hold the statistics of all views and densify identically."""
    if world_size() == 1:
        return
    dist.all_reduce(grad_norm_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)


def add_densification_stats(viewspace_points_grad, radii, xyz_gradient_accum, denom, max_radii2D):
    """In-place update of the three per-Gaussian densification statistics for one view in ONE launch and without the
    boolean-mask indexing of the reference (gaussian_model.py:728-730, train.py:403-406), which costs a host sync per step:
    where radii > 0:  xyz_gradient_accum += |grad.xy|, denom += 1, max_radii2D = max(max_radii2D, radii)."""
    import ctypes as C
    from . import _lib as L
    if radii.device.type != "cuda":
        raise L.EmdError("add_densification_stats needs tensors on a ROCm device; there is no CPU path")
    g = viewspace_points_grad.contiguous().float()
    r = radii.contiguous().to(torch.int32)
    for t in (xyz_gradient_accum, denom, max_radii2D):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32 and t.numel() == r.numel())
    L.check(L.load().emd_densification_stats(r.numel(), r.data_ptr(), g.data_ptr(), L.ptr(xyz_gradient_accum), L.ptr(denom),
                                             L.ptr(max_radii2D), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "emd_densification_stats")


def densification_stats(viewspace_points_grad, radii):
    """Per-view statistics exactly as S3Gaussian/scene/gaussian_model.py:728-730 and train.py:406 compute them."""
    vis = radii > 0
    g = torch.zeros(radii.shape[0], 1, device=radii.device)
    g[vis] = torch.norm(viewspace_points_grad[vis, :2], dim=-1, keepdim=True)
    return g, vis.float()[:, None], radii.float()
