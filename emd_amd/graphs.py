"""Training steps replayed from hipGraphs: the host side of a step -- ~170 launches issued from Python and autograd, 15-19 ms for
S3Gaussian's fine stage against 17.6 ms of GPU work -- shrinks to one `hipGraphLaunch` (0.06 ms).

`StepGraphs` records ONE graph per key (a frame of the clip, a camera of the rig: whatever changes the HOST constants of a step -- camera
matrices, frame time, sky rays) into one shared memory pool and replays them by key -- or, with `inputs=StepInputs(...)` (round 4), ONE
graph for ALL keys: the per-view constants live in device tables, the graph's first node copies the selected row to the addresses the step
reads, and a density-control event (every 100 iterations until 15 000, S3Gaussian/arguments/gaussian_options.py:112-117) costs one
re-capture instead of one per frame of the clip.  What the reference does every iteration
(S3Gaussian/train.py:203-229 + 366-430: render, losses, backward, densification statistics, optimizer.step) can all be inside the recorded
function.  The recorder takes care of what a capture must not contain or depend on (each found the hard way):

  * autograd graphs of earlier eager steps still referenced (their AccumulateGrad nodes are bound to the eager stream): collected first;
  * a refresh of a HexPlaneField's cached visiting orders inside a capture (the new tensors would belong to that graph's pool and be recycled
    by the next graph): `freeze=[field, ...]` suspends the refresh while recording;
  * lazily formed host constants that need a device-to-host copy (the sky pass's per-camera ray parameters): `prime(key)` runs before any
    capture, eagerly and without gradients, for every key;
  * optimiser state created by the first step: `emd_amd.optim.Adam(..., capturable=True)`'s state is allocated before recording.

The gradients a recorded step leaves in `param.grad` live in the graphs' pool: they are valid after the replay of THAT key until the next
replay, which is what an optimiser step inside the recorded function (or right behind the replay) needs.  Rasterizer calls inside must be
built with `RasterOptions(no_sync=True)` (they raise otherwise)."""
import ctypes as C
import gc
import types

import torch

from . import _lib as L


def select_step_inputs(sel, table, out_row, frames=None, frame_out=None, t_out=None, num_frames=1, k_sched=None, k_fine_out=None,
                       status=None, status_log=None, prev_sel=None, next_sel=None):
    """emd_select_step_inputs: everything a replayed step reads at fixed device addresses, written by ONE launch (row `sel[0]` of `table`
    -> `out_row`, its frame index / frame time, the coarse-to-fine level of the row's step, the status words of the step before into a
    log, and -- with `next_sel` -- the row of the next replay)."""
    a = L.EmdStepSelect()
    a.sel, a.rows, a.row_floats = sel.data_ptr(), table.shape[0], table.shape[1]
    a.table, a.out_row = table.data_ptr(), out_row.data_ptr()
    a.frames, a.frame_out, a.t_out, a.num_frames = L.ptr(frames), L.ptr(frame_out), L.ptr(t_out), int(num_frames)
    a.k_min, a.k_max, a.k_until = k_sched if k_sched is not None else (1, 1, 1)
    a.steps, a.k_fine_out = None, L.ptr(k_fine_out)
    a.status, a.status_log, a.prev_sel, a.next_sel = L.ptr(status), L.ptr(status_log), L.ptr(prev_sel), L.ptr(next_sel)
    L.check(L.load().emd_select_step_inputs(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emd_select_step_inputs")


class StepInputs:
    """Everything that distinguishes the VIEWS of a training loop, in device-resident tables -- one row per view -- and the fixed device
    addresses a recorded step reads them from.  `StepGraphs(step_fn, keys, inputs=...)` then records ONE graph for all views: a replay is
    one small host-to-device write of the row index, the graph's first node (emd_select_step_inputs) copies that row, and nothing of a
    camera is baked into the capture.  A density-control event costs one re-capture, not one per frame of the clip.

    A row holds: bg[3], viewmatrix[16], projmatrix[16], campos[3], tanfovx, tanfovy (the 40-float settings block of the rasterizer,
    EmdFwdArgs.settings_dev + EMD_FLAG_SDEV_TANFOV), the sky pass's ray constants Kinv[9], R[9], T[3] (EmdSkyArgs.camera_dev; zeros for a
    camera without intrinsics), the view's time, and `extra` caller-defined floats.  Per row as well: the frame index (int32; the row of
    the per-frame actor pose tables).

    `camera` is the camera object of the CURRENT row for emd_amd.model.render / raster_settings_for / the sky model: every per-view field
    is a view of the selected row.  Image size is common to all views (it shapes the launch grids).  What stays a host constant of the
    capture: the image size, `cam_no` (the index of S3Gaussian's per-camera time offset: views that differ in it need a graph each) and the
    iteration-dependent coarse-to-fine level of the deformation network (re-record when it changes, as at a density-control event)."""
    SETTINGS, SKY, TIME = 40, 21, 1

    def __init__(self, cameras, bg, frames=None, times=None, extra=None, device="cuda"):
        from .sky import sky_ray_constants
        dev = torch.device(device)
        cams = list(cameras)
        if not cams:
            raise ValueError("StepInputs needs at least one camera")
        H, W = int(cams[0].image_height), int(cams[0].image_width)
        if any((int(c.image_height), int(c.image_width)) != (H, W) for c in cams):
            raise ValueError("the views of one recorded step share the image size")
        # host constants of the capture that model.render reads from the camera object (`cam_no` indexes S3Gaussian's per-camera
        # time_offset parameter, `time_diff` scales it): one value for all rows, or the recorded step would silently apply camera 0's
        cam_nos = {int(getattr(c, "cam_no", 0)) for c in cams}
        time_diffs = {float(getattr(c, "time_diff", 0.0)) for c in cams}
        if len(cam_nos) > 1 or len(time_diffs) > 1:
            raise ValueError(f"the views of one recorded step share cam_no and time_diff (got cam_no {sorted(cam_nos)}, time_diff "
                             f"{sorted(time_diffs)}): group the views by camera and record one StepGraphs per group")
        rows = []
        bg = torch.as_tensor(bg, dtype=torch.float32).reshape(-1).cpu()
        for i, c in enumerate(cams):
            K = getattr(c, "intrinsic", None)
            rays = sky_ray_constants(K.cpu(), c.world_view_transform.cpu()) if K is not None else torch.zeros(21)
            t = float(times[i]) if times is not None else float(getattr(c, "time", 0.0))
            x = torch.as_tensor(extra[i], dtype=torch.float32).reshape(-1) if extra is not None else torch.zeros(0)
            rows.append(torch.cat([bg, c.world_view_transform.reshape(-1).float().cpu(), c.full_proj_transform.reshape(-1).float().cpu(),
                                   c.camera_center.reshape(-1).float().cpu(), torch.tensor([float(c.tanfovx), float(c.tanfovy)]), rays,
                                   torch.tensor([t]), x]))
        self.table = torch.stack(rows).to(dev).contiguous()
        self.rows, self.height, self.width = len(cams), H, W
        self.frames = torch.tensor([int(f) for f in (frames if frames is not None else [0] * len(cams))], dtype=torch.int32, device=dev)
        self.sel = torch.zeros(1, dtype=torch.int64, device=dev)
        self.row = torch.zeros(self.table.shape[1], device=dev)
        self.frame = torch.zeros(1, dtype=torch.int32, device=dev)
        r = self.row
        self.bg, self.time, self.extra = r[0:3], r[61:62], r[62:]
        self.camera = types.SimpleNamespace(image_height=H, image_width=W, tanfovx=r[38:39], tanfovy=r[39:40], world_view_transform=r[3:19].view(4, 4),
                                            full_proj_transform=r[19:35].view(4, 4), camera_center=r[35:38], sky_rays=r[40:61], time=r[61:62],
                                            cam_no=cam_nos.pop(), time_diff=time_diffs.pop())

    def launch_select(self):
        """The launch that copies row sel[0] to the fixed addresses (the first node of a recorded step; also usable eagerly)."""
        select_step_inputs(self.sel, self.table, self.row, self.frames, self.frame)

    def select(self, row):
        """Name the row of the next replay / eager step (one tiny fill launch with the index as its argument: no host buffer that a host
        running ahead of the GPU could overwrite before it is read, no synchronisation)."""
        if not 0 <= int(row) < self.rows:
            raise IndexError(f"row {row} of {self.rows}")
        self.sel.fill_(int(row))


class StepGraphs:
    def __init__(self, step_fn, keys, prime=None, freeze=(), optimizers=(), warmup=1, inputs=None, segmented=False):
        """step_fn(key) -> None records one training step.  `warmup` eager calls of the FIRST key run before recording, on the capture
        stream (lazy allocations: workspaces, caches); `prime(key)` runs for every key, without gradients (per-key host constants).
        `inputs` (a StepInputs whose rows are the keys, in order): ONE graph for all keys -- step_fn(inputs) reads the view from
        `inputs.camera / .bg / .frame / .time` (device memory the graph's first node fills from the selected row) and must not bake anything
        else of a view into its launches; `replay(key)` selects the row and replays the one graph.
        `segmented`: the recorded step may call `self.cut` -- from any thread, typically as `RasterCall.on_sh_factor`, which the rasterizer's
        backward calls between its halves from autograd's device thread -- to end the current graph and begin the next: a step then is a
        LIST of graphs, and `replay(key, between=fn)` calls `fn(i)` on the host between segment i and i + 1.  That is where a view-parallel
        loop issues the collectives that cannot be captured (RCCL): `between=lambda i: exchange.start_factors(rec)` puts the SH-factor
        gathers under the projection backward of a REPLAYED step.  Stream capture then runs in relaxed mode (the only
        mode in which begin and end may come from different threads)."""
        self.keys = list(keys)
        self.segmented = bool(segmented)
        self._recording = None
        self.inputs = inputs
        if inputs is not None:
            if len(self.keys) != inputs.rows:
                raise ValueError(f"{len(self.keys)} keys for {inputs.rows} rows of inputs")
            self._row_of = {k: i for i, k in enumerate(self.keys)}
            user_fn = step_fn

            def step_fn(_key):                       # noqa: E306  (the recorded unit: select launch + the step on the selected row)
                inputs.launch_select()
                user_fn(inputs)
            self._record_keys = [self.keys[0]]
        if not self.keys:
            raise ValueError("StepGraphs needs at least one key")
        if not torch.cuda.is_available():
            raise RuntimeError("StepGraphs needs a ROCm device; there is no CPU path")
        if prime is not None:
            with torch.no_grad():
                for k in self.keys:
                    prime(k)
        for o in optimizers:
            if hasattr(o, "_capturable_state"):
                if not getattr(o, "capturable", False):
                    raise ValueError("an optimiser step inside a recorded step needs emd_amd.optim.Adam(..., capturable=True)")
                o._capturable_state()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # The warm-up runs ON THE CAPTURE STREAM: what a step keeps per stream (the rasterizer's kept-clean backward workspace) is then
        # formed eagerly, before recording, and shared by every graph -- a buffer first allocated inside one capture would be cleared by
        # that graph's replay only.  The warm-up steps are real training steps, so only `warmup` of them run, on the first key; state that
        # is formed lazily PER KEY (host constants that need a device-to-host copy) belongs into `prime(key)`.
        if inputs is not None:
            inputs.select(0)
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 0)):
                step_fn(self.keys[0])
        torch.cuda.synchronize()
        saved = [(f, f.reorder_every) for f in freeze]
        self.graphs, self.pool = {}, None
        self._frozen = []
        try:
            for f in freeze:
                if getattr(f, "_order_cache", None) is None and hasattr(f, "_visiting_order"):
                    raise RuntimeError("a frozen HexPlaneField has no visiting orders yet: they would be built inside the first capture and their "
                                       "memory recycled by the next one -- run one eager step first (warmup >= 1)")
                if hasattr(f, "_aabb_host"):
                    f._aabb_host()                     # (the host copy of the box: a device-to-host read when it is formed)
                f.reorder_every = 1 << 60
            gc.collect()
            with torch.cuda.stream(side):
                for k in (self.keys if inputs is None else self._record_keys):
                    g = torch.cuda.CUDAGraph()
                    if not self.segmented:
                        with torch.cuda.graph(g, pool=self.pool, stream=side):
                            step_fn(k)
                        self.pool = g.pool()
                        self.graphs[k] = g
                        continue
                    torch.cuda.synchronize()
                    self._recording = [g]
                    if self.pool is None:
                        self.pool = torch.cuda.graph_pool_handle()        # (a graph's own pool() is only known once its capture has ended)
                    g.capture_begin(pool=self.pool, capture_error_mode="relaxed")
                    try:
                        step_fn(k)
                    except BaseException:
                        segs, self._recording = self._recording, None
                        try:
                            segs[-1].capture_end()             # leave the stream out of capture mode; the step's own error is the one to report
                        except Exception:
                            pass
                        raise
                    segs, self._recording = self._recording, None
                    segs[-1].capture_end()
                    self.graphs[k] = segs
        except BaseException:
            # a failed recording must not leave the fields frozen: the caller gets no object to release() them with
            for f, every in saved:
                f.reorder_every = every
            self.graphs.clear()
            self.pool = None
            raise
        # the recorded graphs hold the orders they were captured with: the fields keep them (a refresh would be harmless for the graphs,
        # which read the old tensors, but would cost the eager path its cache), so the interval stays suspended while the graphs live
        self._frozen = saved
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def cut(self, *_):
        """Inside a recording of a `segmented` StepGraphs: ends the current graph and begins the next, on the capture stream and in the same
        memory pool (callable from any thread; the arguments a hook is called with are ignored).  Outside a recording: nothing."""
        segs = self._recording
        if segs is None:
            return
        segs[-1].capture_end()
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self.pool, capture_error_mode="relaxed")
        segs.append(g)

    def segments(self, key=None):
        """Number of graphs the step of `key` was recorded as (1 unless `segmented` and the step called `cut`)."""
        g = self.graphs[self._record_keys[0] if self.inputs is not None else (self.keys[0] if key is None else key)]
        return len(g) if isinstance(g, list) else 1

    def replay(self, key, between=None):
        if self.inputs is not None:
            self.inputs.select(self._row_of[key])
            g = self.graphs[self._record_keys[0]]
        else:
            g = self.graphs[key]
        if not isinstance(g, list):
            g.replay()
            return
        for i, seg in enumerate(g):
            seg.replay()
            if between is not None and i + 1 < len(g):
                between(i)

    def release(self):
        """Drop the graphs (and their pool) and give the frozen fields their refresh interval back."""
        self.graphs.clear()
        self.pool = None
        for f, every in self._frozen:
            f.reorder_every = every
        self._frozen = []
