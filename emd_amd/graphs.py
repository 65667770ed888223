"""Training steps replayed from hipGraphs: the host side of a step -- ~170 launches issued from Python and autograd, 15-19 ms for
S3Gaussian's fine stage against 17.6 ms of GPU work -- shrinks to one `hipGraphLaunch` (0.06 ms).

`StepGraphs` records ONE graph per key (a frame of the clip, a camera of the rig: whatever changes the HOST constants of a step -- camera
matrices, frame time, sky rays) into one shared memory pool and replays them by key.  What the reference does every iteration
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
import gc

import torch


class StepGraphs:
    def __init__(self, step_fn, keys, prime=None, freeze=(), optimizers=(), warmup=1):
        """step_fn(key) -> None records one training step.  `warmup` eager calls per key... of the FIRST key run before recording (lazy
        allocations: workspaces, caches)."""
        self.keys = list(keys)
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
        for _ in range(max(int(warmup), 0)):
            step_fn(self.keys[0])
        torch.cuda.synchronize()
        saved = [(f, f.reorder_every) for f in freeze]
        for f in freeze:
            if getattr(f, "_order_cache", None) is None and hasattr(f, "_visiting_order"):
                raise RuntimeError("a frozen HexPlaneField has no visiting orders yet: they would be built inside the first capture and their "
                                   "memory recycled by the next one -- run one eager step first (warmup >= 1)")
            if hasattr(f, "_aabb_host"):
                f._aabb_host()                     # (the host copy of the box: a device-to-host read when it is formed)
            f.reorder_every = 1 << 60
        gc.collect()
        self.graphs, self.pool = {}, None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for k in self.keys:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=self.pool, stream=side):
                        step_fn(k)
                    self.pool = g.pool()
                    self.graphs[k] = g
        finally:
            # the recorded graphs hold the orders they were captured with: the fields keep them (a refresh would be harmless for the graphs,
            # which read the old tensors, but would cost the eager path its cache), so the interval stays suspended while the graphs live
            self._frozen = saved
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def replay(self, key):
        self.graphs[key].replay()

    def release(self):
        """Drop the graphs (and their pool) and give the frozen fields their refresh interval back."""
        self.graphs.clear()
        self.pool = None
        for f, every in self._frozen:
            f.reorder_every = every
        self._frozen = []
