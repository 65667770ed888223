"""EMD explicit motion, host side: per-actor pose table + learned track offsets, and the per-Gaussian transform
through the HIP kernels (`emd_motion_forward/backward`, or fused into the projection kernel via
`GaussianRasterizer(..., actor_ids=, actor_pose=)`).

Mirrors, with the same names and argument meaning (behaviour restated, not copied):
  RigidNodes.transform_means / transform_quats / get_pts_valid_mask   OmniRe/models/nodes/rigid.py:42-46,478-568
  embedding_track_trans_offset / embedding_track_rot_offset           OmniRe/models/nodes/rigid.py:203-246
  get_temporal_embed / query_time / int_lininterp                     OmniRe/models/nodes/rigid.py:147-192
  quat_mult / interpolate_quats / quat_act                            OmniRe/models/gaussians/basics.py:53-110, vanilla.py:145-146
  DeformableNodes residual add                                        OmniRe/models/nodes/deformable.py:57-69

What changes versus the reference is WHERE the work runs, not what is computed: the reference loops over
instances in Python (rigid.py:520-530,550-562; ~15 tiny launches per instance, twice); here the per-actor
quantities are batched over actors in a handful of torch ops and packed into one [A,12] table, and the per-point
gather + rigid transform + quaternion composition + validity mask is one HIP kernel (or no kernel at all when fused).
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib as L


def quat_act(q):
    return F.normalize(q, dim=-1)


def quat_mult(q1, q2):
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)


def interpolate_quats(q1, q2, fraction=0.5):
    """Slerp with the reference's branches (basics.py:53-81): sign flip for dot < 0, lerp when dot > 0.9995."""
    q1 = q1 / torch.norm(q1, dim=-1, keepdim=True)
    q2 = q2 / torch.norm(q2, dim=-1, keepdim=True)
    dot = (q1 * q2).sum(dim=-1).clamp(-1, 1)
    neg = dot < 0
    q2 = torch.where(neg[..., None], -q2, q2)
    dot = torch.where(neg, -dot, dot)
    similar = dot > 0.9995
    lerp = q1 + fraction * (q2 - q1)
    theta_0 = torch.acos(dot)
    theta = theta_0 * fraction
    s2 = torch.sin(theta) / torch.sin(theta_0)
    s1 = torch.cos(theta) - dot * s2
    slerp = s1[..., None] * q1 + s2[..., None] * q2
    return torch.where(similar[..., None], lerp, slerp)


class DeviceStep:
    """Training-step quantities that live on the DEVICE, for a step replayed from a hipGraph: `k_fine` (int32 [1], the
    coarse-to-fine row count of the step, rigid.py:194-201) and `t` (float32 [1], the normalised frame time), as
    `emd_select_step_inputs` writes them; or just `step` (an integer tensor [1]), from which k_fine is derived with torch ops.
    Pass it wherever a python `step` / `iteration` is accepted by the track heads."""
    __slots__ = ("step", "k_fine", "t")

    def __init__(self, step=None, k_fine=None, t=None):
        self.step, self.k_fine, self.t = step, k_fine, t


class TrackOffsetHeads(torch.nn.Module):
    """Per-actor learned track offsets, batched over actors (rigid.py:108-122,150-246).

    Delta t = W_c h_c + W_f h_f,   theta_{c,f} = w_{c,f} . h_{c,f},   Delta q = (cos th_c,0,0,sin th_c) (x) (cos th_f,0,0,sin th_f)
    with h = cat[TE_k(t) from the actor's temporal table, mean of the actor's Gaussian embeddings];
    k = 30 for the coarse head, int_lininterp(step, 30, 150, 25000) for the fine head.
    """

    def __init__(self, num_actors, temporal_embedding_dim=32, gaussian_embedding_dim=4, max_embeddings=150,
                 min_embeddings=30, c2f_temporal_iter=25000):
        super().__init__()
        self.fdim, self.edim = temporal_embedding_dim, gaussian_embedding_dim
        self.max_embeddings, self.min_embeddings, self.c2f_temporal_iter = max_embeddings, min_embeddings, c2f_temporal_iter
        self.weight = torch.nn.Parameter(torch.randn(num_actors, max_embeddings, temporal_embedding_dim) * 0.01 / temporal_embedding_dim ** 0.5)
        d = temporal_embedding_dim + gaussian_embedding_dim
        mk = lambda o: torch.nn.Linear(d, o)
        self.track_rot_c, self.track_rot_f, self.track_trans_c, self.track_trans_f = mk(1), mk(1), mk(3), mk(3)
        for lin in (self.track_rot_c, self.track_rot_f, self.track_trans_c, self.track_trans_f):  # zero init, rigid.py:113-122
            torch.nn.init.zeros_(lin.weight)
            torch.nn.init.zeros_(lin.bias)

    def int_lininterp(self, t, init_val, final_val, until):
        return int(init_val + (final_val - init_val) * min(max(t, 0), until) / until)

    def invalidate(self):
        """Forget the cached per-actor point counts / segment starts (call after any change of the point set: densification,
        pruning, in-place edits of point_ids).  `GaussianModel`-style stores call it from their restructuring code; the cache key
        below also carries the tensor's version counter, so in-place edits are seen without it."""
        self._ids_key = None

    def _prepare_ids(self, point_ids):
        A = self.weight.shape[0]
        # (the allocator may hand a NEW id tensor the address and size of the old one after a densify + prune: the key carries the
        #  object identity and its in-place version counter as well)
        key = (id(point_ids), point_ids._version, point_ids.data_ptr(), point_ids.numel(), point_ids.dtype)
        if getattr(self, "_ids_key", None) != key:             # ids as int32 + points per actor: constant between densifications
            ids32 = point_ids.to(torch.int32).contiguous()
            cnt = torch.zeros(A, device=ids32.device).index_add_(0, point_ids.long().clamp_min(0), (point_ids >= 0).float())
            # the reference stores an actor's points contiguously: then segment starts replace the atomics of the embedding sums
            # (checked once per point set, i.e. once per densification: the only host read of this module)
            seg = None
            if ids32.numel() > 0 and bool((ids32[1:] >= ids32[:-1]).all()) and int(ids32[0]) >= 0:
                seg = torch.searchsorted(ids32, torch.arange(A + 1, device=ids32.device, dtype=torch.int32)).to(torch.int32).contiguous()
            self._ids_key, self._ids_ref, self._ids32, self._cnt, self._seg = key, point_ids, ids32, cnt, seg
        return self._ids32, self._cnt, self._seg

    def _time_and_k(self, frame, num_frames, step):
        # (frame - start_frame) / (end_frame - start_frame), rigid.py:204,241; a device frame index keeps the time on the device
        if isinstance(step, DeviceStep) and step.t is not None:
            t = step.t
        elif isinstance(frame, torch.Tensor):
            t = frame.to(torch.float32) / float(max(num_frames - 1, 1))
        else:
            t = float(frame) / float(max(num_frames - 1, 1))
        if isinstance(step, DeviceStep):
            k_f = step.k_fine if step.k_fine is not None else self.k_fine_from_step(step.step)
        elif isinstance(step, torch.Tensor):        # device step counter (hipGraph replay): k_fine is evaluated on the device
            k_f = self.k_fine_from_step(step)
        else:
            k_f = self.int_lininterp(step, self.min_embeddings, self.max_embeddings, self.c2f_temporal_iter)
        return t, k_f

    def k_fine_from_step(self, step_dev):
        """int_lininterp(step, min, max, until) of a DEVICE step counter -> int32 device tensor [1] (same truncation as the host form)."""
        s = step_dev.to(torch.float64).clamp(0, self.c2f_temporal_iter)
        return (self.min_embeddings + (self.max_embeddings - self.min_embeddings) * s / self.c2f_temporal_iter).to(torch.int32).reshape(1)

    def forward(self, frame, num_frames, embeddings, point_ids, step):
        """-> (track_trans [A,3], track_rot [A,4]).  embeddings [n,4] of the actor Gaussians; point_ids [n] their actor (device
        tensors).  One HIP launch (+ the per-actor embedding sums) forward and one backward for all actors and both levels
        (`emd_track_heads_forward/backward`); no host synchronisation: the frame time is a by-value scalar."""
        if self.weight.device.type != "cuda":
            raise L.EmdError("TrackOffsetHeads needs its parameters on a ROCm device; there is no CPU path "
                             "(the checker's restatement is oracle/torch_ref.track_offsets)")
        ids32, cnt, seg = self._prepare_ids(point_ids)
        t, k_f = self._time_and_k(frame, num_frames, step)
        return _TrackHeads.apply(self.weight, embeddings, self.track_trans_c.weight, self.track_trans_c.bias, self.track_trans_f.weight,
                                 self.track_trans_f.bias, self.track_rot_c.weight, self.track_rot_c.bias, self.track_rot_f.weight,
                                 self.track_rot_f.bias, ids32, cnt, seg, t, self.min_embeddings, k_f)

    def pose_table(self, instances_quats, instances_trans, instances_fv, frame, embeddings, point_ids, step):
        """The frame's [A,12] pose table WITH the learned offsets, the whole per-actor chain (embedding sums -> heads -> pose row) in
        ONE launch each way (`emd_tracked_pose_forward/backward`) when an actor's points are contiguous (the reference's layout) and
        the embedding is at most 8 wide; otherwise the three-launch path `actor_pose_table(..., *self(...))`.  Same values either way."""
        if self.weight.device.type != "cuda":
            raise L.EmdError("TrackOffsetHeads needs its parameters on a ROCm device; there is no CPU path")
        ids32, cnt, seg = self._prepare_ids(point_ids)
        num_frames = instances_quats.shape[0]
        if seg is None or embeddings.shape[1] > 8 or embeddings.shape[1] == 0:
            tt, trq = self(frame, num_frames, embeddings, point_ids, step)
            return actor_pose_table(instances_quats, instances_trans, instances_fv, frame, tt, trq)
        t, k_f = self._time_and_k(frame, num_frames, step)
        return _TrackedPose.apply(self.weight, embeddings, self.track_trans_c.weight, self.track_trans_c.bias, self.track_trans_f.weight,
                                  self.track_trans_f.bias, self.track_rot_c.weight, self.track_rot_c.bias, self.track_rot_f.weight,
                                  self.track_rot_f.bias, instances_quats, instances_trans, instances_fv, ids32, cnt, seg, t,
                                  self.min_embeddings, k_f, frame)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _TrackHeads(torch.autograd.Function):
    @staticmethod
    def _args(weight, emb, heads, ids32, cnt, seg, t, k_c, k_f, emb_sum, trans, rot):
        a = L.EmdTrackArgs()
        a.num_actors, a.rows, a.dim = weight.shape
        a.embed_dim, a.num_points = (emb.shape[1], emb.shape[0]) if emb is not None else (0, 0)
        if isinstance(k_f, torch.Tensor):
            a.k_coarse, a.k_fine, a.k_fine_dev = int(k_c), 1, k_f.data_ptr()
        else:
            a.k_coarse, a.k_fine, a.k_fine_dev = int(k_c), int(k_f), None
        if isinstance(t, torch.Tensor):
            a.t, a.t_dev = 0.0, t.data_ptr()
        else:
            a.t, a.t_dev = float(t), None
        a.weight, a.embeddings, a.point_ids, a.count = weight.data_ptr(), L.ptr(emb), ids32.data_ptr(), cnt.data_ptr()
        a.segment_start = L.ptr(seg)
        for h in range(4):
            a.head_w[h], a.head_b[h] = heads[2 * h].data_ptr(), heads[2 * h + 1].data_ptr()
        a.emb_sum, a.trans, a.rot = emb_sum.data_ptr(), L.ptr(trans), L.ptr(rot)
        return a

    @staticmethod
    def forward(ctx, weight, emb, wtc, btc, wtf, btf, wrc, brc, wrf, brf, ids32, cnt, seg, t, k_c, k_f):
        dev = weight.device
        c = lambda x: x.detach().contiguous().float()
        weight_c, emb_c = c(weight), c(emb)
        heads = [c(x) for x in (wtc, btc, wtf, btf, wrc, brc, wrf, brf)]
        A, E = weight_c.shape[0], emb_c.shape[1]
        if weight_c.shape[2] + E > 64 or any(h.shape[-1] != weight_c.shape[2] + E for h in heads[0::2]):
            raise ValueError("track heads: temporal dim + embedding dim must be <= 64 and match the head widths")
        emb_sum = (torch.empty if seg is not None else torch.zeros)(A, max(E, 1), device=dev)
        trans, rot = torch.empty(A, 3, device=dev), torch.empty(A, 4, device=dev)
        a = _TrackHeads._args(weight_c, emb_c, heads, ids32, cnt, seg, t, k_c, k_f, emb_sum, trans, rot)
        L.check(L.load().emd_track_heads_forward(C.byref(a), _stream()), "emd_track_heads_forward")
        ctx.save_for_backward(weight_c, emb_c, ids32, cnt, emb_sum, *heads)
        ctx.scal, ctx.seg = (t, k_c, k_f), seg
        return trans, rot

    @staticmethod
    def backward(ctx, g_trans, g_rot):
        weight, emb, ids32, cnt, emb_sum, *heads = ctx.saved_tensors
        dev = weight.device
        A, E = weight.shape[0], emb.shape[1]
        z = lambda t_: torch.empty_like(t_)
        g_trans = torch.zeros(A, 3, device=dev) if g_trans is None else g_trans.contiguous().float()
        g_rot = torch.zeros(A, 4, device=dev) if g_rot is None else g_rot.contiguous().float()
        # everything the kernel accumulates into comes from ONE zero-filled allocation
        sizes = [weight.numel()] + [h.numel() for h in heads]
        flat = torch.zeros(sum(sizes), device=dev)
        parts = torch.split(flat, sizes)
        d_weight, d_heads = parts[0].view_as(weight), [p.view_as(h) for p, h in zip(parts[1:], heads)]
        d_emb, d_mean = z(emb), torch.empty(A, max(E, 1), device=dev)
        a = _TrackHeads._args(weight, emb, heads, ids32, cnt, ctx.seg, *ctx.scal, emb_sum, None, None)
        g = L.EmdTrackGrads()
        g.g_trans, g.g_rot, g.d_weight, g.d_embeddings, g.d_mean = g_trans.data_ptr(), g_rot.data_ptr(), d_weight.data_ptr(), d_emb.data_ptr(), d_mean.data_ptr()
        for h in range(4):
            g.d_head_w[h], g.d_head_b[h] = d_heads[2 * h].data_ptr(), d_heads[2 * h + 1].data_ptr()
        L.check(L.load().emd_track_heads_backward(C.byref(a), C.byref(g), _stream()), "emd_track_heads_backward")
        return (d_weight, d_emb, *d_heads, None, None, None, None, None, None)


class _TrackedPose(torch.autograd.Function):
    """embedding sums -> track heads -> pose row: one launch forward, one backward (csrc/embed.hip: k_tracked_pose)."""

    @staticmethod
    def forward(ctx, weight, emb, wtc, btc, wtf, btf, wrc, brc, wrf, brf, q_all, t_all, fv, ids32, cnt, seg, t, k_c, k_f, frame):
        dev = weight.device
        c = lambda x: x.detach().contiguous().float()
        weight_c, emb_c, q_c, t_c = c(weight), c(emb), c(q_all), c(t_all)
        heads = [c(x) for x in (wtc, btc, wtf, btf, wrc, brc, wrf, brf)]
        A, E = weight_c.shape[0], emb_c.shape[1]
        if weight_c.shape[2] + E > 64 or any(h.shape[-1] != weight_c.shape[2] + E for h in heads[0::2]):
            raise ValueError("track heads: temporal dim + embedding dim must be <= 64 and match the head widths")
        if q_c.shape[1] != A or t_c.shape[:2] != q_c.shape[:2]:
            raise ValueError("pose tables must be [F, A, 4] / [F, A, 3] with A = number of temporal tables")
        on_dev = isinstance(frame, torch.Tensor)
        if on_dev and (frame.dtype != torch.int32 or frame.device != dev):
            raise ValueError("a device frame index must be an int32 tensor on the tables' device")
        valid = None if fv is None else fv.contiguous().view(torch.uint8)
        emb_sum = torch.empty(A, max(E, 1), device=dev)
        pose = torch.empty(A, L.ACTOR_STRIDE, device=dev, dtype=torch.float32)
        # the shared head gradients of the backward are accumulated into this buffer, which the forward launch clears (no fill launch)
        head_acc = torch.empty(sum(h.numel() for h in heads), device=dev) if any(ctx.needs_input_grad[:10]) else None
        p = _TrackedPose._args(weight_c, emb_c, heads, ids32, cnt, seg, t, k_c, k_f, emb_sum, q_c, t_c, valid, frame, pose, head_acc)
        L.check(L.load().emd_tracked_pose_forward(C.byref(p), _stream()), "emd_tracked_pose_forward")
        ctx.save_for_backward(weight_c, emb_c, ids32, cnt, seg, emb_sum, q_c, t_c, *heads)
        ctx.scal, ctx.valid, ctx.head_acc, ctx.acc_used = (t, k_c, k_f, frame), valid, head_acc, False
        ctx.shapes = (q_all.shape, t_all.shape)
        return pose

    @staticmethod
    def _args(weight, emb, heads, ids32, cnt, seg, t, k_c, k_f, emb_sum, q_all, t_all, valid, frame, pose, head_acc=None):
        p = L.EmdTrackedPoseArgs()
        p.track = _TrackHeads._args(weight, emb, heads, ids32, cnt, seg, t, k_c, k_f, emb_sum, None, None)
        p.q_all, p.t_all, p.valid_all = q_all.data_ptr(), t_all.data_ptr(), L.ptr(valid)
        p.num_frames = q_all.shape[0]
        if isinstance(frame, torch.Tensor):
            p.frame, p.frame_dev = 0, frame.data_ptr()
        else:
            p.frame, p.frame_dev = int(frame), None
        p.pose = L.ptr(pose)
        p.head_acc, p.head_acc_floats = L.ptr(head_acc), 0 if head_acc is None else head_acc.numel()
        return p

    @staticmethod
    def backward(ctx, g_pose):
        weight, emb, ids32, cnt, seg, emb_sum, q_all, t_all, *heads = ctx.saved_tensors
        dev = weight.device
        A, E, width = weight.shape[0], emb.shape[1], weight.shape[2] + emb.shape[1]
        t, k_c, k_f, frame = ctx.scal
        e = lambda ref: torch.empty_like(ref)
        # dense outputs are written in full by the kernel; the head gradients are added into the accumulator the forward launch cleared
        d_weight, d_emb, d_q, d_t = e(weight), e(emb), e(q_all), e(t_all)
        acc = ctx.head_acc
        if acc is None:
            acc = torch.zeros(sum(h.numel() for h in heads), device=dev)
        elif ctx.acc_used:           # a second backward through the same forward (retain_graph): the kernel's clearing happened only once
            acc = torch.zeros_like(acc)
        ctx.acc_used = True
        d_heads = [v.view_as(h) for v, h in zip(torch.split(acc, [h.numel() for h in heads]), heads)]
        p = _TrackedPose._args(weight, emb, heads, ids32, cnt, seg, t, k_c, k_f, emb_sum, q_all, t_all, ctx.valid, frame, None)
        g = L.EmdTrackedPoseGrads()
        g_pose_c = g_pose.contiguous().float()           # (kept in a local until the launch has been issued: a temporary could be freed)
        g.g_pose = g_pose_c.data_ptr()
        g.d_q_all, g.d_t_all, g.d_weight, g.d_embeddings = d_q.data_ptr(), d_t.data_ptr(), d_weight.data_ptr(), d_emb.data_ptr()
        for h in range(4):
            g.d_head_w[h], g.d_head_b[h] = d_heads[2 * h].data_ptr(), d_heads[2 * h + 1].data_ptr()
        L.check(L.load().emd_tracked_pose_backward(C.byref(p), C.byref(g), _stream()), "emd_tracked_pose_backward")
        return (d_weight, d_emb, *d_heads, d_q.view(ctx.shapes[0]), d_t.view(ctx.shapes[1]), *([None] * 8))


def build_actor_pose(instances_quats, instances_trans, instances_fv, cur_frame, track_trans=None, track_rot=None,
                     in_test_set=False):
    """[A,12] rows (q_mean[4], trans[3], valid, q_rot[4]) for one frame -- the table the HIP kernels gather from.

    Reproduces the reference's asymmetry: the mean's rotation uses the (test-time interpolated) pose quaternion
    WITHOUT the learned rotation offset (rigid.py:485-503), the quaternion composition uses the un-interpolated
    pose quaternion WITH it (rigid.py:547-566); NaN offsets are skipped (rigid.py:528,559).
    """
    num_frames = instances_quats.shape[0]
    q_cur, t_cur = instances_quats[cur_frame], instances_trans[cur_frame]
    q_mean, trans = q_cur, t_cur
    if in_test_set and (cur_frame - 1 > 0 and cur_frame + 1 < num_frames):
        ok = (instances_fv[cur_frame - 1] & instances_fv[cur_frame + 1])[:, None]
        q_mean = torch.where(ok, interpolate_quats(instances_quats[cur_frame - 1], instances_quats[cur_frame + 1]), q_cur)
        trans = torch.where(ok, (instances_trans[cur_frame - 1] + instances_trans[cur_frame + 1]) * 0.5, t_cur)
    q_mean = quat_act(q_mean)
    if track_trans is not None:
        bad = track_trans.isnan().any(dim=-1, keepdim=True)
        trans = trans + torch.where(bad, torch.zeros_like(track_trans), track_trans)
    q_rot = q_cur
    if track_rot is not None:
        bad = track_rot.isnan().any(dim=-1, keepdim=True)
        ident = torch.zeros_like(track_rot)
        ident[:, 0] = 1.0
        # skipped rows multiply by the identity, so no NaN reaches the value or the gradient (rigid.py:559 `continue`)
        q_rot = quat_mult(q_cur, torch.where(bad, ident, track_rot))
    q_rot = quat_act(q_rot)
    valid = instances_fv[cur_frame].to(q_mean.dtype)[:, None]
    return torch.cat([q_mean, trans, valid, q_rot], dim=1).contiguous()


class _ActorPose(torch.autograd.Function):
    """`frame`: python int, or a 1-element int32 DEVICE tensor (the row is then selected inside the kernels: the call can be replayed
    from a hipGraph for another frame by rewriting that tensor)."""

    @staticmethod
    def forward(ctx, instances_quats, instances_trans, instances_fv, frame, track_trans, track_rot):
        lib = L.load()
        if instances_quats.device.type != "cuda":
            raise L.EmdError("actor_pose_table needs tensors on a ROCm device; there is no CPU path")
        on_dev = isinstance(frame, torch.Tensor)
        if on_dev:
            if frame.dtype != torch.int32 or frame.device != instances_quats.device:
                raise ValueError("a device frame index must be an int32 tensor on the tables' device")
            q_f, t_f = instances_quats.detach().contiguous(), instances_trans.detach().contiguous()
            valid = None if instances_fv is None else instances_fv.contiguous().view(torch.uint8)
            A = q_f.shape[1]
        else:
            q_f, t_f = instances_quats[frame].contiguous(), instances_trans[frame].contiguous()
            valid = None if instances_fv is None else instances_fv[frame].contiguous().view(torch.uint8)   # bool bytes, no copy kernel
            A = q_f.shape[0]
        pose = torch.empty(A, L.ACTOR_STRIDE, device=q_f.device, dtype=torch.float32)
        L.check(lib.emd_actor_pose_forward(A, q_f.data_ptr(), t_f.data_ptr(), L.ptr(valid), L.ptr(track_trans), L.ptr(track_rot),
                                           pose.data_ptr(), frame.data_ptr() if on_dev else None, _stream()), "emd_actor_pose_forward")
        ctx.frame, ctx.shapes = (frame if on_dev else int(frame)), (instances_quats.shape, instances_trans.shape)
        ctx.save_for_backward(q_f, track_trans, track_rot)
        return pose

    @staticmethod
    def backward(ctx, g_pose):
        lib = L.load()
        q_f, track_trans, track_rot = ctx.saved_tensors
        on_dev = isinstance(ctx.frame, torch.Tensor)
        A = q_f.shape[1] if on_dev else q_f.shape[0]
        nq, nt = ctx.shapes[0].numel(), ctx.shapes[1].numel()
        flat = torch.zeros(nq + nt, device=q_f.device, dtype=torch.float32)          # one fill for both dense clip gradients
        d_q, d_t = flat[:nq].view(ctx.shapes[0]), flat[nq:].view(ctx.shapes[1])
        d_dt = torch.empty_like(track_trans) if track_trans is not None else None
        d_dq = torch.empty_like(track_rot) if track_rot is not None else None
        dq_row, dt_row = (d_q, d_t) if on_dev else (d_q[ctx.frame], d_t[ctx.frame])
        L.check(lib.emd_actor_pose_backward(A, q_f.data_ptr(), L.ptr(track_trans), L.ptr(track_rot), g_pose.contiguous().data_ptr(),
                                            dq_row.data_ptr(), dt_row.data_ptr(), L.ptr(d_dt), L.ptr(d_dq),
                                            ctx.frame.data_ptr() if on_dev else None, _stream()), "emd_actor_pose_backward")
        return d_q, d_t, None, None, d_dt, d_dq


def actor_pose_table(instances_quats, instances_trans, instances_fv, frame, track_trans=None, track_rot=None):
    """[A,12] pose table of one training frame in ONE HIP launch (and one for its backward): same result as
    `build_actor_pose(..., in_test_set=False)`; the test-time interpolation branch stays in `build_actor_pose`."""
    c = lambda t: None if t is None else t.contiguous().float()
    return _ActorPose.apply(instances_quats, instances_trans, instances_fv, frame if isinstance(frame, torch.Tensor) else int(frame),
                            c(track_trans), c(track_rot))


class _MotionTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, opacities, actor_pose, residual_dx, residual_dq, actor_ids):
        lib = L.load()
        if means.device.type != "cuda":
            raise L.EmdError("transform_gaussians needs tensors on a ROCm device; there is no CPU path")
        N = means.shape[0]
        wm = torch.empty_like(means)
        wq = torch.empty_like(quats) if quats is not None else None
        wo = torch.empty_like(opacities) if opacities is not None else None
        m = L.EmdMotion(L.ptr(actor_ids), L.ptr(actor_pose), int(actor_pose.shape[0]), L.ptr(residual_dx), L.ptr(residual_dq))
        L.check(lib.emd_motion_forward(N, means.data_ptr(), L.ptr(quats), L.ptr(opacities), C.byref(m), wm.data_ptr(),
                                       L.ptr(wq), L.ptr(wo), _stream()), "emd_motion_forward")
        ctx.save_for_backward(means, quats, opacities, actor_pose, residual_dx, residual_dq, actor_ids)
        return wm, wq, wo

    @staticmethod
    def backward(ctx, g_wm, g_wq, g_wo):
        lib = L.load()
        means, quats, opacities, actor_pose, residual_dx, residual_dq, actor_ids = ctx.saved_tensors
        N = means.shape[0]
        c = lambda t: None if t is None else t.contiguous()
        g_wm, g_wq, g_wo = c(g_wm), c(g_wq), c(g_wo)
        d_means = torch.empty_like(means)
        d_quats = torch.empty_like(quats) if quats is not None else None
        d_opac = torch.empty_like(opacities) if opacities is not None else None
        d_pose = torch.empty_like(actor_pose)
        d_rdx = torch.empty_like(residual_dx) if residual_dx is not None else None
        d_rdq = torch.empty_like(residual_dq) if residual_dq is not None else None
        m = L.EmdMotion(L.ptr(actor_ids), L.ptr(actor_pose), int(actor_pose.shape[0]), L.ptr(residual_dx), L.ptr(residual_dq))
        L.check(lib.emd_motion_backward(N, means.data_ptr(), L.ptr(quats), L.ptr(opacities), C.byref(m), L.ptr(g_wm),
                                        L.ptr(g_wq), L.ptr(g_wo), d_means.data_ptr(), L.ptr(d_quats), L.ptr(d_opac),
                                        d_pose.data_ptr(), L.ptr(d_rdx), L.ptr(d_rdq), _stream()), "emd_motion_backward")
        return d_means, d_quats, d_opac, d_pose, d_rdx, d_rdq, None


def transform_gaussians(means, quats, opacities, actor_ids, actor_pose, residual_dx=None, residual_dq=None):
    """world_means, world_quats (already `quat_act`-ed), opacities * valid -- one HIP kernel, differentiable.

    Equivalent of RigidNodes.transform_means + transform_quats + the validity mask of get_gaussians
    (rigid.py:575-576,589-591); points with actor_id < 0 pass through unchanged (static background).
    """
    f = lambda t: None if t is None else t.contiguous().float()
    return _MotionTransform.apply(f(means), f(quats), None if opacities is None else f(opacities).reshape(-1),
                                  f(actor_pose), f(residual_dx), f(residual_dq), actor_ids.to(torch.int32).contiguous())
