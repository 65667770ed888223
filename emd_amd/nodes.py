"""Per-class Gaussian dictionaries of the OmniRe flow on the HIP path.

`rigid_gaussians` is RigidNodes.get_gaussians (OmniRe/models/nodes/rigid.py:570-615): world means / quaternions through the
per-actor rigid transform, opacity x validity of (frame, actor), SH colour on DETACHED world view directions clamped to [0, 1],
exp scales -- the dictionary `collect_gaussians` (models/trainers/base.py:342-383) concatenates over classes before the
`rasterization` call.  `deformable_gaussians` is DeformableNodes.get_gaussians (models/nodes/deformable.py:49-114): the learned
residual of `ConditionalDeformNetwork` is added to the (detached) canonical means and to the normalised quaternions first.
Two HIP launches per class (`emd_motion_forward`, `emd_sh_forward`) instead of the reference's Python loop over actors; the
NaN / Inf guard of the reference (ten `.any()` host syncs per class and step) is one reduction and one sync, and can be
switched off.  (When the caller uses this repository's GaussianRasterizer directly, the same transform and SH evaluation are
fused into its projection kernel and none of this is needed: EMD_FLAG_MOTION.)"""
import torch

from .gsplat_api import spherical_harmonics
from .motion import transform_gaussians


def _check_finite(gs, step):
    bad = torch.stack([~torch.isfinite(v).all() for v in gs.values()])
    if bool(bad.any()):                                          # one host sync (the reference: two per entry)
        for (k, v), b in zip(gs.items(), bad.tolist()):
            if b:
                kind = "NaN" if bool(torch.isnan(v).any()) else "Inf"
                raise ValueError(f"{kind} detected in gaussian {k} at step {step}")


def _colors(world_means, features_dc, features_rest, camera_center, sh_degree, step, sh_degree_interval):
    colors = torch.cat((features_dc[:, None, :], features_rest), dim=1)
    if sh_degree > 0:
        viewdirs = world_means.detach() - camera_center                           # (N, 3): no gradient to the means through colour
        viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
        n = min(step // sh_degree_interval, sh_degree)
        return torch.clamp(spherical_harmonics(n, viewdirs, colors) + 0.5, 0.0, 1.0)
    return torch.sigmoid(colors[:, 0, :])


def rigid_gaussians(means, quats, opacity_logits, log_scales, features_dc, features_rest, point_ids, actor_pose, camera_center,
                    sh_degree=3, step=0, sh_degree_interval=1000, check_finite=True):
    """-> dict(_means, _opacities [N,1], _rgbs, _scales, _quats).  actor_pose: the [A,12] table of motion.build_actor_pose /
    actor_pose_table for the current frame (pose, learned track offsets, validity)."""
    ids = point_ids[..., 0] if point_ids.dim() == 2 else point_ids
    world_means, world_quats, opac = transform_gaussians(means, quats, torch.sigmoid(opacity_logits), ids, actor_pose)
    gs = dict(_means=world_means, _opacities=opac.reshape(-1, 1),
              _rgbs=_colors(world_means, features_dc, features_rest, camera_center, sh_degree, step, sh_degree_interval),
              _scales=torch.exp(log_scales), _quats=world_quats)
    if check_finite:
        _check_finite(gs, step)
    return gs


def deformable_gaussians(network, means, quats, opacity_logits, log_scales, features_dc, features_rest, point_ids, actor_pose,
                         camera_center, instances_size, instances_embedding, t, sh_degree=3, step=0, sh_degree_interval=1000,
                         use_deformation=True, stop_optimizing_canonical_xyz=True, check_finite=True):
    """DeformableNodes.get_gaussians: delta_xyz on the canonical means (detached when `stop_optimizing_canonical_xyz`), delta_quat on
    the normalised quaternions, delta_scale (when the network has that head) on the activated scales; then as rigid_gaussians."""
    from .deformation import nonrigid_deformation
    ids = point_ids[..., 0] if point_ids.dim() == 2 else point_ids
    dx = dq = ds = None
    if use_deformation:
        dx, dq, ds = nonrigid_deformation(network, means, ids, instances_size, instances_embedding, t)
    base = means.detach() if (dx is not None and stop_optimizing_canonical_xyz) else means
    # the residuals enter the HIP transform as `residual_dx` / `residual_dq`, added before the rigid motion; the reference adds
    # delta_quat to get_quats = q / |q| (deformable.py:69), so the quaternions are normalised first when a residual is present
    q_in = quats / quats.norm(dim=-1, keepdim=True) if dq is not None else quats
    world_means, world_quats, opac = transform_gaussians(base, q_in, torch.sigmoid(opacity_logits), ids, actor_pose,
                                                         residual_dx=dx, residual_dq=dq)
    scales = torch.exp(log_scales)
    if ds is not None:
        scales = scales + ds
    gs = dict(_means=world_means, _opacities=opac.reshape(-1, 1),
              _rgbs=_colors(world_means, features_dc, features_rest, camera_center, sh_degree, step, sh_degree_interval),
              _scales=scales, _quats=world_quats)
    if check_finite:
        _check_finite(gs, step)
    return gs
