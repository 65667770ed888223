"""Synthetic street scenes for tests and bench.py (no dataset can be fetched here).

Gaussians (seed 0): world box x in [0,120] m forward, y in [-30,30], z in [-2,10]; 70 % lie on a ground
sheet and two facade sheets (street-like tile occupancy), 30 % are uniform.  log-scale ~ N(log 0.05, 0.5^2)
clipped to [log 0.005, log 0.5] (reference init is kNN-derived, S3Gaussian/scene/gaussian_model.py:163-164),
unit quaternions from N(0,1)^4, opacity logit ~ N(0,1.5^2) (gaussian_model.py:168), SH dc = RGB2SH(U(0,1)),
rest ~ N(0,0.05^2), SH degree 3.
Actors (seed 1): boxes 4.5 x 2 x 1.6 m with up to 5000 Gaussians each (OmniRe/configs/paper_legacy/omnire.yaml:93),
straight-line motion 0-15 m/s at 10 Hz plus a yaw rate <= 0.1 rad/frame.
"""
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .camera import look_at_camera

SH_C0 = 0.28209479177387814
RIG_YAWS = (0.0, 45.0, -45.0, 90.0, -90.0, 180.0)


@dataclass
class GaussianScene:
    means: torch.Tensor        # [N,3] (local coordinates for actor points)
    log_scales: torch.Tensor   # [N,3]
    quats: torch.Tensor        # [N,4] raw (w,x,y,z)
    opacity_logits: torch.Tensor  # [N,1]
    shs: torch.Tensor          # [N,16,3]
    actor_id: Optional[torch.Tensor] = None     # [N] int32, -1 static
    actor_quats: Optional[torch.Tensor] = None  # [F,A,4]
    actor_trans: Optional[torch.Tensor] = None  # [F,A,3]
    actor_valid: Optional[torch.Tensor] = None  # [F,A] bool

    @property
    def N(self):
        return self.means.shape[0]

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        return GaussianScene(**kw)


def make_static_scene(n, seed=0, sh_coeffs=16):
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    rn = lambda *s: torch.randn(*s, generator=g)
    n_sheet = int(0.7 * n)
    n_ground = n_sheet // 2
    n_wall = n_sheet - n_ground
    ground = torch.stack([u(n_ground) * 120.0, u(n_ground) * 60.0 - 30.0, -1.5 + 0.2 * rn(n_ground)], 1)
    side = torch.where(u(n_wall) < 0.5, -15.0, 15.0)
    wall = torch.stack([u(n_wall) * 120.0, side + 0.2 * rn(n_wall), u(n_wall) * 12.0 - 2.0], 1)
    n_uni = n - n_sheet
    uni = torch.stack([u(n_uni) * 120.0, u(n_uni) * 60.0 - 30.0, u(n_uni) * 12.0 - 2.0], 1)
    means = torch.cat([ground, wall, uni], 0)
    means = means[torch.randperm(n, generator=g)]
    log_scales = (math.log(0.05) + 0.5 * rn(n, 3)).clamp(math.log(0.005), math.log(0.5))
    quats = rn(n, 4)
    quats = quats / quats.norm(dim=1, keepdim=True)
    opacity = 1.5 * rn(n, 1)
    shs = torch.zeros(n, sh_coeffs, 3)
    shs[:, 0] = (u(n, 3) - 0.5) / SH_C0
    if sh_coeffs > 1:
        shs[:, 1:] = 0.05 * rn(n, sh_coeffs - 1, 3)
    return GaussianScene(means.float(), log_scales.float(), quats.float(), opacity.float(), shs.float())


def add_actors(scene, num_actors=32, pts_per_actor=5000, num_frames=50, seed=1):
    """Re-label the first A*pts Gaussians as actor points (local box coordinates) and attach per-frame poses."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    A, Pn = num_actors, pts_per_actor
    n_dyn = min(A * Pn, scene.N)
    Pn = n_dyn // A
    n_dyn = A * Pn
    size = torch.tensor([4.5, 2.0, 1.6])
    local = (u(n_dyn, 3) - 0.5) * size
    means = scene.means.clone()
    means[:n_dyn] = local
    actor_id = torch.full((scene.N,), -1, dtype=torch.int32)
    actor_id[:n_dyn] = torch.arange(A, dtype=torch.int32).repeat_interleave(Pn)
    start = torch.stack([u(A) * 100.0 + 5.0, u(A) * 24.0 - 12.0, torch.full((A,), -0.7)], 1)
    heading = u(A) * 2 * math.pi
    speed = u(A) * 15.0 / 10.0  # m per frame at 10 Hz
    yaw_rate = (u(A) - 0.5) * 0.2
    f = torch.arange(num_frames, dtype=torch.float32)[:, None]
    yaw = heading[None] + yaw_rate[None] * f
    trans = start[None] + torch.stack([torch.cos(heading)[None] * speed[None] * f,
                                       torch.sin(heading)[None] * speed[None] * f,
                                       torch.zeros(num_frames, A)], -1)
    quats = torch.stack([torch.cos(yaw / 2), torch.zeros_like(yaw), torch.zeros_like(yaw), torch.sin(yaw / 2)], -1)
    valid = torch.ones(num_frames, A, dtype=torch.bool)
    return GaussianScene(means, scene.log_scales, scene.quats, scene.opacity_logits, scene.shs, actor_id,
                         quats.float(), trans.float(), valid)


def rig_camera(frame=0, cam=0, height=1066, width=1600, fx=1700.0, fy=1700.0):
    """Ego at (1*frame, 0, 1.5) m, rig yaw RIG_YAWS[cam]."""
    return look_at_camera((1.0 * frame, 0.0, 1.5), RIG_YAWS[cam % len(RIG_YAWS)], height, width, fx, fy)


def small_camera(height=64, width=96, eye=(0.0, 0.0, 1.5), yaw=0.0, focal=None):
    focal = focal if focal is not None else 1.0625 * width
    return look_at_camera(eye, yaw, height, width, focal, focal)
