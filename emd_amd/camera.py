"""Camera model of the hot path: the matrices GaussianRasterizationSettings carries.

Mirrors (same names, argument meaning and conventions; written from the behaviour, not copied):
  getWorld2View2        S3Gaussian/utils/graphics_utils.py:58-70
  getProjectionMatrix   S3Gaussian/utils/graphics_utils.py:72-92
  focal2fov / fov2focal S3Gaussian/utils/graphics_utils.py:94-98
  Camera                S3Gaussian/scene/cameras.py:55-66  (world_view_transform = W2C^T,
                        full_proj_transform = world_view_transform @ P^T, camera_center)
  OmniRe cameras        c2w + K (OmniRe/models/trainers/base.py:393-408) via `from_c2w_K`
"""
import math
from typing import NamedTuple

import numpy as np
import torch


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """R is stored transposed (camera-to-world rotation), t is the W2C translation; fp64 inverse, fp32 result."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """Perspective matrix with z mapped to [0,1] and P[3,2] = 1 (column-vector form; callers transpose it)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top, right = tanHalfFovY * znear, tanHalfFovX * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera(NamedTuple):
    """The per-view constants the rasterizer settings are built from."""
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # [4,4] = W2C^T
    projection_matrix: torch.Tensor     # [4,4] = P^T
    full_proj_transform: torch.Tensor   # [4,4] = world_view_transform @ projection_matrix
    camera_center: torch.Tensor         # [3]
    time: float = 0.0
    cam_no: int = 0

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)


def make_camera(R, T, FoVx, FoVy, height, width, znear=0.01, zfar=100.0, time=0.0, cam_no=0,
                trans=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """Same construction as S3Gaussian/scene/cameras.py:55-66 (znear 0.01, zfar 100)."""
    wvt = torch.tensor(getWorld2View2(R, T, trans, scale)).transpose(0, 1)
    proj = getProjectionMatrix(znear=znear, zfar=zfar, fovX=FoVx, fovY=FoVy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    return Camera(int(height), int(width), float(FoVx), float(FoVy), wvt, proj, full, center, float(time), int(cam_no))


def projection_from_K(K, width, height, znear=0.01, zfar=100.0):
    """P^T (row-vector form) for a pinhole K with an arbitrary principal point, in the same NDC
    convention as getProjectionMatrix: pixel = ((ndc + 1) * size - 1) / 2."""
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * fx / width
    P[1, 1] = 2.0 * fy / height
    # pixel x = fx X/Z + cx - 0.5  (pixel centres at integers)  =>  ndc offset (2 cx - W) / W
    P[0, 2] = (2.0 * cx - width) / width
    P[1, 2] = (2.0 * cy - height) / height
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P.transpose(0, 1)


def from_c2w_K(c2w, K, width, height, znear=0.01, zfar=100.0):
    """OmniRe-style camera (camtoworlds [4,4], Ks [3,3]; base.py:399-400) -> Camera."""
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    K = torch.as_tensor(K, dtype=torch.float32)
    w2c = torch.linalg.inv(c2w)
    wvt = w2c.transpose(0, 1).contiguous()
    proj = projection_from_K(K, width, height, znear, zfar)
    full = wvt @ proj
    FoVx = focal2fov(float(K[0, 0]), width)
    FoVy = focal2fov(float(K[1, 1]), height)
    return Camera(int(height), int(width), FoVx, FoVy, wvt, proj, full, c2w[:3, 3].clone())


def look_at_camera(eye, yaw_deg, height, width, fx, fy, pitch_deg=0.0, **kw):
    """Street-rig helper: camera at `eye` (world: x forward, y left, z up), yawed about +z.
    OpenCV camera axes (x right, y down, z forward)."""
    yaw, pitch = math.radians(yaw_deg), math.radians(pitch_deg)
    fwd = np.array([math.cos(yaw) * math.cos(pitch), math.sin(yaw) * math.cos(pitch), math.sin(pitch)])
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w_R = np.stack([right, down, fwd], 1)  # columns = camera axes in world
    w2c_R = c2w_R.T
    T = -w2c_R @ np.asarray(eye, np.float64)
    # reference convention: Camera.R is the transposed W2C rotation (dataset_readers.py:137-139)
    return make_camera(c2w_R, T, focal2fov(fx, width), focal2fov(fy, height), height, width, **kw)
