"""ctypes front-end of oracle/raster_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (emd_amd) never does; it fails loudly when its HIP extension is missing.

Parity status: see the header of raster_oracle.c ("PARITY UNPINNED" for the tile rasterizer
itself -- the reference ships no rasterizer source, tests or golden vectors; the Python-level
pieces are pinned by tests/golden/*.npz generated from the imported reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F_NORMAL, F_MOTION, F_ABSGRAD, F_CLAMP01 = 1, 2, 4, 16
TILE = 16
ACTOR_STRIDE = 12


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_bin.restype = C.c_int64
    return _LIB


class OrcSettings(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("bg", C.c_float * 3), ("scale_modifier", C.c_float), ("view", C.c_float * 16),
                ("proj", C.c_float * 16), ("sh_degree", C.c_int32), ("campos", C.c_float * 3),
                ("near_plane", C.c_float)]


def make_settings(H, W, tanfovx, tanfovy, bg, viewmatrix, projmatrix, sh_degree, campos, scale_modifier=1.0,
                  near_plane=0.2):
    s = OrcSettings()
    s.H, s.W = int(H), int(W)
    s.tanfovx, s.tanfovy = float(tanfovx), float(tanfovy)
    s.bg[:] = [float(x) for x in np.asarray(bg, np.float32).reshape(3)]
    s.scale_modifier = float(scale_modifier)
    s.view[:] = [float(x) for x in np.asarray(viewmatrix, np.float32).reshape(16)]
    s.proj[:] = [float(x) for x in np.asarray(projmatrix, np.float32).reshape(16)]
    s.sh_degree = int(sh_degree)
    s.campos[:] = [float(x) for x in np.asarray(campos, np.float32).reshape(3)]
    s.near_plane = float(near_plane)
    return s


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def motion_forward(means, quats, opac, actor_id, pose, rdx=None, rdq=None):
    means = _f32(means); quats = _f32(quats); opac = _f32(opac); pose = _f32(pose)
    rdx = _f32(rdx); rdq = _f32(rdq)
    actor_id = np.ascontiguousarray(actor_id, np.int32)
    N = means.shape[0]
    wm = np.zeros((N, 3), np.float32)
    wq = np.zeros((N, 4), np.float32) if quats is not None else None
    wo = np.zeros((N,), np.float32) if opac is not None else None
    lib().orc_motion_forward(N, _p(means), _p(quats), _p(opac), _p(actor_id), _p(pose), _p(rdx), _p(rdq), _p(wm),
                             _p(wq), _p(wo))
    return wm, wq, wo


def sh_forward(degree, dirs, coeffs):
    dirs = _f32(dirs); coeffs = _f32(coeffs)
    N, M = coeffs.shape[0], coeffs.shape[1]
    rgb = np.zeros((N, 3), np.float32)
    lib().orc_sh_forward(N, int(degree), M, _p(dirs), _p(coeffs), _p(rgb))
    return rgb


def cov3d(scales, mod, rots):
    scales = _f32(scales); rots = _f32(rots)
    out = np.zeros((scales.shape[0], 6), np.float32)
    lib().orc_cov3d(scales.shape[0], _p(scales), C.c_float(mod), _p(rots), _p(out))
    return out


class Scene:
    """Plain container of the rasterizer inputs (numpy, fp32)."""

    def __init__(self, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None, actor_id=None, actor_pose=None, residual_dx=None, residual_dq=None):
        self.means3D = _f32(means3D)
        self.opacities = _f32(opacities).reshape(-1)
        self.shs = _f32(shs)
        self.colors_precomp = _f32(colors_precomp)
        self.scales = _f32(scales)
        self.rotations = _f32(rotations)
        self.cov3D_precomp = _f32(cov3D_precomp)
        self.actor_id = None if actor_id is None else np.ascontiguousarray(actor_id, np.int32)
        self.actor_pose = _f32(actor_pose)
        self.residual_dx = _f32(residual_dx)
        self.residual_dq = _f32(residual_dq)

    @property
    def N(self):
        return self.means3D.shape[0]

    @property
    def M(self):
        return 0 if self.shs is None else self.shs.shape[1]

    @property
    def A(self):
        return 0 if self.actor_pose is None else self.actor_pose.shape[0]


def preprocess(S, sc, flags=0):
    N = sc.N
    out = dict(means2D=np.zeros((N, 2), np.float32), depths=np.zeros(N, np.float32),
               conic_opacity=np.zeros((N, 4), np.float32), rgb=np.zeros((N, 3), np.float32),
               normal=np.zeros((N, 3), np.float32), radii=np.zeros(N, np.int32),
               tiles_touched=np.zeros(N, np.uint32), rect=np.zeros((N, 4), np.int32),
               clamped=np.zeros((N, 3), np.uint8), cov3D=np.zeros((N, 6), np.float32))
    lib().orc_preprocess(C.byref(S), N, sc.M, int(flags), _p(sc.means3D), _p(sc.shs), _p(sc.colors_precomp),
                         _p(sc.opacities), _p(sc.scales), _p(sc.rotations), _p(sc.cov3D_precomp), _p(sc.actor_id),
                         _p(sc.actor_pose), _p(sc.residual_dx), _p(sc.residual_dq), _p(out["means2D"]),
                         _p(out["depths"]), _p(out["conic_opacity"]), _p(out["rgb"]), _p(out["normal"]),
                         _p(out["radii"]), _p(out["tiles_touched"]), _p(out["rect"]), _p(out["clamped"]),
                         _p(out["cov3D"]))
    return out


def bin_tiles(S, pre):
    N = pre["depths"].shape[0]
    gx, gy = (S.W + TILE - 1) // TILE, (S.H + TILE - 1) // TILE
    D = int(pre["tiles_touched"].astype(np.int64).sum())
    keys = np.zeros(max(D, 1), np.uint64)
    ids = np.zeros(max(D, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    got = lib().orc_bin(C.byref(S), N, _p(pre["depths"]), _p(pre["tiles_touched"]), _p(pre["rect"]), C.c_int64(D),
                        _p(keys), _p(ids), _p(ranges))
    assert got == D
    return dict(D=D, keys=keys[:D], ids=ids[:D], ranges=ranges)


def pair_quadrant_hits(S, pre, binning):
    """Checker of the product's footprint culling: per entry of the sorted list the 4-bit mask of the tile's 8x8 quadrants in which at
    least one pixel passes the render loop's skip tests (brute force over the pixels; orc_pair_quadrant_hits)."""
    D = binning["D"]
    out = np.zeros(max(D, 1), np.uint8)
    if D:
        lib().orc_pair_quadrant_hits(C.byref(S), C.c_int64(D), _p(binning["keys"]), _p(binning["ids"]), _p(pre["means2D"]),
                                     _p(pre["conic_opacity"]), _p(out))
    return out[:D]


def render_forward(S, pre, binning, flags=0):
    H, W = S.H, S.W
    out = dict(color=np.zeros((3, H, W), np.float32), depth=np.zeros((1, H, W), np.float32),
               normal=np.zeros((3, H, W), np.float32), alpha=np.zeros((1, H, W), np.float32),
               n_contrib=np.zeros((H, W), np.uint32), final_T=np.zeros((H, W), np.float32))
    ids = binning["ids"] if binning["D"] > 0 else np.zeros(1, np.uint32)
    lib().orc_render_forward(C.byref(S), int(flags), _p(binning["ranges"]), _p(ids), _p(pre["means2D"]),
                             _p(pre["conic_opacity"]), _p(pre["rgb"]), _p(pre["depths"]), _p(pre["normal"]),
                             _p(out["color"]), _p(out["depth"]), _p(out["normal"]), _p(out["alpha"]),
                             _p(out["n_contrib"]), _p(out["final_T"]))
    return out


def forward(S, sc, flags=0):
    pre = preprocess(S, sc, flags)
    b = bin_tiles(S, pre)
    img = render_forward(S, pre, b, flags)
    return pre, b, img


def render_backward(S, sc, pre, binning, img, dL_dcolor=None, dL_ddepth=None, dL_dalpha=None, dL_dnormal=None, flags=0):
    """K7: per-Gaussian sums of the per-(pixel, Gaussian) partial derivatives -> dict(mean2D[N,2] (pixel units), abs[N,2], conic[N,3],
    opacity[N], rgb[N,3], depth[N], normal[N,3])."""
    N = sc.N
    dL_dcolor = _f32(dL_dcolor); dL_ddepth = _f32(dL_ddepth); dL_dalpha = _f32(dL_dalpha); dL_dnormal = _f32(dL_dnormal)
    g = dict(mean2D=np.zeros((N, 2), np.float32), abs=np.zeros((N, 2), np.float32),
             conic=np.zeros((N, 3), np.float32), opacity=np.zeros(N, np.float32), rgb=np.zeros((N, 3), np.float32),
             depth=np.zeros(N, np.float32), normal=np.zeros((N, 3), np.float32))
    ids = binning["ids"] if binning["D"] > 0 else np.zeros(1, np.uint32)
    lib().orc_render_backward(C.byref(S), N, int(flags), _p(binning["ranges"]), _p(ids), _p(pre["means2D"]),
                              _p(pre["conic_opacity"]), _p(pre["rgb"]), _p(pre["depths"]), _p(pre["normal"]),
                              _p(img["n_contrib"]), _p(img["final_T"]), _p(dL_dcolor), _p(dL_ddepth), _p(dL_dalpha),
                              _p(dL_dnormal), _p(g["mean2D"]), _p(g["abs"]), _p(g["conic"]), _p(g["opacity"]),
                              _p(g["rgb"]), _p(g["depth"]), _p(g["normal"]))
    return g


def preprocess_backward(S, sc, pre, g, flags=0):
    """K8: chain the render gradients `g` (as render_backward returns them) to the inputs."""
    N = sc.N
    g = {k: np.ascontiguousarray(v, np.float32) for k, v in g.items()}
    M, A = sc.M, sc.A
    out = dict(means3D=np.zeros((N, 3), np.float32), means2D=np.zeros((N, 3), np.float32),
               shs=np.zeros((N, M, 3), np.float32) if M else None,
               colors=np.zeros((N, 3), np.float32), opacities=np.zeros(N, np.float32),
               scales=np.zeros((N, 3), np.float32), rotations=np.zeros((N, 4), np.float32),
               cov3D=np.zeros((N, 6), np.float32),
               actor_pose=np.zeros((max(A, 1), ACTOR_STRIDE), np.float32),
               residual_dx=np.zeros((N, 3), np.float32), residual_dq=np.zeros((N, 4), np.float32))
    lib().orc_preprocess_backward(C.byref(S), N, M, int(flags), _p(sc.means3D), _p(sc.shs), _p(sc.colors_precomp),
                                  _p(sc.opacities), _p(sc.scales), _p(sc.rotations), _p(sc.cov3D_precomp),
                                  _p(sc.actor_id), _p(sc.actor_pose), A, _p(sc.residual_dx), _p(sc.residual_dq),
                                  _p(pre["radii"]), _p(pre["clamped"]), _p(g["mean2D"]), _p(g["conic"]),
                                  _p(g["opacity"]), _p(g["rgb"]), _p(g["depth"]), _p(out["means3D"]),
                                  _p(out["means2D"]), _p(out["shs"]), _p(out["colors"]), _p(out["opacities"]),
                                  _p(out["scales"]), _p(out["rotations"]), _p(out["cov3D"]), _p(out["actor_pose"]),
                                  _p(out["residual_dx"]), _p(out["residual_dq"]))
    out["means2D_abs"] = np.stack([0.5 * S.W * g["abs"][:, 0], 0.5 * S.H * g["abs"][:, 1]], 1)
    out["render_grads"] = g
    out["actor_pose"] = out["actor_pose"][:A]
    return out


def backward(S, sc, pre, binning, img, dL_dcolor=None, dL_ddepth=None, dL_dalpha=None, dL_dnormal=None, flags=0):
    return preprocess_backward(S, sc, pre, render_backward(S, sc, pre, binning, img, dL_dcolor, dL_ddepth, dL_dalpha, dL_dnormal, flags), flags)
