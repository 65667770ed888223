"""`GaussianModel` -- the reference's parameter store, its adaptive density control and its on-disk formats, on the HIP path.

Mirrors, with the same attribute / method names and argument meaning (behaviour restated, not copied):
  parameter layout, capture / restore (16-tuple)     S3Gaussian/scene/gaussian_model.py:54-118
  training_setup (named Adam groups, lr schedules)    :183-243
  construct_list_of_attributes / save_ply / load_ply  :245-298, :378-425
  reset_opacity / replace_tensor_to_optimizer         :373-376, :427-439
  prune_points / _prune_optimizer                     :441-479
  densification_postfix / cat_tensors_to_optimizer    :480-530
  densify_and_split / densify_and_clone / densify     :532-603, :696-701
  prune                                               :683-695
  add_densification_stats + max_radii2D update        :728-730, train.py:403-406

What differs from the reference is WHERE the work runs.  The reference edits every parameter and both Adam moments with
boolean-mask indexing and `torch.cat` / `repeat` (a device-to-host sync per mask, ~60 launches and as many temporaries per
event) and draws the split samples from the global CUDA generator.  Here one event is four launches (`emd_densify_*`,
csrc/densify.hip): decide -> prefix sums -> output index -> ONE gather that writes all 7 parameters, 14 Adam moments and the
statistics, with a single host read (the new point count, needed to size the new tensors).  The split samples come from
Philox4x32-10 keyed by (seed, source Gaussian index, replica): every rank of a data-parallel job draws the same samples without
communication, so replicas stay identical ("DP semantic change").  The ORDER of the resulting points is the
reference's: survivors, clones, split samples replica 0, replica 1.

On-disk formats: `save_ply` writes the binary little-endian PLY that `plyfile` produces for the attribute list of
`construct_list_of_attributes` (x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_* embedding_*), `load_ply` reads it by
property name exactly as the reference does.  (The reference's own `save_ply` concatenates 62 columns for these 66 names and
raises inside numpy whenever `gaussian_embedding_dim > 0` -- gaussian_model.py:282-291; files written here carry all 66 columns
and load in the reference unchanged.)  `capture` / `restore` use the reference's 16-tuple, so `torch.save((gaussians.capture(),
iteration), path)` checkpoints are interchangeable.
"""
import ctypes as C
import os
import struct

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .optim import Adam, expon_lr

SH_C0 = 0.28209479177387814


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- PLY (binary_little_endian 1.0, one `vertex` element of float properties) ---------------------------------------------------
def write_ply(path, names, columns):
    """columns: float32 [N, len(names)]."""
    columns = np.ascontiguousarray(columns, dtype="<f4")
    assert columns.ndim == 2 and columns.shape[1] == len(names)
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % columns.shape[0]
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(columns.tobytes())


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
              "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4"}


def read_ply(path):
    """-> dict property name -> numpy array, of the first element (`vertex`).  binary little / big endian and ascii."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first, n_elements = None, 0, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                n_elements += 1
                in_first = n_elements == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not part of the Gaussian format")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            return {n: data[:, i].astype(t) for i, (n, t) in enumerate(props)}
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, end + t) for n, t in props])
        arr = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
        return {n: np.asarray(arr[n]) for n, _ in props}


def restructure_rows(mode, args, jobs, N, seed=0, samples=None, front_rows=0, scaling=None, rotation=None):
    """One density-control event on `N` rows of an arbitrary set of per-point tensors: decide -> scan -> index -> ONE gather (csrc/densify.hip).
    `args`: an EmdDensifyArgs with the decision inputs filled (pointers to the N rows); `jobs`: [(tensor [N, ...] float32 contiguous, role)];
    `scaling` [N,3] / `rotation` [N,4]: the source log-scales and quaternions a split sample's position is drawn from (needed in DENSIFY mode);
    `front_rows`: every output tensor gets that many extra rows in FRONT of the gathered ones, left for the caller to fill (rows of a store
    that do not take part: the actors of emd_amd.model.density_control).  -> (outs or None when nothing changes, (n_keep, n_clone, n_split)).
    The event's single host read is the three totals."""
    lib = L.load()
    if N == 0:
        return None, (0, 0, 0)
    dev = jobs[0][0].device
    code = torch.empty(N, dtype=torch.int32, device=dev)
    inc = torch.empty(3, (N + 255) // 256, dtype=torch.int32, device=dev)          # per-block counts, then exclusive block offsets
    totals = torch.empty(3, dtype=torch.int32, device=dev)
    args.num_points, args.mode = N, mode
    L.check(lib.emd_densify_decide(C.byref(args), code.data_ptr(), inc.data_ptr(), _stream()), "emd_densify_decide")
    L.check(lib.emd_densify_scan(N, 3, inc.data_ptr(), totals.data_ptr(), _stream()), "emd_densify_scan")
    n_keep, n_clone, n_split = (int(v) for v in totals.tolist())
    if (mode == L.DENSIFY_MODE_DENSIFY and n_clone == 0 and n_split == 0) or (mode == L.DENSIFY_MODE_PRUNE and n_keep == N):
        return None, (n_keep, n_clone, n_split)
    M = n_keep + n_clone + 2 * n_split
    src = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    kind = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    L.check(lib.emd_densify_index(N, M, code.data_ptr(), inc.data_ptr(), totals.data_ptr(), src.data_ptr(), kind.data_ptr(), _stream()), "emd_densify_index")
    g = L.EmdDensifyGather()
    g.num_out, g.mode, g.num_split = M, mode, n_split
    g.src, g.kind = src.data_ptr(), kind.data_ptr()
    g.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    keep_alive = [src, kind, code, inc, totals, scaling, rotation]
    g.scaling, g.rotation = L.ptr(scaling), L.ptr(rotation)
    if samples is not None:
        samples = samples.to(dev).float().contiguous()
        assert samples.shape == (2, n_split, 3), (tuple(samples.shape), n_split)
        rank = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
        L.check(lib.emd_densify_split_rank(M, n_keep, n_clone, n_split, rank.data_ptr(), _stream()), "emd_densify_split_rank")
        g.samples, g.split_rank = samples.data_ptr(), rank.data_ptr()
        keep_alive += [samples, rank]
    assert len(jobs) <= L.DENSIFY_MAX_TENSORS
    outs = []
    for k, (t, role) in enumerate(jobs):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == N
        width = t.numel() // N
        out = torch.empty((front_rows + M,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev)
        g.tensors[k].src, g.tensors[k].dst, g.tensors[k].width = t.data_ptr(), out.data_ptr() + 4 * width * front_rows, width
        g.tensors[k].role = role
        outs.append(out)
    g.num_tensors = len(jobs)
    L.check(lib.emd_densify_gather(C.byref(g), _stream()), "emd_densify_gather")
    del keep_alive
    return outs, (n_keep, n_clone, n_split)


class GaussianModel:
    """Parameters in the reference's layout: `_xyz [N,3]`, `_features_dc [N,1,3]`, `_features_rest [N,15,3]`, `_scaling [N,3]` (log),
    `_rotation [N,4]` (raw), `_opacity [N,1]` (logit), `_embedding [N,E]`; statistics `max_radii2D [N]`, `xyz_gradient_accum [N,1]`,
    `denom [N,1]`; `_deformation_table [N]` bool.  `deformation` / `sky_model` are optional modules whose `state_dict()` rides in
    capture() as in the reference (emd_amd.deformation.deform_network, emd_amd.sky.SkyCubeMap)."""

    GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "embedding")
    _ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
             "rotation": "_rotation", "embedding": "_embedding"}

    def __init__(self, sh_degree=3, gaussian_embedding_dim=4, device="cuda", deformation=None, sky_model=None, densify_seed=0):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.gaussian_embedding_dim = gaussian_embedding_dim
        self.device = torch.device(device)
        e = lambda *s: torch.empty(*s, device=self.device)
        self._xyz, self._features_dc, self._features_rest = e(0, 3), e(0, 1, 3), e(0, (sh_degree + 1) ** 2 - 1, 3)
        self._scaling, self._rotation, self._opacity, self._embedding = e(0, 3), e(0, 4), e(0, 1), e(0, gaussian_embedding_dim)
        self.max_radii2D, self.xyz_gradient_accum, self.denom = e(0), e(0, 1), e(0, 1)
        self._deformation_table = torch.empty(0, dtype=torch.bool, device=self.device)
        self._deformation, self._sky_model = deformation, sky_model
        self.optimizer = None
        self.percent_dense = 0.0
        self.spatial_lr_scale = 0.0
        self.densify_seed = int(densify_seed)      # Philox key of the split samples (identical on all ranks)
        self.densify_events = 0                    # mixed into the key: a fresh draw per event

    # ---- accessors (gaussian_model.py:120-146) ------------------------------------------------------------------------------------
    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: torch.exp(self._scaling))
    get_rotation = property(lambda self: torch.nn.functional.normalize(self._rotation))
    get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    get_embedding = property(lambda self: self._embedding)
    get_features = property(lambda self: torch.cat((self._features_dc, self._features_rest), dim=1))

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- construction ----------------------------------------------------------------------------------------------------------------
    def create_from_tensors(self, xyz, rgb, scales_log, spatial_lr_scale=1.0, opacity=0.1):
        """create_from_pcd (gaussian_model.py:152-181) with the kNN-derived initial scales supplied by the caller
        (`distCUDA2` belongs to the absent simple_knn dependency): SH dc = RGB2SH(rgb), identity rotations, opacity 0.1."""
        self.spatial_lr_scale = float(spatial_lr_scale)
        dev, n = self.device, xyz.shape[0]
        P = lambda t: nn.Parameter(t.to(dev).float().contiguous().requires_grad_(True))
        self._xyz = P(xyz)
        self._features_dc = P(((rgb.to(dev).float() - 0.5) / SH_C0)[:, None, :])
        self._features_rest = P(torch.zeros(n, (self.max_sh_degree + 1) ** 2 - 1, 3))
        self._scaling = P(scales_log.reshape(n, -1).expand(n, 3) if scales_log.dim() < 2 or scales_log.shape[1] == 1 else scales_log)
        rots = torch.zeros(n, 4)
        rots[:, 0] = 1
        self._rotation = P(rots)
        self._opacity = P(inverse_sigmoid(opacity * torch.ones(n, 1)))
        self._embedding = P(torch.zeros(n, self.gaussian_embedding_dim))
        self.max_radii2D = torch.zeros(n, device=dev)
        self._deformation_table = torch.ones(n, dtype=torch.bool, device=dev)

    def training_setup(self, training_args):
        """The ten named groups of gaussian_model.py:188-201 (deformation / grid / sky only when those modules are attached); the
        optimiser is emd_amd.optim.Adam (one HIP launch per step), `lr=0.0, eps=1e-15` as in the reference."""
        a = training_args
        n = self._xyz.shape[0]
        self.percent_dense = a.percent_dense
        self.xyz_gradient_accum = torch.zeros(n, 1, device=self.device)
        self.denom = torch.zeros(n, 1, device=self.device)
        s = self.spatial_lr_scale
        groups = [{"params": [self._xyz], "lr": a.position_lr_init * s, "name": "xyz"}]
        if self._deformation is not None:
            groups += [{"params": list(self._deformation.get_mlp_parameters()), "lr": a.deformation_lr_init * s, "name": "deformation"},
                       {"params": list(self._deformation.get_grid_parameters()), "lr": a.grid_lr_init * s, "name": "grid"}]
        groups += [{"params": [self._features_dc], "lr": a.feature_lr, "name": "f_dc"},
                   {"params": [self._features_rest], "lr": a.feature_lr / 20.0, "name": "f_rest"},
                   {"params": [self._opacity], "lr": a.opacity_lr, "name": "opacity"},
                   {"params": [self._scaling], "lr": a.scaling_lr, "name": "scaling"},
                   {"params": [self._rotation], "lr": a.rotation_lr, "name": "rotation"},
                   {"params": [self._embedding], "lr": a.feature_lr, "name": "embedding"}]
        if self._sky_model is not None:
            groups.append({"params": [self._sky_model.sky_cube_map], "lr": a.sky_cube_map_lr_init, "name": "sky_cube_map"})
        # (`capturable_optimizer`, not a reference option: step counts and learning rates on the device, so that optimizer.step() can be part of
        #  an iteration replayed from a hipGraph -- emd_amd.StepGraphs; update_learning_rate then also uploads the new rates)
        self.optimizer = Adam(groups, lr=0.0, eps=1e-15, capturable=bool(getattr(a, "capturable_optimizer", False)))
        self.xyz_scheduler_args = expon_lr(a.position_lr_init * s, a.position_lr_final * s, lr_delay_mult=a.position_lr_delay_mult,
                                           max_steps=a.position_lr_max_steps)
        self.deformation_scheduler_args = expon_lr(a.deformation_lr_init * s, a.deformation_lr_final * s,
                                                   lr_delay_mult=a.deformation_lr_delay_mult, max_steps=a.position_lr_max_steps)
        self.grid_scheduler_args = expon_lr(a.grid_lr_init * s, a.grid_lr_final * s, lr_delay_mult=a.deformation_lr_delay_mult,
                                            max_steps=a.position_lr_max_steps)
        self.sky_cube_map_scheduler_args = expon_lr(a.sky_cube_map_lr_init, a.sky_cube_map_lr_final, max_steps=a.sky_cube_map_max_steps)

    def update_learning_rate(self, iteration):
        try:
            return self._update_learning_rate(iteration)
        finally:
            if getattr(self.optimizer, "capturable", False):
                self.optimizer.push_lrs()

    def _update_learning_rate(self, iteration):
        lr_pos = None
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                lr_pos = g["lr"] = self.xyz_scheduler_args(iteration)
            if "grid" in g["name"]:
                g["lr"] = self.grid_scheduler_args(iteration)
            elif g["name"] == "deformation":
                g["lr"] = self.deformation_scheduler_args(iteration)
            elif g["name"] == "sky_cube_map":
                g["lr"] = self.sky_cube_map_scheduler_args(iteration)
        return lr_pos

    # ---- checkpoints (gaussian_model.py:74-118) ----------------------------------------------------------------------------------------
    def capture(self):
        sd = lambda m: {} if m is None else m.state_dict()
        return (self.active_sh_degree, self._xyz, sd(self._deformation), self._deformation_table, sd(self._sky_model), self._features_dc,
                self._features_rest, self._scaling, self._rotation, self._opacity, self._embedding, self.max_radii2D,
                self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(), self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, self._xyz, deform_state, self._deformation_table, sky_state, self._features_dc, self._features_rest,
         self._scaling, self._rotation, self._opacity, self._embedding, self.max_radii2D, xyz_gradient_accum, denom, opt_dict,
         self.spatial_lr_scale) = model_args
        if self._deformation is not None:
            self._deformation.load_state_dict(deform_state)
        if self._sky_model is not None:
            self._sky_model.load_state_dict(sky_state)
        self.training_setup(training_args)
        self.xyz_gradient_accum, self.denom = xyz_gradient_accum, denom
        self.optimizer.load_state_dict(opt_dict)

    # ---- PLY (gaussian_model.py:245-298, 378-425) ----------------------------------------------------------------------------------------
    def construct_list_of_attributes(self):
        l = ["x", "y", "z", "nx", "ny", "nz"]
        l += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        l += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        l.append("opacity")
        l += [f"scale_{i}" for i in range(self._scaling.shape[1])]
        l += [f"rot_{i}" for i in range(self._rotation.shape[1])]
        l += [f"embedding_{i}" for i in range(self._embedding.shape[1])]
        return l

    def save_ply(self, path):
        c = lambda t: t.detach().cpu().numpy()
        xyz = c(self._xyz)
        cols = [xyz, np.zeros_like(xyz), c(self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous()),
                c(self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous()), c(self._opacity), c(self._scaling),
                c(self._rotation), c(self._embedding)]
        write_ply(path, self.construct_list_of_attributes(), np.concatenate(cols, axis=1))

    def load_ply(self, path):
        d = read_ply(path)
        n = d["x"].shape[0]
        col = lambda names: np.stack([np.asarray(d[k], np.float32) for k in names], axis=1) if names else np.zeros((n, 0), np.float32)
        by_index = lambda prefix: sorted((k for k in d if k.startswith(prefix)), key=lambda x: int(x.split("_")[-1]))
        rest_names = by_index("f_rest_")
        assert len(rest_names) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        f_dc = col(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(n, 3, 1)
        f_rest = col(rest_names).reshape(n, 3, (self.max_sh_degree + 1) ** 2 - 1)
        P = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float, device=self.device).requires_grad_(True))
        self._xyz = P(col(["x", "y", "z"]))
        self._features_dc = nn.Parameter(torch.tensor(f_dc, dtype=torch.float, device=self.device).transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(torch.tensor(f_rest, dtype=torch.float, device=self.device).transpose(1, 2).contiguous().requires_grad_(True))
        self._opacity = P(col(["opacity"]))
        self._scaling = P(col(by_index("scale_")))
        self._rotation = P(col(by_index("rot")))
        self._embedding = P(col(by_index("embedding")))
        self.active_sh_degree = self.max_sh_degree
        self.max_radii2D = torch.zeros(n, device=self.device)
        self._deformation_table = torch.ones(n, dtype=torch.bool, device=self.device)

    # ---- statistics (gaussian_model.py:728-730, train.py:403-406) -----------------------------------------------------------------------
    def add_densification_stats(self, viewspace_point_tensor_grad, radii):
        """xyz_gradient_accum += |grad.xy|, denom += 1, max_radii2D = max(., radii) where radii > 0: one launch, no mask sync."""
        from . import dp
        dp.add_densification_stats(viewspace_point_tensor_grad, radii, self.xyz_gradient_accum, self.denom, self.max_radii2D)

    # ---- device-side surgery ---------------------------------------------------------------------------------------------------------
    def _state(self, group):
        st = self.optimizer.state.get(group["params"][0], None) if self.optimizer is not None else None
        return st if st else None

    def _point_groups(self):
        if self.optimizer is None:
            return {}
        return {g["name"]: g for g in self.optimizer.param_groups if g["name"] in self.GROUPS and len(g["params"]) == 1}

    def _restructure(self, mode, args, samples=None):
        """One event: decide -> scan -> index -> gather (see the module docstring).  Returns (n_keep, n_clone, n_split)."""
        lib, dev = L.load(), self.device
        N = self._xyz.shape[0]
        if N == 0:
            return 0, 0, 0
        code = torch.empty(N, dtype=torch.int32, device=dev)
        nblocks = (N + 255) // 256
        inc = torch.empty(3, nblocks, dtype=torch.int32, device=dev)          # per-block counts, then exclusive block offsets
        totals = torch.empty(3, dtype=torch.int32, device=dev)
        args.num_points, args.mode = N, mode
        L.check(lib.emd_densify_decide(C.byref(args), code.data_ptr(), inc.data_ptr(), _stream()), "emd_densify_decide")
        L.check(lib.emd_densify_scan(N, 3, inc.data_ptr(), totals.data_ptr(), _stream()), "emd_densify_scan")
        n_keep, n_clone, n_split = (int(v) for v in totals.tolist())          # the event's single host read
        if mode == L.DENSIFY_MODE_DENSIFY and n_clone == 0 and n_split == 0:
            # nothing selected: the reference's densify_and_clone still runs densification_postfix, which clears the three statistics
            # (gaussian_model.py:526-530) -- stale ones would leak into the next interval and into prune()'s max_radii2D test
            for t in (getattr(self, "xyz_gradient_accum", None), getattr(self, "denom", None), getattr(self, "max_radii2D", None)):
                if t is not None:
                    t.zero_()
            return n_keep, 0, 0
        if mode == L.DENSIFY_MODE_PRUNE and n_keep == N:
            return n_keep, 0, 0
        M = n_keep + n_clone + 2 * n_split
        src = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
        kind = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
        L.check(lib.emd_densify_index(N, M, code.data_ptr(), inc.data_ptr(), totals.data_ptr(), src.data_ptr(), kind.data_ptr(), _stream()), "emd_densify_index")
        g = L.EmdDensifyGather()
        g.num_out, g.mode, g.num_split = M, mode, n_split
        g.src, g.kind = src.data_ptr(), kind.data_ptr()
        g.scaling, g.rotation = self._scaling.data_ptr(), self._rotation.data_ptr()
        g.seed = (self.densify_seed * 0x9E3779B97F4A7C15 + self.densify_events) & 0xFFFFFFFFFFFFFFFF
        keep_alive = [src, kind, code, inc, totals]
        if samples is not None:
            samples = samples.to(dev).float().contiguous()
            assert samples.shape == (2, n_split, 3), (tuple(samples.shape), n_split)
            rank = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
            L.check(lib.emd_densify_split_rank(M, n_keep, n_clone, n_split, rank.data_ptr(), _stream()), "emd_densify_split_rank")
            g.samples, g.split_rank = samples.data_ptr(), rank.data_ptr()
            keep_alive += [samples, rank]
        jobs = []          # (source tensor, role, setter)
        roles = {"xyz": L.DENSIFY_ROLE_XYZ, "scaling": L.DENSIFY_ROLE_SCALING}
        groups = self._point_groups()
        for name in self.GROUPS:
            attr = self._ATTR[name]
            p = getattr(self, attr)
            jobs.append((p, roles.get(name, L.DENSIFY_ROLE_COPY), ("param", name)))
            st = self._state(groups[name]) if name in groups else None
            if st is not None:
                jobs.append((st["exp_avg"], L.DENSIFY_ROLE_STATE, ("exp_avg", name)))
                jobs.append((st["exp_avg_sq"], L.DENSIFY_ROLE_STATE, ("exp_avg_sq", name)))
        for attr in ("xyz_gradient_accum", "denom", "max_radii2D"):
            t = getattr(self, attr)
            if t.dim() >= 1 and t.shape[0] == N:
                jobs.append((t, L.DENSIFY_ROLE_ZERO, ("stat", attr)))
        table_f = self._deformation_table.to(torch.float32)
        jobs.append((table_f, L.DENSIFY_ROLE_COPY, ("table", None)))
        assert len(jobs) <= L.DENSIFY_MAX_TENSORS
        outs = []
        for k, (t, role, _) in enumerate(jobs):
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            width = t.numel() // N
            out = torch.empty((M,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev)
            g.tensors[k].src, g.tensors[k].dst, g.tensors[k].width, g.tensors[k].role = t.data_ptr(), out.data_ptr(), width, role
            keep_alive.append(t)
            outs.append(out)
        g.num_tensors = len(jobs)
        L.check(lib.emd_densify_gather(C.byref(g), _stream()), "emd_densify_gather")
        # install the new tensors: parameters become fresh leaves, the optimiser keeps its groups and per-parameter state
        new_params, new_state = {}, {}
        for (t, role, (what, name)), out in zip(jobs, outs):
            if what == "param":
                new_params[name] = nn.Parameter(out.requires_grad_(True))
            elif what in ("exp_avg", "exp_avg_sq"):
                new_state.setdefault(name, {})[what] = out
            elif what == "stat":
                setattr(self, name, out)
            else:
                self._deformation_table = out > 0.5
        for name in self.GROUPS:
            old = getattr(self, self._ATTR[name])
            if name in groups:
                grp = groups[name]
                st = self.optimizer.state.pop(grp["params"][0], None)
                grp["params"][0] = new_params[name]
                if st:
                    st["exp_avg"], st["exp_avg_sq"] = new_state[name]["exp_avg"], new_state[name]["exp_avg_sq"]
                    self.optimizer.state[new_params[name]] = st
            setattr(self, self._ATTR[name], new_params[name])
            del old
        if mode == L.DENSIFY_MODE_DENSIFY:
            self.densify_events += 1
        return n_keep, n_clone, n_split

    def densify(self, max_grad, min_opacity, extent, max_screen_size, density_threshold=None, displacement_scale=None, model_path=None,
                iteration=None, stage=None, samples=None):
        """grads = xyz_gradient_accum / denom (NaN -> 0); clone where grads >= max_grad and the Gaussian is small
        (max scale <= percent_dense * extent), split into two samples where it is large; the statistics restart at zero.
        (`min_opacity`, `max_screen_size` and the trailing arguments are accepted and unused, as in the reference's densify.)
        `samples [2, n_split, 3]`: standard normals to use instead of the Philox draw (tests)."""
        a = L.EmdDensifyArgs()
        a.scaling, a.grad_accum, a.denom = self._scaling.data_ptr(), self.xyz_gradient_accum.data_ptr(), self.denom.data_ptr()
        a.grad_threshold, a.percent_dense, a.scene_extent = float(max_grad), float(self.percent_dense), float(extent)
        return self._restructure(L.DENSIFY_MODE_DENSIFY, a, samples)

    def prune(self, max_grad, min_opacity, extent, max_screen_size):
        """Drop Gaussians with opacity < min_opacity and, once `max_screen_size` is set, those larger than it on screen
        (max_radii2D) or larger than 0.1 * extent in the world; statistics and Adam moments of the survivors are kept."""
        return self._prune(min_opacity, extent, max_screen_size, None)

    def prune_points(self, mask):
        """Drop the rows where `mask` is True (prune_points of the reference, for callers with their own criterion)."""
        return self._prune(-1.0, 1.0, None, mask)

    def _prune(self, min_opacity, extent, max_screen_size, mask):
        a = L.EmdDensifyArgs()
        a.scaling, a.opacity = self._scaling.data_ptr(), self._opacity.data_ptr()
        a.max_radii2D = self.max_radii2D.data_ptr()
        m8 = None
        if mask is not None:
            m8 = mask.to(self.device).to(torch.uint8).contiguous()
            a.extra_drop = m8.data_ptr()
        a.min_opacity, a.scene_extent = float(min_opacity), float(extent)
        a.max_screen_size = float(max_screen_size) if max_screen_size else 0.0
        return self._restructure(L.DENSIFY_MODE_PRUNE, a)

    def reset_opacity(self):
        """opacity <- logit(min(sigmoid(opacity), 0.01)); its Adam moments restart at zero (gaussian_model.py:373-376,427-439)."""
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self._opacity) * 0.01)).detach()
        p = nn.Parameter(new.requires_grad_(True))
        grp = self._point_groups().get("opacity")
        if grp is not None:
            st = self.optimizer.state.pop(grp["params"][0], None)
            grp["params"][0] = p
            if st:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
                self.optimizer.state[p] = st
        self._opacity = p
