"""ctypes binding of libemd_raster.so (include/emd_raster.h).

The library is built in-tree by `emd_amd.build.build_native()` (hipcc --offload-arch=gfx950).  There is NO
fallback: if the shared object is missing or fails to load, every operator raises -- a silent CPU/eager path
would void the parity claims of this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libemd_raster.so")

ABI_VERSION = 27
MAX_EXTRA = 2
SETTINGS_DEV_FLOATS = 38
TILE = 16
ACTOR_STRIDE = 12
BWD_STRIDE = 12

EMD_OK, EMD_ERR_INVALID, EMD_ERR_CAPACITY, EMD_ERR_HIP, EMD_ERR_WORKSPACE, EMD_ERR_DEPTH_RANGE = 0, -1, -2, -3, -4, -5
FLAG_NORMAL, FLAG_MOTION, FLAG_ABSGRAD, FLAG_NO_SYNC, FLAG_CLAMP_RGB01, FLAG_RAW_PARAMS, FLAG_SDEV_TANFOV = 1, 2, 4, 8, 16, 32, 64
FLAG_WIDE_DEPTH_SORT = 128
FLAG_BWD_WS_CLEAN = 256
FLAG_KEEP_ALL_PAIRS = 512
FLAG_BWD_RENDER_ONLY = 1024
FLAG_BWD_PROJECT_ONLY = 2048

_f = C.c_void_p  # device pointers are passed as integers


class EmdSettings(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float),
                ("tanfovy", C.c_float), ("bg", C.c_float * 3), ("scale_modifier", C.c_float),
                ("viewmatrix", C.c_float * 16), ("projmatrix", C.c_float * 16), ("sh_degree", C.c_int32),
                ("campos", C.c_float * 3), ("prefiltered", C.c_int32), ("debug", C.c_int32),
                ("near_plane", C.c_float)]


class EmdMotion(C.Structure):
    _fields_ = [("actor_id", _f), ("actor_pose", _f), ("num_actors", C.c_int32), ("residual_dx", _f),
                ("residual_dq", _f)]


class EmdDims(C.Structure):
    _fields_ = [("num_gaussians", C.c_int32), ("image_height", C.c_int32), ("image_width", C.c_int32),
                ("bin_capacity", C.c_int64), ("flags", C.c_int32), ("num_extra", C.c_int32)]


class EmdFwdArgs(C.Structure):
    _fields_ = [("s", EmdSettings), ("num_gaussians", C.c_int32), ("sh_coeffs", C.c_int32), ("flags", C.c_int32),
                ("bin_capacity", C.c_int64),
                ("means3D", _f), ("shs", _f), ("colors_precomp", _f), ("opacities", _f), ("scales", _f),
                ("rotations", _f), ("cov3D_precomp", _f), ("motion", EmdMotion),
                ("out_color", _f), ("out_depth", _f), ("out_normal", _f), ("out_alpha", _f), ("radii", _f),
                ("geom_ws", _f), ("geom_bytes", C.c_size_t), ("bin_ws", _f), ("bin_bytes", C.c_size_t),
                ("img_ws", _f), ("img_bytes", C.c_size_t), ("status", _f),
                ("num_rendered", C.c_int64), ("num_visible", C.c_int64), ("settings_dev", _f),
                ("num_extra", C.c_int32), ("colors_extra", _f * 2), ("out_extra", _f * 2), ("aux_stream", _f), ("shs_residual", _f * 2), ("loop_stats", _f)]


class EmdBwdArgs(C.Structure):
    _fields_ = [("s", EmdSettings), ("num_gaussians", C.c_int32), ("sh_coeffs", C.c_int32), ("flags", C.c_int32),
                ("bin_capacity", C.c_int64), ("num_rendered", C.c_int64),
                ("means3D", _f), ("shs", _f), ("colors_precomp", _f), ("opacities", _f), ("scales", _f),
                ("rotations", _f), ("cov3D_precomp", _f), ("motion", EmdMotion), ("radii", _f),
                ("geom_ws", _f), ("geom_bytes", C.c_size_t), ("bin_ws", _f), ("bin_bytes", C.c_size_t),
                ("img_ws", _f), ("img_bytes", C.c_size_t), ("status", _f),
                ("out_color", _f), ("out_depth", _f), ("out_normal", _f),
                ("dL_dcolor", _f), ("dL_ddepth", _f), ("dL_dalpha", _f), ("dL_dnormal", _f),
                ("bwd_ws", _f), ("bwd_bytes", C.c_size_t),
                ("dL_dmeans3D", _f), ("dL_dmeans2D", _f), ("dL_dmeans2D_abs", _f), ("dL_dshs", _f),
                ("dL_dcolors", _f), ("dL_dopacities", _f), ("dL_dscales", _f), ("dL_drotations", _f),
                ("dL_dcov3D", _f), ("dL_dactor_pose", _f), ("dL_dresidual_dx", _f), ("dL_dresidual_dq", _f),
                ("dL_dsh_color", _f), ("settings_dev", _f),
                ("num_extra", C.c_int32), ("colors_extra", _f * 2), ("out_extra", _f * 2), ("dL_dextra", _f * 2), ("dL_dcolors_extra", _f * 2),
                ("pair_stats", _f)]


SKY_CLAMP01, SKY_BLEND_S3G, SKY_BLEND_ADD, SKY_INTERLEAVED = 1, 2, 4, 8


class EmdSkyArgs(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("resolution", C.c_int32), ("flags", C.c_int32),
                ("cube", _f), ("dirs", _f), ("Kinv", C.c_float * 9), ("R", C.c_float * 9), ("T", C.c_float * 3),
                ("jitter", _f), ("acc", _f), ("mask_threshold", C.c_float), ("fill", C.c_float), ("fg", _f),
                ("sky", _f), ("out", _f), ("camera_dev", _f)]


class EmdSkyBwdArgs(C.Structure):
    _fields_ = [("f", EmdSkyArgs), ("dL_dout", _f), ("dL_dsky", _f), ("dL_dcube", _f), ("dL_dacc", _f), ("dL_dfg", _f)]


class EmdLossArgs(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("image", _f), ("gt", _f), ("depth", _f), ("gt_depth", _f),
                ("mask", _f), ("weight", _f), ("sky_mask", _f), ("lambda_dssim", C.c_float), ("lambda_depth", C.c_float),
                ("lambda_sky", C.c_float), ("max_depth", C.c_float), ("losses", _f), ("dL_dimage", _f), ("dL_ddepth", _f),
                ("dL_dweight", _f)]


HEX_MAX_SCALES = 8


class EmdHexArgs(C.Structure):
    _fields_ = [("num_points", C.c_int32), ("channels", C.c_int32), ("num_scales", C.c_int32), ("times_broadcast", C.c_int32),
                ("res", (C.c_int32 * 4) * HEX_MAX_SCALES), ("planes", (_f * 6) * HEX_MAX_SCALES), ("pts", _f), ("times", _f),
                ("aabb", C.c_float * 6), ("out", _f), ("order", _f), ("time_tables", _f)]


class EmdHexGrads(C.Structure):
    _fields_ = [("dL_dout", _f), ("dL_dplanes", (_f * 6) * HEX_MAX_SCALES), ("dL_dpts", _f), ("dL_dtimes", _f),
                ("order2d", C.c_void_p * 3), ("pos2d", C.c_void_p * 3), ("defer_rows", _f), ("defer_mask", C.c_uint32), ("reserved", C.c_uint32), ("dL_dtime_sum", _f)]


class EmdDeformInArgs(C.Structure):
    _fields_ = [("num_points", C.c_int32), ("num_freqs_x", C.c_int32), ("num_freqs_t", C.c_int32), ("embed_dim", C.c_int32),
                ("ld", C.c_int32), ("reserved", C.c_int32), ("means", _f), ("point_ids", _f), ("inst_size", _f), ("inst_embed", _f),
                ("t", _f), ("out", _f)]


DENSIFY_MODE_DENSIFY, DENSIFY_MODE_PRUNE, DENSIFY_MODE_REFINE = 0, 1, 2
DENSIFY_ROLE_COPY, DENSIFY_ROLE_XYZ, DENSIFY_ROLE_SCALING, DENSIFY_ROLE_STATE, DENSIFY_ROLE_ZERO = 0, 1, 2, 3, 4
DENSIFY_MAX_TENSORS = 40


class EmdDensifyArgs(C.Structure):
    _fields_ = [("num_points", C.c_int32), ("mode", C.c_int32), ("scaling", _f), ("opacity", _f), ("grad_accum", _f), ("denom", _f),
                ("max_radii2D", _f), ("extra_drop", _f), ("grad_threshold", C.c_float), ("percent_dense", C.c_float),
                ("scene_extent", C.c_float), ("min_opacity", C.c_float), ("max_screen_size", C.c_float)]


class EmdRefineArgs(C.Structure):
    _fields_ = [("num_points", C.c_int32), ("do_densify", C.c_int32), ("do_cull", C.c_int32), ("use_split_screen", C.c_int32), ("cull_big", C.c_int32),
                ("use_cull_screen", C.c_int32), ("scaling", _f), ("opacity", _f), ("grad_norm", _f), ("vis_counts", _f), ("max_2Dsize", _f),
                ("grad_threshold", C.c_float), ("size_threshold", C.c_float), ("split_screen", C.c_float), ("cull_alpha", C.c_float),
                ("cull_size", C.c_float), ("cull_screen", C.c_float)]


class EmdDensifyTensor(C.Structure):
    _fields_ = [("src", _f), ("dst", _f), ("width", C.c_int32), ("role", C.c_int32)]


class EmdDensifyGather(C.Structure):
    _fields_ = [("num_out", C.c_int32), ("num_tensors", C.c_int32), ("mode", C.c_int32), ("num_split", C.c_int32), ("src", _f), ("kind", _f),
                ("scaling", _f), ("rotation", _f), ("seed", C.c_uint64), ("samples", _f), ("split_rank", _f),
                ("tensors", EmdDensifyTensor * DENSIFY_MAX_TENSORS)]


class EmdTrackArgs(C.Structure):
    _fields_ = [("num_actors", C.c_int32), ("rows", C.c_int32), ("dim", C.c_int32), ("embed_dim", C.c_int32), ("k_coarse", C.c_int32),
                ("k_fine", C.c_int32), ("num_points", C.c_int32), ("reserved", C.c_int32), ("t", C.c_float), ("t_dev", _f), ("k_fine_dev", _f), ("weight", _f),
                ("embeddings", _f), ("point_ids", _f), ("count", _f), ("segment_start", _f), ("head_w", _f * 4), ("head_b", _f * 4), ("emb_sum", _f),
                ("trans", _f), ("rot", _f)]


class EmdTrackGrads(C.Structure):
    _fields_ = [("g_trans", _f), ("g_rot", _f), ("d_weight", _f), ("d_embeddings", _f), ("d_head_w", _f * 4), ("d_head_b", _f * 4),
                ("d_mean", _f)]


class EmdStepSelect(C.Structure):
    _fields_ = [("sel", _f), ("rows", C.c_int32), ("row_floats", C.c_int32), ("table", _f), ("out_row", _f), ("frames", _f),
                ("frame_out", _f), ("t_out", _f), ("num_frames", C.c_int32), ("k_min", C.c_int32), ("k_max", C.c_int32),
                ("k_until", C.c_int32), ("steps", _f), ("k_fine_out", _f), ("status", _f), ("status_log", _f), ("prev_sel", _f),
                ("next_sel", _f)]


class EmdTrackedPoseArgs(C.Structure):
    _fields_ = [("track", EmdTrackArgs), ("q_all", _f), ("t_all", _f), ("valid_all", _f), ("num_frames", C.c_int32), ("frame", C.c_int32),
                ("frame_dev", _f), ("pose", _f), ("head_acc", _f), ("head_acc_floats", C.c_int32), ("reserved", C.c_int32)]


class EmdTrackedPoseGrads(C.Structure):
    _fields_ = [("g_pose", _f), ("d_q_all", _f), ("d_t_all", _f), ("d_weight", _f), ("d_embeddings", _f), ("d_head_w", _f * 4),
                ("d_head_b", _f * 4)]


MLP_MAX_BRANCHES = 6


class EmdMlpTrunk(C.Structure):
    _fields_ = [("num_points", C.c_int32), ("ka", C.c_int32), ("kb", C.c_int32), ("ld_w", C.c_int32), ("col_a", C.c_int32),
                ("col_b", C.c_int32), ("reserved0", C.c_int32), ("reserved1", C.c_int32), ("xa", _f), ("xb", _f), ("w", _f), ("b", _f), ("h", _f)]


class EmdMlpTrunkGrads(C.Structure):
    _fields_ = [("num_gh", C.c_int32), ("reserved", C.c_int32), ("g_h", _f * MLP_MAX_BRANCHES), ("d_xa", _f), ("d_xb", _f), ("d_w", _f), ("d_b", _f)]


class EmdMlpBranch(C.Structure):
    _fields_ = [("num_points", C.c_int32), ("depth", C.c_int32), ("relu_input", C.c_int32), ("out_dim", C.c_int32), ("h", _f),
                ("w_hidden", _f * 2), ("b_hidden", _f * 2), ("w_out", _f), ("b_out", _f), ("out", _f), ("l1_sum", _f),
                ("xb", _f), ("w_in", _f), ("b_in", _f), ("kb_in", C.c_int32), ("ld_w_in", C.c_int32), ("col_in", C.c_int32), ("reserved", C.c_int32)]


class EmdMlpBranchGrads(C.Structure):
    _fields_ = [("g_out", _f), ("g_h", _f), ("d_w_hidden", _f * 2), ("d_b_hidden", _f * 2), ("d_w_out", _f), ("d_b_out", _f),
                ("l1_grad", _f), ("out", _f), ("g_h_in", _f)]


ADAM_MAX_TENSORS = 32


class EmdAdamTensor(C.Structure):
    _fields_ = [("param", _f), ("grad", _f), ("exp_avg", _f), ("exp_avg_sq", _f), ("numel", C.c_int64), ("step_size", C.c_float),
                ("bias_correction2_sqrt", C.c_float), ("one_minus_beta1", C.c_float), ("beta2", C.c_float), ("one_minus_beta2", C.c_float),
                ("eps", C.c_float), ("step_dev", _f), ("lr_dev", _f)]


class EmdAdamArgs(C.Structure):
    _fields_ = [("num_tensors", C.c_int32), ("reserved", C.c_int32), ("tensors", EmdAdamTensor * ADAM_MAX_TENSORS)]


# every symbol include/emd_raster.h declares
EXPORTED_SYMBOLS = ("emd_abi_version", "emd_last_error", "emd_raster_workspace_size", "emd_raster_forward",
                    "emd_raster_backward", "emd_raster_export_binning", "emd_raster_export_geometry",
                    "emd_motion_forward", "emd_motion_backward", "emd_sh_forward", "emd_sh_backward",
                    "emd_profile_enable", "emd_profile_read", "emd_profile_stage_name", "emd_activations_forward", "emd_actor_pose_forward", "emd_actor_pose_backward", "emd_l1_loss",
                    "emd_sky_forward", "emd_sky_backward", "emd_image_loss_workspace", "emd_image_loss", "emd_hexplane_forward", "emd_hexplane_backward", "emd_hexplane_order_keys", "emd_sh_grad_from_factors", "emd_densification_stats",
                    "emd_temporal_embed_forward", "emd_temporal_embed_backward", "emd_deform_input_width", "emd_deform_input_forward",
                    "emd_deform_input_backward", "emd_adam_step", "emd_track_heads_forward", "emd_track_heads_backward",
                    "emd_densify_decide", "emd_densify_index", "emd_densify_split_rank", "emd_densify_gather",
                    "emd_refine_decide", "emd_refine_index", "emd_after_train_stats", "emd_densify_scan",
                    "emd_mlp_trunk_forward", "emd_mlp_trunk_backward", "emd_mlp_branch_forward", "emd_mlp_branch_backward",
                    "emd_abs_mean_backward", "emd_residual_l1_backward", "emd_tracked_pose_forward", "emd_tracked_pose_backward",
                    "emd_select_step_inputs", "emd_compact_rows", "emd_scatter_rows", "emd_l1_loss_ws")
PROF_STAGES = 8

_lib = None


class EmdError(RuntimeError):
    pass


def load():
    """Load the HIP extension; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("EMD_LIB_PATH") or LIB_PATH          # (EMD_LIB_PATH: an alternative BUILD of the same library, for A/B timing)
    if not os.path.exists(path):
        raise EmdError(f"HIP extension not built: {path} is missing. Run `python -c 'import __graft_entry__ as g; "
                       f"g.build()'` (or `make -C emd_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name in EXPORTED_SYMBOLS:
        if not hasattr(lib, name):
            raise EmdError(f"{path} does not export {name}")
    lib.emd_abi_version.restype = C.c_int
    lib.emd_last_error.restype = C.c_char_p
    v = lib.emd_abi_version()
    if v != ABI_VERSION:
        raise EmdError(f"ABI mismatch: library {v}, binding {ABI_VERSION}")
    lib.emd_raster_workspace_size.argtypes = [C.POINTER(EmdDims), C.POINTER(C.c_size_t)]
    lib.emd_raster_forward.argtypes = [C.POINTER(EmdFwdArgs), C.c_void_p]
    lib.emd_raster_backward.argtypes = [C.POINTER(EmdBwdArgs), C.c_void_p]
    lib.emd_raster_export_binning.argtypes = [C.POINTER(EmdDims), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_raster_export_geometry.argtypes = [C.POINTER(EmdDims), C.c_void_p, C.c_size_t] + [C.c_void_p] * 7
    lib.emd_motion_forward.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(EmdMotion),
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_motion_backward.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(EmdMotion)] + \
        [C.c_void_p] * 10
    lib.emd_sh_forward.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_sh_backward.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6
    lib.emd_activations_forward.argtypes = [C.c_int32] + [C.c_void_p] * 7
    lib.emd_actor_pose_forward.argtypes = [C.c_int32] + [C.c_void_p] * 8
    lib.emd_actor_pose_backward.argtypes = [C.c_int32] + [C.c_void_p] * 10
    lib.emd_l1_loss.argtypes = [C.c_int64] + [C.c_void_p] * 5
    lib.emd_l1_loss_ws.argtypes = [C.c_int64] + [C.c_void_p] * 6
    lib.emd_profile_enable.argtypes = [C.c_int]
    lib.emd_image_loss_workspace.argtypes = [C.c_int, C.c_int]
    lib.emd_image_loss_workspace.restype = C.c_size_t
    lib.emd_image_loss.argtypes = [C.POINTER(EmdLossArgs), C.c_void_p, C.c_size_t, C.c_void_p]
    lib.emd_sh_grad_from_factors.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(EmdMotion), C.c_int32,
                                             C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    lib.emd_densification_stats.argtypes = [C.c_int32] + [C.c_void_p] * 6
    lib.emd_compact_rows.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
    lib.emd_scatter_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.c_float,
                                     C.c_void_p, C.c_void_p]
    lib.emd_hexplane_forward.argtypes = [C.POINTER(EmdHexArgs), C.c_void_p]
    lib.emd_hexplane_backward.argtypes = [C.POINTER(EmdHexArgs), C.POINTER(EmdHexGrads), C.c_void_p]
    lib.emd_hexplane_order_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.emd_temporal_embed_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_temporal_embed_backward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
    lib.emd_deform_input_width.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.emd_deform_input_forward.argtypes = [C.POINTER(EmdDeformInArgs), C.c_void_p]
    lib.emd_deform_input_backward.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_adam_step.argtypes = [C.POINTER(EmdAdamArgs), C.c_void_p]
    lib.emd_densify_decide.argtypes = [C.POINTER(EmdDensifyArgs), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_densify_index.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_densify_scan.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_densify_split_rank.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.emd_densify_gather.argtypes = [C.POINTER(EmdDensifyGather), C.c_void_p]
    lib.emd_refine_decide.argtypes = [C.POINTER(EmdRefineArgs), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_refine_index.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_after_train_stats.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    lib.emd_abs_mean_backward.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emd_residual_l1_backward.argtypes = [C.c_int64] + [C.c_void_p] * 9
    lib.emd_mlp_trunk_forward.argtypes = [C.POINTER(EmdMlpTrunk), C.c_void_p]
    lib.emd_mlp_trunk_backward.argtypes = [C.POINTER(EmdMlpTrunk), C.POINTER(EmdMlpTrunkGrads), C.c_void_p]
    lib.emd_mlp_branch_forward.argtypes = [C.POINTER(EmdMlpBranch), C.c_void_p]
    lib.emd_mlp_branch_backward.argtypes = [C.POINTER(EmdMlpBranch), C.POINTER(EmdMlpBranchGrads), C.c_void_p]
    lib.emd_track_heads_forward.argtypes = [C.POINTER(EmdTrackArgs), C.c_void_p]
    lib.emd_track_heads_backward.argtypes = [C.POINTER(EmdTrackArgs), C.POINTER(EmdTrackGrads), C.c_void_p]
    lib.emd_select_step_inputs.argtypes = [C.POINTER(EmdStepSelect), C.c_void_p]
    lib.emd_tracked_pose_forward.argtypes = [C.POINTER(EmdTrackedPoseArgs), C.c_void_p]
    lib.emd_tracked_pose_backward.argtypes = [C.POINTER(EmdTrackedPoseArgs), C.POINTER(EmdTrackedPoseGrads), C.c_void_p]
    lib.emd_sky_forward.argtypes = [C.POINTER(EmdSkyArgs), C.c_void_p]
    lib.emd_sky_backward.argtypes = [C.POINTER(EmdSkyBwdArgs), C.c_void_p]
    lib.emd_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    lib.emd_profile_stage_name.argtypes = [C.c_int]
    lib.emd_profile_stage_name.restype = C.c_char_p
    _lib = lib
    return lib


def profile_enable(on=True):
    load().emd_profile_enable(1 if on else 0)


def profile_read():
    """{stage name: (total ms, launches)} since the last read (HIP events on the launch stream)."""
    lib = load()
    ms = (C.c_double * PROF_STAGES)()
    cnt = (C.c_int64 * PROF_STAGES)()
    lib.emd_profile_read(ms, cnt, PROF_STAGES)
    return {lib.emd_profile_stage_name(i).decode(): (ms[i], cnt[i]) for i in range(PROF_STAGES)}


def check(rc, what):
    if rc != 0:
        msg = load().emd_last_error().decode("utf-8", "replace")
        raise EmdError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def workspace_sizes(N, H, W, capacity, flags=0, num_extra=0):
    d = EmdDims(int(N), int(H), int(W), int(capacity), int(flags), int(num_extra))
    out = (C.c_size_t * 4)()
    check(load().emd_raster_workspace_size(C.byref(d), out), "emd_raster_workspace_size")
    return tuple(int(x) for x in out)
