/*
 * emd_raster.h -- C ABI of the MI355X (gfx950) street-Gaussian rasterizer.
 *
 * This is the drop-in boundary for EMD's hot path.  The reference binds this path
 * through a third-party CUDA extension that is NOT vendored in the reference tree
 * (`from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer`,
 * S3Gaussian/gaussian_renderer/__init__.py:14; `gsplat.rendering.rasterization`,
 * OmniRe/models/gaussians/basics.py:12).  The entry points below are what a Python
 * `ctypes` binding of that path binds instead:
 *
 *   emd_raster_forward   <- GaussianRasterizer.forward(...)   S3Gaussian/gaussian_renderer/__init__.py:145-155
 *                           rasterization(...)                OmniRe/models/trainers/base.py:393-408
 *   emd_raster_backward  <- autograd of the same call         S3Gaussian/train.py:366
 *   emd_motion_forward / emd_motion_backward
 *                        <- RigidNodes.transform_means/quats  OmniRe/models/nodes/rigid.py:478-568
 *                           DeformableNodes residual add      OmniRe/models/nodes/deformable.py:57-69
 *                           (when EMD_FLAG_MOTION is set the same transform is fused
 *                            into the projection kernel of emd_raster_forward)
 *   emd_sh_forward / emd_sh_backward
 *                        <- gsplat spherical_harmonics(...)   OmniRe/models/nodes/rigid.py:584
 *
 * and, further down, the callers either side of that path, each with its own block comment:
 *   emd_sky_forward / backward          <- SkyCubeMap.forward + blend    S3Gaussian/scene/sky_cubemap.py:41-87, gaussian_renderer/__init__.py:299-301
 *                                          EnvLight.forward              OmniRe/models/modules.py:174-208
 *   emd_image_loss                      <- l1 + D-SSIM + depth + sky BCE S3Gaussian/utils/loss_utils.py:21-98, train.py:226-363
 *   emd_hexplane_forward / backward     <- HexPlaneField.get_density     S3Gaussian/scene/hexplane.py:18-183
 *   emd_mlp_trunk / emd_mlp_branch      <- feature_out + *_deform heads + dino_head  S3Gaussian/scene/deformation.py:100-185,254-337
 *   emd_temporal_embed_forward/backward <- get_temporal_embed            S3Gaussian/scene/deformation.py:208-221, OmniRe/models/nodes/rigid.py:150-164
 *   emd_deform_input_forward/backward   <- get_embedder + get_deformation OmniRe/models/modules.py:318-366, nodes/deformable.py:35-47
 *   emd_densification_stats             <- add_densification_stats       S3Gaussian/scene/gaussian_model.py:728-730, train.py:403-406
 *   emd_adam_step                       <- optimizer.step()              S3Gaussian/scene/gaussian_model.py:188-201, train.py:428
 *   emd_sh_grad_from_factors            (multi-GPU: rebuilds the SH gradient from all-gathered rank-one factors; no reference counterpart)
 *   emd_compact_rows / emd_scatter_rows (multi-GPU: visibility-compacted gradient rows for 2 / 4 ranks; no reference counterpart)
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions across the ABI; every pointer is a DEVICE
 *     pointer to contiguous fp32/int32 memory unless marked "host".
 *   - the library never allocates or frees device memory: all outputs, scratch and
 *     saved-for-backward state live in caller-owned buffers whose sizes come from
 *     emd_raster_workspace_size().
 *   - every function returns 0 on success, a negative EMD_ERR_* code otherwise;
 *     emd_last_error() returns a thread-local message for the last failure.
 *   - kernels are enqueued on the hipStream_t passed by the caller (as void*).
 *   - matrices use the row-vector convention of S3Gaussian/scene/cameras.py:61-65:
 *     viewmatrix = W2C^T, projmatrix = viewmatrix @ P^T, both row-major 4x4, so that
 *     p_view = [x y z 1] @ viewmatrix.
 */
#ifndef EMD_RASTER_H
#define EMD_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMD_ABI_VERSION 27

/* tile geometry is part of the sort-key contract (tile_id << 32 | depth bits) */
#define EMD_TILE_X 16
#define EMD_TILE_Y 16

enum {
    EMD_OK = 0,
    EMD_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, both/neither of shs & colors...) */
    EMD_ERR_CAPACITY = -2,  /* bin_capacity too small; EmdFwdArgs.num_rendered holds the needed count */
    EMD_ERR_HIP = -3,       /* a HIP runtime call or kernel launch failed */
    EMD_ERR_WORKSPACE = -4, /* a workspace buffer is smaller than emd_raster_workspace_size() reports */
    EMD_ERR_DEPTH_RANGE = -5 /* a visible Gaussian lies more than 65 536 x the near plane away: the three-pass depth sort does not
                                cover it; call again with EMD_FLAG_WIDE_DEPTH_SORT */
};

enum {
    EMD_FLAG_NORMAL   = 1 << 0, /* also composite the view-space normal image (out_normal) */
    EMD_FLAG_MOTION   = 1 << 1, /* fuse the per-actor rigid transform + residual in front of the projection */
    EMD_FLAG_ABSGRAD  = 1 << 2, /* backward also accumulates sum |d L/d mean2D| (gsplat absgrad) */
    EMD_FLAG_NO_SYNC  = 1 << 3, /* never read the duplicate count back to the host; overflow is reported
                                   through EmdStatus.overflow only (graph-capturable path) */
    EMD_FLAG_CLAMP_RGB01 = 1 << 4, /* SH colour clamped to [0,1] (OmniRe, rigid.py:585) instead of >= 0 */
    EMD_FLAG_RAW_PARAMS = 1 << 5,  /* inputs are the raw parameters: scales = log-scales (exp applied here), rotations
                                      un-normalised (normalised here), opacities = logits (sigmoid here); gradients are
                                      returned w.r.t. the raw parameters.  Fuses S3Gaussian/gaussian_renderer/__init__.py:99-101 */
    EMD_FLAG_WIDE_DEPTH_SORT = 1 << 7, /* sort the Gaussians by all 32 depth bits in four passes instead of by the 27 bits above the
                                      near plane in three (needed only when depths exceed 65 536 x the near plane: EMD_ERR_DEPTH_RANGE) */
    EMD_FLAG_SDEV_TANFOV = 1 << 6, /* settings_dev holds two more floats, tanfovx and tanfovy (cameras given as device-resident
                                      intrinsics, OmniRe/models/trainers/base.py:399-400): they replace the by-value fields */
    EMD_FLAG_BWD_WS_CLEAN = 1 << 8, /* backward only (ABI 18): bwd_ws ARRIVES zero-filled (the caller's promise) and is LEFT zero-filled --
                                      the projection backward clears every accumulator row it reads, so a caller that keeps one
                                      workspace across steps pays no 48 N-byte zero fill per backward.  Without the flag the library
                                      clears the workspace itself, as before. */
    EMD_FLAG_BWD_RENDER_ONLY = 1 << 10, /* emd_raster_backward (ABI 21): only the render backward (K7).  With dL_dsh_color set, the clamp-masked
                                      colour gradient of every Gaussian -- the factor of its rank-one dL/dshs -- is extracted from the
                                      accumulator rows right behind it (one small launch), so that a view-parallel step can start the
                                      all-gather of the factors while the projection backward is still to run */
    EMD_FLAG_BWD_PROJECT_ONLY = 1 << 11, /* ... and only the projection backward (K8) on the accumulator rows of such a call: the two halves
                                      of one backward as two calls, with whatever the caller enqueues in between (emd_amd.dp: collectives
                                      on the communication stream).  dL_dsh_color is then left alone (pass NULL). */
    EMD_FLAG_KEEP_ALL_PAIRS = 1 << 9 /* (ABI 21) enumerate every tile of upstream's tile rectangle (the 3-sigma square of the largest
                                      eigenvalue): the sorted keys then ARE upstream's (tile << 32 | depth bits) list, entry for entry.
                                      By default a Gaussian enumerates only the tiles of that rectangle which its alpha >= 1/255 bounding
                                      box (half extents sqrt(2 ln(255 o) cov_xx), sqrt(2 ln(255 o) cov_yy)) reaches: the pairs left out are
                                      pairs upstream's render loop skips at every pixel of the tile (its alpha < 1/255 `continue`), so
                                      images, radii and gradients are the same bit for bit and the sorted list is upstream's list with
                                      those entries left out -- same keys, same order, 25 % fewer entries on the street scenes of
                                      BASELINE.json. */
};

/* The 12 fields of GaussianRasterizationSettings (S3Gaussian/gaussian_renderer/__init__.py:49-62), by value. */
typedef struct EmdSettings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float bg[3];
    float scale_modifier;
    float viewmatrix[16];
    float projmatrix[16];
    int32_t sh_degree;      /* active degree 0..3 */
    float campos[3];
    int32_t prefiltered;    /* accepted, ignored (reference always passes False) */
    int32_t debug;          /* !=0: synchronise and check after every stage */
    float near_plane;       /* cull view z <= near_plane; 0.2 for the diff_gauss surface, 0.1 for OmniRe */
} EmdSettings;

/* Per-actor pose row for the fused explicit-motion transform (12 floats, 48 B):
 *   q_mean[4]  unit quaternion (w,x,y,z) rotating local means     rigid.py:499-503
 *   trans[3]   translation incl. learned track offset             rigid.py:519-532
 *   valid      1.0 / 0.0, multiplies opacity                      rigid.py:589-591
 *   q_rot[4]   unit quaternion composed onto local quats (q_mean x track rot offset, rigid.py:562-566)
 */
#define EMD_ACTOR_STRIDE 12
#define EMD_SETTINGS_DEV_FLOATS 38
#define EMD_MAX_EXTRA 2          /* extra colour sets composited by one call (S3Gaussian renders feat_c and feat_f) */

typedef struct EmdMotion {
    const int32_t* actor_id;   /* [N]; -1 = static (identity) */
    const float* actor_pose;   /* [A, EMD_ACTOR_STRIDE] */
    int32_t num_actors;
    const float* residual_dx;  /* [N,3] or NULL: added to the local mean before the rigid transform */
    const float* residual_dq;  /* [N,4] or NULL: added to the local quaternion before normalisation */
} EmdMotion;

/* Small device-side status block written by the forward pass. */
typedef struct EmdStatus {
    uint32_t num_rendered;  /* D = (tile, Gaussian) pairs of the sorted list: the sum over the Gaussians of the tiles their alpha >= 1/255
                               bounding box reaches inside upstream's tile rectangle (ABI 21; with EMD_FLAG_KEEP_ALL_PAIRS: upstream's
                               sum of tiles touched) */
    uint32_t overflow;      /* bit 0: D > bin_capacity; bit 1: depth range beyond the three-pass sort (either: results invalid,
                               the image is the background) */
    uint32_t num_visible;   /* V = Gaussians with radii > 0 */
    uint32_t reserved;      /* internal: visible Gaussians as counted by the depth sort's compacting first pass (== num_visible) */
} EmdStatus;

typedef struct EmdDims {
    int32_t num_gaussians;
    int32_t image_height;
    int32_t image_width;
    int64_t bin_capacity;   /* max (tile, Gaussian) pairs the binning workspace can hold */
    int32_t flags;
    int32_t num_extra;      /* extra colour sets of the call (sizes bwd_ws) */
} EmdDims;

typedef struct EmdFwdArgs {
    EmdSettings s;
    int32_t num_gaussians;
    int32_t sh_coeffs;            /* coefficients stored per Gaussian in shs ([N, sh_coeffs, 3]) */
    int32_t flags;
    int64_t bin_capacity;
    /* inputs */
    const float* means3D;         /* [N,3] (local means when EMD_FLAG_MOTION) */
    const float* shs;             /* [N,sh_coeffs,3] or NULL */
    const float* colors_precomp;  /* [N,3] or NULL; exactly one of shs / colors_precomp */
    const float* opacities;       /* [N] activated */
    const float* scales;          /* [N,3] activated, or NULL */
    const float* rotations;       /* [N,4] (w,x,y,z), or NULL */
    const float* cov3D_precomp;   /* [N,6] or NULL; exactly one of (scales,rotations) / cov3D_precomp */
    EmdMotion motion;
    /* outputs */
    float* out_color;             /* [3,H,W] */
    float* out_depth;             /* [1,H,W] */
    float* out_normal;            /* [3,H,W] (EMD_FLAG_NORMAL) or NULL */
    float* out_alpha;             /* [1,H,W] */
    int32_t* radii;               /* [N] */
    /* caller-owned scratch; kept alive by the caller until backward has run */
    void* geom_ws;  size_t geom_bytes;
    void* bin_ws;   size_t bin_bytes;
    void* img_ws;   size_t img_bytes;
    EmdStatus* status;            /* device, 16 B */
    /* host-side results (filled unless EMD_FLAG_NO_SYNC) */
    int64_t num_rendered;
    int64_t num_visible;
    /* Optional DEVICE copy of the camera-dependent settings, EMD_SETTINGS_DEV_FLOATS floats laid out as
     * bg[3], viewmatrix[16], projmatrix[16], campos[3] (+ tanfovx, tanfovy with EMD_FLAG_SDEV_TANFOV); NULL = use the by-value
     * fields of `s`.  The reference keeps exactly
     * these four on the GPU (`.cuda()` at S3Gaussian/gaussian_renderer/__init__.py:54-59): with this pointer its call site
     * costs no device-to-host copy, so a forward with EMD_FLAG_NO_SYNC never touches the host. */
    const float* settings_dev;
    /* Extra colour sets: the reference's "fine" stage renders the same Gaussians two more times with colors_precomp = ddict["coarse"]["feat"]
     * and ddict["fine"]["feat"] (S3Gaussian/gaussian_renderer/__init__.py:170-201).  With num_extra > 0 those images come out of THIS
     * call: one projection, one sort, one walk over every tile list; out_extra[k] equals, bit for bit, the colour image of a separate
     * call with colors_precomp = colors_extra[k]. */
    int32_t num_extra;                       /* 0 .. EMD_MAX_EXTRA */
    const float* colors_extra[EMD_MAX_EXTRA];  /* [N,3] each */
    float* out_extra[EMD_MAX_EXTRA];           /* [3,H,W] each */
    /* optional second hipStream_t (ABI 18).  With it the forward splits its projection kernel: the geometry half on the call's stream,
     * the colour half (SH colour, clamp bits, colour Jacobian -- read by the compositing kernel only) on aux_stream, beside the binning
     * stage; fork and join are events, i.e. graph edges under stream capture.  The stream must belong to the same device and must not be
     * the call's own; NULL (or debug = 1) keeps everything on one stream.  Same results bit for bit either way. */
    void* aux_stream;
    /* Optional residuals of the SH coefficients (ABI 20), [N,sh_coeffs,3] each or NULL: the colour is evaluated on
     * (shs + shs_residual[0]) + shs_residual[1], added in that order while the rows are staged.  The reference's fine stage forms
     * `shs_final = shs + dshs_coarse + dshs_fine` as two element-wise passes over [N,16,3] before every render
     * (S3Gaussian/scene/deformation.py:468-481); here the sum exists only for the Gaussians that are visible, inside the projection
     * kernel.  dL/dshs of the backward is the gradient of each of the three terms (the sum's Jacobian is the identity). */
    const float* shs_residual[2];
    /* diagnostic (ABI 27): device uint64[6], ADDED to by the compositing kernel's counting instantiation: [0] scan steps (64 list words each),
     * [1] cull steps (64 queued entries each), [2] drain iterations (one entry per 16-pixel row each), [3] entries the rows held, [4] scan -> cull ->
     * drain rounds, [5] waves that ran.  NULL in every training path (the counters are compiled out); num_extra must be 0. */
    uint64_t* loop_stats;
} EmdFwdArgs;

typedef struct EmdBwdArgs {
    EmdSettings s;
    int32_t num_gaussians;
    int32_t sh_coeffs;
    int32_t flags;
    int64_t bin_capacity;
    int64_t num_rendered;         /* D from forward; <0 = read it from status on device */
    /* forward inputs again */
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    EmdMotion motion;
    const int32_t* radii;
    /* forward state */
    const void* geom_ws;  size_t geom_bytes;
    const void* bin_ws;   size_t bin_bytes;
    const void* img_ws;   size_t img_bytes;
    const EmdStatus* status;
    /* forward outputs again (the backward pass needs the composited totals per pixel) */
    const float* out_color;       /* [3,H,W] */
    const float* out_depth;       /* [1,H,W] */
    const float* out_normal;      /* [3,H,W] or NULL */
    /* incoming gradients (any may be NULL = zero) */
    const float* dL_dcolor;       /* [3,H,W] */
    const float* dL_ddepth;       /* [1,H,W] */
    const float* dL_dalpha;       /* [1,H,W] */
    const float* dL_dnormal;      /* [3,H,W]  (propagated to the blended normal only) */
    /* scratch for backward: [N, 12 + 4 num_extra] floats, zeroed by the library */
    void* bwd_ws;  size_t bwd_bytes;
    /* outgoing gradients (NULL = not wanted) */
    float* dL_dmeans3D;           /* [N,3] */
    float* dL_dmeans2D;           /* [N,3]  x,y in NDC-scaled pixel units (x 0.5 W, 0.5 H), z = 0 */
    float* dL_dmeans2D_abs;       /* [N,2]  (EMD_FLAG_ABSGRAD) */
    float* dL_dshs;               /* [N,sh_coeffs,3] */
    float* dL_dcolors;            /* [N,3] */
    float* dL_dopacities;         /* [N] */
    float* dL_dscales;            /* [N,3] */
    float* dL_drotations;         /* [N,4] */
    float* dL_dcov3D;             /* [N,6] */
    /* motion gradients (EMD_FLAG_MOTION) */
    float* dL_dactor_pose;        /* [A, EMD_ACTOR_STRIDE], zeroed by the library */
    float* dL_dresidual_dx;       /* [N,3] */
    float* dL_dresidual_dq;       /* [N,4] */
    float* dL_dsh_color;          /* [N,3] or NULL: the clamp-masked dL/d(SH colour) of every Gaussian (0 when not visible). With it
                                   * dL_dshs may be NULL: dL/dshs[n][k][c] = basis_k(view direction) * dL_dsh_color[n][c] is rank one and is
                                   * rebuilt (and summed over views) by emd_sh_grad_from_factors -- 12 bytes per Gaussian to exchange
                                   * between GPUs instead of 192 */
    const float* settings_dev;    /* as in EmdFwdArgs (same buffer, kept alive by the caller) */
    /* extra colour sets (see EmdFwdArgs): forward inputs / outputs again, incoming and outgoing gradients; bwd_ws then holds
     * [N, 12 + 4 num_extra] floats (emd_raster_workspace_size with EmdDims.num_extra) */
    int32_t num_extra;
    const float* colors_extra[EMD_MAX_EXTRA];
    const float* out_extra[EMD_MAX_EXTRA];
    const float* dL_dextra[EMD_MAX_EXTRA];        /* [3,H,W] or NULL */
    float* dL_dcolors_extra[EMD_MAX_EXTRA];       /* [N,3] or NULL */
    /* diagnostic (ABI 18; four words since ABI 25): device uint64[4], ADDED to by the render backward: [0] (pixel, list entry) pairs its
     * waves evaluated, [1] pairs that contributed (alpha >= 1/255 in front of the pixel's last contributor), [2] accumulator rows it sent
     * to memory as float atomics (one per (quadrant, survivor) with a non-zero entry), [3] float atomics issued.  NULL in production: the counting
     * instantiation of the kernel is only launched when the pointer is set (plain call only: no extra sets / absgrad / dL_dnormal) */
    uint64_t* pair_stats;
} EmdBwdArgs;

/* Per-step inputs of a training step that is replayed from ONE hipGraph (ABI 18).  A captured step reads everything that changes
 * from step to step at fixed device addresses; this op fills them from device-resident tables in one launch:
 *   out_row[0..row_floats)  = table[sel[0]]                 e.g. the 38-float settings block of EmdFwdArgs.settings_dev
 *   frame_out[0]            = frames[sel[0]],  t_out[0] = frame / max(num_frames - 1, 1)          (rigid.py:204,241)
 *   k_fine_out[0]           = int(k_min + (k_max - k_min) * clamp(step, 0, k_until) / k_until)    (rigid.py:147-148,194-201)
 *                             with step = steps[sel[0]] (or the row index when steps is NULL)
 * and, for callers that want the device status words of every step without reading them back per step:
 *   status_log[prev_sel[0]] = status[0..4)  (the words the step BEFORE left behind), then prev_sel[0] = sel[0].
 * A call with sel[0] outside [0, rows) only flushes the pending status row.  With next_sel the launch finally sets sel[0] =
 * next_sel[row]: a loop of graph replays walks a schedule of rows without the host touching device memory between replays. */
typedef struct EmdStepSelect {
    int64_t* sel;                    /* device [1]: the row to select; with next_sel it is advanced to next_sel[row] by the launch */
    int32_t rows, row_floats;
    const float* table;              /* [rows, row_floats] */
    float* out_row;                  /* [row_floats] */
    const int32_t* frames;           /* [rows] or NULL */
    int32_t* frame_out;              /* [1] or NULL */
    float* t_out;                    /* [1] or NULL */
    int32_t num_frames;
    int32_t k_min, k_max, k_until;
    const int64_t* steps;            /* [rows] or NULL */
    int32_t* k_fine_out;             /* [1] or NULL */
    const int32_t* status;           /* [4] or NULL */
    int32_t* status_log;             /* [rows, 4] or NULL */
    int64_t* prev_sel;               /* device [1], -1 = nothing pending; or NULL */
    const int64_t* next_sel;         /* [rows] or NULL: successor of every row */
} EmdStepSelect;
int emd_select_step_inputs(const EmdStepSelect* args, void* hip_stream);

int emd_abi_version(void);
const char* emd_last_error(void);

/* out[0..3] = bytes of geom_ws, bin_ws, img_ws, bwd_ws */
int emd_raster_workspace_size(const EmdDims* dims, size_t out[4]);

int emd_raster_forward(EmdFwdArgs* args, void* hip_stream);
int emd_raster_backward(const EmdBwdArgs* args, void* hip_stream);

/* Copy binning state out for parity tests: sorted keys (tile<<32 | depth bits), sorted Gaussian ids, the quadrant mask of every
 * entry (bit q = qy * 2 + qx: the Gaussian's alpha >= 1/255 footprint reaches the 8x8 quadrant q of the tile; ABI 21),
 * per-tile [start,end) ranges.  Any output may be NULL.  keys / ids / quad_masks hold num_entries = EmdStatus.num_rendered entries.
 * The sort itself moves (tile id, Gaussian id) pairs of Gaussians pre-ordered by depth; the 64-bit keys of the
 * reference are rebuilt here from the tile id and the depth bits kept per Gaussian in geom_ws. */
int emd_raster_export_binning(const EmdDims* dims, const void* geom_ws, size_t geom_bytes, const void* bin_ws, size_t bin_bytes,
                              int64_t num_entries, uint64_t* keys, uint32_t* ids, uint32_t* ranges /*[T,2]*/,
                              uint32_t* quad_masks, void* hip_stream);

/* Copy per-Gaussian projection state out for parity tests.  Any output may be NULL. */
int emd_raster_export_geometry(const EmdDims* dims, const void* geom_ws, size_t geom_bytes,
                               float* means2D /*[N,2]*/, float* depths /*[N]*/, float* conic_opacity /*[N,4]*/,
                               float* rgb /*[N,3]*/, float* normal /*[N,3]*/, uint32_t* tiles_touched /*[N]*/,
                               void* hip_stream);

/* Stand-alone explicit-motion transform (same arithmetic as the fused path):
 *   world_mean = R(q_mean[a]) (mean + dx) + trans[a];  world_quat = q_rot[a] (x) normalize(quat + dq);
 *   opacity_out = opacity * valid[a].  Outputs may be NULL. */
int emd_motion_forward(int32_t n, const float* means, const float* quats, const float* opacities,
                       const EmdMotion* motion, float* world_means, float* world_quats, float* opacities_out,
                       void* hip_stream);
int emd_motion_backward(int32_t n, const float* means, const float* quats, const float* opacities,
                        const EmdMotion* motion, const float* dL_dworld_means, const float* dL_dworld_quats,
                        const float* dL_dopacities_out, float* dL_dmeans, float* dL_dquats, float* dL_dopacities,
                        float* dL_dactor_pose /*[A,12] zeroed here*/, float* dL_dresidual_dx, float* dL_dresidual_dq,
                        void* hip_stream);

/* Spherical harmonics colour (no +0.5, no clamp): rgb = SH_deg(dirs / |dirs|) . coeffs */
int emd_sh_forward(int32_t n, int32_t degree, int32_t sh_coeffs, const float* dirs /*[N,3]*/,
                   const float* coeffs /*[N,K,3]*/, float* rgb /*[N,3]*/, void* hip_stream);
int emd_sh_backward(int32_t n, int32_t degree, int32_t sh_coeffs, const float* dirs, const float* coeffs,
                    const float* dL_drgb, float* dL_dcoeffs, float* dL_ddirs /*or NULL*/, void* hip_stream);

/* Dense SH-coefficient gradient from per-view factors (view-parallel data parallelism):
 *   dL_dshs[n][k][c] = scale * sum_v basis_k(normalize(world_mean_n - campos[v])) * sh_color_grads[v][n][c]      k < (degree+1)^2
 * with world_mean_n the mean after the explicit-motion transform (motion may be NULL: static scene).  Every rank gathers
 * the [N,3] factors and the camera centres of all views and rebuilds the same dense, averaged gradient locally.
 * pose_per_view != 0: the views belong to different timestamps (six cameras on eight ranks), motion->actor_pose is then
 * [V, A, EMD_ACTOR_STRIDE] -- one gathered pose table per view -- and an actor's Gaussians are placed per view. */
int emd_sh_grad_from_factors(int32_t n, int32_t num_views, int32_t degree, int32_t sh_coeffs, const float* means3D,
                             const EmdMotion* motion, int32_t pose_per_view, const float* campos /*[V,3]*/,
                             const float* sh_color_grads /*[V,N,3]*/, float scale, float* dL_dshs /*[N,sh_coeffs,3]*/,
                             void* hip_stream);

/* Visibility-compacted rows for the view-parallel gradient exchange (ABI 25; no reference counterpart -- the
 * reference trains one view per step on one GPU, S3Gaussian/train.py:203).  A view's gradient is zero for every Gaussian it does not see:
 * a rank sends (index, values) rows of its visible Gaussians with an all-gather instead of all-reducing dense [N, .] tensors, and every
 * rank adds the gathered rows into a dense buffer in rank order (a fixed order of float additions: replicas stay bit-identical).
 *
 * emd_compact_rows: rows is uint32 [(1 + capacity) * row_words], row_words = 1 + sum(widths).  rows[0 .. row_words) is the header
 * (count of data rows, overflow flag: more visible Gaussians than capacity -- the surplus rows are dropped and the receiver must not
 * trust the step -- then zeros); data row j = (Gaussian index, the widths[k] floats of sources[k][index] for k = 0 .. num_sources - 1) for
 * the Gaussians with radii > 0, in no particular order.  counter: one device uint32 of scratch.
 * emd_scatter_rows: one gathered view: for j < header.count: dests[k][index] = (add ? dests[k][index] : 0) + scale * value.  Indices are
 * unique within a view (no atomics); launch the views one after the other.  overflow_out (optional device uint32) gets bit 0 set when the
 * header carries the overflow flag. */
#define EMD_ROW_SOURCES 4
int emd_compact_rows(int32_t n, const int32_t* radii, int32_t num_sources, const float* const* sources, const int32_t* widths,
                     int64_t capacity, uint32_t* rows, uint32_t* counter, void* hip_stream);
int emd_scatter_rows(const uint32_t* rows, int64_t capacity, int32_t num_dst_rows, int32_t num_dests, float* const* dests,
                     const int32_t* widths, int32_t add, float scale, uint32_t* overflow_out, void* hip_stream);

/* Densification statistics of one view (the per-step part): for every Gaussian with radii > 0
 * grad_accum += |dL_dmeans2D.xy|, denom += 1, max_radii2D = max(max_radii2D, radii), in place -- gaussian_model.py:728-730 and
 * train.py:403-406 without their boolean-mask indexing (a host sync per step).  Any of the three outputs may be NULL. */
int emd_densification_stats(int32_t n, const int32_t* radii, const float* dL_dmeans2D /*[N,3]*/, float* grad_accum /*[N]*/,
                            float* denom /*[N]*/, float* max_radii2D /*[N]*/, void* hip_stream);

/* Per-frame actor pose table, training branch of RigidNodes.transform_means / transform_quats
 * (OmniRe/models/nodes/rigid.py:499-503,519-532,547-566): pose[a] = (normalize(q_f[a]), t_f[a] + dt[a], valid[a],
 * normalize(q_f[a] (x) dq[a])).  q_f [A,4] raw pose quaternions of the frame, t_f [A,3], valid [A] bytes or NULL,
 * dt [A,3] / dq [A,4] learned track offsets or NULL (NaN rows are skipped as the reference does).
 * frame_dev (optional): a DEVICE int32 holding the frame index; q_f / t_f / valid (and dL_dq_f / dL_dt_f) are then the whole
 * [F, A, .] tables and the row is selected on the device -- the call is then replayable from a hipGraph with a new frame. */
int emd_actor_pose_forward(int32_t num_actors, const float* q_f, const float* t_f, const uint8_t* valid, const float* dt,
                           const float* dq, float* pose /*[A,12]*/, const int32_t* frame_dev, void* hip_stream);
int emd_actor_pose_backward(int32_t num_actors, const float* q_f, const float* dt, const float* dq, const float* dL_dpose,
                            float* dL_dq_f, float* dL_dt_f, float* dL_ddt /*or NULL*/, float* dL_ddq /*or NULL*/,
                            const int32_t* frame_dev, void* hip_stream);

/* L1 photometric loss of the training step (S3Gaussian/utils/loss_utils.py:21-22, train.py:226):
 * loss[0] = mean |a - b| over n elements, grad[i] = sign(a[i] - b[i]) / n (grad may be NULL).  One launch.
 * b == NULL: loss[0] = mean |a| -- the residual regularisers of the fine stage (the L1 norms of dx / do / dshs, S3Gaussian/train.py);
 * emd_abs_mean_backward writes their gradient sign(x[i]) * g[0] / n with the upstream gradient g read on the device. */
int emd_l1_loss(int64_t n, const float* a, const float* b, float* loss /*[1]*/, float* grad /*[n] or NULL*/, void* hip_stream);
/* ABI 25.  The same with a caller-kept scratch table (EMD_L1_SCRATCH_WORDS 4-byte words, 8-byte aligned, ZERO at the first call and left zero by
 * every call; used by one stream at a time): every workgroup publishes its partial sum there as one 8-byte {value, tag} granule, workgroup 0
 * adds them in workgroup order and writes `loss` -- no zero fill of `loss` in front of the kernel (a launch of its own: 4.6 us of a 1.3 ms
 * step), and a sum that does not depend on the order of float atomics.  scratch == NULL is emd_l1_loss. */
#define EMD_L1_SCRATCH_WORDS 1024
int emd_l1_loss_ws(int64_t n, const float* a, const float* b, float* loss /*[1]*/, float* grad /*[n] or NULL*/, uint32_t* scratch /*[EMD_L1_SCRATCH_WORDS]*/,
                   void* hip_stream);
int emd_abs_mean_backward(int64_t n, const float* x, const float* g /*[1], device*/, float* grad /*[n]*/, void* hip_stream);
/* ABI 20.  The backward of a PAIR of regularised residuals that also feed the renderer (dshs_coarse / dshs_fine of the fine stage,
 * S3Gaussian/train.py:238-310 with scene/deformation.py:468-481): grad_a[i] = up_a[i] + sign(x_a[i]) g_a[0] / n, likewise b, in one
 * pass; up_a / up_b (the renderer's dL/dshs, usually one tensor: read once when the pointers are equal) and g_a / g_b may be NULL (= 0). */
int emd_residual_l1_backward(int64_t n, const float* up_a, const float* up_b, const float* x_a, const float* x_b, const float* g_a /*[1], device*/,
                             const float* g_b /*[1], device*/, float* grad_a /*[n]*/, float* grad_b /*[n]*/, void* hip_stream);

/* The activations exactly as EMD_FLAG_RAW_PARAMS applies them (exp, F.normalize, sigmoid); any pair may be NULL. */
int emd_activations_forward(int32_t n, const float* log_scales, float* scales, const float* raw_quats, float* quats,
                            const float* opacity_logits, float* opacities, void* hip_stream);

/* Per-stage device timing with HIP events recorded on the caller's stream around each stage (bench.py's roofline
 * leg).  Stages: see emd_profile_stage_name().  emd_profile_read synchronises on the recorded events, adds the
 * elapsed milliseconds and launch counts per stage into ms[]/count[] (up to max_stages entries) and clears the
 * recorded events; returns the number of stages. */
#define EMD_PROF_STAGES 8
int emd_profile_enable(int on);
int emd_profile_read(double* ms, int64_t* count, int max_stages);
const char* emd_profile_stage_name(int stage);


/* ==== widening rows of the scope table, each behind the reference's own module surface ==== */

/* ---- sky cube map + final blend -------------------------------------------------
 * Replaces S3Gaussian/scene/sky_cubemap.py:41-87 (SkyCubeMap.forward: get_rays_torch, mask, nvdiffrast dr.texture
 * 'linear'/'cube', clamp) with the blend of gaussian_renderer/__init__.py:299-301, and OmniRe/models/modules.py:174-208
 * (EnvLight.forward) with the blend of models/trainers/base.py:491-497. */
#define EMD_SKY_CLAMP01     1   /* clamp the looked-up colour to [0,1] (S3G) */
#define EMD_SKY_BLEND_S3G   2   /* out = fg * acc + sky * (1 - acc) */
#define EMD_SKY_BLEND_ADD   4   /* out = fg + sky * (1 - acc) (OmniRe) */
#define EMD_SKY_INTERLEAVED 8   /* images are [P,3]; default planar [3,P] */

typedef struct EmdSkyArgs {
    int32_t height, width;      /* P = height * width pixels (a flat list of directions: height 1) */
    int32_t resolution;         /* cube faces are resolution x resolution texels */
    int32_t flags;              /* EMD_SKY_* */
    const float* cube;          /* [6,res,res,3] faces +x,-x,+y,-y,+z,-z (OpenGL orientation) */
    const float* dirs;          /* [P,3] lookup directions in the cube's frame, or NULL: pinhole rays below */
    float Kinv[9], R[9], T[3];  /* row-major inverse intrinsics, w2c rotation, w2c translation: get_rays_torch(H,W,K,R,T) */
    const float* jitter;        /* [P,2] sub-pixel offsets (training: U[0,1)), or NULL: pixel centres (+0.5) */
    const float* acc;           /* [P] foreground opacity ("weight"), or NULL */
    float mask_threshold;       /* with acc: texture sampled where (1 - acc) > threshold (S3G: 1e-3), else sky = fill; <0: always */
    float fill;                 /* sky colour of masked-out pixels */
    const float* fg;            /* foreground image for the blend, or NULL */
    float* sky;                 /* out: sky colour, or NULL */
    float* out;                 /* out: blended image, or NULL */
    const float* camera_dev;    /* optional (ABI 21): Kinv[9], R[9], T[3] on the DEVICE (21 floats) -- they replace the by-value fields, so a pass
                                   recorded into a hipGraph can serve every camera of a rig (emd_amd.StepInputs) */
} EmdSkyArgs;

typedef struct EmdSkyBwdArgs {
    EmdSkyArgs f;               /* the forward arguments (sky / out unused) */
    const float* dL_dout;       /* gradient of the blended image, or NULL */
    const float* dL_dsky;       /* gradient of the sky colour output, or NULL */
    float* dL_dcube;            /* [6,res,res,3], zeroed here and accumulated with float atomics; or NULL */
    float* dL_dacc;             /* [P] through the blend only (the mask is not differentiated), or NULL */
    float* dL_dfg;              /* like fg, or NULL */
} EmdSkyBwdArgs;

int emd_sky_forward(const EmdSkyArgs* args, void* hip_stream);
int emd_sky_backward(const EmdSkyBwdArgs* args, void* hip_stream);

/* ---- image-loss tail -----------------------------------------------------------------
 * loss = L1 + lambda_depth * depth L2 + lambda_dssim * (1 - SSIM 11x11) + lambda_sky * sky BCE, with the gradients with
 * respect to the rendered image, depth and weight in the planar layouts emd_raster_backward consumes.  Replaces
 * S3Gaussian/utils/loss_utils.py:21-98 (compute_depth "l2", l1_loss, ssim) as used in S3Gaussian/train.py:226-363. */
typedef struct EmdLossArgs {
    int32_t height, width;
    const float* image;        /* [3,H,W] rendered */
    const float* gt;           /* [3,H,W] target */
    const float* depth;        /* [H,W] rendered depth, or NULL (no depth term) */
    const float* gt_depth;     /* [H,W] lidar depth (0 where absent) */
    const float* mask;         /* [H,W] multiplies both depths (train.py: ~sky_mask), or NULL = ones */
    const float* weight;       /* [H,W] rendered opacity, or NULL (no sky term) */
    const uint8_t* sky_mask;   /* [H,W] 1 = sky */
    float lambda_dssim, lambda_depth, lambda_sky, max_depth;   /* reference: 0.2, 0.5, 0.05, 80 */
    float* losses;             /* out [5]: total, l1, ssim, depth, sky */
    float* dL_dimage;          /* out [3,H,W] or NULL */
    float* dL_ddepth;          /* out [H,W] or NULL */
    float* dL_dweight;         /* out [H,W] or NULL */
} EmdLossArgs;

size_t emd_image_loss_workspace(int height, int width);
int emd_image_loss(const EmdLossArgs* args, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- HexPlane feature lookup -------------------------------------------------------------
 * Replaces HexPlaneField.get_density / interpolate_ms_features (S3Gaussian/scene/hexplane.py:18-110,150-183): for every
 * scale the product over the six coordinate planes (xy, xz, xt, yz, yt, zt) of a bilinear, align_corners, border-padded
 * grid_sample, concatenated over scales.  Planes are passed CHANNEL-LAST: [res_h][res_w][C]. */
#define EMD_HEX_MAX_SCALES 8
typedef struct EmdHexArgs {
    int32_t num_points, channels, num_scales;
    int32_t times_broadcast;                     /* (ABI 24) non-zero: `times` holds ONE timestamp, shared by all points */
    int32_t res[EMD_HEX_MAX_SCALES][4];          /* per scale: resolution along x, y, z, t */
    const float* planes[EMD_HEX_MAX_SCALES][6];  /* per scale: planes of the pairs (0,1),(0,2),(0,3),(1,2),(1,3),(2,3), [res[b]][res[a]][C] */
    const float* pts;                            /* [N,3] */
    const float* times;                          /* [N], or one float with times_broadcast */
    float aabb[6];                               /* aabb[0] (3 floats) then aabb[1] (3 floats), as HexPlaneField stores them */
    float* out;                                  /* [N, num_scales * C] (forward) */
    const int32_t* order;                        /* [N] a permutation of the points, or NULL: the order the kernels visit them in.
                                                    A spatially coherent one (e.g. Morton order of pts) lets the backward
                                                    aggregate the plane gradients in LDS before they reach HBM; results are
                                                    the same up to the order of the float sums. */
    float* time_tables;                          /* (ABI 23) NULL, or scratch of C * sum over scales of (res_x + res_y + res_z) floats.  Non-NULL is
                                                    the caller's PROMISE that every timestamp equals times[0] (one frame per step: always, in
                                                    training): emd_hexplane_forward then blends the two time rows of the planes (x,t), (y,t),
                                                    (z,t) into 1-D tables once per call and reads two taps instead of four on those planes --
                                                    18 instead of 24 tap rows per point and scale.  Same values up to rounding ((1-fx) [(1-ft) nw
                                                    + ft sw] + fx [...] instead of the four-term sum).  The backward does not use it. */
} EmdHexArgs;

typedef struct EmdHexGrads {
    const float* dL_dout;                        /* [N, num_scales * C] */
    float* dL_dplanes[EMD_HEX_MAX_SCALES][6];    /* channel-last like planes; ZEROED BY THE CALLER, accumulated with float atomics; may be NULL */
    float* dL_dpts;                              /* [N,3] or NULL */
    float* dL_dtimes;                            /* [N] or NULL (reaches the reference's time_offset parameter, deformation.py:325-328) */
    /* Optional per-plane pass for the fine scales (ABI 19; needs args->order and 16 or 32 channels).  On the scales named by
     * defer_mask a run of 256 points that is compact in 3-D still spans tens of cells of a SPATIAL plane and puts ~1 tap into a cell,
     * so nothing aggregates and every tap row is a global atomic.  With the three extra orders below the backward instead writes, per
     * point and spatial plane, the 128-byte row dL/d(interpolated plane value) into defer_rows at the point's position in THAT
     * plane's order, and a second launch walks each plane in its own order -- runs that are compact in the plane's two coordinates
     * share cells -- and adds the taps through LDS windows.  Any permutations give the same sums (up to float rounding); all of
     * order2d / pos2d / defer_rows must be non-NULL when defer_mask != 0. */
    const int32_t* order2d[3];                   /* [N] each: visiting order of the planes xy, xz, yz */
    const int32_t* pos2d[3];                     /* [N] each: the inverse permutations, pos2d[p][order2d[p][i]] = i */
    float* defer_rows;                           /* scratch of popcount(defer_mask) * 3 * N * C + 6 * N floats (the rows, then the planes' coordinates); written and read by the call */
    uint32_t defer_mask;                         /* bit s set: the spatial planes of scale s go through the per-plane pass */
    uint32_t reserved;
    float* dL_dtime_sum;                         /* (ABI 24) NULL, or one float ZEROED BY THE CALLER that receives sum_n dL/dtimes[n] (float atomics, one per
                                                    workgroup): the gradient of a broadcast timestamp -- S3Gaussian's time_offset parameter,
                                                    deformation.py:325-328 -- without the [N] gradient and its reduction; used when dL_dtimes is NULL */
} EmdHexGrads;

int emd_hexplane_forward(const EmdHexArgs* args, void* hip_stream);
int emd_hexplane_backward(const EmdHexArgs* args, const EmdHexGrads* grads, void* hip_stream);

/* Sort keys of the visiting orders the aggregating backward is handed (EmdHexArgs.order, EmdHexGrads.order2d): in ONE launch, per point, the
 * 30-bit Z-order key of its box-normalised position (10 bits per axis) -> keys[0 .. N) and the 24-bit HILBERT keys of its (x, y), (x, z),
 * (y, z) coordinates (12 bits per axis) -> keys[N .. 4 N).  Runs of consecutive points of the sorted keys are compact in 3-D / in that
 * plane; a Hilbert curve has no jumps, so a run of the per-plane pass stays inside its LDS window (a Z-order run leaves it with 7 % of its
 * taps at resolution 512).  aabb: six floats, the two corners of the box in either order (HexPlaneField.aabb); coordinates are clamped
 * into the box, NaN goes to cell 0.  [host side: emd_amd/hexplane.py VisitingOrders.build; no reference counterpart -- the reference's
 * grid_sample backward scatters in the order the points come] */
int emd_hexplane_order_keys(const float* pts, const float* aabb, int64_t num_points, int32_t* keys, void* hip_stream);

/* ---- embedding ops in front of the deformation MLPs (rows a3, a12, a15) ---------------------
 * emd_temporal_embed_*: one row of the coarse-to-fine temporal embedding -- each of the num_tables [rows, dim] tables resized
 *   to k rows (bilinear, align_corners) and sampled at time t[0] (bilinear, align_corners, reflection padding) -> out
 *   [num_tables, dim].  Replaces
 *   Deformation.get_temporal_embed (S3Gaussian/scene/deformation.py:208-221) and RigidNodes.get_temporal_embed
 *   (OmniRe/models/nodes/rigid.py:150-164).  t is a DEVICE scalar (no host sync).  The backward ACCUMULATES into dL_dweight
 *   [rows, dim] and dL_dt [1] (either may be NULL); the caller zeroes them.
 * emd_deform_input_*: the input matrix of OmniRe's ConditionalDeformNetwork (models/modules.py:411-457 with get_embedder
 *   :318-366, fed by DeformableNodes.get_deformation, models/nodes/deformable.py:35-47): row n =
 *   [x, sin(x 2^0), cos(x 2^0), ..., t, sin(t 2^0), ..., embed[id(n)]] with x = means[n] / inst_size[id(n)][2] * 2.
 *   The backward ACCUMULATES columns [col0, col0 + embed_dim) of dL_din into dL_dembed [A, embed_dim] (positions and the time
 *   are detached in the reference). */
int emd_temporal_embed_forward(const float* weight, int num_tables, int rows, int dim, int k, const float* t, float* out,
                               void* hip_stream);
int emd_temporal_embed_backward(const float* weight, int num_tables, int rows, int dim, int k, const float* t, const float* dL_dout,
                                float* dL_dweight, float* dL_dt, void* hip_stream);

/* Learned per-actor track offsets, all actors and both levels in one launch each way
 * (embedding_track_trans_offset / embedding_track_rot_offset / query_coarse_to_fine / get_temporal_embed,
 * OmniRe/models/nodes/rigid.py:150-246): see the block comment in csrc/embed.hip.  head_w[h] is [rows_h, dim + embed_dim]
 * row-major, head_b[h] is [rows_h], h = 0 track_trans_c (3 rows), 1 track_trans_f (3), 2 track_rot_c (1), 3 track_rot_f (1). */
typedef struct EmdTrackArgs {
    int32_t num_actors, rows, dim, embed_dim;    /* temporal tables [A, rows, dim]; dim + embed_dim <= 64 */
    int32_t k_coarse, k_fine, num_points, reserved;
    float t;                                     /* normalised frame (frame - start) / (end - start), rigid.py:204,241 */
    const float* t_dev;                          /* optional DEVICE copy of t (overrides it): hipGraph replay with a new frame */
    const int32_t* k_fine_dev;                   /* optional DEVICE copy of k_fine (overrides it): the coarse-to-fine row count follows the
                                                    training step (rigid.py:194-201), which a replayed graph reads from the device (ABI 18) */
    const float* weight;                         /* [A, rows, dim] */
    const float* embeddings;                     /* [num_points, embed_dim] of the actor Gaussians */
    const int32_t* point_ids;                    /* [num_points] actor of every point (-1: none) */
    const float* count;                          /* [A] points per actor, as floats */
    const int32_t* segment_start;                /* [A+1] or NULL: point_ids are sorted and actor a owns points [start[a], start[a+1]) -- the
                                                    sums are then formed without atomics and emb_sum need not be zero-filled */
    const float* head_w[4];
    const float* head_b[4];
    float* emb_sum;                              /* [A, embed_dim] scratch: per-actor embedding sums; ZERO-FILLED by the caller before forward, kept for backward */
    float* trans;                                /* out [A,3] */
    float* rot;                                  /* out [A,4] (w,x,y,z) */
} EmdTrackArgs;

typedef struct EmdTrackGrads {
    const float* g_trans;                        /* [A,3] */
    const float* g_rot;                          /* [A,4] */
    float* d_weight;                             /* [A, rows, dim]; accumulated into: ZERO-FILLED by the caller, like d_head_w / d_head_b */
    float* d_embeddings;                         /* [num_points, embed_dim] or NULL */
    float* d_head_w[4];
    float* d_head_b[4];
    float* d_mean;                               /* [A, embed_dim] scratch */
} EmdTrackGrads;

int emd_track_heads_forward(const EmdTrackArgs* args, void* hip_stream);
int emd_track_heads_backward(const EmdTrackArgs* args, const EmdTrackGrads* grads, void* hip_stream);

/* The per-actor chain of a training step in ONE launch each way (ABI 18): embedding sums -> track heads -> pose row
 * (= emd_track_heads_forward + emd_actor_pose_forward: RigidNodes.transform_means / transform_quats with the learned offsets,
 * OmniRe/models/nodes/rigid.py:150-246,499-566), one 1024-thread workgroup per actor.  Needs track.segment_start (an actor's
 * points contiguous, as the reference stores them) and embed_dim <= 8.  track.trans / track.rot are optional outputs here.
 * The backward WRITES d_q_all / d_t_all (the dense [F, A, .] clip gradients, zero outside the frame), d_weight and d_embeddings in
 * full and ADDS the head gradients into d_head_w / d_head_b, which must be views of the `head_acc` buffer the FORWARD call of the
 * same step was given (it clears those head_acc_floats floats): the caller zero-fills nothing.  A second backward through one
 * forward must clear head_acc itself. */
typedef struct EmdTrackedPoseArgs {
    EmdTrackArgs track;
    const float* q_all;                          /* [F, A, 4] raw per-frame pose quaternions (instances_quats) */
    const float* t_all;                          /* [F, A, 3] (instances_trans) */
    const uint8_t* valid_all;                    /* [F, A] bytes or NULL (instances_fv) */
    int32_t num_frames, frame;                   /* frame: row of the tables, unless frame_dev is given */
    const int32_t* frame_dev;                    /* optional DEVICE frame index (hipGraph replay) */
    float* pose;                                 /* out [A, EMD_ACTOR_STRIDE] */
    float* head_acc;                             /* forward: cleared (may be NULL when no backward will follow) */
    int32_t head_acc_floats, reserved;
} EmdTrackedPoseArgs;

typedef struct EmdTrackedPoseGrads {
    const float* g_pose;                         /* [A, EMD_ACTOR_STRIDE] */
    float* d_q_all;                              /* [F, A, 4] */
    float* d_t_all;                              /* [F, A, 3] */
    float* d_weight;                             /* [A, rows, dim] */
    float* d_embeddings;                         /* [num_points, embed_dim] or NULL */
    float* d_head_w[4];                          /* accumulated into: views of the forward's head_acc */
    float* d_head_b[4];
} EmdTrackedPoseGrads;

int emd_tracked_pose_forward(const EmdTrackedPoseArgs* args, void* hip_stream);
int emd_tracked_pose_backward(const EmdTrackedPoseArgs* args, const EmdTrackedPoseGrads* grads, void* hip_stream);

typedef struct EmdDeformInArgs {
    int32_t num_points, num_freqs_x, num_freqs_t, embed_dim;
    int32_t ld, reserved;                        /* row stride of out in floats (>= emd_deform_input_width) */
    const float* means;                          /* [N,3] local means */
    const int32_t* point_ids;                    /* [N] actor of each point; NULL: row n of inst_size / inst_embed belongs to point n */
    const float* inst_size;                      /* [A,3] actor box sizes (height = [.,2]); NULL: means are used as given */
    const float* inst_embed;                     /* [A, embed_dim] */
    const float* t;                              /* device scalar: normalised time of the frame */
    float* out;                                  /* [N, ld] */
} EmdDeformInArgs;

int emd_deform_input_width(int num_freqs_x, int num_freqs_t, int embed_dim);
int emd_deform_input_forward(const EmdDeformInArgs* args, void* hip_stream);
int emd_deform_input_backward(int num_points, int embed_dim, int ld, int col0, const int32_t* point_ids, const float* dL_din,
                              float* dL_dembed, void* hip_stream);

/* ---- The width-64 MLPs of the EMD deformation network as fused fp32-MFMA kernels -------------------
 * Deformation.feature_out (defor_depth = 1) + the pos / scales / rotations / opacity / shs heads + dino_head of
 * S3Gaussian/scene/deformation.py:100-185,254-337: see the block comment of csrc/mlp.hip.  One trunk call and one branch call per
 * head replace ~12 GEMMs and ~30 element-wise launches per level and direction; every intermediate stays in registers.
 *   trunk   h = b + w[:, col_a : col_a + ka] xa + w[:, col_b : col_b + kb] xb      (ka a multiple of 4 up to 128, kb <= 8; b = the layer's bias +
 *           the contribution of the temporal-embedding row, which is the same for every Gaussian)
 *   branch  out = w_out act(... act(w_hidden[0] in + b_hidden[0]) ...) + b_out, `depth` hidden layers of width 64 with ReLU,
 *           in = relu(h) when relu_input (the deformation heads: nn.Sequential(ReLU, Linear, ReLU, Linear)) else h (dino_head)
 * Backward: each branch writes ITS contribution to dL/dh into its own [N,64] buffer (the ReLU mask of relu_input applied) and
 * ACCUMULATES its weight / bias gradients with float atomics (the caller zero-fills them); the trunk sums the contributions,
 * writes dL/dxa, dL/dxb and accumulates the xa / xb column blocks of d_w and d_b.  All matrices row-major fp32; xa, h, g_h, d_xa
 * 16-byte aligned. */
#define EMD_MLP_MAX_BRANCHES 6
typedef struct EmdMlpTrunk {
    int32_t num_points, ka, kb, ld_w;            /* ld_w: row stride of w (the nn.Linear weight [64, ld_w]) */
    int32_t col_a, col_b, reserved0, reserved1;  /* first column of the xa / xb block inside w */
    const float* xa;                             /* [N, ka] or NULL */
    const float* xb;                             /* [N, kb] or NULL */
    const float* w;                              /* [64, ld_w] */
    const float* b;                              /* [64] effective bias */
    float* h;                                    /* [N, 64] pre-activation: written by the forward; the backward does not read it (may be NULL there, ABI 26) */
} EmdMlpTrunk;

typedef struct EmdMlpTrunkGrads {
    int32_t num_gh, reserved;
    const float* g_h[EMD_MLP_MAX_BRANCHES];      /* [N,64] each: summed */
    float* d_xa;                                 /* [N, ka] or NULL */
    float* d_xb;                                 /* [N, kb] or NULL */
    float* d_w;                                  /* [64, ld_w] accumulated (xa / xb column blocks only) or NULL */
    float* d_b;                                  /* [64] accumulated or NULL */
} EmdMlpTrunkGrads;

typedef struct EmdMlpBranch {
    int32_t num_points, depth, relu_input, out_dim;   /* depth 1 or 2; out_dim 1..64 */
    const float* h;                              /* [N,64] */
    const float* w_hidden[2];                    /* [64,64] each */
    const float* b_hidden[2];                    /* [64] */
    const float* w_out;                          /* [out_dim,64] */
    const float* b_out;                          /* [out_dim] */
    float* out;                                  /* [N,out_dim] (forward) */
    float* l1_sum;                               /* optional (ABI 20), [1], ZEROED BY THE CALLER: the forward adds mean |out| to it -- the L1 regulariser
                                                    of a residual head (S3Gaussian/train.py:238-310) formed while the outputs are stored */
    /* optional (ABI 26), depth 1 only: a level WITHOUT HexPlane features (`no_fine_hexplane_features`, the reference's run script: the trunk's
     * input is the per-Gaussian embedding alone, S3Gaussian/scene/deformation.py:256-296).  h = b_in + w_in[:, col_in : col_in + kb_in] xb is then
     * eight fp32 MFMAs per 32 rows: with `xb` set the head forms h itself, in the trunk kernel's operation order (same bits), and `h` may be NULL --
     * no [N,64] tensor is written by a trunk launch and read back by every head, forward and backward. */
    const float* xb;                             /* [N,kb_in] */
    const float* w_in;                           /* the trunk's weight [64, ld_w_in] */
    const float* b_in;                           /* [64]: its bias plus whatever is constant over the Gaussians */
    int32_t kb_in, ld_w_in, col_in, reserved;    /* kb_in 1..8 */
} EmdMlpBranch;

typedef struct EmdMlpBranchGrads {
    const float* g_out;                          /* [N,out_dim] */
    float* g_h;                                  /* [N,64] out */
    float* d_w_hidden[2];                        /* accumulated; any may be NULL */
    float* d_b_hidden[2];
    float* d_w_out;
    float* d_b_out;
    /* optional (ABI 20): the gradient of l1_sum's mean |out|, folded into g_out while it is loaded: g_out[i] + sign(out[i]) l1_grad[0] / (N out_dim).
     * `out` is the forward's output tensor; g_out may then be NULL (no other gradient reaches the head). */
    const float* l1_grad;                        /* [1], device */
    const float* out;                            /* [N,out_dim] */
    /* optional (ABI 20): g_h = g_h_in + this head's contribution (g_h_in may be g_h itself: every element is read and written by the same
     * lane).  Chaining the heads of a level through ONE buffer leaves the trunk's backward a single [N,64] tensor to read instead of one
     * per head.  Depth 1 only (the two-hidden-layer kernel has no registers for the extra tile: EMD_ERR_INVALID). */
    const float* g_h_in;                         /* [N,64] or NULL */
} EmdMlpBranchGrads;

int emd_mlp_trunk_forward(const EmdMlpTrunk* args, void* hip_stream);
int emd_mlp_trunk_backward(const EmdMlpTrunk* args, const EmdMlpTrunkGrads* grads, void* hip_stream);
int emd_mlp_branch_forward(const EmdMlpBranch* args, void* hip_stream);
int emd_mlp_branch_backward(const EmdMlpBranch* args, const EmdMlpBranchGrads* grads, void* hip_stream);

/* ---- Adaptive density control on the device ----------------------------------------------------
 * densify = densify_and_clone + densify_and_split, prune = prune / prune_points of S3Gaussian/scene/gaussian_model.py:441-603,
 * 683-701 (OmniRe: models/gaussians/vanilla.py:206-376), rewriting the SoA parameters, both Adam moments and the statistics in
 * one gather launch -- see the block comment of csrc/densify.hip.  Call sequence of one event:
 *   emd_densify_decide  -> code[N], block_counts[3][ceil(N / 256)] (per 256-point block: survivors, clones, splits)
 *   emd_densify_scan    -> the counts become exclusive block offsets; totals[3]: num_out = keep + clone + 2 split is the event's one host read
 *   emd_densify_index   -> src[num_out], kind[num_out] (0 survivor, 1 clone, 2 / 3 split sample replica 0 / 1), in the
 *                          reference's output order: survivors, clones, replica 0, replica 1
 *   emd_densify_gather  -> every output tensor */
enum { EMD_DENSIFY_MODE_DENSIFY = 0, EMD_DENSIFY_MODE_PRUNE = 1,
       EMD_DENSIFY_MODE_REFINE = 2   /* OmniRe's per-class refinement (VanillaGaussians.refinement_after, models/gaussians/vanilla.py:206-297): split (the original
                                        STAYS, its log-scale reduced in place), duplicate and cull as ONE event -- see EmdRefineArgs below */ };
enum {
    EMD_DENSIFY_ROLE_COPY = 0,     /* parameter copied row for row (features, opacity, rotation, embedding, deformation table ...) */
    EMD_DENSIFY_ROLE_XYZ = 1,      /* means: split samples get R(q) (exp(scaling) * n) + xyz, n ~ N(0, 1)   gaussian_model.py:541-545 */
    EMD_DENSIFY_ROLE_SCALING = 2,  /* log-scales: split samples get log(exp(s) / (0.8 * 2))                  gaussian_model.py:546 */
    EMD_DENSIFY_ROLE_STATE = 3,    /* Adam moment: survivors keep theirs, new rows start at zero             gaussian_model.py:480-500 */
    EMD_DENSIFY_ROLE_ZERO = 4      /* statistics: reset to zero by a densification (gaussian_model.py:526-530), carried by a prune */
};

typedef struct EmdDensifyArgs {
    int32_t num_points, mode;
    const float* scaling;          /* [N,3] log-scales */
    const float* opacity;          /* [N] logits (prune) */
    const float* grad_accum;       /* [N] xyz_gradient_accum (densify) */
    const float* denom;            /* [N] (densify) */
    const float* max_radii2D;      /* [N] (prune with max_screen_size > 0) */
    const uint8_t* extra_drop;     /* [N] or NULL: additional rows to drop in prune mode (prune_points(mask) with a caller's mask) */
    float grad_threshold, percent_dense, scene_extent;      /* densify */
    float min_opacity, max_screen_size;                     /* prune; max_screen_size <= 0: the size tests are off */
} EmdDensifyArgs;

typedef struct EmdDensifyTensor {
    const float* src;              /* [N, width] */
    float* dst;                    /* [num_out, width] */
    int32_t width, role;
} EmdDensifyTensor;

#define EMD_DENSIFY_MAX_TENSORS 40
typedef struct EmdDensifyGather {
    int32_t num_out, num_tensors, mode, num_split;
    const int32_t* src;            /* [num_out] */
    const int32_t* kind;           /* [num_out] */
    const float* scaling;          /* [N,3] source log-scales  (split) */
    const float* rotation;         /* [N,4] source quaternions (split) */
    uint64_t seed;                 /* Philox key of the split samples: counter = (source Gaussian index, replica) -- the same samples on
                                      every rank of a data-parallel job without communication */
    const float* samples;          /* [2, num_split, 3] standard normals to use INSTEAD of Philox (tests: the reference's recorded draw) or NULL */
    const int32_t* split_rank;     /* [num_out] rank of a split row inside its replica (emd_densify_split_rank); needed with `samples` */
    EmdDensifyTensor tensors[EMD_DENSIFY_MAX_TENSORS];
} EmdDensifyGather;

/* ---- OmniRe refinement: VanillaGaussians.refinement_after = split_gaussians + dup_gaussians + cull_gaussians, models/gaussians/vanilla.py:206-376,
 * and dup_in_optim / remove_from_optim, models/gaussians/basics.py:198-242 -- fused into ONE decide -> scan -> index -> gather event.
 * Semantics restated from the reference (they differ from S3Gaussian's in every step):
 *   high   = xys_grad_norm / vis_counts > densify_grad_thresh                               (no NaN -> 0 step: NaN compares false)
 *   split  = (max exp(scale) > size_thresh  [or max_2Dsize > split_screen while step < stop_screen_size_at]) and high
 *            -> n_split_samples new rows  R(q^)(exp(scale) * n) + mean, n ~ N(0,1), scale log(exp(s) / 1.6); the ORIGINAL stays and takes the
 *               same reduced scale in place (vanilla.py:345)
 *   dup    = (max exp(scale) <= size_thresh) and high, evaluated AFTER the in-place reduction: a Gaussian just above the threshold is split
 *            AND duplicated (its duplicate carries the reduced scale)                        (vanilla.py:240-254)
 *   cull   = on every row of the grown arrays: sigmoid(opacity) < cull_alpha, or -- past the first opacity reset -- max exp(scale) > cull_size,
 *            or (while step < stop_screen_size_at) max_2Dsize > cull_screen, new rows carrying max_2Dsize 0       (vanilla.py:299-326)
 *   output order: surviving originals, split samples replica 0, replica 1, ..., duplicates (vanilla.py:256-263); Adam moments of new rows zero.
 * Thresholds are the host's products (size_thresh = densify_size_thresh * scene_scale ...) rounded to float once, as torch rounds a Python scalar.
 * Sequence: emd_refine_decide -> code[N], per-block counts of four 0/1 columns (original kept, duplicate kept, samples kept, source is split) ->
 * emd_densify_scan (4 columns; one host read of the totals: num_out = keep + num_samples * samples + dup) -> emd_refine_index ->
 * emd_densify_gather with mode EMD_DENSIFY_MODE_REFINE. */
typedef struct EmdRefineArgs {
    int32_t num_points;
    int32_t do_densify;            /* split / duplicate this event (step < stop_split_at and past the opacity-reset guard, vanilla.py:213-216) */
    int32_t do_cull;               /* cull this event (vanilla.py:281) */
    int32_t use_split_screen;      /* step < stop_screen_size_at: the screen-size split test is on */
    int32_t cull_big;              /* step > reset_alpha_interval: the world-size (and screen-size) cull tests are on */
    int32_t use_cull_screen;       /* ... and step < stop_screen_size_at */
    const float* scaling;          /* [N,3] log-scales */
    const float* opacity;          /* [N] logits */
    const float* grad_norm;        /* [N] xys_grad_norm */
    const float* vis_counts;       /* [N] */
    const float* max_2Dsize;       /* [N] */
    float grad_threshold, size_threshold, split_screen, cull_alpha, cull_size, cull_screen;
} EmdRefineArgs;
int emd_refine_decide(const EmdRefineArgs* args, int32_t* code /*[N]*/, int32_t* block_counts /*[4, ceil(N / 256)]*/, void* hip_stream);
/* kind[j]: 0 original, 1 duplicate, 2 + r split sample of replica r; + 16 when the source row was split (its scale is reduced wherever it is
 * copied); split_rank[j]: rank of a sample's source among ALL split sources (the row of a caller-supplied normal draw [num_samples, n_split, 3]) */
int emd_refine_index(int32_t num_points, int32_t num_out, int32_t num_samples, const int32_t* code, const int32_t* block_offsets /*[4, ceil(N / 256)]*/,
                     const int32_t* totals /*[4]*/, int32_t* src, int32_t* kind, int32_t* split_rank, void* hip_stream);

/* OmniRe's running refinement statistics of one view (VanillaGaussians.after_train, models/gaussians/vanilla.py:163-191), one launch: where radii > 0
 * grad_norm += |xys_grad| (2 components, row stride `grad_stride` floats), vis_counts += 1, max_2Dsize = max(max_2Dsize, radii / last_size). */
int emd_after_train_stats(int32_t n, const int32_t* radii, const float* xys_grad, int32_t grad_stride, float* grad_norm, float* vis_counts,
                          float* max_2Dsize, float last_size, void* hip_stream);

int emd_densify_decide(const EmdDensifyArgs* args, int32_t* code /*[N]*/, int32_t* block_counts /*[3, ceil(N / 256)]*/, void* hip_stream);
/* exclusive prefix sums of the per-block counts over the blocks, in place, + totals[c] = sum of column c (device int32[num_columns]) */
int emd_densify_scan(int32_t num_points, int32_t num_columns, int32_t* block_counts, int32_t* totals, void* hip_stream);
int emd_densify_index(int32_t num_points, int32_t num_out, const int32_t* code, const int32_t* block_offsets /*[3, ceil(N / 256)]*/,
                      const int32_t* totals /*[3]*/, int32_t* src, int32_t* kind, void* hip_stream);
int emd_densify_split_rank(int32_t num_out, int32_t n_keep, int32_t n_clone, int32_t n_split, int32_t* rank, void* hip_stream);
int emd_densify_gather(const EmdDensifyGather* args, void* hip_stream);

/* ---- Adam over all parameter tensors in one launch ---------------------------------------
 * Replaces optimizer.step() of `torch.optim.Adam(l, lr=0.0, eps=1e-15)` (S3Gaussian/scene/gaussian_model.py:188-201,
 * train.py:428; OmniRe builds the same optimiser per class, models/trainers/base.py:213-253): per element
 *   m = m + (1-beta1)(g-m);  v = v beta2 + (1-beta2) g g;  p = p - step_size * m / (sqrt(v) / bias_correction2_sqrt + eps)
 * in torch.optim.Adam's operation order, in place.  The caller supplies the step-dependent scalars (computed in double, as torch
 * does): step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t).  param / grad / exp_avg / exp_avg_sq of one
 * tensor must share one memory layout (they are walked as flat arrays of `numel` floats). */
#define EMD_ADAM_MAX_TENSORS 32
typedef struct EmdAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
    float step_size, bias_correction2_sqrt, one_minus_beta1, beta2, one_minus_beta2, eps;
    /* ABI 20, optional ("capturable" steps: an optimiser step recorded into a hipGraph must not bake the step count or the learning rate
     * into its launch): with step_dev != NULL the kernel forms step_size = lr_dev[0] / (1 - beta1^t) and bias_correction2_sqrt =
     * sqrt(1 - beta2^t) itself from t = step_dev[0] (the step count AFTER this step's increment, a device float as torch's capturable
     * Adam keeps it) and ignores the two by-value fields; lr_dev must then be non-NULL too. */
    const float* step_dev;
    const float* lr_dev;
} EmdAdamTensor;

typedef struct EmdAdamArgs {
    int32_t num_tensors, reserved;
    EmdAdamTensor tensors[EMD_ADAM_MAX_TENSORS];
} EmdAdamArgs;

int emd_adam_step(const EmdAdamArgs* args, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* EMD_RASTER_H */
