// api.hip -- the C ABI of include/emd_raster.h: argument checking, workspace carving, stage sequencing.
// No device memory is allocated or freed here; errors are returned as codes + a thread-local message.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void emd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiling ------------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct ProfRec { int stage; hipEvent_t a, b; bool b_shared; };
bool g_prof_on = false;
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t g_prof_open[EMD_PROF_STAGES];
const char* kStageNames[EMD_PROF_STAGES] = {"preprocess", "scan_duplicate", "radix_sort", "tile_ranges", "render_forward",
                                            "render_backward", "preprocess_backward", "other"};
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void emd_prof_begin(int stage, hipStream_t st) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t e = prof_event();
    if (!e) return;
    (void)hipEventRecord(e, st);
    g_prof_open[stage] = e;
}

void emd_prof_end(int stage, hipStream_t st) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_open[stage]) return;
    hipEvent_t e = prof_event();
    if (!e) return;
    (void)hipEventRecord(e, st);
    g_prof_recs.push_back({stage, g_prof_open[stage], e, false});
    g_prof_open[stage] = nullptr;
}

void emd_prof_switch(int ended, int started, hipStream_t st) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t e = prof_event();
    if (!e) return;
    (void)hipEventRecord(e, st);
    if (g_prof_open[ended]) { g_prof_recs.push_back({ended, g_prof_open[ended], e, true}); g_prof_open[ended] = nullptr; }
    g_prof_open[started] = e;
}

namespace {
__global__ void __launch_bounds__(EMD_BLOCK) k_zero_words(uint32_t* __restrict__ p, size_t head, size_t quads, size_t words) {
    // [0, head) single words up to 16-byte alignment, then `quads` uint4 stores, then the tail words
    const size_t i = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x, stride = (size_t)gridDim.x * EMD_BLOCK;
    if (i < head) p[i] = 0u;
    uint4* q = (uint4*)(p + head);
    for (size_t k = i; k < quads; k += stride) q[k] = make_uint4(0u, 0u, 0u, 0u);
    const size_t tail0 = head + quads * 4;
    if (tail0 + i < words && i < 4) p[tail0 + i] = 0u;
}
}  // namespace

// Per-step inputs of a replayed training step, selected on the device in one launch (see EmdStepSelect in emd_raster.h): what a
// host-driven loop does with a handful of tiny copies and index ops per step (camera upload, frame number, frame time, the
// coarse-to-fine level of the step) when the whole step is one hipGraph whose inputs must live at fixed device addresses.
__global__ void __launch_bounds__(EMD_WAVE) k_select_step_inputs(EmdStepSelect a) {
    const long long row = a.sel[0];
    const int lane = threadIdx.x;
    if (a.status_log && a.status && a.prev_sel) {            // the status words of the step before (its forward is long done): logged by row
        const long long prev = a.prev_sel[0];
        if (prev >= 0 && prev < a.rows && lane < 4) a.status_log[4 * prev + lane] = a.status[lane];
    }
    if (row < 0 || row >= a.rows) {                          // flush-only call (sel = -1 after the last step)
        if (a.prev_sel && lane == 0) a.prev_sel[0] = -1;
        return;
    }
    for (int j = lane; j < a.row_floats; j += EMD_WAVE) a.out_row[j] = a.table[(size_t)row * a.row_floats + j];
    if (lane == 0) {
        if (a.prev_sel) a.prev_sel[0] = row;
        if (a.next_sel) a.sel[0] = a.next_sel[row];           // the row of the NEXT launch: a replay loop then needs no host-side write at all
        if (a.frames && a.frame_out) {
            const int f = a.frames[row];
            a.frame_out[0] = f;
            if (a.t_out) a.t_out[0] = (float)f / (float)(a.num_frames > 1 ? a.num_frames - 1 : 1);
        }
        if (a.k_fine_out) {
            // int_lininterp(step, k_min, k_max, until) of rigid.py:147-148,194-201, with the reference's float division + truncation
            long long step = a.steps ? a.steps[row] : row;
            step = step < 0 ? 0 : (step > a.k_until ? a.k_until : step);
            a.k_fine_out[0] = (int)((double)a.k_min + (double)(a.k_max - a.k_min) * (double)step / (double)a.k_until);
        }
    }
}

int emd_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!p || bytes == 0) return EMD_OK;
    if (((uintptr_t)p & 3) || (bytes & 3)) { emd_set_error("emd_zero_async: pointer / size not a multiple of 4 bytes"); return EMD_ERR_INVALID; }
    const size_t words = bytes / 4;
    size_t head = ((16 - ((uintptr_t)p & 15)) & 15) / 4;
    if (head > words) head = words;
    const size_t quads = (words - head) / 4;
    size_t blocks = (quads + EMD_BLOCK - 1) / EMD_BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)blocks), dim3(EMD_BLOCK), 0, st, (uint32_t*)p, head, quads, words);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_abs_mean_backward(size_t n, const float* x, const float* g, float* out, hipStream_t st);
int emd_launch_residual_l1_backward(size_t n, const float* up_a, const float* up_b, const float* x_a, const float* x_b, const float* g_a,
                                    const float* g_b, float* out_a, float* out_b, hipStream_t st);
int emd_launch_motion_forward(int n, const float* means, const float* quats, const float* opac, const EmdMotion& mo,
                              float* wm, float* wq, float* wo, hipStream_t st);
int emd_launch_motion_backward(int n, const float* means, const float* quats, const float* opac, const EmdMotion& mo,
                               const float* g_wm, const float* g_wq, const float* g_wo, float* d_means, float* d_quats,
                               float* d_opac, float* d_pose, float* d_rdx, float* d_rdq, hipStream_t st);
int emd_launch_sh_forward(int n, int deg, int M, const float* dirs, const float* coeffs, float* rgb, hipStream_t st);
int emd_launch_sh_backward(int n, int deg, int M, const float* dirs, const float* coeffs, const float* g_rgb,
                           float* d_coeffs, float* d_dirs, hipStream_t st);
int emd_launch_sh_grad_from_factors(int n, int V, int deg, int M, const float* means, const EmdMotion& mo, int pose_per_view,
                                    const float* campos, const float* gc, float scale, float* d_shs, hipStream_t st);
int emd_launch_densification_stats(int n, const int32_t* radii, const float* g2d, float* accum, float* denom, float* max_radii,
                                   hipStream_t st);
int emd_launch_actor_pose_forward(int A, const float* q, const float* t, const uint8_t* valid, const float* dt, const float* dq,
                                  float* pose, const int32_t* frame_dev, hipStream_t st);
int emd_launch_actor_pose_backward(int A, const float* q, const float* dt, const float* dq, const float* g_pose, float* d_q,
                                   float* d_t, float* d_dt, float* d_dq, const int32_t* frame_dev, hipStream_t st);
int emd_launch_l1_loss(size_t n, const float* a, const float* b, float* loss, float* grad, uint32_t* scratch, hipStream_t st);
int emd_launch_activations(int n, const float* ls, float* sc, const float* rq, float* q, const float* lo, float* o, hipStream_t st);
int emd_launch_export_geometry(int N, const GeomWs& g, float* means2D, float* depths, float* conic_opacity, float* rgb,
                               float* normal, uint32_t* tiles_touched, hipStream_t st);

extern "C" {

int emd_abi_version(void) { return EMD_ABI_VERSION; }

int emd_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return EMD_OK;
}

const char* emd_profile_stage_name(int stage) { return (stage >= 0 && stage < EMD_PROF_STAGES) ? kStageNames[stage] : ""; }

int emd_profile_read(double* ms, int64_t* count, int max_stages) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && r.stage < max_stages) {
            if (ms) ms[r.stage] += (double)t;
            if (count) count[r.stage] += 1;
        }
        g_prof_pool.push_back(r.a);                 // a shared end event is the next record's `a`: pooled there
        if (!r.b_shared) g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    return EMD_PROF_STAGES;
}

const char* emd_last_error(void) { return g_err; }

int emd_raster_workspace_size(const EmdDims* dims, size_t out[4]) {
    if (!dims || !out) { emd_set_error("workspace_size: null argument"); return EMD_ERR_INVALID; }
    if (dims->num_gaussians < 0 || dims->image_height <= 0 || dims->image_width <= 0 || dims->bin_capacity < 0) {
        emd_set_error("workspace_size: bad dims N=%d H=%d W=%d cap=%lld", dims->num_gaussians, dims->image_height,
                      dims->image_width, (long long)dims->bin_capacity);
        return EMD_ERR_INVALID;
    }
    GeomWs g; BinWs b; ImgWs im;
    const int gx = (dims->image_width + EMD_TILE_X - 1) / EMD_TILE_X, gy = (dims->image_height + EMD_TILE_Y - 1) / EMD_TILE_Y;
    emd_carve_geom(nullptr, dims->num_gaussians, &g);
    emd_carve_bin(nullptr, dims->bin_capacity, gx * gy, &b);
    emd_carve_img(nullptr, dims->image_height, dims->image_width, &im);
    out[0] = g.bytes; out[1] = b.bytes; out[2] = im.bytes;
    if (dims->num_extra < 0 || dims->num_extra > EMD_MAX_EXTRA) { emd_set_error("workspace_size: num_extra %d not in 0..%d", dims->num_extra, EMD_MAX_EXTRA); return EMD_ERR_INVALID; }
    out[3] = (size_t)(dims->num_gaussians > 0 ? dims->num_gaussians : 1) * emd_bwd_stride(dims->num_extra) * sizeof(float);
    return EMD_OK;
}

static int check_common(const EmdSettings& s, int N, int M, const float* means3D, const float* shs,
                        const float* colors, const float* opac, const float* scales, const float* rots,
                        const float* cov, int flags, const EmdMotion& mo) {
    if (N < 0 || s.image_height <= 0 || s.image_width <= 0) { emd_set_error("bad sizes N=%d H=%d W=%d", N, s.image_height, s.image_width); return EMD_ERR_INVALID; }
    if (N > 0 && (!means3D || !opac)) { emd_set_error("means3D / opacities must not be null"); return EMD_ERR_INVALID; }
    if ((shs != nullptr) == (colors != nullptr) && N > 0) { emd_set_error("provide exactly one of shs / colors_precomp"); return EMD_ERR_INVALID; }
    const bool sr = scales && rots;
    if (((scales != nullptr) != (rots != nullptr)) || (sr == (cov != nullptr) && N > 0)) {
        emd_set_error("provide exactly one of (scales, rotations) / cov3D_precomp"); return EMD_ERR_INVALID;
    }
    if (s.sh_degree < 0 || s.sh_degree > 3) { emd_set_error("sh_degree %d not in 0..3", s.sh_degree); return EMD_ERR_INVALID; }
    if (shs && M < (s.sh_degree + 1) * (s.sh_degree + 1)) { emd_set_error("shs holds %d coefficients, degree %d needs %d", M, s.sh_degree, (s.sh_degree + 1) * (s.sh_degree + 1)); return EMD_ERR_INVALID; }
    if (!(flags & EMD_FLAG_SDEV_TANFOV) && (!(s.tanfovx > 0.f) || !(s.tanfovy > 0.f))) { emd_set_error("tanfov must be positive"); return EMD_ERR_INVALID; }
    if (flags & EMD_FLAG_MOTION) {
        if (mo.actor_id && (!mo.actor_pose || mo.num_actors <= 0)) { emd_set_error("motion: actor_id given without actor_pose"); return EMD_ERR_INVALID; }
        if (cov) { emd_set_error("motion: cov3D_precomp cannot be combined with the fused motion transform"); return EMD_ERR_INVALID; }
    }
    if ((flags & EMD_FLAG_RAW_PARAMS) && cov) { emd_set_error("raw params: cov3D_precomp has no raw form"); return EMD_ERR_INVALID; }
    return EMD_OK;
}

int emd_raster_forward(EmdFwdArgs* a, void* hip_stream) {
    if (!a) { emd_set_error("forward: null args"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    const int N = a->num_gaussians;
    int rc = check_common(a->s, N, a->sh_coeffs, a->means3D, a->shs, a->colors_precomp, a->opacities, a->scales,
                          a->rotations, a->cov3D_precomp, a->flags, a->motion);
    if (rc) return rc;
    if (!a->out_color || !a->out_depth || !a->out_alpha || !a->radii || !a->status || !a->geom_ws || !a->bin_ws || !a->img_ws) {
        emd_set_error("forward: null output / workspace pointer"); return EMD_ERR_INVALID;
    }
    if ((a->flags & EMD_FLAG_NORMAL) && !a->out_normal) { emd_set_error("forward: EMD_FLAG_NORMAL without out_normal"); return EMD_ERR_INVALID; }
    if (a->bin_capacity < 0) { emd_set_error("forward: negative bin_capacity"); return EMD_ERR_INVALID; }
    if ((int64_t)N >= (int64_t)1 << 28) { emd_set_error("forward: %d Gaussians; a list word holds a 28-bit Gaussian id beside the pair's quadrant mask", N); return EMD_ERR_INVALID; }
    if (a->num_extra < 0 || a->num_extra > EMD_MAX_EXTRA) { emd_set_error("forward: num_extra %d not in 0..%d", a->num_extra, EMD_MAX_EXTRA); return EMD_ERR_INVALID; }
    for (int k = 0; k < a->num_extra; k++)
        if (!a->colors_extra[k] || !a->out_extra[k]) { emd_set_error("forward: extra colour set %d: null colours / output", k); return EMD_ERR_INVALID; }
    if ((a->flags & EMD_FLAG_SDEV_TANFOV) && !a->settings_dev) { emd_set_error("forward: EMD_FLAG_SDEV_TANFOV without settings_dev"); return EMD_ERR_INVALID; }
    const int gx = (a->s.image_width + EMD_TILE_X - 1) / EMD_TILE_X, gy = (a->s.image_height + EMD_TILE_Y - 1) / EMD_TILE_Y;
    GeomWs g; BinWs b; ImgWs im;
    emd_carve_geom(a->geom_ws, N, &g);
    emd_carve_bin(a->bin_ws, a->bin_capacity, gx * gy, &b);
    emd_carve_img(a->img_ws, a->s.image_height, a->s.image_width, &im);
    if (g.bytes > a->geom_bytes || b.bytes > a->bin_bytes || im.bytes > a->img_bytes) {
        emd_set_error("forward: workspace too small (geom %zu/%zu bin %zu/%zu img %zu/%zu)", a->geom_bytes, g.bytes,
                      a->bin_bytes, b.bytes, a->img_bytes, im.bytes);
        return EMD_ERR_WORKSPACE;
    }
    const bool dbg = a->s.debug != 0;
#define STAGE_SYNC(name)                                                                                     \
    if (dbg) {                                                                                               \
        hipError_t e_ = hipStreamSynchronize(st);                                                            \
        if (e_ != hipSuccess) { emd_set_error("stage %s failed: %s", name, hipGetErrorString(e_)); return EMD_ERR_HIP; } \
    }
    PreArgs pa;   // (the status word is written by the binning stage)
    pa.s = a->s; pa.N = N; pa.M = a->sh_coeffs; pa.flags = a->flags;
    pa.means3D = a->means3D; pa.shs = a->shs; pa.colors_precomp = a->colors_precomp; pa.opacities = a->opacities;
    pa.scales = a->scales; pa.rotations = a->rotations; pa.cov3D_precomp = a->cov3D_precomp;
    pa.motion = a->motion; pa.radii = a->radii; pa.g = g; pa.status = a->status; pa.sdev = a->settings_dev;
    pa.shs_res0 = a->shs_residual[0]; pa.shs_res1 = a->shs_residual[1];
    if ((pa.shs_res0 || pa.shs_res1) && !a->shs) { emd_set_error("raster_forward: shs_residual needs shs"); return EMD_ERR_INVALID; }
    if (pa.shs_res1 && !pa.shs_res0) { pa.shs_res0 = pa.shs_res1; pa.shs_res1 = nullptr; }
    if (!(a->flags & EMD_FLAG_MOTION)) memset(&pa.motion, 0, sizeof(pa.motion));
    emd_prof_begin(PROF_PREPROCESS, st);
    // With an auxiliary stream the colour half of K1 (SH colour, clamp bits, colour Jacobian: needed by K6 only) runs BESIDE the binning
    // stage: fork after the geometry half, join in front of K6.  (Measured in round 3 and OFF by default in the binding: the halves cost
    // 0.075 + 0.122 ms apart against 0.159 ms fused, and beside the colour half's 31 250 workgroups the binning kernels wait for slots:
    // 684 against 700 it/s.  Kept as an option for callers whose binning is longer.)  Both edges are events, so a capturing stream records them as graph
    // dependencies.  Every return between fork and join joins first: the caller may free or reuse the workspaces on an error.
    hipStream_t aux = (hipStream_t)a->aux_stream;
    const bool split = aux != nullptr && aux != st && N > 0 && !dbg;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    auto join = [&]() -> int {
        if (!ev_join) return EMD_OK;
        hipError_t e1 = hipStreamWaitEvent(st, ev_join, 0);
        (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join);
        ev_fork = ev_join = nullptr;
        if (e1 != hipSuccess) { emd_set_error("forward: joining the auxiliary stream failed: %s", hipGetErrorString(e1)); return EMD_ERR_HIP; }
        return EMD_OK;
    };
    rc = emd_launch_preprocess(pa, split ? 1 : 0, st);
    if (rc) return rc;
    if (split) {
        EMD_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        EMD_HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        EMD_HIP_CHECK(hipEventRecord(ev_fork, st));
        EMD_HIP_CHECK(hipStreamWaitEvent(aux, ev_fork, 0));
        rc = emd_launch_preprocess(pa, 2, aux);
        hipError_t e2 = hipEventRecord(ev_join, aux);
        if (rc || e2 != hipSuccess) { (void)join(); if (!rc) { emd_set_error("forward: hipEventRecord on the auxiliary stream failed"); rc = EMD_ERR_HIP; } return rc; }
    }
    STAGE_SYNC("preprocess");
    rc = emd_launch_binning(a->s, a->flags, N, g, b, a->bin_capacity, a->status, st);
    if (rc) { (void)join(); return rc; }
    STAGE_SYNC("binning");
    a->num_rendered = -1;
    a->num_visible = -1;
    if (!(a->flags & EMD_FLAG_NO_SYNC)) {
        EmdStatus hs;
        EMD_HIP_CHECK(hipMemcpyAsync(&hs, a->status, sizeof(hs), hipMemcpyDeviceToHost, st));
        EMD_HIP_CHECK(hipStreamSynchronize(st));
        a->num_rendered = hs.num_rendered;
        a->num_visible = hs.num_visible;
        if (hs.overflow & 2u) {
            (void)join();
            emd_set_error("forward: a visible Gaussian lies beyond 65 536 x the near plane; repeat with EMD_FLAG_WIDE_DEPTH_SORT");
            return EMD_ERR_DEPTH_RANGE;
        }
        if (hs.overflow) {
            (void)join();
            emd_set_error("forward: %u (tile, Gaussian) pairs exceed bin_capacity %lld", hs.num_rendered, (long long)a->bin_capacity);
            return EMD_ERR_CAPACITY;
        }
    }
    { int jrc = join(); if (jrc) return jrc; }          // K6 reads the colours
    emd_prof_switch(PROF_RANGES, PROF_RENDER_FWD, st);
    EmdExtra ex;
    memset(&ex, 0, sizeof(ex));
    ex.num = a->num_extra;
    for (int k = 0; k < a->num_extra; k++) { ex.colors[k] = a->colors_extra[k]; ex.out[k] = a->out_extra[k]; }
    if (a->loop_stats && a->num_extra > 0) { emd_set_error("forward: loop_stats (diagnostic counters) only without extra colour sets"); return EMD_ERR_INVALID; }
    rc = emd_launch_render_forward(a->s, a->settings_dev, a->flags, g, b, im, a->out_color, a->out_depth, a->out_normal, a->out_alpha, &ex,
                                   (unsigned long long*)a->loop_stats, st);
    emd_prof_end(PROF_RENDER_FWD, st);
    if (rc) return rc;
    STAGE_SYNC("render_forward");
    return EMD_OK;
}

int emd_raster_backward(const EmdBwdArgs* a, void* hip_stream) {
    if (!a) { emd_set_error("backward: null args"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    const int N = a->num_gaussians;
    int rc = check_common(a->s, N, a->sh_coeffs, a->means3D, a->shs, a->colors_precomp, a->opacities, a->scales,
                          a->rotations, a->cov3D_precomp, a->flags, a->motion);
    if (rc) return rc;
    if (!a->radii || !a->geom_ws || !a->bin_ws || !a->img_ws || !a->bwd_ws || !a->status || !a->out_color || !a->out_depth) {
        emd_set_error("backward: null state pointer"); return EMD_ERR_INVALID;
    }
    const int gx = (a->s.image_width + EMD_TILE_X - 1) / EMD_TILE_X, gy = (a->s.image_height + EMD_TILE_Y - 1) / EMD_TILE_Y;
    GeomWs g; BinWs b; ImgWs im;
    emd_carve_geom((void*)a->geom_ws, N, &g);
    emd_carve_bin((void*)a->bin_ws, a->bin_capacity, gx * gy, &b);
    emd_carve_img((void*)a->img_ws, a->s.image_height, a->s.image_width, &im);
    if (a->num_extra < 0 || a->num_extra > EMD_MAX_EXTRA) { emd_set_error("backward: num_extra %d not in 0..%d", a->num_extra, EMD_MAX_EXTRA); return EMD_ERR_INVALID; }
    const size_t need = (size_t)(N > 0 ? N : 1) * emd_bwd_stride(a->num_extra) * sizeof(float);
    if (g.bytes > a->geom_bytes || b.bytes > a->bin_bytes || im.bytes > a->img_bytes || need > a->bwd_bytes) {
        emd_set_error("backward: workspace too small"); return EMD_ERR_WORKSPACE;
    }
    if ((a->flags & EMD_FLAG_ABSGRAD) && !a->dL_dmeans2D_abs) { emd_set_error("backward: EMD_FLAG_ABSGRAD without dL_dmeans2D_abs"); return EMD_ERR_INVALID; }
    const bool dbg = a->s.debug != 0;
    // the two halves of the pass may arrive as two calls (EMD_FLAG_BWD_RENDER_ONLY, then EMD_FLAG_BWD_PROJECT_ONLY on the same workspaces)
    const bool do_render = !(a->flags & EMD_FLAG_BWD_PROJECT_ONLY), do_project = !(a->flags & EMD_FLAG_BWD_RENDER_ONLY);
    if (!do_render && !do_project) { emd_set_error("backward: EMD_FLAG_BWD_RENDER_ONLY and EMD_FLAG_BWD_PROJECT_ONLY exclude each other"); return EMD_ERR_INVALID; }
    if (do_render) {
        emd_prof_begin(PROF_OTHER, st);
        if (!(a->flags & EMD_FLAG_BWD_WS_CLEAN)) { int zrc = emd_zero_async(a->bwd_ws, need, st); if (zrc) return zrc; }
        float* pose_grad = nullptr;       // accumulated by K8 with atomics; cleared by K7's first workgroup
        int pose_grad_n = 0;
        if ((a->flags & EMD_FLAG_MOTION) && a->dL_dactor_pose && a->motion.num_actors > 0) {
            pose_grad = a->dL_dactor_pose;
            pose_grad_n = a->motion.num_actors * EMD_ACTOR_STRIDE;
        }
        emd_prof_switch(PROF_OTHER, PROF_RENDER_BWD, st);
        EmdExtra ex;
        memset(&ex, 0, sizeof(ex));
        ex.num = a->num_extra;
        for (int k = 0; k < a->num_extra; k++) {
            if (!a->colors_extra[k] || !a->out_extra[k]) { emd_set_error("backward: extra colour set %d: null colours / forward output", k); return EMD_ERR_INVALID; }
            ex.colors[k] = a->colors_extra[k]; ex.out[k] = (float*)a->out_extra[k]; ex.dL_dout[k] = a->dL_dextra[k];
        }
        if (a->pair_stats && (a->num_extra > 0 || (a->flags & EMD_FLAG_ABSGRAD) || a->dL_dnormal)) {
            emd_set_error("backward: pair_stats (diagnostic counters) only with the plain call: no extra colour sets, absgrad or normal gradient");
            return EMD_ERR_INVALID;
        }
        rc = emd_launch_render_backward(a->s, a->settings_dev, a->flags, g, b, im, a->out_color, a->out_depth, a->out_normal, a->dL_dcolor,
                                        a->dL_ddepth, a->dL_dalpha, a->dL_dnormal, &ex, (float*)a->bwd_ws, pose_grad, pose_grad_n,
                                        (unsigned long long*)a->pair_stats, st);
        if (rc) return rc;
        if (!do_project && a->dL_dsh_color) {      // the SH factor right behind K7: a view-parallel step starts gathering it under K8
            rc = emd_launch_sh_factor(N, a->radii, g, (const float*)a->bwd_ws, emd_bwd_stride(a->num_extra), a->dL_dsh_color, st);
            if (rc) return rc;
        }
        STAGE_SYNC("render_backward");
        if (!do_project) { emd_prof_end(PROF_RENDER_BWD, st); return EMD_OK; }
    }
    PreBwdArgs pb;
    pb.s = a->s; pb.N = N; pb.M = a->sh_coeffs; pb.flags = a->flags;
    pb.means3D = a->means3D; pb.shs = a->shs; pb.colors_precomp = a->colors_precomp; pb.opacities = a->opacities;
    pb.scales = a->scales; pb.rotations = a->rotations; pb.cov3D_precomp = a->cov3D_precomp;
    pb.motion = a->motion; pb.radii = a->radii; pb.g = g; pb.grad_rec = (float*)a->bwd_ws;
    if (!(a->flags & EMD_FLAG_MOTION)) memset(&pb.motion, 0, sizeof(pb.motion));
    pb.dL_dmeans3D = a->dL_dmeans3D; pb.dL_dmeans2D = a->dL_dmeans2D; pb.dL_dmeans2D_abs = a->dL_dmeans2D_abs;
    pb.dL_dshs = a->dL_dshs; pb.dL_dcolors = a->dL_dcolors; pb.dL_dopacities = a->dL_dopacities;
    pb.dL_dscales = a->dL_dscales; pb.dL_drotations = a->dL_drotations; pb.dL_dcov3D = a->dL_dcov3D;
    pb.dL_dactor_pose = a->dL_dactor_pose; pb.dL_dresidual_dx = a->dL_dresidual_dx; pb.dL_dresidual_dq = a->dL_dresidual_dq;
    pb.dL_dsh_color = a->dL_dsh_color;
    pb.sdev = a->settings_dev;
    pb.bwd_stride = emd_bwd_stride(a->num_extra); pb.num_extra = a->num_extra;
    for (int k = 0; k < EMD_MAX_EXTRA; k++) pb.dL_dextra[k] = k < a->num_extra ? a->dL_dcolors_extra[k] : nullptr;
    if (do_render) emd_prof_switch(PROF_RENDER_BWD, PROF_PREPROCESS_BWD, st); else emd_prof_begin(PROF_PREPROCESS_BWD, st);
    rc = emd_launch_preprocess_backward(pb, st);
    emd_prof_end(PROF_PREPROCESS_BWD, st);
    if (rc) return rc;
    STAGE_SYNC("preprocess_backward");
    return EMD_OK;
}

int emd_raster_export_binning(const EmdDims* dims, const void* geom_ws, size_t geom_bytes, const void* bin_ws, size_t bin_bytes,
                              int64_t num_rendered, uint64_t* keys, uint32_t* ids, uint32_t* ranges, uint32_t* quad_masks,
                              void* hip_stream) {
    if (!dims || !bin_ws || !geom_ws) { emd_set_error("export_binning: null argument"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    const int gx = (dims->image_width + EMD_TILE_X - 1) / EMD_TILE_X, gy = (dims->image_height + EMD_TILE_Y - 1) / EMD_TILE_Y;
    GeomWs g; BinWs b;
    emd_carve_geom((void*)geom_ws, dims->num_gaussians, &g);
    emd_carve_bin((void*)bin_ws, dims->bin_capacity, gx * gy, &b);
    if (g.bytes > geom_bytes || b.bytes > bin_bytes || num_rendered > dims->bin_capacity || num_rendered < 0) { emd_set_error("export_binning: bad sizes"); return EMD_ERR_WORKSPACE; }
    // the sort moves (tile id, Gaussian id) pairs; upstream's 64-bit key is tile id << 32 | depth bits of the Gaussian
    if ((keys || ids || quad_masks) && num_rendered) { int rc = emd_launch_export_keys(num_rendered, g, b, keys, ids, quad_masks, st); if (rc) return rc; }
    if (ranges) EMD_HIP_CHECK(hipMemcpyAsync(ranges, b.ranges, (size_t)gx * gy * 8, hipMemcpyDeviceToDevice, st));
    return EMD_OK;
}

int emd_raster_export_geometry(const EmdDims* dims, const void* geom_ws, size_t geom_bytes, float* means2D,
                               float* depths, float* conic_opacity, float* rgb, float* normal, uint32_t* tiles_touched,
                               void* hip_stream) {
    if (!dims || !geom_ws) { emd_set_error("export_geometry: null argument"); return EMD_ERR_INVALID; }
    GeomWs g;
    emd_carve_geom((void*)geom_ws, dims->num_gaussians, &g);
    if (g.bytes > geom_bytes) { emd_set_error("export_geometry: workspace too small"); return EMD_ERR_WORKSPACE; }
    return emd_launch_export_geometry(dims->num_gaussians, g, means2D, depths, conic_opacity, rgb, normal, tiles_touched,
                                      (hipStream_t)hip_stream);
}

int emd_motion_forward(int32_t n, const float* means, const float* quats, const float* opacities,
                       const EmdMotion* motion, float* world_means, float* world_quats, float* opacities_out,
                       void* hip_stream) {
    if (n < 0 || !means || !motion) { emd_set_error("motion_forward: bad argument"); return EMD_ERR_INVALID; }
    if (motion->actor_id && (!motion->actor_pose || motion->num_actors <= 0)) { emd_set_error("motion_forward: actor_id without actor_pose"); return EMD_ERR_INVALID; }
    return emd_launch_motion_forward(n, means, quats, opacities, *motion, world_means, world_quats, opacities_out,
                                     (hipStream_t)hip_stream);
}

int emd_motion_backward(int32_t n, const float* means, const float* quats, const float* opacities,
                        const EmdMotion* motion, const float* dL_dworld_means, const float* dL_dworld_quats,
                        const float* dL_dopacities_out, float* dL_dmeans, float* dL_dquats, float* dL_dopacities,
                        float* dL_dactor_pose, float* dL_dresidual_dx, float* dL_dresidual_dq, void* hip_stream) {
    if (n < 0 || !means || !motion) { emd_set_error("motion_backward: bad argument"); return EMD_ERR_INVALID; }
    if (motion->actor_id && (!motion->actor_pose || motion->num_actors <= 0)) { emd_set_error("motion_backward: actor_id without actor_pose"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    if (dL_dactor_pose && motion->num_actors > 0)
        { int zrc = emd_zero_async(dL_dactor_pose, (size_t)motion->num_actors * EMD_ACTOR_STRIDE * sizeof(float), st); if (zrc) return zrc; }
    return emd_launch_motion_backward(n, means, quats, opacities, *motion, dL_dworld_means, dL_dworld_quats,
                                      dL_dopacities_out, dL_dmeans, dL_dquats, dL_dopacities, dL_dactor_pose,
                                      dL_dresidual_dx, dL_dresidual_dq, st);
}

int emd_sh_forward(int32_t n, int32_t degree, int32_t sh_coeffs, const float* dirs, const float* coeffs, float* rgb,
                   void* hip_stream) {
    if (n < 0 || degree < 0 || degree > 3 || sh_coeffs < (degree + 1) * (degree + 1) || !dirs || !coeffs || !rgb) {
        emd_set_error("sh_forward: bad argument (n=%d degree=%d K=%d)", n, degree, sh_coeffs); return EMD_ERR_INVALID;
    }
    return emd_launch_sh_forward(n, degree, sh_coeffs, dirs, coeffs, rgb, (hipStream_t)hip_stream);
}

int emd_sh_backward(int32_t n, int32_t degree, int32_t sh_coeffs, const float* dirs, const float* coeffs,
                    const float* dL_drgb, float* dL_dcoeffs, float* dL_ddirs, void* hip_stream) {
    if (n < 0 || degree < 0 || degree > 3 || sh_coeffs < (degree + 1) * (degree + 1) || !dirs || !coeffs || !dL_drgb) {
        emd_set_error("sh_backward: bad argument (n=%d degree=%d K=%d)", n, degree, sh_coeffs); return EMD_ERR_INVALID;
    }
    return emd_launch_sh_backward(n, degree, sh_coeffs, dirs, coeffs, dL_drgb, dL_dcoeffs, dL_ddirs, (hipStream_t)hip_stream);
}

int emd_sh_grad_from_factors(int32_t n, int32_t num_views, int32_t degree, int32_t sh_coeffs, const float* means3D,
                             const EmdMotion* motion, int32_t pose_per_view, const float* campos, const float* sh_color_grads,
                             float scale, float* dL_dshs, void* hip_stream) {
    if (n < 0 || num_views < 1 || degree < 0 || degree > 3 || sh_coeffs < (degree + 1) * (degree + 1) || sh_coeffs > 16 ||
        (n > 0 && (!means3D || !campos || !sh_color_grads || !dL_dshs))) {
        emd_set_error("sh_grad_from_factors: bad argument"); return EMD_ERR_INVALID;
    }
    EmdMotion mo;
    memset(&mo, 0, sizeof(mo));
    if (motion) mo = *motion;
    return emd_launch_sh_grad_from_factors(n, num_views, degree, sh_coeffs, means3D, mo, pose_per_view ? 1 : 0, campos, sh_color_grads, scale, dL_dshs,
                                           (hipStream_t)hip_stream);
}

int emd_densification_stats(int32_t n, const int32_t* radii, const float* dL_dmeans2D, float* grad_accum, float* denom,
                            float* max_radii2D, void* hip_stream) {
    if (n < 0 || (n > 0 && (!radii || !dL_dmeans2D))) { emd_set_error("densification_stats: bad argument"); return EMD_ERR_INVALID; }
    return emd_launch_densification_stats(n, radii, dL_dmeans2D, grad_accum, denom, max_radii2D, (hipStream_t)hip_stream);
}

int emd_actor_pose_forward(int32_t num_actors, const float* q_f, const float* t_f, const uint8_t* valid, const float* dt,
                           const float* dq, float* pose, const int32_t* frame_dev, void* hip_stream) {
    if (num_actors < 0 || (num_actors > 0 && (!q_f || !t_f || !pose))) { emd_set_error("actor_pose_forward: bad argument"); return EMD_ERR_INVALID; }
    return emd_launch_actor_pose_forward(num_actors, q_f, t_f, valid, dt, dq, pose, frame_dev, (hipStream_t)hip_stream);
}

int emd_actor_pose_backward(int32_t num_actors, const float* q_f, const float* dt, const float* dq, const float* dL_dpose,
                            float* dL_dq_f, float* dL_dt_f, float* dL_ddt, float* dL_ddq, const int32_t* frame_dev, void* hip_stream) {
    if (num_actors < 0 || (num_actors > 0 && (!q_f || !dL_dpose || !dL_dq_f || !dL_dt_f))) { emd_set_error("actor_pose_backward: bad argument"); return EMD_ERR_INVALID; }
    return emd_launch_actor_pose_backward(num_actors, q_f, dt, dq, dL_dpose, dL_dq_f, dL_dt_f, dL_ddt, dL_ddq, frame_dev, (hipStream_t)hip_stream);
}

int emd_select_step_inputs(const EmdStepSelect* a, void* hip_stream) {
    if (!a || !a->sel) { emd_set_error("select_step_inputs: null args"); return EMD_ERR_INVALID; }
    if (a->rows < 0 || a->row_floats < 0 || (a->row_floats > 0 && (!a->table || !a->out_row)) || (a->k_fine_out && a->k_until < 1)) {
        emd_set_error("select_step_inputs: bad sizes / null table"); return EMD_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_select_step_inputs, dim3(1), dim3(EMD_WAVE), 0, (hipStream_t)hip_stream, *a);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_l1_loss_ws(int64_t n, const float* a, const float* b, float* loss, float* grad, uint32_t* scratch, void* hip_stream) {
    if (n < 0 || !loss || (n > 0 && !a)) { emd_set_error("l1_loss: bad argument"); return EMD_ERR_INVALID; }
    if (((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)grad & 15)) { emd_set_error("l1_loss: a, b, grad must be 16-byte aligned"); return EMD_ERR_INVALID; }
    if ((uintptr_t)scratch & 7) { emd_set_error("l1_loss: scratch must be 8-byte aligned"); return EMD_ERR_INVALID; }
    return emd_launch_l1_loss((size_t)n, a, b, loss, grad, scratch, (hipStream_t)hip_stream);
}

int emd_l1_loss(int64_t n, const float* a, const float* b, float* loss, float* grad, void* hip_stream) {
    return emd_l1_loss_ws(n, a, b, loss, grad, nullptr, hip_stream);
}

int emd_abs_mean_backward(int64_t n, const float* x, const float* g, float* grad, void* hip_stream) {
    if (n < 0 || (n > 0 && (!x || !g || !grad))) { emd_set_error("abs_mean_backward: bad argument"); return EMD_ERR_INVALID; }
    if (((uintptr_t)x & 15) || ((uintptr_t)grad & 15)) { emd_set_error("abs_mean_backward: x, grad must be 16-byte aligned"); return EMD_ERR_INVALID; }
    return emd_launch_abs_mean_backward((size_t)n, x, g, grad, (hipStream_t)hip_stream);
}

int emd_residual_l1_backward(int64_t n, const float* up_a, const float* up_b, const float* x_a, const float* x_b, const float* g_a,
                             const float* g_b, float* grad_a, float* grad_b, void* hip_stream) {
    if (n < 0 || (n > 0 && (!x_a || !x_b || !grad_a || !grad_b))) { emd_set_error("residual_l1_backward: bad argument"); return EMD_ERR_INVALID; }
    if ((((uintptr_t)up_a) | ((uintptr_t)up_b) | ((uintptr_t)x_a) | ((uintptr_t)x_b) | ((uintptr_t)grad_a) | ((uintptr_t)grad_b)) & 15) {
        emd_set_error("residual_l1_backward: the tensors must be 16-byte aligned"); return EMD_ERR_INVALID;
    }
    return emd_launch_residual_l1_backward((size_t)n, up_a, up_b, x_a, x_b, g_a, g_b, grad_a, grad_b, (hipStream_t)hip_stream);
}

int emd_activations_forward(int32_t n, const float* log_scales, float* scales, const float* raw_quats, float* quats,
                            const float* opacity_logits, float* opacities, void* hip_stream) {
    if (n < 0) { emd_set_error("activations_forward: negative n"); return EMD_ERR_INVALID; }
    return emd_launch_activations(n, log_scales, scales, raw_quats, quats, opacity_logits, opacities, (hipStream_t)hip_stream);
}

}  // extern "C"
