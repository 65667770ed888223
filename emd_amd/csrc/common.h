// common.h -- shared host/device definitions of the gfx950 street-Gaussian rasterizer.
// Wave = 64 lanes, 256-thread workgroups, 16x16 tiles.  No CUDA compatibility layer: HIP for CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/emd_raster.h"

#define EMD_WAVE 64
#define EMD_BLOCK 256
#define EMD_NUM_TILE_PIX (EMD_TILE_X * EMD_TILE_Y)

// Per-Gaussian projected record: 4 x float4 = 64 B, one half cache line, gathered by the render kernels.
//   r0 = (x_pix, y_pix, depth, opacity)   r1 = (conic A, B, C, bits)   r2 = (r, g, b, 0)   r3 = (nx, ny, nz, 0)
// r1.w bits: 0..2 = SH colour channel clamped (zero gradient).
#define EMD_REC_F4 4

// Per-Gaussian gradient accumulator written by the render backward: 12 floats = 48 B (3 x float4)
//   [0..1] d/d mean2D (pixel units)  [2] d/d depth  [3] d/d opacity
//   [4..6] d/d conic (A,B,C)         [7..9] d/d rgb
//   [10..11] sum |d/d mean2D| (EMD_FLAG_ABSGRAD)
#define EMD_BWD_STRIDE 12

struct GeomWs {
    float4* rec;             // [N][4]
    float4* shjac;           // [N][3] d rgb_c / d (unit view direction): row c = (d/dx, d/dy, d/dz, -); written by K1 for
                             // visible Gaussians so that K8 need not re-read the 192 B of SH coefficients
    uint2* binrec;           // [N] by Gaussian id (K1): x = ENUMERATED tile rectangle x0 | y0 << 10 | width << 20 (upstream's rectangle cut down to the
                             //   alpha >= 1/255 box) | skip-left / skip-right << 30, y = its height | skip-top / skip-bottom << 10 | upstream's tiles
                             //   touched << 12.  A skip bit: the outer quadrant column / row of the rectangle's first / last tile lies outside the box
    uint32_t* depth_key;     // [N] by Gaussian id: float bits of the view depth, 0xFFFFFFFF when not visible (K1)
    uint32_t* gkeys[2];      // [N] ping-pong of the Gaussian depth sort
    uint32_t* gvals[2];      // [N] Gaussian ids in depth order after the sort (gvals[1])
    uint2* bin_s;            // [N] the same records in depth order
    uint2* bin_t;            // [N] ping-pong partner of bin_s: the records travel through the depth passes as a second value (round 5)
    uint32_t* ghist;         // radix histograms of the depth sort [bins][ceil(N / sort tile)]
    uint32_t* sort_count;    // [1] visible Gaussians V: published by the first depth pass (which compacts), read by the later ones
    uint32_t* block_sums;    // [ceil(N/256)] pairs emitted per block of 256 depth-ordered Gaussians (added up by the last depth pass)
    size_t bytes;
};

struct BinWs {
    uint32_t* tkeys[2];      // [capacity] ping-pong: tile id of every (tile, Gaussian) pair
    uint32_t* vals[2];       // [capacity] list word: Gaussian id | quadrant mask of the pair << 28 (footprint.h)
    uint32_t* ranges;        // [T][2]
    uint32_t* tile_order;    // [T] tile ids by descending list length: dispatch order of the render kernels
    uint32_t* hist;          // radix histograms [bins][num_sort_blocks]
    uint32_t* surv;          // [4 * capacity] per (tile, quadrant): the Gaussian ids of the list entries whose footprint reaches the quadrant, in
                             //   list order, written by the render forward for the render backward (at 4 * range start + quadrant * list length)
    uint32_t* quad_need;     // [4 * T] how many of them lie in front of the quadrant's deepest contributor (what the backward walks)
    int sorted_buf;          // which ping-pong buffer holds the sorted list after forward (fixed by #passes)
    size_t bytes;
};

struct ImgWs {
    float* final_T;          // [H*W]
    uint32_t* n_contrib;     // [H*W]
    size_t bytes;
};

static inline size_t emd_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// sort geometry
#define EMD_SORT_ITEMS 8                        // keys per thread
#define EMD_SORT_TILE (EMD_BLOCK * EMD_SORT_ITEMS)
#define EMD_RADIX_BITS 8
#define EMD_RADIX_BINS (1 << EMD_RADIX_BITS)

static inline int emd_tile_bits(int num_tiles) {
    int b = 0;
    while ((1 << b) < num_tiles) b++;
    return b;
}
// Upstream key = tile_id << 32 | depth bits.  Sorted here as radix passes over the depth bits of the visible Gaussians,
// then ceil(tile bits / 8) passes of equal width over the tile ids of the duplicates (binning.hip).
// Depth bits: a visible Gaussian lies beyond the near plane, and positive floats order like their bit patterns, so the sort runs on
// (bits - bits(near plane)), which fits 27 bits for depths up to 65 536 x the near plane (13 km at 0.2 m): three passes of
// nine bits.  A key outside that range raises bit 1 of EmdStatus.overflow and the caller repeats the call with
// EMD_FLAG_WIDE_DEPTH_SORT: four passes of eight bits over all 32.
#define EMD_DEPTH_BITS_NARROW 9
#define EMD_DEPTH_PASSES_NARROW 3
#define EMD_DEPTH_RANGE_NARROW (EMD_DEPTH_BITS_NARROW * EMD_DEPTH_PASSES_NARROW)
#define EMD_DEPTH_PASSES_WIDE (32 / EMD_RADIX_BITS)
#define EMD_DEPTH_BINS_MAX (1 << EMD_DEPTH_BITS_NARROW)
static inline int emd_tile_passes(int num_tiles) { return (emd_tile_bits(num_tiles) + EMD_RADIX_BITS - 1) / EMD_RADIX_BITS; }
static inline int emd_tile_pass_bits(int num_tiles) {
    const int p = emd_tile_passes(num_tiles);
    return p ? (emd_tile_bits(num_tiles) + p - 1) / p : 0;
}

static inline void emd_carve_geom(void* base, int N, GeomWs* w) {
    char* p = (char*)base;
    size_t off = 0;
    const size_t n = (size_t)(N > 0 ? N : 1);
    w->rec = (float4*)(p + off); off = emd_align_up(off + n * EMD_REC_F4 * sizeof(float4), 256);
    w->shjac = (float4*)(p + off); off = emd_align_up(off + n * 3 * sizeof(float4), 256);
    w->binrec = (uint2*)(p + off); off = emd_align_up(off + n * 8, 256);
    w->depth_key = (uint32_t*)(p + off); off = emd_align_up(off + n * 4, 256);
    for (int i = 0; i < 2; i++) { w->gkeys[i] = (uint32_t*)(p + off); off = emd_align_up(off + n * 4, 256); }
    for (int i = 0; i < 2; i++) { w->gvals[i] = (uint32_t*)(p + off); off = emd_align_up(off + n * 4, 256); }
    w->bin_s = (uint2*)(p + off); off = emd_align_up(off + n * 8, 256);
    w->bin_t = (uint2*)(p + off); off = emd_align_up(off + n * 8, 256);
    size_t nsb = (n + EMD_SORT_TILE - 1) / EMD_SORT_TILE;
    w->ghist = (uint32_t*)(p + off); off = emd_align_up(off + nsb * EMD_DEPTH_BINS_MAX * 4, 256);
    w->sort_count = (uint32_t*)(p + off); off = emd_align_up(off + 16, 256);
    size_t nb = (n + EMD_BLOCK - 1) / EMD_BLOCK;
    w->block_sums = (uint32_t*)(p + off); off = emd_align_up(off + (nb + 8) * 4, 256);
    w->bytes = off + 256;
}

static inline void emd_carve_bin(void* base, int64_t capacity, int num_tiles, BinWs* w) {
    char* p = (char*)base;
    size_t off = 0;
    size_t cap = (size_t)(capacity > 0 ? capacity : 1);
    for (int i = 0; i < 2; i++) { w->tkeys[i] = (uint32_t*)(p + off); off = emd_align_up(off + cap * 4, 256); }
    for (int i = 0; i < 2; i++) { w->vals[i] = (uint32_t*)(p + off); off = emd_align_up(off + cap * 4, 256); }
    w->ranges = (uint32_t*)(p + off); off = emd_align_up(off + (size_t)num_tiles * 8, 256);
    w->tile_order = (uint32_t*)(p + off); off = emd_align_up(off + (size_t)num_tiles * 4, 256);
    size_t nsb = (cap + EMD_SORT_TILE - 1) / EMD_SORT_TILE;
    w->hist = (uint32_t*)(p + off); off = emd_align_up(off + nsb * EMD_RADIX_BINS * 4, 256);
    w->surv = (uint32_t*)(p + off); off = emd_align_up(off + cap * 16, 256);
    w->quad_need = (uint32_t*)(p + off); off = emd_align_up(off + (size_t)num_tiles * 16, 256);
    w->sorted_buf = emd_tile_passes(num_tiles) & 1;
    w->bytes = off + 256;
}

static inline void emd_carve_img(void* base, int H, int W, ImgWs* w) {
    char* p = (char*)base;
    size_t off = 0, hw = (size_t)H * W;
    w->final_T = (float*)(p + off); off = emd_align_up(off + hw * 4, 256);
    w->n_contrib = (uint32_t*)(p + off); off = emd_align_up(off + hw * 4, 256);
    w->bytes = off + 256;
}

// ---- error plumbing (api.hip) -------------------------------------------------------------------
void emd_set_error(const char* fmt, ...);
#define EMD_HIP_CHECK(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            emd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return EMD_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)
#define EMD_LAUNCH_CHECK() EMD_HIP_CHECK(hipGetLastError())

// Zero `bytes` (a multiple of 4, `p` 4-byte aligned) with a KERNEL on `st` (api.hip).  Not hipMemsetAsync: a memset node captured into a
// hipGraph on ROCm 7.2 / gfx950 replays with a corrupt 16-byte fill pattern from the second replay on (every fourth word of the
// destination came back as garbage: tests/test_boundary_gpu.py's graph test caught it), and a kernel node costs the same.
int emd_zero_async(void* p, size_t bytes, hipStream_t st);

// ---- per-stage HIP-event timing (api.hip); no-ops unless emd_profile_enable(1) ------------------------------
enum { PROF_PREPROCESS = 0, PROF_DUPLICATE, PROF_SORT, PROF_RANGES, PROF_RENDER_FWD, PROF_RENDER_BWD, PROF_PREPROCESS_BWD, PROF_OTHER };
void emd_prof_begin(int stage, hipStream_t st);
void emd_prof_end(int stage, hipStream_t st);
void emd_prof_switch(int ended, int started, hipStream_t st);   // one event closes `ended` and opens `started`

// ---- stage launchers (one per translation unit) -------------------------------------------------
struct PreArgs {
    EmdSettings s;
    int N, M, flags;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    EmdMotion motion;
    int32_t* radii;
    GeomWs g;
    EmdStatus* status;
    const float* sdev;       // device copy of bg / viewmatrix / projmatrix / campos (EmdFwdArgs.settings_dev) or null
    const float *shs_res0, *shs_res1;   // optional residuals of the SH coefficients (EmdFwdArgs.shs_residual) or null
};
int emd_launch_preprocess(const PreArgs& a, int part, hipStream_t st);       // preprocess.hip; part 0 = whole kernel, 1 = geometry half, 2 = colour half
int emd_launch_binning(const EmdSettings& s, int flags, int N, const GeomWs& g, const BinWs& b, int64_t capacity, EmdStatus* status,
                       hipStream_t st);                                       // binning.hip
int emd_launch_export_keys(int64_t D, const GeomWs& g, const BinWs& b, uint64_t* keys, uint32_t* ids, uint32_t* quad_masks, hipStream_t st);  // binning.hip
// extra colour sets composited by the same list walk as the main colours (EmdFwdArgs.colors_extra ...)
struct EmdExtra {
    int num;
    const float* colors[EMD_MAX_EXTRA];    // [N,3]
    float* out[EMD_MAX_EXTRA];             // [3,H,W]
    const float* dL_dout[EMD_MAX_EXTRA];   // [3,H,W] or null (backward)
};
// accumulator row of the render backward: EMD_BWD_STRIDE floats + (r, g, b, -) per extra colour set
static inline int emd_bwd_stride(int num_extra) { return EMD_BWD_STRIDE + 4 * num_extra; }
int emd_launch_render_forward(const EmdSettings& s, const float* sdev, int flags, const GeomWs& g, const BinWs& b, const ImgWs& im,
                              float* out_color, float* out_depth, float* out_normal, float* out_alpha, const EmdExtra* x,
                              unsigned long long* loop_stats, hipStream_t st);   // render.hip
int emd_launch_render_backward(const EmdSettings& s, const float* sdev, int flags, const GeomWs& g, const BinWs& b, const ImgWs& im,
                               const float* out_color, const float* out_depth, const float* out_normal,
                               const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                               const float* dL_dnormal, const EmdExtra* x, float* grad_rec, float* zero_buf, int zero_n,
                               unsigned long long* pair_stats, hipStream_t st);  // render.hip (zero_buf: small table cleared by block 0 for K8)
struct PreBwdArgs {
    EmdSettings s;
    int N, M, flags;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    EmdMotion motion;
    const int32_t* radii;
    GeomWs g;
    float* grad_rec;        // [N][bwd_stride]; read, and cleared row by row under EMD_FLAG_BWD_WS_CLEAN
    int bwd_stride, num_extra;
    float* dL_dextra[EMD_MAX_EXTRA];   // [N,3] gradient of every extra colour set
    float *dL_dmeans3D, *dL_dmeans2D, *dL_dmeans2D_abs, *dL_dshs, *dL_dcolors, *dL_dopacities, *dL_dscales,
        *dL_drotations, *dL_dcov3D, *dL_dactor_pose, *dL_dresidual_dx, *dL_dresidual_dq, *dL_dsh_color;
    const float* sdev;
};

// Camera-dependent settings from their device copy (uniform scalar loads), replacing the by-value fields.
__device__ __forceinline__ void emd_settings_from_device(EmdSettings& S, const float* __restrict__ sdev, int flags) {
    if (!sdev) return;
    if (flags & EMD_FLAG_SDEV_TANFOV) { S.tanfovx = sdev[38]; S.tanfovy = sdev[39]; }
#pragma unroll
    for (int k = 0; k < 3; k++) S.bg[k] = sdev[k];
#pragma unroll
    for (int k = 0; k < 16; k++) { S.viewmatrix[k] = sdev[3 + k]; S.projmatrix[k] = sdev[19 + k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) S.campos[k] = sdev[35 + k];
}
int emd_launch_preprocess_backward(const PreBwdArgs& a, hipStream_t st);     // preprocess.hip
// the clamp-masked colour gradient of every Gaussian (the factor of its rank-one dL/dshs) from the render backward's accumulator rows
int emd_launch_sh_factor(int N, const int32_t* radii, const GeomWs& g, const float* grad_rec, int bwd_stride, float* dL_dsh_color, hipStream_t st);
