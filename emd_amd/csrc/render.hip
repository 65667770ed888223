// render.hip -- K6 per-tile front-to-back alpha compositing and K7 its backward.  gfx950.
// [UPSTREAM K6/K7; call site S3Gaussian/gaussian_renderer/__init__.py:145-155,
//  outputs consumed at :158-168,299-301 and S3Gaussian/train.py:226-368]
//
// Work decomposition (wave64):
//   - one 256-thread workgroup per 16x16 tile; each of its 4 waves owns an 8x8 pixel quadrant (spatially compact,
//     so "no lane of this wave is touched by Gaussian j" is common and whole waves skip work on a ballot);
//   - the tile's depth-sorted list is staged through LDS 256 entries at a time: lane t gathers the 48/64-byte
//     projected record of entry t (3-4 dwordx4 loads from one half cache line), then every lane walks the staged
//     chunk with broadcast LDS reads;
//   - K7 walks the same list back to front.  Per-(pixel, Gaussian) partial derivatives are reduced across the wave
//     with DPP row shifts/broadcasts (6 v_add_f32_dpp per value, no LDS traffic), lane 63 of each wave adds the
//     wave sums into an LDS accumulator row for the staged chunk, and after the chunk one lane per entry issues
//     the global float atomics: at most one atomic row per (Gaussian, tile) instead of one per (Gaussian, pixel).
#include <stdlib.h>

#include "common.h"
#include "device_utils.h"

namespace {

struct RenderDims {
    int W, H, gx, gy;
    float bg[3];
};

__device__ __forceinline__ void tile_pixel(const RenderDims& d, uint32_t tile, uint32_t tid, int& px, int& py) {
    const uint32_t tx = tile % (uint32_t)d.gx, ty = tile / (uint32_t)d.gx;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    px = (int)(tx * EMD_TILE_X + (wave & 1) * 8 + (lane & 7));
    py = (int)(ty * EMD_TILE_Y + (wave >> 1) * 8 + (lane >> 3));
}

// blockIdx -> tile.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one), so runs of
// XCD_CHUNK horizontally adjacent tiles are handed to the same XCD: neighbouring tiles share most of their
// Gaussians, which then stay in that XCD's L2.  Fine-grained interleave (not one band per XCD) keeps the load
// of sky rows and ground rows spread over all XCDs.  Speed only, never correctness; the grid is padded to a
// multiple of 8 * XCD_CHUNK and surplus workgroups exit.
#define XCD_CHUNK 4
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b) {
    const uint32_t xcd = b % 8, k = b / 8;
    return ((k / XCD_CHUNK) * 8 + xcd) * XCD_CHUNK + (k % XCD_CHUNK);
}
static inline unsigned padded_tile_grid(int T) { return (unsigned)((T + 8 * XCD_CHUNK - 1) / (8 * XCD_CHUNK) * (8 * XCD_CHUNK)); }

// Gaussian exponent, evaluated identically (explicit FMAs, no further contraction) in K6 and K7 so that both
// kernels take the same skip decisions for the same (pixel, Gaussian) pair.
__device__ __forceinline__ float gauss_power(float A, float B, float C, float dx, float dy) {
#pragma clang fp contract(off)
    const float q = __builtin_fmaf(A * dx, dx, (C * dy) * dy);
    return __builtin_fmaf(-0.5f, q, -((B * dx) * dy));
}

// ---------------------------------------------------------------------------------------------------
// Exact-conservative footprint culling.  A pixel receives alpha >= 1/255 from a Gaussian only if
// power >= -tau with tau = ln(255 * opacity), i.e. inside the ellipse d^T Conic d <= 2 tau, whose axis-aligned
// bounding box has half-extents sqrt(2 tau * cov_xx), sqrt(2 tau * cov_yy) (cov = Conic^-1).  That box is much
// tighter than the 3-sigma square of the largest eigenvalue that defines the tile list (anisotropic or faint
// Gaussians), so: the lane that stages entry t computes the box once and a 4-bit mask of the 8x8 quadrants it
// overlaps; ballots compact the 256 staged entries into one index list per quadrant (order preserved), and each
// wave walks only its own list.  Pixels outside the box would have been rejected by the alpha test anyway, so the
// result is unchanged; the box carries a small margin so float rounding can never cull a contributing pixel.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quadrant_mask(const float4& r0, const float4& r1, float tile_x0, float tile_y0) {
    const float o = r0.w;
    if (!(o >= (1.f / 255.f))) return 0u;            // can never reach alpha >= 1/255 (also catches NaN)
    const float det = r1.x * r1.z - r1.y * r1.y;
    if (!(det > 0.f)) return 0xFu;                   // degenerate conic: no culling
    const float tau2 = 2.f * __logf(255.f * o);
    const float inv = 1.f / det;
    const float bx = sqrtf(tau2 * r1.z * inv) * 1.0001f + 0.01f;
    const float by = sqrtf(tau2 * r1.x * inv) * 1.0001f + 0.01f;
    if (!(bx == bx) || !(by == by)) return 0xFu;
    const float lx = r0.x - bx, hx = r0.x + bx, ly = r0.y - by, hy = r0.y + by;
    // quadrant q covers pixel centres [x0 + 8(q&1), +7] x [y0 + 8(q>>1), +7]
    const bool xl = lx <= tile_x0 + 7.f && hx >= tile_x0, xr = lx <= tile_x0 + 15.f && hx >= tile_x0 + 8.f;
    const bool yt = ly <= tile_y0 + 7.f && hy >= tile_y0, yb = ly <= tile_y0 + 15.f && hy >= tile_y0 + 8.f;
    return (xl && yt ? 1u : 0u) | (xr && yt ? 2u : 0u) | (xl && yb ? 4u : 0u) | (xr && yb ? 8u : 0u);
}

// Compacts the staged slots whose qmask has bit q set into s_list[q][...] (ascending slot order).
// Needs one barrier before (counts) and one after (lists); returns this wave's list length.
__device__ __forceinline__ uint32_t build_quadrant_lists(uint32_t qmask, uint16_t (*s_list)[EMD_BLOCK],
                                                         uint32_t (*s_qcnt)[4]) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned long long bal[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bal[q] = __ballot((qmask >> q) & 1u);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) s_qcnt[wave][q] = (uint32_t)__popcll(bal[q]);
    }
    __syncthreads();  // counts visible; also orders the staged records before any reader
    uint32_t my_n = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t c0 = s_qcnt[0][q], c1 = s_qcnt[1][q], c2 = s_qcnt[2][q], c3 = s_qcnt[3][q];
        const uint32_t base = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u);
        if ((qmask >> q) & 1u) s_list[q][base + (uint32_t)__popcll(bal[q] & lt)] = (uint16_t)threadIdx.x;
        if ((uint32_t)q == wave) my_n = c0 + c1 + c2 + c3;
    }
    __syncthreads();
    return my_n;
}

// exp(x) for x <= 0, fully specified (bit-exact twin of pinned_exp in oracle/raster_oracle.c): a pixel's colour depends
// discontinuously on alpha >= 1/255 and T (1 - alpha) >= 1e-4, so the hardware v_exp_f32 (1 ulp, unspecified) cannot
// be part of a contract that must hold on all 1.7 M pixels of a 1066 x 1600 image.
__device__ __forceinline__ float pinned_exp(float x) {
#pragma clang fp contract(off)
    const float t = x * 1.44269504088896341f;
    const float n = __builtin_rintf(t);
    const float f = t - n;
    float p = 1.54035304e-4f;
    p = __builtin_fmaf(p, f, 1.33335581e-3f);
    p = __builtin_fmaf(p, f, 9.61812911e-3f);
    p = __builtin_fmaf(p, f, 5.55041087e-2f);
    p = __builtin_fmaf(p, f, 2.40226507e-1f);
    p = __builtin_fmaf(p, f, 6.93147181e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

// ---------------------------------------------------------------------------------------------------
// K6 with 4x4 granularity.  The 64 lanes of a wave are four DPP rows of 16 lanes; each row owns one 4x4 pixel
// sub-block of the tile (16 sub-blocks = 4 waves x 4 rows) and walks ITS OWN compacted list, so one wave iteration
// evaluates four different (entry, sub-block) pairs.  ds_read_b128 serves a wave as 4 groups of 16 lanes anyway, so
// per-row LDS addresses cost nothing extra.  Measured on the bench scene: 4.29 of 16 sub-blocks per entry survive the
// tight-footprint test (vs 1.59 of 4 quadrants), i.e. 1.07 D instead of 1.59 D wave iterations at 71 % lane use.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t subblock_mask(const float4& r0, const float4& r1, float tile_x0, float tile_y0) {
    const float o = r0.w;
    if (!(o >= (1.f / 255.f))) return 0u;
    const float det = r1.x * r1.z - r1.y * r1.y;
    if (!(det > 0.f)) return 0xFFFFu;
    const float tau2 = 2.f * __logf(255.f * o);
    const float inv = 1.f / det;
    const float bx = sqrtf(tau2 * r1.z * inv) * 1.0001f + 0.01f;
    const float by = sqrtf(tau2 * r1.x * inv) * 1.0001f + 0.01f;
    if (!(bx == bx) || !(by == by)) return 0xFFFFu;
    const float lx = r0.x - bx, hx = r0.x + bx, ly = r0.y - by, hy = r0.y + by;
    uint32_t cm = 0, rm = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        cm |= (lx <= tile_x0 + (float)(4 * c + 3) && hx >= tile_x0 + (float)(4 * c)) ? (1u << c) : 0u;
        rm |= (ly <= tile_y0 + (float)(4 * c + 3) && hy >= tile_y0 + (float)(4 * c)) ? (1u << c) : 0u;
    }
    // sub-block sb = sby * 4 + sbx  ->  bit sb
    return ((rm & 1u) ? cm : 0u) | ((rm & 2u) ? cm << 4 : 0u) | ((rm & 4u) ? cm << 8 : 0u) | ((rm & 8u) ? cm << 12 : 0u);
}

// Compacts the staged slots into 16 per-sub-block index lists (ascending slot order); s_cnt[w][sb] holds per-wave
// counts afterwards, s_tot[sb] the list lengths.  One barrier before (counts) and one after (lists).
__device__ __forceinline__ void build_subblock_lists(uint32_t mask16, uint16_t (*s_list)[EMD_BLOCK], uint32_t (*s_cnt)[16],
                                                     uint32_t* s_tot) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t before[16];
#pragma unroll
    for (int sb = 0; sb < 16; sb++) {
        const unsigned long long bal = __ballot((mask16 >> sb) & 1u);
        before[sb] = (uint32_t)__popcll(bal & lt);
        if (lane == 0) s_cnt[wave][sb] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < 16; sb++) {
        const uint32_t c0 = s_cnt[0][sb], c1 = s_cnt[1][sb], c2 = s_cnt[2][sb], c3 = s_cnt[3][sb];
        const uint32_t base = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u);
        if ((mask16 >> sb) & 1u) s_list[sb][base + before[sb]] = (uint16_t)threadIdx.x;
        if (threadIdx.x == (uint32_t)sb) s_tot[sb] = c0 + c1 + c2 + c3;
    }
    __syncthreads();
}

template <bool NORMAL>
__global__ void __launch_bounds__(EMD_BLOCK) k_render_forward(RenderDims d, const uint32_t* __restrict__ ranges,
                                                              const uint32_t* __restrict__ point_list,
                                                              const float4* __restrict__ rec, float* __restrict__ out_color,
                                                              float* __restrict__ out_depth, float* __restrict__ out_normal,
                                                              float* __restrict__ out_alpha, float* __restrict__ final_T,
                                                              uint32_t* __restrict__ n_contrib) {
#pragma clang fp contract(off)   // the forward image is a bit-exact contract: only the explicit fma calls below fuse
    __shared__ float4 s0[EMD_BLOCK], s1[EMD_BLOCK], s2[EMD_BLOCK];
    __shared__ float4 s3[NORMAL ? EMD_BLOCK : 1];
    __shared__ uint16_t s_list[16][EMD_BLOCK];
    __shared__ uint32_t s_cnt[4][16];
    __shared__ uint32_t s_tot[16];
    const uint32_t tile = xcd_tile(blockIdx.x);
    if (tile >= (uint32_t)(d.gx * d.gy)) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t sb = wave * 4 + (lane >> 4), l = lane & 15;          // sub-block of this DPP row, pixel inside it
    const float tile_x0 = (float)((tile % (uint32_t)d.gx) * EMD_TILE_X), tile_y0 = (float)((tile / (uint32_t)d.gx) * EMD_TILE_Y);
    const int px = (int)tile_x0 + (int)((sb & 3) * 4 + (l & 3)), py = (int)tile_y0 + (int)((sb >> 2) * 4 + (l >> 2));
    const bool inside = px < d.W && py < d.H;
    const float pfx = (float)px, pfy = (float)py;
    const uint32_t start = ranges[2 * tile], end = ranges[2 * tile + 1];
    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dz = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    uint32_t last = 0;
    // Register double buffer: the (index -> record) gather of chunk c+1 is issued before chunk c is composited, so
    // the two dependent global loads overlap the compositing instead of sitting exposed between two barriers.
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 p0 = zero4, p1 = zero4, p2 = zero4, p3 = zero4;
    if (start + threadIdx.x < end) {
        const float4* r = rec + (size_t)point_list[start + threadIdx.x] * EMD_REC_F4;
        p0 = r[0]; p1 = r[1]; p2 = r[2];
        if (NORMAL) p3 = r[3];
    }
    for (uint32_t base = start; base < end; base += EMD_BLOCK) {
        if (__syncthreads_count(done) == EMD_BLOCK) break;   // also: previous chunk fully consumed
        const uint32_t idx = base + threadIdx.x;
        uint32_t mask16 = 0;
        if (idx < end) {
            s0[threadIdx.x] = p0;
            s1[threadIdx.x] = p1;
            s2[threadIdx.x] = p2;
            if (NORMAL) s3[threadIdx.x] = p3;
            mask16 = subblock_mask(p0, p1, tile_x0, tile_y0);
        }
        if (idx + EMD_BLOCK < end) {
            const float4* r = rec + (size_t)point_list[idx + EMD_BLOCK] * EMD_REC_F4;
            p0 = r[0]; p1 = r[1]; p2 = r[2];
            if (NORMAL) p3 = r[3];
        }
        build_subblock_lists(mask16, s_list, s_cnt, s_tot);
        if (__ballot(!done) == 0ull) continue;  // whole wave finished: keep feeding the barriers only
        const uint32_t n = s_tot[sb];
        // trip count of the wave = longest of its four row lists
        const uint32_t nmax = max(max(readlane_u32(n, 0), readlane_u32(n, 16)), max(readlane_u32(n, 32), readlane_u32(n, 48)));
        if (nmax == 0) continue;
        const uint16_t* list = s_list[sb];
        // software pipeline: the index and record of entry i+1 are in flight while entry i is evaluated
        uint32_t j = n ? list[0] : 0u;
        float4 g0 = s0[j], g1 = s1[j];
        for (uint32_t i = 0; i < nmax; i++) {
            const bool active = i < n;
            const uint32_t jn = (i + 1 < n) ? list[i + 1] : j;
            const float4 g0n = s0[jn], g1n = s1[jn];
            const float dx = g0.x - pfx, dy = g0.y - pfy;
            const float power = gauss_power(g1.x, g1.y, g1.z, dx, dy);
            const float alpha = fminf(0.99f, g0.w * pinned_exp(power));
            // Branch-free: predication (v_cndmask) instead of exec-mask branches; arithmetic is lane-wise identical to
            // the sequential definition, and w = 0 leaves every accumulator bit-identical (fma(c, 0, acc) = acc).
            const bool hit = active && !done && power <= 0.f && alpha >= (1.f / 255.f);
            const float test_T = T * (1.f - alpha);
            const bool stop = hit && test_T < 0.0001f;
            const bool take = hit && !stop;
            const float w = take ? alpha * T : 0.f;
            const float4 g2 = s2[j];
            C0 = __builtin_fmaf(g2.x, w, C0); C1 = __builtin_fmaf(g2.y, w, C1); C2 = __builtin_fmaf(g2.z, w, C2);
            Dz = __builtin_fmaf(g0.z, w, Dz);
            if (NORMAL) {
                const float4 g3 = s3[j];
                N0 = __builtin_fmaf(g3.x, w, N0); N1 = __builtin_fmaf(g3.y, w, N1); N2 = __builtin_fmaf(g3.z, w, N2);
            }
            T = take ? test_T : T;
            last = take ? base - start + j + 1 : last;   // 1-based position in the tile list
            done = done || stop;
            j = jn; g0 = g0n; g1 = g1n;
        }
    }
    if (inside) {
        const size_t HW = (size_t)d.H * d.W, pix = (size_t)py * d.W + px;
        out_color[pix] = __builtin_fmaf(T, d.bg[0], C0);
        out_color[HW + pix] = __builtin_fmaf(T, d.bg[1], C1);
        out_color[2 * HW + pix] = __builtin_fmaf(T, d.bg[2], C2);
        out_depth[pix] = Dz;
        if (NORMAL) { out_normal[pix] = N0; out_normal[HW + pix] = N1; out_normal[2 * HW + pix] = N2; }
        out_alpha[pix] = 1.f - T;
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

// ---------------------------------------------------------------------------------------------------
// K6, wave-autonomous variant: one wave (64-thread workgroup) per 8x8 quadrant, four DPP rows = its four 4x4 sub-blocks.
//
// Same compositing loop as above, but the wave feeds itself: it scans the tile list 64 entries per step (records
// prefetched one step ahead), keeps the entries whose tight footprint touches the quadrant in a 128-slot LDS ring and
// appends the slot to the byte list of every sub-block it touches.  Once more than 64 entries are queued (or the list
// ends) the four rows drain their lists.  No workgroup barriers, no 256-entry chunk granularity, and the scan stops as
// soon as the 64 pixels of THIS quadrant are saturated rather than all 256 of the tile.
// The position of an entry in the tile list (needed for n_contrib) rides in the unused .w of the colour record.
// ---------------------------------------------------------------------------------------------------
#define FQ_RING 128

// bit r (r = sby * 2 + sbx) set when the tight alpha >= 1/255 box overlaps 4x4 sub-block r of the quadrant at (qx0, qy0)
__device__ __forceinline__ uint32_t quad_subblock_mask(const float4& r0, const float4& r1, float qx0, float qy0) {
    const float o = r0.w;
    if (!(o >= (1.f / 255.f))) return 0u;
    const float det = r1.x * r1.z - r1.y * r1.y;
    if (!(det > 0.f)) return 0xFu;
    const float tau2 = 2.f * __logf(255.f * o);
    const float inv = 1.f / det;
    const float bx = sqrtf(tau2 * r1.z * inv) * 1.0001f + 0.01f;
    const float by = sqrtf(tau2 * r1.x * inv) * 1.0001f + 0.01f;
    if (!(bx == bx) || !(by == by)) return 0xFu;
    const float lx = r0.x - bx, hx = r0.x + bx, ly = r0.y - by, hy = r0.y + by;
    const bool c0 = lx <= qx0 + 3.f && hx >= qx0, c1 = lx <= qx0 + 7.f && hx >= qx0 + 4.f;
    const bool w0 = ly <= qy0 + 3.f && hy >= qy0, w1 = ly <= qy0 + 7.f && hy >= qy0 + 4.f;
    return (c0 && w0 ? 1u : 0u) | (c1 && w0 ? 2u : 0u) | (c0 && w1 ? 4u : 0u) | (c1 && w1 ? 8u : 0u);
}

template <bool NORMAL>
__global__ void __launch_bounds__(EMD_WAVE) k_render_forward_q(RenderDims d, const uint32_t* __restrict__ ranges,
                                                               const uint32_t* __restrict__ point_list,
                                                               const float4* __restrict__ rec, float* __restrict__ out_color,
                                                               float* __restrict__ out_depth, float* __restrict__ out_normal,
                                                               float* __restrict__ out_alpha, float* __restrict__ final_T,
                                                               uint32_t* __restrict__ n_contrib) {
#pragma clang fp contract(off)   // the forward image is a bit-exact contract: only the explicit fma calls below fuse
    __shared__ float4 s0[FQ_RING], s1[FQ_RING], s2[FQ_RING];
    __shared__ float4 s3[NORMAL ? FQ_RING : 1];
    __shared__ uint8_t s_list[4][FQ_RING];
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b % 8, k = b / 8, tl = k >> 2, quad = k & 3u;
    const uint32_t tile = ((tl / XCD_CHUNK) * 8 + xcd) * XCD_CHUNK + (tl % XCD_CHUNK);
    if (tile >= (uint32_t)(d.gx * d.gy)) return;
    const uint32_t lane = threadIdx.x, row = lane >> 4, l = lane & 15;
    const int qxi = (int)((tile % (uint32_t)d.gx) * EMD_TILE_X + (quad & 1) * 8), qyi = (int)((tile / (uint32_t)d.gx) * EMD_TILE_Y + (quad >> 1) * 8);
    const float qx0 = (float)qxi, qy0 = (float)qyi;
    const int px = qxi + (int)((row & 1) * 4 + (l & 3)), py = qyi + (int)((row >> 1) * 4 + (l >> 2));
    const bool inside = px < d.W && py < d.H;
    const float pfx = (float)px, pfy = (float)py;
    const uint32_t start = ranges[2 * tile], n_tile = ranges[2 * tile + 1] - start;
    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dz = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    uint32_t last = 0;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 p0 = zero4, p1 = zero4, p2 = zero4, p3 = zero4;
    if (lane < n_tile) {
        const float4* r = rec + (size_t)point_list[start + lane] * EMD_REC_F4;
        p0 = r[0]; p1 = r[1]; p2 = r[2];
        if (NORMAL) p3 = r[3];
    }
    uint32_t scanned = 0;
    while (scanned < n_tile) {
        if (__ballot(!done) == 0ull) break;
        // ---- scan: queue entries until more than 64 wait or the list ends ----
        uint32_t head = 0, len0 = 0, len1 = 0, len2 = 0, len3 = 0;
        while (head <= EMD_WAVE && scanned < n_tile) {
            const uint32_t idx = scanned + lane;
            const float4 c0r = p0, c1r = p1, c3r = p3;
            float4 c2r = p2;
            if (idx + EMD_WAVE < n_tile) {
                const float4* r = rec + (size_t)point_list[start + idx + EMD_WAVE] * EMD_REC_F4;
                p0 = r[0]; p1 = r[1]; p2 = r[2];
                if (NORMAL) p3 = r[3];
            }
            const uint32_t m4 = idx < n_tile ? quad_subblock_mask(c0r, c1r, qx0, qy0) : 0u;
            const unsigned long long bal = __ballot(m4 != 0u);
            const uint32_t slot = head + (uint32_t)__popcll(bal & lt);
            if (m4) {
                c2r.w = __uint_as_float(idx + 1);           // 1-based position in the tile list
                s0[slot] = c0r; s1[slot] = c1r; s2[slot] = c2r;
                if (NORMAL) s3[slot] = c3r;
            }
            const unsigned long long b0 = __ballot(m4 & 1u), b1 = __ballot(m4 & 2u), b2 = __ballot(m4 & 4u), b3 = __ballot(m4 & 8u);
            if (m4 & 1u) s_list[0][len0 + (uint32_t)__popcll(b0 & lt)] = (uint8_t)slot;
            if (m4 & 2u) s_list[1][len1 + (uint32_t)__popcll(b1 & lt)] = (uint8_t)slot;
            if (m4 & 4u) s_list[2][len2 + (uint32_t)__popcll(b2 & lt)] = (uint8_t)slot;
            if (m4 & 8u) s_list[3][len3 + (uint32_t)__popcll(b3 & lt)] = (uint8_t)slot;
            len0 += (uint32_t)__popcll(b0); len1 += (uint32_t)__popcll(b1); len2 += (uint32_t)__popcll(b2); len3 += (uint32_t)__popcll(b3);
            head += (uint32_t)__popcll(bal);
            scanned += EMD_WAVE;
        }
        __syncthreads();
        const uint32_t nmax = max(max(len0, len1), max(len2, len3));
        if (nmax) {
            const uint32_t n = row == 0 ? len0 : row == 1 ? len1 : row == 2 ? len2 : len3;
            const uint8_t* list = s_list[row];
            // software pipeline: the index and record of entry i+1 are in flight while entry i is evaluated
            uint32_t j = n ? list[0] : 0u;
            float4 g0 = s0[j], g1 = s1[j];
#pragma unroll 2
            for (uint32_t i = 0; i < nmax; i++) {
                const bool active = i < n;
                const uint32_t jn = (i + 1 < n) ? list[i + 1] : j;
                const float4 g0n = s0[jn], g1n = s1[jn];
                const float dx = g0.x - pfx, dy = g0.y - pfy;
                const float power = gauss_power(g1.x, g1.y, g1.z, dx, dy);
                const float alpha = fminf(0.99f, g0.w * pinned_exp(power));
                const bool hit = active && !done && power <= 0.f && alpha >= (1.f / 255.f);
                const float test_T = T * (1.f - alpha);
                const bool stop = hit && test_T < 0.0001f;
                const bool take = hit && !stop;
                const float w = take ? alpha * T : 0.f;
                const float4 g2 = s2[j];
                C0 = __builtin_fmaf(g2.x, w, C0); C1 = __builtin_fmaf(g2.y, w, C1); C2 = __builtin_fmaf(g2.z, w, C2);
                Dz = __builtin_fmaf(g0.z, w, Dz);
                if (NORMAL) {
                    const float4 g3 = s3[j];
                    N0 = __builtin_fmaf(g3.x, w, N0); N1 = __builtin_fmaf(g3.y, w, N1); N2 = __builtin_fmaf(g3.z, w, N2);
                }
                T = take ? test_T : T;
                last = take ? __float_as_uint(g2.w) : last;
                done = done || stop;
                j = jn; g0 = g0n; g1 = g1n;
            }
        }
        __syncthreads();   // ring and lists fully consumed before the next scan overwrites them
    }
    if (inside) {
        const size_t HW = (size_t)d.H * d.W, pix = (size_t)py * d.W + px;
        out_color[pix] = __builtin_fmaf(T, d.bg[0], C0);
        out_color[HW + pix] = __builtin_fmaf(T, d.bg[1], C1);
        out_color[2 * HW + pix] = __builtin_fmaf(T, d.bg[2], C2);
        out_depth[pix] = Dz;
        if (NORMAL) { out_normal[pix] = N0; out_normal[HW + pix] = N1; out_normal[2 * HW + pix] = N2; }
        out_alpha[pix] = 1.f - T;
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

// ---------------------------------------------------------------------------------------------------
// K7, entry-parallel formulation for wave64.
//
// Upstream walks each pixel's list back to front and needs, per (pixel, Gaussian), ten partial derivatives summed
// over the pixels of the tile: a 64-lane reduction of ten values per list entry.  On CDNA4 that reduction (60
// v_add_f32_dpp per entry and wave) costs as much as all the arithmetic (measured: profiles/r01_ablation.txt).
// Here the roles are swapped: a lane owns one list ENTRY of the wave's quadrant list, the wave loops over the 64
// PIXELS of its quadrant, and the per-pixel recurrences become wave prefix scans in list order:
//     T_k  = prod_{i<k} (1 - alpha_i)                 exclusive scan-product   (6 v_mul_f32_dpp + 1 wave_shr)
//     S_k  = sum_{i<=k} g_i alpha_i T_i               inclusive scan-sum       (6 v_add_f32_dpp)
//     dL/dalpha_k = g_k T_k + (Q + S_k) / (1 - alpha_k),   g = colour . dL/dC + depth dL/dD (+ normal . dL/dN),
//     Q = T_final (dL/dalpha_img - bg . dL/dC) - S_total,  S_total from the forward outputs.
// (algebraically the upstream gradient: sum_{i>k} g_i alpha_i T_i = S_total - S_k; no T / (1 - alpha)
// reconstruction by division).  The ten derivative sums then accumulate in the lane's registers over the 64 pixels
// with no cross-lane traffic at all; per 64 (pixel, entry) pairs the wave spends 13 DPP ops instead of 60.
// Running T and S per pixel are carried across batches of 64 entries in lane p (v_readlane / lane select).
// ---------------------------------------------------------------------------------------------------
#define NV_BASE 10

template <bool NORMAL, bool ABS>
__global__ void __launch_bounds__(EMD_BLOCK) k_render_backward(RenderDims d, const uint32_t* __restrict__ ranges,
                                                               const uint32_t* __restrict__ point_list,
                                                               const float4* __restrict__ rec,
                                                               const float* __restrict__ final_T,
                                                               const uint32_t* __restrict__ n_contrib,
                                                               const float* __restrict__ out_color,
                                                               const float* __restrict__ out_depth,
                                                               const float* __restrict__ out_normal,
                                                               const float* __restrict__ dL_dcolor,
                                                               const float* __restrict__ dL_ddepth,
                                                               const float* __restrict__ dL_dalpha,
                                                               const float* __restrict__ dL_dnormal,
                                                               float* __restrict__ grad_rec) {
    constexpr int NV = NV_BASE + (ABS ? 2 : 0);
    __shared__ float4 s0[EMD_BLOCK], s1[EMD_BLOCK], s2[EMD_BLOCK];
    __shared__ float4 s3[NORMAL ? EMD_BLOCK : 1];
    __shared__ uint32_t s_id[EMD_BLOCK];
    __shared__ float s_acc[EMD_BLOCK][EMD_BWD_STRIDE + 1];  // odd row stride: conflict-free row and column walks
    __shared__ uint16_t s_list[4][EMD_BLOCK];
    __shared__ uint32_t s_qcnt[4][4];
    __shared__ uint32_t s_max[4];
    const uint32_t tile = xcd_tile(blockIdx.x);
    if (tile >= (uint32_t)(d.gx * d.gy)) return;
    int px, py;
    tile_pixel(d, tile, threadIdx.x, px, py);
    const bool inside = px < d.W && py < d.H;
    const float tile_x0 = (float)((tile % (uint32_t)d.gx) * EMD_TILE_X), tile_y0 = (float)((tile / (uint32_t)d.gx) * EMD_TILE_Y);
    const uint32_t start = ranges[2 * tile], end = ranges[2 * tile + 1];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float qx0 = tile_x0 + (float)((wave & 1) * 8), qy0 = tile_y0 + (float)((wave >> 1) * 8);
    const size_t HW = (size_t)d.H * d.W, pix = (size_t)py * d.W + px;
    // ---- per-pixel constants live in lane p of the wave that owns the pixel ----
    const uint32_t my_n = inside ? n_contrib[pix] : 0u;
    {
        uint32_t m = my_n;
        for (int off = 32; off; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
        if (lane == 0) s_max[wave] = m;
    }
    __syncthreads();
    const uint32_t tile_n = min(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])), end - start);
    if (tile_n == 0) return;
    float dC0 = 0.f, dC1 = 0.f, dC2 = 0.f, dD = 0.f, dN0 = 0.f, dN1 = 0.f, dN2 = 0.f, Q = 0.f;
    if (inside) {
        const float Tf = final_T[pix];
        float dA = 0.f;
        if (dL_dcolor) { dC0 = dL_dcolor[pix]; dC1 = dL_dcolor[HW + pix]; dC2 = dL_dcolor[2 * HW + pix]; }
        if (dL_ddepth) dD = dL_ddepth[pix];
        if (dL_dalpha) dA = dL_dalpha[pix];
        if (NORMAL && dL_dnormal) { dN0 = dL_dnormal[pix]; dN1 = dL_dnormal[HW + pix]; dN2 = dL_dnormal[2 * HW + pix]; }
        const float bgdot = d.bg[0] * dC0 + d.bg[1] * dC1 + d.bg[2] * dC2;
        float Stot = (out_color[pix] - Tf * d.bg[0]) * dC0 + (out_color[HW + pix] - Tf * d.bg[1]) * dC1 +
                     (out_color[2 * HW + pix] - Tf * d.bg[2]) * dC2 + out_depth[pix] * dD;
        if (NORMAL && dL_dnormal) Stot += out_normal[pix] * dN0 + out_normal[HW + pix] * dN1 + out_normal[2 * HW + pix] * dN2;
        Q = Tf * (dA - bgdot) - Stot;
    }
    float cT = 1.f, cS = 0.f;  // running transmittance / running S of pixel `lane`, carried across batches
    for (uint32_t done_cnt = 0; done_cnt < tile_n; done_cnt += EMD_BLOCK) {
        const uint32_t cnt = min((uint32_t)EMD_BLOCK, tile_n - done_cnt);
        __syncthreads();  // previous chunk fully consumed (s_acc flushed) before restaging
        uint32_t qmask = 0;
        if (threadIdx.x < cnt) {
            const uint32_t id = point_list[start + done_cnt + threadIdx.x];
            const float4* r = rec + (size_t)id * EMD_REC_F4;
            const float4 r0 = r[0], r1 = r[1];
            s_id[threadIdx.x] = id;
            s0[threadIdx.x] = r0;
            s1[threadIdx.x] = r1;
            s2[threadIdx.x] = r[2];
            if (NORMAL) s3[threadIdx.x] = r[3];
            qmask = quadrant_mask(r0, r1, tile_x0, tile_y0);
        }
#pragma unroll
        for (int v = 0; v < EMD_BWD_STRIDE; v++) s_acc[threadIdx.x][v] = 0.f;
        const uint32_t n = build_quadrant_lists(qmask, s_list, s_qcnt);
        const uint16_t* list = s_list[wave];
        for (uint32_t b = 0; b < n; b += EMD_WAVE) {
            const uint32_t i = b + lane;
            const bool valid = i < n;
            const uint32_t j = list[valid ? i : n - 1];
            const float4 g0 = s0[j], g1 = s1[j], g2 = s2[j];
            float4 g3 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NORMAL) g3 = s3[j];
            const uint32_t pos = done_cnt + j;                       // 0-based position in the tile list
            const uint32_t first_pos = readlane_u32(pos, 0);         // list is ascending: lane 0 is the front-most
            // per-entry sums over the quadrant's pixels.  With u = G dL/dG the conic / mean derivatives are moments of u:
            //   m0 = sum u, m1 = sum u d, m2 = sum u d d^T  =>  d mean = -Conic m1, d conic = -(1/2, 1, 1/2) m2, d opacity = m0 / o
            float m0 = 0.f, m1x = 0.f, m1y = 0.f, m2xx = 0.f, m2xy = 0.f, m2yy = 0.f, a_dz = 0.f, a_r = 0.f, a_g = 0.f,
                  a_b = 0.f, a_ax = 0.f, a_ay = 0.f;
            for (int p = 0; p < EMD_WAVE; p++) {
                const uint32_t n_p = readlane_u32(my_n, p);
                if (n_p <= first_pos) continue;                      // pixel p terminated before this batch
                const float pxs = qx0 + (float)(p & 7), pys = qy0 + (float)(p >> 3);
                const float dx = g0.x - pxs, dy = g0.y - pys;
                const float power = gauss_power(g1.x, g1.y, g1.z, dx, dy);
                const float G = pinned_exp(power);
                const float alpha = fminf(0.99f, g0.w * G);
                const bool hit = valid && pos < n_p && power <= 0.f && alpha >= (1.f / 255.f);
                if (__ballot(hit) == 0ull) continue;
                const float a = hit ? alpha : 0.f;
                const float om = 1.f - a;
                const float t_incl = wave_scan_mul_f32_asm(om);
                const float cTp = readlane_f32(cT, p), cSp = readlane_f32(cS, p);
                const float Tk = cTp * wave_shift_up1_f32(t_incl, 1.f);
                const float w = a * Tk;
                const float c0 = readlane_f32(dC0, p), c1 = readlane_f32(dC1, p), c2 = readlane_f32(dC2, p),
                            cd = readlane_f32(dD, p);
                float g = g2.x * c0 + g2.y * c1 + g2.z * c2 + g0.z * cd;
                if (NORMAL) g += g3.x * readlane_f32(dN0, p) + g3.y * readlane_f32(dN1, p) + g3.z * readlane_f32(dN2, p);
                const float s_incl = wave_scan_add_f32_asm(g * w);
                const float Sk = cSp + s_incl;
                const float inv = __builtin_amdgcn_rcpf(om);
                float dL_da = g * Tk + inv * (readlane_f32(Q, p) + Sk);
                dL_da = hit ? dL_da : 0.f;
                const float u = G * (g0.w * dL_da);                  // G dL/dG
                const float ux = u * dx, uy = u * dy;
                m0 += u; m1x += ux; m1y += uy;
                m2xx += ux * dx; m2xy += ux * dy; m2yy += uy * dy;
                if (ABS) {
                    a_ax += fabsf(ux * g1.x + uy * g1.y);
                    a_ay += fabsf(uy * g1.z + ux * g1.y);
                }
                a_dz += w * cd;
                a_r += w * c0; a_g += w * c1; a_b += w * c2;
                // carry the running T and S of pixel p to the next batch
                const float nT = cTp * readlane_f32(t_incl, 63), nS = cSp + readlane_f32(s_incl, 63);
                cT = (lane == (uint32_t)p) ? nT : cT;
                cS = (lane == (uint32_t)p) ? nS : cS;
            }
            if (valid) {
                float* row = s_acc[j];   // the 4 waves may meet on one entry: LDS float atomics
                atomicAdd(row + 0, -(g1.x * m1x + g1.y * m1y)); atomicAdd(row + 1, -(g1.z * m1y + g1.y * m1x));
                atomicAdd(row + 2, a_dz); atomicAdd(row + 3, m0 * __builtin_amdgcn_rcpf(g0.w));
                atomicAdd(row + 4, -0.5f * m2xx); atomicAdd(row + 5, -m2xy); atomicAdd(row + 6, -0.5f * m2yy);
                atomicAdd(row + 7, a_r); atomicAdd(row + 8, a_g); atomicAdd(row + 9, a_b);
                if (ABS) { atomicAdd(row + 10, a_ax); atomicAdd(row + 11, a_ay); }
            }
        }
        __syncthreads();
        // flush: consecutive lanes add consecutive floats of one accumulator row (48-byte contiguous segments per
        // Gaussian) -- the shape global float atomics like -- instead of one lane per scattered row.
        for (uint32_t idx = threadIdx.x; idx < cnt * EMD_BWD_STRIDE; idx += EMD_BLOCK) {
            const uint32_t e = idx / EMD_BWD_STRIDE, v = idx % EMD_BWD_STRIDE;
            const float val = s_acc[e][v];
            if (val != 0.f) atomicAdd(grad_rec + (size_t)s_id[e] * EMD_BWD_STRIDE + v, val);
        }
    }
    (void)NV;
}

// ---------------------------------------------------------------------------------------------------
// K7, wave-autonomous variant: one wave (64-thread workgroup) per 8x8 quadrant.
//
// The workgroup-per-tile kernel above stages 256 list entries at a time and cuts each quadrant's list at the chunk
// boundary, so the 64-entry batches of the entry-parallel loop are only ~78 % full, and four waves wait on three
// barriers per chunk.  Here a wave scans the tile list by itself, 64 entries per step (records prefetched one step
// ahead), keeps the entries whose tight footprint overlaps ITS quadrant in a small LDS queue, and runs the pixel loop
// whenever 64 entries are queued: every batch but the last is full, there is no workgroup barrier, and the scan stops
// at the deepest contributor of this quadrant rather than of the whole tile.  Gradients leave per batch as
// row-shaped global float atomics (48 contiguous bytes per Gaussian) through a per-wave LDS staging tile.
// ---------------------------------------------------------------------------------------------------
#define BQ_QUEUE 128

__device__ __forceinline__ uint32_t xcd_quadrant_block(uint32_t b, uint32_t* quad) {
    // the four quadrants of a tile and runs of XCD_CHUNK neighbouring tiles stay on one XCD (blocks b, b+8, ... share it)
    const uint32_t xcd = b % 8, k = b / 8, tl = k >> 2;
    *quad = k & 3u;
    return ((tl / XCD_CHUNK) * 8 + xcd) * XCD_CHUNK + (tl % XCD_CHUNK);
}

template <bool NORMAL, bool ABS>
__global__ void __launch_bounds__(EMD_WAVE) k_render_backward_q(RenderDims d, const uint32_t* __restrict__ ranges,
                                                                const uint32_t* __restrict__ point_list,
                                                                const float4* __restrict__ rec,
                                                                const float* __restrict__ final_T,
                                                                const uint32_t* __restrict__ n_contrib,
                                                                const float* __restrict__ out_color,
                                                                const float* __restrict__ out_depth,
                                                                const float* __restrict__ out_normal,
                                                                const float* __restrict__ dL_dcolor,
                                                                const float* __restrict__ dL_ddepth,
                                                                const float* __restrict__ dL_dalpha,
                                                                const float* __restrict__ dL_dnormal,
                                                                float* __restrict__ grad_rec) {
    // queue slot s lives in half s>>6: the gradient staging tile (64 rows x 12 floats = 3 KB) overlays the records of
    // the lower half, which are dead (held in registers) by the time a batch's gradients are staged.
    constexpr int NR = NORMAL ? 4 : 3;
    __shared__ float4 q_rec[2][NR][EMD_WAVE];
    __shared__ uint32_t q_id[BQ_QUEUE], q_pos[BQ_QUEUE];
    float* const s_stage = reinterpret_cast<float*>(&q_rec[0][0][0]);
    // per-pixel constants and running state, read back as wave-uniform (broadcast) LDS loads in the pixel loop: keeps
    // ~15 v_readlane / v_mov / v_cndmask per pixel-iteration off the VALU, which is what bounds this kernel.
    //   [p][0] = (dL/dC rgb, dL/dD)   [p][1] = (Q, running T, running S, -)   [p][2] = dL/dN
    __shared__ float4 s_pix[EMD_WAVE][NORMAL ? 3 : 2];
#define QREC(r, s) q_rec[(s) >> 6][r][(s) & 63]
    uint32_t quad;
    const uint32_t tile = xcd_quadrant_block(blockIdx.x, &quad);
    if (tile >= (uint32_t)(d.gx * d.gy)) return;
    const uint32_t lane = threadIdx.x;
    const float tile_x0 = (float)((tile % (uint32_t)d.gx) * EMD_TILE_X), tile_y0 = (float)((tile / (uint32_t)d.gx) * EMD_TILE_Y);
    const float qx0 = tile_x0 + (float)((quad & 1) * 8), qy0 = tile_y0 + (float)((quad >> 1) * 8);
    const int px = (int)qx0 + (int)(lane & 7), py = (int)qy0 + (int)(lane >> 3);
    const bool inside = px < d.W && py < d.H;
    const uint32_t start = ranges[2 * tile], end = ranges[2 * tile + 1];
    const size_t HW = (size_t)d.H * d.W, pix = (size_t)py * d.W + px;
    const uint32_t my_n = inside ? n_contrib[pix] : 0u;
    uint32_t wave_n = my_n;
    for (int off = 32; off; off >>= 1) wave_n = max(wave_n, (uint32_t)__shfl_xor((int)wave_n, off));
    wave_n = min(wave_n, end - start);       // deepest contributor of THIS quadrant
    if (wave_n == 0) return;
    float dC0 = 0.f, dC1 = 0.f, dC2 = 0.f, dD = 0.f, dN0 = 0.f, dN1 = 0.f, dN2 = 0.f, Q = 0.f;
    if (inside) {
        const float Tf = final_T[pix];
        float dA = 0.f;
        if (dL_dcolor) { dC0 = dL_dcolor[pix]; dC1 = dL_dcolor[HW + pix]; dC2 = dL_dcolor[2 * HW + pix]; }
        if (dL_ddepth) dD = dL_ddepth[pix];
        if (dL_dalpha) dA = dL_dalpha[pix];
        if (NORMAL && dL_dnormal) { dN0 = dL_dnormal[pix]; dN1 = dL_dnormal[HW + pix]; dN2 = dL_dnormal[2 * HW + pix]; }
        const float bgdot = d.bg[0] * dC0 + d.bg[1] * dC1 + d.bg[2] * dC2;
        float Stot = (out_color[pix] - Tf * d.bg[0]) * dC0 + (out_color[HW + pix] - Tf * d.bg[1]) * dC1 +
                     (out_color[2 * HW + pix] - Tf * d.bg[2]) * dC2 + out_depth[pix] * dD;
        if (NORMAL && dL_dnormal) Stot += out_normal[pix] * dN0 + out_normal[HW + pix] * dN1 + out_normal[2 * HW + pix] * dN2;
        Q = Tf * (dA - bgdot) - Stot;
    }
    s_pix[lane][0] = make_float4(dC0, dC1, dC2, dD);
    s_pix[lane][1] = make_float4(Q, 1.f, 0.f, 0.f);   // running transmittance / running S, carried across batches
    if (NORMAL) s_pix[lane][2] = make_float4(dN0, dN1, dN2, 0.f);
    __syncthreads();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // One batch: lane = queue slot; identical arithmetic to the workgroup-per-tile kernel's pixel loop.
    auto process_batch = [&](uint32_t nb) {
        const bool valid = lane < nb;
        const uint32_t sl = valid ? lane : 0u;
        const float4 g0 = QREC(0, sl), g1 = QREC(1, sl), g2 = QREC(2, sl);
        float4 g3 = zero4;
        if (NORMAL) g3 = QREC(3, sl);
        const uint32_t pos = q_pos[sl];
        const uint32_t first_pos = readlane_u32(pos, 0);
        float m0 = 0.f, m1x = 0.f, m1y = 0.f, m2xx = 0.f, m2xy = 0.f, m2yy = 0.f, a_dz = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f,
              a_ax = 0.f, a_ay = 0.f;
        for (int p = 0; p < EMD_WAVE; p++) {
            const uint32_t n_p = readlane_u32(my_n, p);
            if (n_p <= first_pos) continue;                      // pixel p terminated before this batch
            const float4 pa = s_pix[p][0], pb = s_pix[p][1];
            const float pxs = qx0 + (float)(p & 7), pys = qy0 + (float)(p >> 3);
            const float dx = g0.x - pxs, dy = g0.y - pys;
            const float power = gauss_power(g1.x, g1.y, g1.z, dx, dy);
            const float G = pinned_exp(power);
            const float alpha = fminf(0.99f, g0.w * G);
            const bool hit = valid && pos < n_p && power <= 0.f && alpha >= (1.f / 255.f);
            if (__ballot(hit) == 0ull) continue;
            const float a = hit ? alpha : 0.f;
            const float om = 1.f - a;
            const float t_incl = wave_scan_mul_f32_asm(om);
            const float Tk = pb.y * wave_shift_up1_f32(t_incl, 1.f);
            const float w = a * Tk;
            float g = g2.x * pa.x + g2.y * pa.y + g2.z * pa.z + g0.z * pa.w;
            if (NORMAL) { const float4 pn = s_pix[p][2]; g += g3.x * pn.x + g3.y * pn.y + g3.z * pn.z; }
            const float s_incl = wave_scan_add_f32_asm(g * w);
            const float Sk = pb.z + s_incl;
            const float inv = __builtin_amdgcn_rcpf(om);
            float dL_da = g * Tk + inv * (pb.x + Sk);
            dL_da = hit ? dL_da : 0.f;
            const float u = G * (g0.w * dL_da);                  // G dL/dG
            const float ux = u * dx, uy = u * dy;
            m0 += u; m1x += ux; m1y += uy;
            m2xx += ux * dx; m2xy += ux * dy; m2yy += uy * dy;
            if (ABS) { a_ax += fabsf(ux * g1.x + uy * g1.y); a_ay += fabsf(uy * g1.z + ux * g1.y); }
            a_dz += w * pa.w;
            a_r += w * pa.x; a_g += w * pa.y; a_b += w * pa.z;
            // lane 63 holds the batch totals: carry the running T and S of pixel p to the next batch
            if (lane == 63) *reinterpret_cast<float2*>(&s_pix[p][1].y) = make_float2(pb.y * t_incl, Sk);
        }
        // rows through LDS so that consecutive lanes add consecutive floats of one 48-byte accumulator row
        __syncthreads();   // every lane holds its record in registers: the lower half may be overwritten
        float4* row = reinterpret_cast<float4*>(s_stage + lane * EMD_BWD_STRIDE);
        row[0] = make_float4(-(g1.x * m1x + g1.y * m1y), -(g1.z * m1y + g1.y * m1x), a_dz, m0 * __builtin_amdgcn_rcpf(g0.w));
        row[1] = make_float4(-0.5f * m2xx, -m2xy, -0.5f * m2yy, a_r);
        row[2] = make_float4(a_g, a_b, ABS ? a_ax : 0.f, ABS ? a_ay : 0.f);
        __syncthreads();
        for (uint32_t idx = lane; idx < nb * EMD_BWD_STRIDE; idx += EMD_WAVE) {
            const uint32_t e = idx / EMD_BWD_STRIDE, v = idx % EMD_BWD_STRIDE;
            const float val = s_stage[idx];
            if (val != 0.f) atomicAdd(grad_rec + (size_t)q_id[e] * EMD_BWD_STRIDE + v, val);
        }
        __syncthreads();
    };

    // prefetch of the first step
    uint32_t head = 0;
    float4 p0 = zero4, p1 = zero4, p2 = zero4, p3 = zero4;
    uint32_t pid = 0;
    if (lane < wave_n) {
        pid = point_list[start + lane];
        const float4* r = rec + (size_t)pid * EMD_REC_F4;
        p0 = r[0]; p1 = r[1]; p2 = r[2];
        if (NORMAL) p3 = r[3];
    }
    for (uint32_t base = 0; base < wave_n; base += EMD_WAVE) {
        const uint32_t idx = base + lane;
        const float4 c0r = p0, c1r = p1, c2r = p2, c3r = p3;
        const uint32_t cid = pid;
        if (idx + EMD_WAVE < wave_n) {           // next step's records in flight while this one is queued / processed
            pid = point_list[start + idx + EMD_WAVE];
            const float4* r = rec + (size_t)pid * EMD_REC_F4;
            p0 = r[0]; p1 = r[1]; p2 = r[2];
            if (NORMAL) p3 = r[3];
        }
        const bool keep = idx < wave_n && ((quadrant_mask(c0r, c1r, tile_x0, tile_y0) >> quad) & 1u);
        const unsigned long long bal = __ballot(keep);
        if (keep) {
            const uint32_t slot = head + (uint32_t)__popcll(bal & lt);
            QREC(0, slot) = c0r; QREC(1, slot) = c1r; QREC(2, slot) = c2r;
            if (NORMAL) QREC(3, slot) = c3r;
            q_id[slot] = cid; q_pos[slot] = idx;
        }
        head += (uint32_t)__popcll(bal);
        __syncthreads();
        if (head >= EMD_WAVE) {
            process_batch(EMD_WAVE);
            const uint32_t rest = head - EMD_WAVE;
            float4 t0 = zero4, t1 = zero4, t2 = zero4, t3 = zero4;
            uint32_t ti = 0, tp = 0;
            if (lane < rest) {
                t0 = q_rec[1][0][lane]; t1 = q_rec[1][1][lane]; t2 = q_rec[1][2][lane];
                if (NORMAL) t3 = q_rec[1][3][lane];
                ti = q_id[EMD_WAVE + lane]; tp = q_pos[EMD_WAVE + lane];
            }
            __syncthreads();
            if (lane < rest) {
                q_rec[0][0][lane] = t0; q_rec[0][1][lane] = t1; q_rec[0][2][lane] = t2;
                if (NORMAL) q_rec[0][3][lane] = t3;
                q_id[lane] = ti; q_pos[lane] = tp;
            }
            head = rest;
            __syncthreads();
        }
    }
    if (head > 0) process_batch(head);
}

RenderDims make_dims(const EmdSettings& s) {
    RenderDims d;
    d.W = s.image_width; d.H = s.image_height;
    d.gx = (d.W + EMD_TILE_X - 1) / EMD_TILE_X; d.gy = (d.H + EMD_TILE_Y - 1) / EMD_TILE_Y;
    d.bg[0] = s.bg[0]; d.bg[1] = s.bg[1]; d.bg[2] = s.bg[2];
    return d;
}

}  // namespace

int emd_launch_render_forward(const EmdSettings& s, int flags, const GeomWs& g, const BinWs& b, const ImgWs& im,
                              float* out_color, float* out_depth, float* out_normal, float* out_alpha,
                              hipStream_t st) {
    const RenderDims d = make_dims(s);
    const int T = d.gx * d.gy;
    if (T <= 0) return EMD_OK;
    const uint32_t* pl = b.vals[b.sorted_buf];
    if (flags & EMD_FLAG_NORMAL)
        hipLaunchKernelGGL(k_render_forward_q<true>, dim3(4 * padded_tile_grid(T)), dim3(EMD_WAVE), 0, st, d, b.ranges, pl, g.rec, out_color,
                           out_depth, out_normal, out_alpha, im.final_T, im.n_contrib);
    else
        hipLaunchKernelGGL(k_render_forward_q<false>, dim3(4 * padded_tile_grid(T)), dim3(EMD_WAVE), 0, st, d, b.ranges, pl, g.rec, out_color,
                           out_depth, out_normal, out_alpha, im.final_T, im.n_contrib);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_render_backward(const EmdSettings& s, int flags, const GeomWs& g, const BinWs& b, const ImgWs& im,
                               const float* out_color, const float* out_depth, const float* out_normal,
                               const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                               const float* dL_dnormal, float* grad_rec, hipStream_t st) {
    const RenderDims d = make_dims(s);
    const int T = d.gx * d.gy;
    if (T <= 0) return EMD_OK;
    const uint32_t* pl = b.vals[b.sorted_buf];
    const bool nrm = (flags & EMD_FLAG_NORMAL) && dL_dnormal && out_normal, ab = flags & EMD_FLAG_ABSGRAD;
#define LAUNCH_BWD(N_, A_)                                                                                          \
    hipLaunchKernelGGL((k_render_backward_q<N_, A_>), dim3(4 * padded_tile_grid(T)), dim3(EMD_WAVE), 0, st, d, b.ranges, pl, g.rec,   \
                       im.final_T, im.n_contrib, out_color, out_depth, out_normal, dL_dcolor, dL_ddepth, dL_dalpha,    \
                       dL_dnormal, grad_rec)
    if (nrm && ab) LAUNCH_BWD(true, true);
    else if (nrm) LAUNCH_BWD(true, false);
    else if (ab) LAUNCH_BWD(false, true);
    else LAUNCH_BWD(false, false);
#undef LAUNCH_BWD
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
