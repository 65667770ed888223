// render.hip -- K6 per-tile front-to-back alpha compositing and K7 its backward.  gfx950.
// [UPSTREAM K6/K7; call site S3Gaussian/gaussian_renderer/__init__.py:145-155,
//  outputs consumed at :158-168,299-301 and S3Gaussian/train.py:226-368]
//
// Work decomposition (wave64), both kernels wave-autonomous -- no workgroup barrier anywhere:
//   - one 64-thread workgroup (= one wave) per 8x8-pixel quadrant of a 16x16 tile; blockIdx -> (position in the longest-first
//     tile order, quadrant) keeps the four quadrant waves of a tile on one XCD (ordered_quadrant_block below);
//   - the wave scans the tile's depth-sorted list itself, 64 entries per step (records prefetched ahead), applies the
//     exact-conservative footprint culling below and keeps the survivors in an LDS queue;
//   - K6 (k_render_forward_q): the four 16-lane DPP rows of the wave own the quadrant's four 4x4 sub-blocks; each row drains
//     the byte list of its sub-block with the pinned exp / FMA compositing arithmetic -- lane = pixel,
//     front to back, early exit per quadrant; up to EMD_MAX_EXTRA further colour sets ride along in the same walk;
//   - K7 (k_render_backward_q): lane = list ENTRY, the wave loops over the quadrant's 64 pixels (two per iteration in packed
//     fp32).  A batch of 64 queued entries is laid out BACK TO FRONT (lane 0 = deepest): the transmittance in front of entry k
//     is T_final times the inclusive prefix product of 1 / (1 - alpha) over the lanes, the colour behind it the exclusive prefix
//     sum of g alpha T (two 6-step DPP scans per pixel), so every quantity is built up from the small end and carries a RELATIVE
//     rounding error (the front-to-back form subtracted running sums from totals: absolute error, visible in the deepest
//     entries' gradients).  Per-entry derivative sums accumulate in registers as moments of u = G dL/dG; after its batches the
//     lane adds ONE 48-byte row (+16 B per extra colour set) to its Gaussian with global float atomics.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "device_utils.h"
#include "footprint.h"

namespace {

struct RenderDims {
    int W, H, gx, gy;
    float bg[3];
    const float* bg_dev;     // device copy of the background colour (EmdFwdArgs.settings_dev) or null
    // extra colour sets composited by the same list walk (EmdFwdArgs.colors_extra): [N,3] inputs, [3,H,W] outputs, no background
    const float* xcol[EMD_MAX_EXTRA];
    float* xout[EMD_MAX_EXTRA];
    const float* xgrad[EMD_MAX_EXTRA];      // dL/d(extra image) [3,H,W] or null (backward)
};

// blockIdx -> (tile, quadrant).  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one).  The tiles are
// taken from `order` (descending list length, binning.hip) at position tl * 8 + xcd, so every XCD works through the order
// front to back and the four quadrant waves of a tile share an XCD: they read the same Gaussian records, which stay in that
// XCD's L2.  Speed only, never correctness; the grid is 4 x the tile count padded to a multiple of 32 and surplus workgroups exit.
#define XCD_CHUNK 4
__device__ __forceinline__ uint32_t ordered_quadrant_block(uint32_t b, uint32_t T, const uint32_t* __restrict__ order, uint32_t* quad) {
    const uint32_t xcd = b % 8, k = b / 8, tl = k >> 2;
    *quad = k & 3u;
    const uint32_t pos = tl * 8 + xcd;
    return pos < T ? order[pos] : 0xFFFFFFFFu;
}
static inline unsigned padded_tile_grid(int T) { return (unsigned)((T + 8 * XCD_CHUNK - 1) / (8 * XCD_CHUNK) * (8 * XCD_CHUNK)); }

// Gaussian exponent, evaluated identically (explicit FMAs, no further contraction) in K6 and K7 so that both
// kernels take the same skip decisions for the same (pixel, Gaussian) pair.
__device__ __forceinline__ float gauss_power(float A, float B, float C, float dx, float dy) {
#pragma clang fp contract(off)
    const float q = __builtin_fmaf(A * dx, dx, (C * dy) * dy);
    return __builtin_fmaf(-0.5f, q, -((B * dx) * dy));
}

// Two-wide forms of gauss_power / pinned_exp (same operations per component, so bit-identical results): CDNA3/4 issue
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on register pairs at the rate of the scalar forms, which halves the VALU
// slots of everything that is not a DPP scan, a compare or a transcendental.  K7 evaluates two pixels per iteration.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f gauss_power2(float A, float B, float C, v2f dx, float dy) {
#pragma clang fp contract(off)
    const v2f q = __builtin_elementwise_fma(splat2(A) * dx, dx, splat2((C * dy) * dy));
    return __builtin_elementwise_fma(splat2(-0.5f), q, -((splat2(B) * dx) * splat2(dy)));
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

__device__ __forceinline__ v2f pinned_exp2(v2f x) {
#pragma clang fp contract(off)
    const v2f t = x * splat2(1.44269504088896341f);
    const v2f n = __builtin_elementwise_rint(t);
    const v2f f = t - n;
    v2f p = splat2(1.54035304e-4f);
    p = __builtin_elementwise_fma(p, f, splat2(1.33335581e-3f));
    p = __builtin_elementwise_fma(p, f, splat2(9.61812911e-3f));
    p = __builtin_elementwise_fma(p, f, splat2(5.55041087e-2f));
    p = __builtin_elementwise_fma(p, f, splat2(2.40226507e-1f));
    p = __builtin_elementwise_fma(p, f, splat2(6.93147181e-1f));
    p = __builtin_elementwise_fma(p, f, splat2(1.0f));
    return (v2f){__builtin_ldexpf(p.x, (int)n.x), __builtin_ldexpf(p.y, (int)n.y)};
}

// ---------------------------------------------------------------------------------------------------
// K6: one wave (64-thread workgroup) per 8x8 quadrant, four DPP rows = its four 4x4 sub-blocks.
//
// The wave feeds itself: it scans the tile list 64 entries per step (records
// prefetched one step ahead), keeps the entries whose tight footprint touches the quadrant in a 128-slot LDS ring and
// appends the slot to the byte list of every sub-block it touches.  Once more than 64 entries are queued (or the list
// ends) the four rows drain their lists.  No workgroup barriers, no 256-entry chunk granularity, and the scan stops as
// soon as the 64 pixels of THIS quadrant are saturated rather than all 256 of the tile.
// The position of an entry in the tile list (needed for n_contrib) rides in the unused .w of the colour record.
// ---------------------------------------------------------------------------------------------------
#define FQ_RING 128

// STATS (diagnostic instantiation, EmdFwdArgs.loop_stats): trip counts of the kernel's loops, summed over the waves -- with the static instruction
// counts of each loop's body (profiles/make_isa_mix.py) they split the kernel's instructions into scan / cull / queue and compositing.
template <bool NORMAL, int NX, bool STATS = false>
__global__ void __launch_bounds__(EMD_WAVE) k_render_forward_q(RenderDims d, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ ranges,
                                                               const uint32_t* __restrict__ point_list,
                                                               const float4* __restrict__ rec, float* __restrict__ out_color,
                                                               float* __restrict__ out_depth, float* __restrict__ out_normal,
                                                               float* __restrict__ out_alpha, float* __restrict__ final_T,
                                                               uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ surv,
                                                               uint32_t* __restrict__ quad_need, unsigned long long* __restrict__ loop_stats) {
#pragma clang fp contract(off)   // the forward image is a bit-exact contract: only the explicit fma calls below fuse
    unsigned long long st_scan = 0ull, st_cull = 0ull, st_drain = 0ull, st_useful = 0ull, st_round = 0ull;
    __shared__ float4 s0[FQ_RING], s1[FQ_RING], s2[FQ_RING];
    __shared__ float4 s3[NORMAL ? FQ_RING : 1];
    __shared__ float4 sx[NX ? NX : 1][NX ? FQ_RING : 1];       // extra colour sets: (r, g, b, -) per ring slot
    __shared__ uint8_t s_list[4][FQ_RING];
    uint32_t quad;
    const uint32_t tile = ordered_quadrant_block(blockIdx.x, (uint32_t)(d.gx * d.gy), tile_order, &quad);
    if (tile >= (uint32_t)(d.gx * d.gy)) return;
    const float bg0 = d.bg_dev ? d.bg_dev[0] : d.bg[0], bg1 = d.bg_dev ? d.bg_dev[1] : d.bg[1], bg2 = d.bg_dev ? d.bg_dev[2] : d.bg[2];
    const uint32_t lane = threadIdx.x, row = lane >> 4, l = lane & 15;
    const int qxi = (int)((tile % (uint32_t)d.gx) * EMD_TILE_X + (quad & 1) * 8), qyi = (int)((tile / (uint32_t)d.gx) * EMD_TILE_Y + (quad >> 1) * 8);
    const float qx0 = (float)qxi, qy0 = (float)qyi;
    const int px = qxi + (int)((row & 1) * 4 + (l & 3)), py = qyi + (int)((row >> 1) * 4 + (l >> 2));
    const bool inside = px < d.W && py < d.H;
    const float pfx = (float)px, pfy = (float)py;
    const uint32_t start = ranges[2 * tile], n_tile = ranges[2 * tile + 1] - start;
    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dz = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float X[NX ? NX : 1][3];
#pragma unroll
    for (int k = 0; k < (NX ? NX : 1); k++) X[k][0] = X[k][1] = X[k][2] = 0.f;
    uint32_t last = 0;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 p0 = zero4, p1 = zero4, p2 = zero4, p3 = zero4;
    float4 px_[NX ? NX : 1];
#pragma unroll
    for (int k = 0; k < (NX ? NX : 1); k++) px_[k] = zero4;
    // Which entries of the tile list can reach THIS quadrant was decided when the list was built: the duplicate kernel (binning.hip) derives the
    // four quadrant bits of a (tile, Gaussian) pair from the alpha >= 1/255 BOUNDING BOX the projection kernel left as four skip bits per
    // Gaussian (the exact ellipse-rectangle test of footprint.h per pair was measured there and dropped: its inputs are a 32-byte gather per
    // Gaussian) and leaves them in the top bits of the list word -- a looser mask than the ellipse test; the exact test runs below, on the
    // queued entries against the 4 x 4 sub-blocks.  The scan here reads 4 bytes per entry and fetches the 64-byte record of the entries whose
    // bit is set only (one step ahead; the list words two steps ahead).
    const uint32_t qbit = 1u << (EMD_ID_BITS + quad);
    uint32_t w_nxt = 0, pgid = 0;
    bool pkeep = false;
    {
        const uint32_t w = lane < n_tile ? point_list[start + lane] : 0u;
        if (lane + EMD_WAVE < n_tile) w_nxt = point_list[start + lane + EMD_WAVE];
        pkeep = (w & qbit) != 0u;
        pgid = w & EMD_ID_MASK;
        if (pkeep) {
            const float4* r = rec + (size_t)pgid * EMD_REC_F4;
            p0 = r[0]; p1 = r[1]; p2 = r[2];
            if (NORMAL) p3 = r[3];
#pragma unroll
            for (int k = 0; k < NX; k++) { const float* c = d.xcol[k] + 3 * (size_t)pgid; px_[k] = make_float4(c[0], c[1], c[2], 0.f); }
        }
    }
    // The survivors are numbered in list order and their Gaussian ids written out: the backward walks exactly this list (no scan of
    // the tile list there), and a pixel's n_contrib is the NUMBER of the last survivor that contributed to it (1-based), not its
    // position in the tile list.
    uint32_t* const sv = surv + 4 * (size_t)start + (size_t)quad * n_tile;
    uint32_t total = 0;
    uint32_t scanned = 0;
    while (scanned < n_tile) {
        if (__ballot(!done) == 0ull) break;
        // ---- scan: queue the entries whose footprint reaches this quadrant until more than 64 wait or the list ends ----
        uint32_t head = 0, len0 = 0, len1 = 0, len2 = 0, len3 = 0;
        if (STATS) st_round++;
        while (head <= EMD_WAVE && scanned < n_tile) {
            if (STATS) st_scan++;
            const uint32_t idx = scanned + lane;
            const float4 c0r = p0, c1r = p1, c3r = p3;
            float4 c2r = p2;
            float4 cxr[NX ? NX : 1];
#pragma unroll
            for (int k = 0; k < NX; k++) cxr[k] = px_[k];
            const uint32_t cgid = pgid;
            const bool keep = pkeep;
            {
                const uint32_t w = w_nxt;                 // (0 beyond the end of the list)
                w_nxt = (idx + 2 * EMD_WAVE < n_tile) ? point_list[start + idx + 2 * EMD_WAVE] : 0u;
                pkeep = (w & qbit) != 0u;
                pgid = w & EMD_ID_MASK;
                if (pkeep) {
                    const float4* r = rec + (size_t)pgid * EMD_REC_F4;
                    p0 = r[0]; p1 = r[1]; p2 = r[2];
                    if (NORMAL) p3 = r[3];
#pragma unroll
                    for (int k = 0; k < NX; k++) { const float* c = d.xcol[k] + 3 * (size_t)pgid; px_[k] = make_float4(c[0], c[1], c[2], 0.f); }
                }
            }
            const unsigned long long bal = __ballot(keep);
            const uint32_t slot = head + (uint32_t)__popcll(bal & lt);
            if (keep) {
                c2r.w = __uint_as_float(cgid);                      // (replaced by the survivor number below)
                s0[slot] = c0r; s1[slot] = c1r; s2[slot] = c2r;
                if (NORMAL) s3[slot] = c3r;
#pragma unroll
                for (int k = 0; k < NX; k++) sx[k][slot] = cxr[k];
            }
            head += (uint32_t)__popcll(bal);
            scanned += EMD_WAVE;
        }
        __syncthreads();
        // ---- the queued entries (not the whole list) get the exact per-sub-block test; one byte list per sub-block.  The entries that
        //      reach at least one sub-block are the quadrant's SURVIVORS: numbered here, in list order, ids written out for the backward
        for (uint32_t base = 0; base < head; base += EMD_WAVE) {
            if (STATS) st_cull++;
            const uint32_t slot = base + lane;
            uint32_t m4 = 0u, gid = 0u;
            if (slot < head) { gid = __float_as_uint(reinterpret_cast<const float*>(&s2[slot])[3]); m4 = ellipse_subblock_mask(s0[slot], s1[slot], qx0, qy0); }
            const unsigned long long b0 = __ballot(m4 & 1u), b1 = __ballot(m4 & 2u), b2 = __ballot(m4 & 4u), b3 = __ballot(m4 & 8u);
            const unsigned long long ba = b0 | b1 | b2 | b3;
            if (m4) {
                const uint32_t num = total + (uint32_t)__popcll(ba & lt);
                sv[num] = gid;
                reinterpret_cast<float*>(&s2[slot])[3] = __uint_as_float(num + 1);      // 1-based number of this survivor in the quadrant's list
            }
            total += (uint32_t)__popcll(ba);
            if (m4 & 1u) s_list[0][len0 + (uint32_t)__popcll(b0 & lt)] = (uint8_t)slot;
            if (m4 & 2u) s_list[1][len1 + (uint32_t)__popcll(b1 & lt)] = (uint8_t)slot;
            if (m4 & 4u) s_list[2][len2 + (uint32_t)__popcll(b2 & lt)] = (uint8_t)slot;
            if (m4 & 8u) s_list[3][len3 + (uint32_t)__popcll(b3 & lt)] = (uint8_t)slot;
            len0 += (uint32_t)__popcll(b0); len1 += (uint32_t)__popcll(b1); len2 += (uint32_t)__popcll(b2); len3 += (uint32_t)__popcll(b3);
        }
        __syncthreads();
        const uint32_t nmax = max(max(len0, len1), max(len2, len3));
        if (STATS) { st_drain += nmax; st_useful += len0 + len1 + len2 + len3; }
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
#pragma unroll
                for (int k = 0; k < NX; k++) {          // the same fma as the main colour: an extra set equals a separate call bit for bit
                    const float4 gx = sx[k][j];
                    X[k][0] = __builtin_fmaf(gx.x, w, X[k][0]); X[k][1] = __builtin_fmaf(gx.y, w, X[k][1]); X[k][2] = __builtin_fmaf(gx.z, w, X[k][2]);
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
        out_color[pix] = __builtin_fmaf(T, bg0, C0);
        out_color[HW + pix] = __builtin_fmaf(T, bg1, C1);
        out_color[2 * HW + pix] = __builtin_fmaf(T, bg2, C2);
        out_depth[pix] = Dz;
        if (NORMAL) { out_normal[pix] = N0; out_normal[HW + pix] = N1; out_normal[2 * HW + pix] = N2; }
#pragma unroll
        for (int k = 0; k < NX; k++) {           // colour + T_final * bg with the same bg, as a separate call with this colour set gives
            d.xout[k][pix] = __builtin_fmaf(T, bg0, X[k][0]);
            d.xout[k][HW + pix] = __builtin_fmaf(T, bg1, X[k][1]);
            d.xout[k][2 * HW + pix] = __builtin_fmaf(T, bg2, X[k][2]);
        }
        out_alpha[pix] = 1.f - T;
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
    // what the backward needs of the survivor list: everything up to the quadrant's deepest contributor
    uint32_t need = inside ? last : 0u;
    for (int off = 32; off; off >>= 1) need = max(need, (uint32_t)__shfl_xor((int)need, off));
    if (lane == 0) quad_need[4 * tile + quad] = need;
    if (STATS && lane == 0) {
        // [0] scan steps (64 list words each)  [1] cull steps (64 queued entries each)  [2] drain iterations (one entry per 16-lane row each)
        // [3] entries the four rows actually held (<= 4 x [2])  [4] scan -> cull -> drain rounds  [5] waves with a non-empty tile
        atomicAdd(loop_stats, st_scan); atomicAdd(loop_stats + 1, st_cull); atomicAdd(loop_stats + 2, st_drain); atomicAdd(loop_stats + 3, st_useful);
        atomicAdd(loop_stats + 4, st_round); atomicAdd(loop_stats + 5, 1ull);
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
// The list is walked back to front, as upstream does, so lane 0 of a batch is its deepest entry:
//     T_k  = T_final / prod_{i>=k} (1 - alpha_i)      inclusive scan-product of the reciprocals   (6 v_mul_f32_dpp)
//     R_k  = sum_{i>k} g_i alpha_i T_i                exclusive scan-sum                          (6 v_add_f32_dpp + 1 wave_shr)
//     dL/dalpha_k = g_k T_k + (Q - R_k) / (1 - alpha_k),   g = colour . dL/dC + depth dL/dD (+ normal . dL/dN),
//     Q = T_final (dL/dalpha_img - bg . dL/dC).
// Both running quantities shrink with the transmittance of the entry they are used for, so every (pixel, Gaussian) partial
// carries a RELATIVE rounding error (a first version ran front to back with S_k = sum_{i<=k} and the forward image's total:
// the difference total - S_k left an absolute error of eps x total on entries whose own weight was a thousandth of it).  The ten derivative sums then accumulate in the lane's registers over the 64 pixels
// with no cross-lane traffic at all; per 64 (pixel, entry) pairs the wave spends 13 DPP ops instead of 60.
// Running T and S per pixel are carried across batches of 64 entries in LDS (broadcast loads, one-lane store).
// ---------------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------------
// K7 kernel: one wave (64-thread workgroup) per 8x8 quadrant.
//
// (A workgroup-per-tile version staged 256 list entries at a time and cut each quadrant's list at the chunk boundary:
// batches only ~78 % full, four waves waiting on three barriers per chunk.)
// Here a wave scans the tile list by itself, 64 entries per step (records prefetched one step
// ahead), keeps the entries whose tight footprint overlaps ITS quadrant in a small LDS queue, and runs the pixel loop
// whenever 64 entries are queued: every batch but the last is full, there is no workgroup barrier, and the scan stops
// at the deepest contributor of this quadrant rather than of the whole tile.  Gradients leave per batch as
// row-shaped global float atomics (48 contiguous bytes per Gaussian) through a per-wave LDS staging tile.
// ---------------------------------------------------------------------------------------------------
#define BQ_QUEUE 128

template <bool NORMAL, bool ABS, int NX, bool STATS = false>
__global__ void __launch_bounds__(EMD_WAVE) __attribute__((amdgpu_waves_per_eu(NX == 0 ? 4 : NX == 1 ? 3 : 2))) k_render_backward_q(RenderDims d, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ ranges,
                                                                const uint32_t* __restrict__ surv,
                                                                const uint32_t* __restrict__ quad_need,
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
                                                                float* __restrict__ grad_rec,
                                                                float* __restrict__ zero_buf, int zero_n,
                                                                unsigned long long* __restrict__ pair_stats) {
    // the tiny actor-pose gradient table K8 accumulates into is cleared here (K8 starts after this kernel): no memset launch
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_n; i += EMD_WAVE) zero_buf[i] = 0.f;
    // Round 3: the forward pass leaves, per quadrant, the ids of the entries whose footprint reaches it ("survivors", in list
    // order) and the number of them in front of its deepest contributor.  The backward walks that list back to front, 64 survivors
    // per batch, every lane keeping ITS entry's record in registers: no scan of the tile list, no footprint test, no queue.
    constexpr int STRIDE = EMD_BWD_STRIDE + 4 * NX;             // accumulator row: 12 floats + (r, g, b, -) per extra set
    constexpr int PB = NORMAL ? 6 : 4;                          // s_pix planes of the main call; two more per extra set
    __shared__ float4 s_stage4[EMD_WAVE * STRIDE / 4];          // gradient staging tile: 64 rows x STRIDE floats
    __shared__ uint32_t q_id[EMD_WAVE];
    float* const s_stage = reinterpret_cast<float*>(s_stage4);
    // per-pixel constants and running state, read back as wave-uniform (broadcast) LDS loads in the pixel loop: keeps
    // ~15 v_readlane / v_mov / v_cndmask per pixel-iteration off the VALU, which is what bounds this kernel.
    // Pixels are handled in horizontal pairs (a, b) = (2 pp, 2 pp + 1); every quantity is stored as the pair:
    //   [pp][0] = (dC0 a,b | dC1 a,b)  [pp][1] = (dC2 a,b | dD a,b)  [pp][2] = (running T a,b | running S a,b)
    //   [pp][3] = (Q a,b | -)          [pp][4] = (dN0 a,b | dN1 a,b) [pp][5] = (dN2 a,b | -)
    //   [pp][PB + 2k] = (dX_k0 a,b | dX_k1 a,b)  [pp][PB + 2k + 1] = (dX_k2 a,b | -)     gradients of extra colour set k
    __shared__ float4 s_pix[EMD_WAVE / 2][PB + 2 * NX];
    __shared__ uint32_t s_myn[EMD_WAVE];                        // n_contrib of the quadrant's pixels (half-wave batches read it per lane)
    uint32_t quad;
    const uint32_t tile = ordered_quadrant_block(blockIdx.x, (uint32_t)(d.gx * d.gy), tile_order, &quad);
    if (tile >= (uint32_t)(d.gx * d.gy)) return;
    const uint32_t lane = threadIdx.x;
    const float tile_x0 = (float)((tile % (uint32_t)d.gx) * EMD_TILE_X), tile_y0 = (float)((tile / (uint32_t)d.gx) * EMD_TILE_Y);
    const float qx0 = tile_x0 + (float)((quad & 1) * 8), qy0 = tile_y0 + (float)((quad >> 1) * 8);
    const int px = (int)qx0 + (int)(lane & 7), py = (int)qy0 + (int)(lane >> 3);
    const bool inside = px < d.W && py < d.H;
    const uint32_t start = ranges[2 * tile], n_tile = ranges[2 * tile + 1] - start;
    const uint32_t wave_n = min(quad_need[4 * tile + quad], n_tile);       // survivors in front of this quadrant's deepest contributor
    if (wave_n == 0) return;
    const uint32_t* const sv = surv + 4 * (size_t)start + (size_t)quad * n_tile;
    const size_t HW = (size_t)d.H * d.W, pix = (size_t)py * d.W + px;
    const uint32_t my_n = inside ? n_contrib[pix] : 0u;                    // 1-based number of the last survivor that contributed to this pixel
    s_myn[lane] = my_n;
    float dC0 = 0.f, dC1 = 0.f, dC2 = 0.f, dD = 0.f, dN0 = 0.f, dN1 = 0.f, dN2 = 0.f, Q = 0.f, Tf = 1.f;
    float dX[NX ? NX : 1][3];
#pragma unroll
    for (int k = 0; k < (NX ? NX : 1); k++) dX[k][0] = dX[k][1] = dX[k][2] = 0.f;
    if (inside) {
        float dA = 0.f;
#ifndef K7_ABL_NO_PROLOGUE       /* ablation build: the per-pixel inputs are constants instead of five image-sized loads (the walk itself is unchanged) */
        Tf = final_T[pix];
        if (dL_dcolor) { dC0 = dL_dcolor[pix]; dC1 = dL_dcolor[HW + pix]; dC2 = dL_dcolor[2 * HW + pix]; }
        if (dL_ddepth) dD = dL_ddepth[pix];
        if (dL_dalpha) dA = dL_dalpha[pix];
#else
        Tf = 0.37f; dC0 = 1e-6f; dC1 = -2e-6f; dC2 = 3e-6f;
#endif
        if (NORMAL && dL_dnormal) { dN0 = dL_dnormal[pix]; dN1 = dL_dnormal[HW + pix]; dN2 = dL_dnormal[2 * HW + pix]; }
        const float bg0 = d.bg_dev ? d.bg_dev[0] : d.bg[0], bg1 = d.bg_dev ? d.bg_dev[1] : d.bg[1], bg2 = d.bg_dev ? d.bg_dev[2] : d.bg[2];
        float bgdot = bg0 * dC0 + bg1 * dC1 + bg2 * dC2;
#pragma unroll
        for (int k = 0; k < NX; k++) {
            if (!d.xgrad[k]) continue;
            dX[k][0] = d.xgrad[k][pix]; dX[k][1] = d.xgrad[k][HW + pix]; dX[k][2] = d.xgrad[k][2 * HW + pix];
            bgdot += bg0 * dX[k][0] + bg1 * dX[k][1] + bg2 * dX[k][2];
        }
        Q = Tf * (dA - bgdot);              // background and alpha image: both depend on T_final = prod (1 - alpha_i)
    }
    {
        float* sp = reinterpret_cast<float*>(&s_pix[lane >> 1][0]) + (lane & 1);
        sp[0] = dC0; sp[2] = dC1; sp[4] = dC2; sp[6] = dD;
        sp[8] = Tf; sp[10] = 0.f;       // transmittance behind / weighted colour behind the entries processed so far (back to front)
        sp[12] = Q; sp[14] = 0.f;
        if (NORMAL) { sp[16] = dN0; sp[18] = dN1; sp[20] = dN2; sp[22] = 0.f; }
#pragma unroll
        for (int k = 0; k < NX; k++) { float* sx_ = sp + 4 * (PB + 2 * k); sx_[0] = dX[k][0]; sx_[2] = dX[k][1]; sx_[4] = dX[k][2]; sx_[6] = 0.f; }
    }
    __syncthreads();
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // STATS (diagnostic instantiation, EmdBwdArgs.pair_stats): (pixel, entry) pairs this wave evaluates / pairs that contribute
    unsigned long long st_eval = 0ull, st_hit = 0ull, st_rows = 0ull, st_atoms = 0ull;       // + atomic rows with a non-zero float / float atomics issued

    // One batch of nb survivors: lane = entry, lane 0 the DEEPEST; `pos` = 0-based number of the lane's survivor, `gid` its Gaussian.
    // HALF (round 4): a batch of at most 32 survivors -- only the deepest step of a walk can be that short -- runs on both halves of the
    // wave at once: lanes 0..31 take the entries against pixel pairs 0..15, lanes 32..63 the SAME entries against pixel pairs 16..31, so the
    // pixel loop has 16 iterations instead of 32 (the scans run inside the 32-lane halves: five DPP steps; the per-pixel constants become
    // two-address LDS loads; the two halves' moment sums of an entry are added at the end).  A quadrant's walk is ~2.4 batches long and its
    // deepest one is half empty on average: the 64-lane form spent a third of its lane-iterations on lanes without an entry.
    auto process_batch = [&](auto half_c, uint32_t nb, const float4 g0, const float4 g1, const float4 g2, const float4 g3, const float4* gxc, uint32_t pos,
                             uint32_t gid) {
        constexpr bool HALF = decltype(half_c)::value;
        const uint32_t el = HALF ? (lane & 31u) : lane, hp = HALF ? (lane >> 5) : 0u;      // entry slot of the lane, its half of the pixels
        const bool valid = el < nb;
        v2f xr_2[NX ? NX : 1], xg_2[NX ? NX : 1], xb_2[NX ? NX : 1];
#pragma unroll
        for (int k = 0; k < (NX ? NX : 1); k++) xr_2[k] = xg_2[k] = xb_2[k] = splat2(0.f);
        const uint32_t last_pos = readlane_u32(pos, (int)nb - 1);     // the batch runs back to front: lane nb-1 is the shallowest entry
        const v2f z2 = splat2(0.f);
        v2f m0_2 = z2, gx_2 = z2, gy_2 = z2, m2xx_2 = z2, m2xy_2 = z2, m2yy_2 = z2, dz_2 = z2, r_2 = z2, g_2 = z2, b_2 = z2;
        float a_ax = 0.f, a_ay = 0.f;
        for (int pp0 = 0; pp0 < (HALF ? EMD_WAVE / 4 : EMD_WAVE / 2); pp0++) {
            const int pp = HALF ? pp0 + (int)(hp * (EMD_WAVE / 4)) : pp0;                    // (per lane half in the HALF form)
            uint32_t n_a, n_b;
            if (HALF) {
                n_a = s_myn[2 * pp]; n_b = s_myn[2 * pp + 1];
                if (__ballot(max(n_a, n_b) > last_pos) == 0ull) continue;
            } else {
                n_a = readlane_u32(my_n, 2 * pp); n_b = readlane_u32(my_n, 2 * pp + 1);
                if (max(n_a, n_b) <= last_pos) continue;         // both pixels terminated before the shallowest entry of this batch
            }
            const float4 c01 = s_pix[pp][0], c2d = s_pix[pp][1], ts = s_pix[pp][2], qq = s_pix[pp][3];
            const float pxs = qx0 + (float)((2 * pp) & 7), pys = qy0 + (float)(pp >> 2);
            const float dy = g0.y - pys;
            const v2f dx = (v2f){g0.x - pxs, g0.x - (pxs + 1.f)};
            const v2f power = gauss_power2(g1.x, g1.y, g1.z, dx, dy);
            // (the hardware v_exp_f32 with a pinned re-evaluation next to the 1/255 threshold was measured in round 3: 0.489 vs 0.494 ms,
            //  i.e. nothing, while its 1-ulp differences cost the end-to-end rotation bar of one sweep scene its margin: not kept)
            const v2f G = pinned_exp2(power);
            const v2f aw = splat2(g0.w) * G;
            const v2f alpha = (v2f){fminf(0.99f, aw.x), fminf(0.99f, aw.y)};
            const bool hit_a = valid && pos < n_a && power.x <= 0.f && alpha.x >= (1.f / 255.f);
            const bool hit_b = valid && pos < n_b && power.y <= 0.f && alpha.y >= (1.f / 255.f);
            if (STATS) { st_eval += (HALF ? 4ull : 2ull) * nb; st_hit += (unsigned long long)(__popcll(__ballot(hit_a)) + __popcll(__ballot(hit_b))); }
            if (__ballot(hit_a || hit_b) == 0ull) continue;
            const v2f a = (v2f){hit_a ? alpha.x : 0.f, hit_b ? alpha.y : 0.f};
            const v2f om = splat2(1.f) - a;
            const v2f inv = (v2f){__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
            // lane 0 is the DEEPEST entry of the batch.  T_k (in front of entry k) = T_behind / prod_{entries from k to the back} (1 - alpha):
            // an inclusive prefix product of the reciprocals over the lanes, times the value carried from the batches behind
            float t_a = inv.x, t_b = inv.y;
            if (HALF) half_scan_mul2_f32_asm(t_a, t_b); else wave_scan_mul2_f32_asm(t_a, t_b);
            const v2f cT = (v2f){ts.x, ts.y}, cR = (v2f){ts.z, ts.w};
            const v2f Tk = cT * (v2f){t_a, t_b};
            const v2f w = a * Tk;
            v2f g = splat2(g2.x) * (v2f){c01.x, c01.y} + splat2(g2.y) * (v2f){c01.z, c01.w} + splat2(g2.z) * (v2f){c2d.x, c2d.y} +
                    splat2(g0.z) * (v2f){c2d.z, c2d.w};
            if (NORMAL) {
                const float4 n01 = s_pix[pp][4], n2 = s_pix[pp][5];
                g += splat2(g3.x) * (v2f){n01.x, n01.y} + splat2(g3.y) * (v2f){n01.z, n01.w} + splat2(g3.z) * (v2f){n2.x, n2.y};
            }
            float4 x01[NX ? NX : 1], x2_[NX ? NX : 1];
#pragma unroll
            for (int k = 0; k < NX; k++) {
                x01[k] = s_pix[pp][PB + 2 * k]; x2_[k] = s_pix[pp][PB + 2 * k + 1];
                g += splat2(gxc[k].x) * (v2f){x01[k].x, x01[k].y} + splat2(gxc[k].y) * (v2f){x01[k].z, x01[k].w} + splat2(gxc[k].z) * (v2f){x2_[k].x, x2_[k].y};
            }
            // R_k = sum over the entries BEHIND k of g_i alpha_i T_i: exclusive prefix sum over the lanes + the carried value
            const v2f gw = g * w;
            float s_a = gw.x, s_b = gw.y;
            if (HALF) half_scan_add2_f32_asm(s_a, s_b); else wave_scan_add2_f32_asm(s_a, s_b);
            float ex_a = wave_shift_up1_f32(s_a, 0.f), ex_b = wave_shift_up1_f32(s_b, 0.f);
            if (HALF && lane == 32u) { ex_a = 0.f; ex_b = 0.f; }                             // (lane 32 is the deepest entry of its half)
            const v2f Rk = cR + (v2f){ex_a, ex_b};
            // dL/dalpha_k = g_k T_k - R_k / (1 - alpha_k) + T_final (dL/dalpha_img - bg . dL/dC) / (1 - alpha_k)
            v2f dL_da = g * Tk + inv * ((v2f){qq.x, qq.y} - Rk);
            dL_da = (v2f){hit_a ? dL_da.x : 0.f, hit_b ? dL_da.y : 0.f};
            const v2f u = G * (splat2(g0.w) * dL_da);           // G dL/dG
            const v2f ux = u * dx, uy = u * splat2(dy);
            // d mean = -sum_pixels u Conic d, formed PER PIXEL (as upstream does) and not as -Conic (sum u d): for an elongated
            // Gaussian the pixels that carry weight lie along its major axis, where A dx + B dy nearly cancels -- summing u dx and
            // u dy separately and combining afterwards loses the anisotropy ratio in significant digits
            const v2f tx = __builtin_elementwise_fma(splat2(g1.x), dx, splat2(g1.y * dy));
            const v2f ty = __builtin_elementwise_fma(splat2(g1.y), dx, splat2(g1.z * dy));
            const v2f ax = u * tx, ay = u * ty;
            m0_2 += u; gx_2 += ax; gy_2 += ay;
            m2xx_2 += ux * dx; m2xy_2 += ux * splat2(dy); m2yy_2 += uy * splat2(dy);
            if (ABS) { a_ax += fabsf(ax.x) + fabsf(ax.y); a_ay += fabsf(ay.x) + fabsf(ay.y); }
            dz_2 += w * (v2f){c2d.z, c2d.w};
            r_2 += w * (v2f){c01.x, c01.y}; g_2 += w * (v2f){c01.z, c01.w}; b_2 += w * (v2f){c2d.x, c2d.y};
#pragma unroll
            for (int k = 0; k < NX; k++) {
                xr_2[k] += w * (v2f){x01[k].x, x01[k].y}; xg_2[k] += w * (v2f){x01[k].z, x01[k].w}; xb_2[k] += w * (v2f){x2_[k].x, x2_[k].y};
            }
            // lane 63 holds the batch totals: carry the transmittance and the weighted colour in front of this batch to the next one
            if ((lane & (HALF ? 31u : 63u)) == (HALF ? 31u : 63u)) s_pix[pp][2] = make_float4(cT.x * t_a, cT.y * t_b, cR.x + s_a, cR.y + s_b);
        }
        if (HALF) {
            // the two pixel halves of every entry: lane e += lane e + 32 (both halves end up with the sums)
            auto fold = [](v2f& x) { x.x += __shfl_xor(x.x, 32); x.y += __shfl_xor(x.y, 32); };
            fold(m0_2); fold(gx_2); fold(gy_2); fold(m2xx_2); fold(m2xy_2); fold(m2yy_2); fold(dz_2); fold(r_2); fold(g_2); fold(b_2);
            if (ABS) { a_ax += __shfl_xor(a_ax, 32); a_ay += __shfl_xor(a_ay, 32); }
#pragma unroll
            for (int k = 0; k < NX; k++) { fold(xr_2[k]); fold(xg_2[k]); fold(xb_2[k]); }
        }
        const float m0 = m0_2.x + m0_2.y, gx = gx_2.x + gx_2.y, gy = gy_2.x + gy_2.y, m2xx = m2xx_2.x + m2xx_2.y,
                    m2xy = m2xy_2.x + m2xy_2.y, m2yy = m2yy_2.x + m2yy_2.y, a_dz = dz_2.x + dz_2.y, a_r = r_2.x + r_2.y,
                    a_g = g_2.x + g_2.y, a_b = b_2.x + b_2.y;
        // rows through LDS so that consecutive lanes add consecutive floats of one 48-byte accumulator row
        q_id[lane] = gid;
        float4* row = reinterpret_cast<float4*>(s_stage + lane * STRIDE);
        row[0] = make_float4(-gx, -gy, a_dz, m0 * __builtin_amdgcn_rcpf(g0.w));
        row[1] = make_float4(-0.5f * m2xx, -m2xy, -0.5f * m2yy, a_r);
        row[2] = make_float4(a_g, a_b, ABS ? a_ax : 0.f, ABS ? a_ay : 0.f);
#pragma unroll
        for (int k = 0; k < NX; k++) row[3 + k] = make_float4(xr_2[k].x + xr_2[k].y, xg_2[k].x + xg_2[k].y, xb_2[k].x + xb_2[k].y, 0.f);
        __syncthreads();
        if (STATS) {
            bool nz = false;
            if (lane < nb)
                for (int v = 0; v < STRIDE; v++) nz = nz || s_stage[lane * STRIDE + v] != 0.f;
            st_rows += (unsigned long long)__popcll(__ballot(nz));
        }
        // (round 6, measured and not kept: walking 10 instead of 12 floats per row without absgrad -- 10 atomic instructions per 64 rows instead of 12 -- changes
        //  nothing: 0.4337 / 0.4362 against 0.4348 / 0.4372 ms, profiles/r06_render_ablations.txt; the lanes of floats 10, 11 were already masked off)
        for (uint32_t idx = lane; idx < nb * STRIDE; idx += EMD_WAVE) {
            const uint32_t e = idx / STRIDE, v = idx % STRIDE;
            const float val = s_stage[idx];
            if (STATS) st_atoms += (unsigned long long)__popcll(__ballot(val != 0.f));
#ifndef K7_ABL_NO_FLUSH          /* ablation build (profiles/r06_render_ablations.txt): the rows are staged but never added to HBM */
            if (val != 0.f) atomicAdd(grad_rec + (size_t)q_id[e] * STRIDE + v, val);
#else
            if (val == 12345.678f) grad_rec[0] = val;
#endif
        }
        __syncthreads();
    };

    // The survivor list is walked BACK TO FRONT, 64 per step; the deepest step is the partial one, so every later batch is full.
    // Software pipeline: Gaussian ids three steps ahead, records two steps ahead of the step being processed, so the dependent
    // id -> record gather (two L2 round trips) is off the critical path of a wave that shares its SIMD with only ~3 others.
    const int steps = (int)((wave_n + EMD_WAVE - 1) / EMD_WAVE);
    auto step_lo = [&](int st_) -> int { return (steps - 1 - st_) * EMD_WAVE; };
    auto step_idx = [&](int st_) -> int { return st_ < steps ? min((int)wave_n, step_lo(st_) + EMD_WAVE) - 1 - (int)lane : -1; };   // this lane's survivor in step st_
    auto in_step = [&](int st_, int i) -> bool { return st_ < steps && i >= step_lo(st_); };
    auto load_id = [&](int st_) -> uint32_t { const int i = step_idx(st_); return in_step(st_, i) ? sv[i] : 0u; };
    float4 a0 = zero4, a1 = zero4, a2 = zero4, a3 = zero4, b0 = zero4, b1 = zero4, b2 = zero4, b3 = zero4;
    float4 ax_[NX ? NX : 1], bx_[NX ? NX : 1];
#pragma unroll
    for (int k = 0; k < (NX ? NX : 1); k++) ax_[k] = bx_[k] = zero4;
    auto load_extra = [&](uint32_t id, float4* o) {
#pragma unroll
        for (int k = 0; k < NX; k++) { const float* c = d.xcol[k] + 3 * (size_t)id; o[k] = make_float4(c[0], c[1], c[2], 0.f); }
    };
    uint32_t idA = load_id(0), idB = load_id(1), idC = load_id(2);
    if (in_step(0, step_idx(0))) {
        const float4* r = rec + (size_t)idA * EMD_REC_F4;
        a0 = r[0]; a1 = r[1]; a2 = r[2];
        if (NORMAL) a3 = r[3];
        load_extra(idA, ax_);
    }
    if (in_step(1, step_idx(1))) {
        const float4* r = rec + (size_t)idB * EMD_REC_F4;
        b0 = r[0]; b1 = r[1]; b2 = r[2];
        if (NORMAL) b3 = r[3];
        load_extra(idB, bx_);
    }
    for (int st_ = 0; st_ < steps; st_++) {
        const int idx = step_idx(st_);
        const float4 c0r = a0, c1r = a1, c2r = a2, c3r = a3;
        float4 cxr[NX ? NX : 1];
#pragma unroll
        for (int k = 0; k < NX; k++) { cxr[k] = ax_[k]; ax_[k] = bx_[k]; }
        const uint32_t cid = idA;
        a0 = b0; a1 = b1; a2 = b2; a3 = b3; idA = idB;
        idB = idC;
        b0 = b1 = b2 = b3 = zero4;
#pragma unroll
        for (int k = 0; k < NX; k++) bx_[k] = zero4;
        if (in_step(st_ + 2, step_idx(st_ + 2))) {
            const float4* r = rec + (size_t)idB * EMD_REC_F4;
            b0 = r[0]; b1 = r[1]; b2 = r[2];
            if (NORMAL) b3 = r[3];
            load_extra(idB, bx_);
        }
        idC = load_id(st_ + 3);
        const uint32_t nb = (uint32_t)(min((int)wave_n, step_lo(st_) + EMD_WAVE) - step_lo(st_));
        if (nb <= EMD_WAVE / 2) {
            // (only the deepest step can be this short) the upper half of the wave takes copies of the lower half's entries
            const int src = (int)(lane & 31u);
            auto dup4 = [&](float4 v) { return make_float4(__shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src), __shfl(v.w, src)); };
            const float4 h0 = dup4(c0r), h1 = dup4(c1r), h2 = dup4(c2r), h3 = NORMAL ? dup4(c3r) : c3r;
            float4 hx[NX ? NX : 1];
#pragma unroll
            for (int k = 0; k < NX; k++) hx[k] = dup4(cxr[k]);
            process_batch(std::true_type{}, nb, h0, h1, h2, h3, hx, (uint32_t)__shfl(max(idx, 0), src), (uint32_t)__shfl((int)cid, src));
        } else {
            process_batch(std::false_type{}, nb, c0r, c1r, c2r, c3r, cxr, (uint32_t)max(idx, 0), cid);
        }
    }
    if (STATS && lane == 0) { atomicAdd(pair_stats, st_eval); atomicAdd(pair_stats + 1, st_hit); atomicAdd(pair_stats + 2, st_rows); atomicAdd(pair_stats + 3, st_atoms); }
}

RenderDims make_dims(const EmdSettings& s, const float* sdev, const EmdExtra* x) {
    RenderDims d;
    d.bg_dev = sdev;
    d.W = s.image_width; d.H = s.image_height;
    d.gx = (d.W + EMD_TILE_X - 1) / EMD_TILE_X; d.gy = (d.H + EMD_TILE_Y - 1) / EMD_TILE_Y;
    d.bg[0] = s.bg[0]; d.bg[1] = s.bg[1]; d.bg[2] = s.bg[2];
    for (int k = 0; k < EMD_MAX_EXTRA; k++) {
        const bool on = x && k < x->num;
        d.xcol[k] = on ? x->colors[k] : nullptr;
        d.xout[k] = on ? x->out[k] : nullptr;
        d.xgrad[k] = on ? x->dL_dout[k] : nullptr;
    }
    return d;
}

}  // namespace

int emd_launch_render_forward(const EmdSettings& s, const float* sdev, int flags, const GeomWs& g, const BinWs& b, const ImgWs& im,
                              float* out_color, float* out_depth, float* out_normal, float* out_alpha, const EmdExtra* x,
                              unsigned long long* loop_stats, hipStream_t st) {
    const RenderDims d = make_dims(s, sdev, x);
    const int T = d.gx * d.gy;
    if (T <= 0) return EMD_OK;
    const uint32_t* pl = b.vals[b.sorted_buf];
    const int nx = x ? x->num : 0;
#define LAUNCH_FWD(N_, X_)                                                                                                          \
    hipLaunchKernelGGL((k_render_forward_q<N_, X_>), dim3(4 * padded_tile_grid(T)), dim3(EMD_WAVE), 0, st, d, b.tile_order, b.ranges, pl, g.rec, \
                       out_color, out_depth, out_normal, out_alpha, im.final_T, im.n_contrib, b.surv, b.quad_need, nullptr)
    const bool nrm = (flags & EMD_FLAG_NORMAL) != 0;
    if (nx == 0 && loop_stats) {          // diagnostic: the same kernel with the loop counters compiled in
        if (nrm) hipLaunchKernelGGL((k_render_forward_q<true, 0, true>), dim3(4 * padded_tile_grid(T)), dim3(EMD_WAVE), 0, st, d, b.tile_order, b.ranges, pl, g.rec,
                                    out_color, out_depth, out_normal, out_alpha, im.final_T, im.n_contrib, b.surv, b.quad_need, loop_stats);
        else hipLaunchKernelGGL((k_render_forward_q<false, 0, true>), dim3(4 * padded_tile_grid(T)), dim3(EMD_WAVE), 0, st, d, b.tile_order, b.ranges, pl, g.rec,
                                out_color, out_depth, out_normal, out_alpha, im.final_T, im.n_contrib, b.surv, b.quad_need, loop_stats);
    }
    else if (nx == 0) { if (nrm) LAUNCH_FWD(true, 0); else LAUNCH_FWD(false, 0); }
    else if (nx == 1) { if (nrm) LAUNCH_FWD(true, 1); else LAUNCH_FWD(false, 1); }
    else { if (nrm) LAUNCH_FWD(true, 2); else LAUNCH_FWD(false, 2); }
#undef LAUNCH_FWD
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_render_backward(const EmdSettings& s, const float* sdev, int flags, const GeomWs& g, const BinWs& b, const ImgWs& im,
                               const float* out_color, const float* out_depth, const float* out_normal,
                               const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                               const float* dL_dnormal, const EmdExtra* x, float* grad_rec, float* zero_buf, int zero_n,
                               unsigned long long* pair_stats, hipStream_t st) {
    const RenderDims d = make_dims(s, sdev, x);
    const int T = d.gx * d.gy;
    if (T <= 0) return EMD_OK;
    const bool nrm = (flags & EMD_FLAG_NORMAL) && dL_dnormal && out_normal, ab = flags & EMD_FLAG_ABSGRAD;
    const int nx = x ? x->num : 0;
#define LAUNCH_BWD(N_, A_, X_)                                                                                          \
    hipLaunchKernelGGL((k_render_backward_q<N_, A_, X_>), dim3(4 * padded_tile_grid(T)), dim3(EMD_WAVE), 0, st, d, b.tile_order, b.ranges, b.surv, b.quad_need, g.rec,   \
                       im.final_T, im.n_contrib, out_color, out_depth, out_normal, dL_dcolor, dL_ddepth, dL_dalpha,    \
                       dL_dnormal, grad_rec, zero_buf, zero_n, pair_stats)
#define LAUNCH_BWD_X(X_)                                   \
    if (nrm && ab) LAUNCH_BWD(true, true, X_);             \
    else if (nrm) LAUNCH_BWD(true, false, X_);             \
    else if (ab) LAUNCH_BWD(false, true, X_);              \
    else LAUNCH_BWD(false, false, X_)
    if (nx == 0 && pair_stats) {          // diagnostic: the same kernel with the pair counters compiled in
        hipLaunchKernelGGL((k_render_backward_q<false, false, 0, true>), dim3(4 * padded_tile_grid(T)), dim3(EMD_WAVE), 0, st, d, b.tile_order, b.ranges, b.surv,
                           b.quad_need, g.rec, im.final_T, im.n_contrib, out_color, out_depth, out_normal, dL_dcolor, dL_ddepth, dL_dalpha, nullptr, grad_rec,
                           zero_buf, zero_n, pair_stats);
    }
    else if (nx == 0) { LAUNCH_BWD_X(0); }
    else if (nx == 1) { LAUNCH_BWD_X(1); }
    else { LAUNCH_BWD_X(2); }
#undef LAUNCH_BWD_X
#undef LAUNCH_BWD
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
