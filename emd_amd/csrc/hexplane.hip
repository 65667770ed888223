// hexplane.hip -- fused multi-scale HexPlane feature lookup, forward and backward.
//
// Replaces HexPlaneField.get_density / interpolate_ms_features (S3Gaussian/scene/hexplane.py:18-110,150-183): per scale six
// F.grid_sample launches (bilinear, align_corners, border padding) + five products + a concat, and their backward
// (six grid_sampler_2d_backward launches that scatter channel-FIRST: one float atomic per channel per tap, each to a
// different cache line).  Here the planes are handed over CHANNEL-LAST ([res_h][res_w][C]) and a point is owned by C
// consecutive lanes (lane = channel): every tap is one coalesced C x 4-byte read, the product over the six planes and
// the concat over scales happen in registers, and the backward scatters C consecutive floats per tap (the row shape
// float atomics like).  One launch forward, one backward, for all scales.
// Three kernels: k_hexplane_fwd2 (two-phase forward, at the L2 -> CU gather rate), k_hexplane_bwd_agg (backward that aggregates
// plane gradients in LDS over a spatially coherent visiting order; VALU-issue-bound) and k_hexplane_bwd (direct float atomics,
// bound by the L2 atomic units; used when no order is given).
#include <limits.h>
#include <string.h>

#include "common.h"
#include "device_utils.h"

namespace {

struct Bilin { int x0, x1, y0, y1; float fx, fy, cx, cy; };   // cx, cy: d(ix)/d(coord) incl. the border-clip mask

// F.grid_sample coordinate handling: align_corners=True, padding_mode='border'
__device__ __forceinline__ void unnormalize(float c, int size, float& idx, float& dscale) {
    const float s = 0.5f * (float)(size - 1);
    float v = (c + 1.f) * s;
    dscale = s;
    if (!(v > 0.f)) { v = 0.f; dscale = 0.f; }                       // clip_coordinates_set_grad: 0 outside [0, size-1]
    else if (!(v < (float)(size - 1))) { v = (float)(size - 1); dscale = 0.f; }
    idx = v;
}

__device__ __forceinline__ Bilin bilin(float cx, float cy, int W, int H) {
    Bilin b;
    float ix, iy;
    unnormalize(cx, W, ix, b.cx);
    unnormalize(cy, H, iy, b.cy);
    const float x0 = floorf(ix), y0 = floorf(iy);
    b.fx = ix - x0; b.fy = iy - y0;
    b.x0 = (int)x0; b.y0 = (int)y0;
    b.x1 = min(b.x0 + 1, W - 1); b.y1 = min(b.y0 + 1, H - 1);      // the out-of-range neighbour has weight 0
    return b;
}

// One axis of a tap.  A scale has four axes (x, y, z, t) and six planes that pair them: the un-normalise / clip / floor work is
// done once per axis and shared by the three planes the axis takes part in (both lookup kernels are VALU-heavy -- at the measured
// 2.35 cycles per plain wave instruction, profiles/r02_issue_rate_microbench.txt, rocprofv3's SQ_INSTS_VALU fills ~60 % of the
// forward's run time -- and every lane of a point repeats this arithmetic).
struct Tap1 { int i0, i1; float f, ds; };
__device__ __forceinline__ Tap1 tap1(float coord, int size) {
    Tap1 t;
    float idx;
    unnormalize(coord, size, idx, t.ds);
    const float i0 = floorf(idx);
    t.f = idx - i0;
    t.i0 = (int)i0;
    t.i1 = min(t.i0 + 1, size - 1);                                  // the out-of-range neighbour has weight 0
    return t;
}
__device__ __forceinline__ Bilin make_bilin(const Tap1& tx, const Tap1& ty) {
    Bilin b;
    b.x0 = tx.i0; b.x1 = tx.i1; b.fx = tx.f; b.cx = tx.ds;
    b.y0 = ty.i0; b.y1 = ty.i1; b.fy = ty.f; b.cy = ty.ds;
    return b;
}

// plane pair p of (0,1),(0,2),(0,3),(1,2),(1,3),(2,3): first index -> width axis, second -> height axis
__device__ __forceinline__ void pair_axes(int p, int& a, int& b) {
    const int A[6] = {0, 0, 0, 1, 1, 2}, B[6] = {1, 2, 3, 2, 3, 3};
    a = A[p]; b = B[p];
}

// element offset of tap (x, y), channel c, in a channel-last plane (32-bit: a plane holds < 2^31 floats, checked on the host)
__device__ __forceinline__ uint32_t tap_at(int x, int y, int W, int C, int c) { return ((uint32_t)y * (uint32_t)W + (uint32_t)x) * (uint32_t)C + (uint32_t)c; }

__device__ __forceinline__ float sample(const float* __restrict__ pl, const Bilin& t, int W, int C, int c) {
    const float nw = pl[tap_at(t.x0, t.y0, W, C, c)], ne = pl[tap_at(t.x1, t.y0, W, C, c)];
    const float sw = pl[tap_at(t.x0, t.y1, W, C, c)], se = pl[tap_at(t.x1, t.y1, W, C, c)];
    // grid_sampler_2d: nw * (1-fx)(1-fy) + ne * fx (1-fy) + sw * (1-fx) fy + se * fx fy
    return nw * ((1.f - t.fx) * (1.f - t.fy)) + ne * (t.fx * (1.f - t.fy)) + sw * ((1.f - t.fx) * t.fy) + se * (t.fx * t.fy);
}

// the sample and its slopes d/d(ix), d/d(iy) (already times the border-clip masks) from one read of the four corners
__device__ __forceinline__ float sample_slopes(const float* __restrict__ pl, const Bilin& t, int W, int C, int c, float& dix, float& diy) {
    const float nw = pl[tap_at(t.x0, t.y0, W, C, c)], ne = pl[tap_at(t.x1, t.y0, W, C, c)];
    const float sw = pl[tap_at(t.x0, t.y1, W, C, c)], se = pl[tap_at(t.x1, t.y1, W, C, c)];
    // the clamped neighbour (x1 == x0 at the border) contributes no slope there: its weight is 0 and cx = 0
    dix = ((ne - nw) * (1.f - t.fy) + (se - sw) * t.fy) * t.cx;
    diy = ((sw - nw) * (1.f - t.fx) + (se - ne) * t.fx) * t.cy;
    return nw * ((1.f - t.fx) * (1.f - t.fy)) + ne * (t.fx * (1.f - t.fy)) + sw * ((1.f - t.fx) * t.fy) + se * (t.fx * t.fy);
}

// direct-atomic backward: one row of float atomics per tap (used when the caller gives no visiting order)
__global__ void __launch_bounds__(EMD_BLOCK) k_hexplane_bwd(EmdHexArgs a, EmdHexGrads g) {
    const int C = a.channels, S = a.num_scales;
    const int group = threadIdx.x / C, c = threadIdx.x % C, per_block = EMD_BLOCK / C;
    const long slot = (long)blockIdx.x * per_block + group;
    if (slot >= a.num_points) return;
    const long n = a.order ? (long)a.order[slot] : slot;       // spatially coherent visiting order: neighbours share cache lines
    float q[4];
#pragma unroll
    for (int k = 0; k < 3; k++) q[k] = (a.pts[3 * n + k] - a.aabb[k]) * (2.f / (a.aabb[3 + k] - a.aabb[k])) - 1.f;
    q[3] = a.times[n];
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_dq = g.dL_dpts || g.dL_dtimes;
    for (int s = 0; s < S; s++) {
        float f[6], dix[6], diy[6];
        Bilin t[6];
        Tap1 axis[4];
#pragma unroll
        for (int k = 0; k < 4; k++) axis[k] = tap1(q[k], a.res[s][k]);
#pragma unroll
        for (int p = 0; p < 6; p++) {
            int ax, ay;
            pair_axes(p, ax, ay);
            const int W = a.res[s][ax];
            t[p] = make_bilin(axis[ax], axis[ay]);
            if (want_dq) f[p] = sample_slopes(a.planes[s][p], t[p], W, C, c, dix[p], diy[p]);   // one read of the corners serves both
            else f[p] = sample(a.planes[s][p], t[p], W, C, c);
        }
        {
            const float go = g.dL_dout[(size_t)n * (S * C) + s * C + c];
            float pre[7], suf[7];
            pre[0] = 1.f; suf[6] = 1.f;
#pragma unroll
            for (int p = 0; p < 6; p++) pre[p + 1] = pre[p] * f[p];
#pragma unroll
            for (int p = 5; p >= 0; p--) suf[p] = suf[p + 1] * f[p];
#pragma unroll
            for (int p = 0; p < 6; p++) {
                int ax, ay;
                pair_axes(p, ax, ay);
                const int W = a.res[s][ax];
                const float gi = go * (pre[p] * suf[p + 1]);             // dL / d interp of plane p, channel c
                const Bilin& b = t[p];
                float* gp = g.dL_dplanes[s][p];
                if (gp && gi != 0.f) {
                    atomicAdd(gp + tap_at(b.x0, b.y0, W, C, c), gi * ((1.f - b.fx) * (1.f - b.fy)));
                    atomicAdd(gp + tap_at(b.x1, b.y0, W, C, c), gi * (b.fx * (1.f - b.fy)));
                    atomicAdd(gp + tap_at(b.x0, b.y1, W, C, c), gi * ((1.f - b.fx) * b.fy));
                    atomicAdd(gp + tap_at(b.x1, b.y1, W, C, c), gi * (b.fx * b.fy));
                }
                if (want_dq) {
                    dq[ax] += gi * dix[p];
                    dq[ay] += gi * diy[p];
                }
            }
        }
    }
    if (want_dq) {
        // sum over the C channel lanes of the point, then through normalize_aabb (the time coordinate is used as given)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float v = dq[k];
            for (int off = C >> 1; off; off >>= 1) v += __shfl_xor(v, off, C);
            if (c == 0) {
                if (k < 3) { if (g.dL_dpts) g.dL_dpts[3 * n + k] = v * (2.f / (a.aabb[3 + k] - a.aabb[k])); }
                else if (g.dL_dtimes) g.dL_dtimes[n] = v;
            }
        }
    }
}

// ---- forward, two-phase ---------------------------------------------------------------------------------------------------
// With lane = channel every lane of a point repeats the point's scalar work (un-normalise, clip, floor, tap offsets, bilinear
// weights: ~125 of the ~220 VALU instructions per two points and scale), and the one-phase kernel above is VALU-issue-bound.
// Here a workgroup takes 64 points; per scale, phase A gives one THREAD to each (point, plane) pair, which writes the four tap
// offsets and weights to LDS; phase B is the lane = channel gather, reading them back as broadcast LDS loads: 4 loads, 4 FMAs and
// a few adds per plane.
#define HEX_F2_POINTS 64
__global__ void __launch_bounds__(EMD_BLOCK) k_hexplane_fwd2(EmdHexArgs a) {
    __shared__ uint4 s_off[HEX_F2_POINTS * 6];          // element offsets of the 2 x 2 taps (without the channel)
    __shared__ float4 s_w[HEX_F2_POINTS * 6];           // their bilinear weights
    __shared__ int s_n[HEX_F2_POINTS];
    const int C = a.channels, S = a.num_scales, tid = threadIdx.x;
    const int group = tid / C, c = tid % C, groups = EMD_BLOCK / C;
    const long first = (long)blockIdx.x * HEX_F2_POINTS;
    const int count = (int)min((long)HEX_F2_POINTS, (long)a.num_points - first);
    if (tid < HEX_F2_POINTS) s_n[tid] = tid < count ? (a.order ? a.order[first + tid] : (int)(first + tid)) : 0;
    __syncthreads();
    for (int s = 0; s < S; s++) {
        // phase A: item = (point, plane)
        for (int item = tid; item < HEX_F2_POINTS * 6; item += EMD_BLOCK) {
            const int j = item / 6, p = item - 6 * j;
            if (j >= count) continue;
            const long n = s_n[j];
            int ax, ay;
            pair_axes(p, ax, ay);
            const float qx = (a.pts[3 * n + ax] - a.aabb[ax]) * (2.f / (a.aabb[3 + ax] - a.aabb[ax])) - 1.f;     // ax < 3 always
            const float qy = ay < 3 ? (a.pts[3 * n + ay] - a.aabb[ay]) * (2.f / (a.aabb[3 + ay] - a.aabb[ay])) - 1.f : a.times[n];
            const int W = a.res[s][ax];
            const Tap1 tx = tap1(qx, W), ty = tap1(qy, a.res[s][ay]);
            s_off[item] = make_uint4(tap_at(tx.i0, ty.i0, W, C, 0), tap_at(tx.i1, ty.i0, W, C, 0), tap_at(tx.i0, ty.i1, W, C, 0),
                                     tap_at(tx.i1, ty.i1, W, C, 0));
            // grid_sampler_2d: nw * (1-fx)(1-fy) + ne * fx (1-fy) + sw * (1-fx) fy + se * fx fy
            s_w[item] = make_float4((1.f - tx.f) * (1.f - ty.f), tx.f * (1.f - ty.f), (1.f - tx.f) * ty.f, tx.f * ty.f);
        }
        __syncthreads();
        // phase B: lane = channel
        for (int j = group; j < count; j += groups) {
            float prod = 1.f;
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const uint4 o = s_off[j * 6 + p];
                const float4 w = s_w[j * 6 + p];
                const float* __restrict__ pl = a.planes[s][p] + c;
                prod = prod * (pl[o.x] * w.x + pl[o.y] * w.y + pl[o.z] * w.z + pl[o.w] * w.w);
            }
            a.out[(size_t)s_n[j] * (S * C) + s * C + c] = prod;
        }
        __syncthreads();
    }
}


// ---- backward with in-LDS aggregation -----------------------------------------------------------------------------------
// The plain backward issues one row of float atomics per tap (N x 24 planes x 4 taps); the L2 atomic units retire roughly
// one dword per clock per channel, so at N = 2 M that is ~20 ms no matter how the rows are spread.  When the caller hands the
// points in a spatially coherent order (`order`: e.g. Morton order of the positions), 256 consecutive points fall into a small
// box, and their taps into a few cells of every plane -- above all on the coarse scales and on the three time planes (every
// point of a step carries the same time).  A 512-thread block therefore accumulates its 256 points into LDS windows
// (12 x 12 cells per spatial plane, 32 x 2 per time plane, anchored at the block's smallest tap) and flushes each touched
// cell row to HBM once.  Taps outside a window take the direct global atomic, so any order is correct.  The LDS adds are
// compare-and-swap loops (device_utils.h: the native ds_add_f32 is ~10x slower on gfx950).
// Measured at N = 2 M on the street scene, 4 scales x 32 channels: 29 ms plain -> 9.1 ms (structure 5.3, LDS adds 2.3, the
// remaining global atomics 1.5); geometries 1024/256/16, 512/128/12, 256/256/8 threads/points/window were within 20 %.
#ifndef HEX_AGG_THREADS
#define HEX_AGG_THREADS 512              /* two blocks per CU (78 KB of LDS each at 32 channels): one block's barriers hide behind the other */
#define HEX_AGG_POINTS 256
#define HEX_SW 12                        /* spatial window: HEX_SW x HEX_SW cells */
#endif
#define HEX_TW 32                        /* time-plane window: HEX_TW x 2 cells */
#define HEX_SCELLS (HEX_SW * HEX_SW)
#define HEX_TCELLS (HEX_TW * 2)
#define HEX_WIN_CELLS (3 * HEX_SCELLS + 3 * HEX_TCELLS)
__device__ __forceinline__ void win_shape(int p, int& base, int& wx, int& wy) {
    // planes 0 (xy), 1 (xz), 3 (yz) are spatial; 2 (xt), 4 (yt), 5 (zt) have the time axis as their second (height) axis
    const int B[6] = {0, HEX_SCELLS, 3 * HEX_SCELLS, 2 * HEX_SCELLS, 3 * HEX_SCELLS + HEX_TCELLS, 3 * HEX_SCELLS + 2 * HEX_TCELLS};
    base = B[p];
    const bool time_plane = (p == 2) || (p >= 4);
    wx = time_plane ? HEX_TW : HEX_SW;
    wy = time_plane ? 2 : HEX_SW;
}

template <int C>
__global__ void __launch_bounds__(HEX_AGG_THREADS) k_hexplane_bwd_agg(EmdHexArgs a, EmdHexGrads g, unsigned chunk_stride) {
    __shared__ float win[HEX_WIN_CELLS * C];
    constexpr int GROUPS = HEX_AGG_THREADS / C, ROUNDS = HEX_AGG_POINTS / GROUPS;
    const int tid = threadIdx.x, group = tid / C, c = tid % C, S = a.num_scales;
    // blocks that run side by side take chunks far apart along the curve (stride coprime with the grid): neighbouring chunks
    // flush to the same plane rows, and float atomics to one cache line from many CUs queue up in a single L2 channel
    const long first = (long)(((unsigned long long)blockIdx.x * chunk_stride) % gridDim.x) * HEX_AGG_POINTS;
    __shared__ int qmin[4];
    for (int i = tid; i < HEX_WIN_CELLS * C; i += HEX_AGG_THREADS) win[i] = 0.f;
    if (tid < 4) qmin[tid] = INT_MAX;
    __syncthreads();
    // the block's smallest coordinate per axis (order-preserving integer image of the float): un-normalise, clip and floor are
    // monotone, so on every scale and plane the smallest tap cell of the block is the tap cell of this corner
    for (int r = 0; r < ROUNDS; r++) {
        const long slot = first + r * GROUPS + group;
        if (slot >= a.num_points || c >= 4) continue;
        const long n = a.order ? (long)a.order[slot] : slot;
        const float qv = c < 3 ? (a.pts[3 * n + c] - a.aabb[c]) * (2.f / (a.aabb[3 + c] - a.aabb[c])) - 1.f : a.times[n];
        const int bits = __float_as_int(qv);
        atomicMin(&qmin[c], bits >= 0 ? bits : bits ^ 0x7fffffff);
    }
    __syncthreads();
    float qlo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int bits = qmin[k]; qlo[k] = __int_as_float(bits >= 0 ? bits : bits ^ 0x7fffffff); }
    const bool want_dq = g.dL_dpts || g.dL_dtimes;
    for (int s = 0; s < S; s++) {
        int anc[12];
#pragma unroll
        for (int p = 0; p < 6; p++) {
            int ax, ay;
            pair_axes(p, ax, ay);
            const Bilin b = bilin(qlo[ax], qlo[ay], a.res[s][ax], a.res[s][ay]);
            anc[2 * p] = b.x0; anc[2 * p + 1] = b.y0;
        }
        // phase 2: per point the six samples, the product rule, and the taps into the windows
        for (int r = 0; r < ROUNDS; r++) {
            const long slot = first + r * GROUPS + group;
            if (slot >= a.num_points) continue;
            const long n = a.order ? (long)a.order[slot] : slot;
            float q[4];
#pragma unroll
            for (int k = 0; k < 3; k++) q[k] = (a.pts[3 * n + k] - a.aabb[k]) * (2.f / (a.aabb[3 + k] - a.aabb[k])) - 1.f;
            q[3] = a.times[n];
            float f[6], dix[6], diy[6], dq[4] = {0.f, 0.f, 0.f, 0.f};
            Tap1 axis[4];
#pragma unroll
            for (int k = 0; k < 4; k++) axis[k] = tap1(q[k], a.res[s][k]);
#pragma unroll
            for (int p = 0; p < 6; p++) {
                int ax, ay;
                pair_axes(p, ax, ay);
                const Bilin b = make_bilin(axis[ax], axis[ay]);
                if (want_dq) f[p] = sample_slopes(a.planes[s][p], b, a.res[s][ax], C, c, dix[p], diy[p]);
                else f[p] = sample(a.planes[s][p], b, a.res[s][ax], C, c);
            }
            const float go = g.dL_dout[(size_t)n * (S * C) + s * C + c];
            float pre[7], suf[7];
            pre[0] = 1.f; suf[6] = 1.f;
#pragma unroll
            for (int p = 0; p < 6; p++) pre[p + 1] = pre[p] * f[p];
#pragma unroll
            for (int p = 5; p >= 0; p--) suf[p] = suf[p + 1] * f[p];
#pragma unroll
            for (int p = 0; p < 6; p++) {
                int ax, ay, wbase, wx, wy;
                pair_axes(p, ax, ay);
                win_shape(p, wbase, wx, wy);
                const int W = a.res[s][ax];
                const float gi = go * (pre[p] * suf[p + 1]);
                const Bilin b = make_bilin(axis[ax], axis[ay]);
                float* gp = g.dL_dplanes[s][p];
                if (gp && gi != 0.f) {
                    const int cx0 = b.x0 - anc[2 * p], cy0 = b.y0 - anc[2 * p + 1], cx1 = b.x1 - anc[2 * p], cy1 = b.y1 - anc[2 * p + 1];
                    const float w00 = gi * ((1.f - b.fx) * (1.f - b.fy)), w10 = gi * (b.fx * (1.f - b.fy));
                    const float w01 = gi * ((1.f - b.fx) * b.fy), w11 = gi * (b.fx * b.fy);
                    if ((unsigned)cx1 < (unsigned)wx && (unsigned)cy1 < (unsigned)wy && cx0 >= 0 && cy0 >= 0) {
                        // the whole 2 x 2 footprint lies in the window (the anchor is the block's smallest tap, so c*0 >= 0): one
                        // address, three strides (0 where the neighbour was clamped onto the same cell at the border; its weight is 0)
                        float* w0 = &win[(wbase + cy0 * wx + cx0) * C + c];
                        const int sx = (cx1 - cx0) * C, sy = (cy1 - cy0) * wx * C;
                        lds_add_f32(w0, w00);
                        lds_add_f32(w0 + sx, w10);
                        lds_add_f32(w0 + sy, w01);
                        lds_add_f32(w0 + sy + sx, w11);
                    } else {
                        const int xs[2] = {b.x0, b.x1}, ys[2] = {b.y0, b.y1}, cxs[2] = {cx0, cx1}, cys[2] = {cy0, cy1};
                        const float ws[4] = {w00, w10, w01, w11};
#pragma unroll
                        for (int j = 0; j < 2; j++)
#pragma unroll
                            for (int i = 0; i < 2; i++) {
                                if ((unsigned)cxs[i] < (unsigned)wx && (unsigned)cys[j] < (unsigned)wy) lds_add_f32(&win[(wbase + cys[j] * wx + cxs[i]) * C + c], ws[2 * j + i]);
                                else atomicAdd(gp + tap_at(xs[i], ys[j], W, C, c), ws[2 * j + i]);
                            }
                    }
                }
                if (want_dq) {
                    dq[ax] += gi * dix[p];
                    dq[ay] += gi * diy[p];
                }
            }
            if (want_dq) {
                // a point belongs to one thread group of one block: plain read-modify-write across the scales
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float v = dq[k];
                    for (int off = C >> 1; off; off >>= 1) v += __shfl_xor(v, off, C);
                    if (c == 0) {
                        if (k < 3) {
                            if (g.dL_dpts) {
                                const float add = v * (2.f / (a.aabb[3 + k] - a.aabb[k]));
                                g.dL_dpts[3 * n + k] = s ? g.dL_dpts[3 * n + k] + add : add;
                            }
                        } else if (g.dL_dtimes) g.dL_dtimes[n] = s ? g.dL_dtimes[n] + v : v;
                    }
                }
            }
        }
        __syncthreads();
        // phase 3: every touched cell row goes to HBM once; the windows are left clean for the next scale
        for (int cell = group; cell < HEX_WIN_CELLS; cell += GROUPS) {
            const float v = win[cell * C + c];
            if (v == 0.f) continue;
            win[cell * C + c] = 0.f;
            const int p = cell < 3 * HEX_SCELLS ? (cell < HEX_SCELLS ? 0 : (cell < 2 * HEX_SCELLS ? 1 : 3))
                                                : (cell < 3 * HEX_SCELLS + HEX_TCELLS ? 2 : (cell < 3 * HEX_SCELLS + 2 * HEX_TCELLS ? 4 : 5));
            int ax, ay, wbase, wx, wy;
            pair_axes(p, ax, ay);
            win_shape(p, wbase, wx, wy);
            const int local = cell - wbase, x = anc[2 * p] + local % wx, y = anc[2 * p + 1] + local / wx;
            atomicAdd(g.dL_dplanes[s][p] + tap_at(x, y, a.res[s][ax], C, c), v);   // (only cells a tap reached are non-zero)
        }
        __syncthreads();
    }
}

template <int C>
void launch_bwd_agg(const EmdHexArgs* a, const EmdHexGrads* g, hipStream_t st) {
    const unsigned blocks = (unsigned)((a->num_points + HEX_AGG_POINTS - 1) / HEX_AGG_POINTS);
    unsigned stride = 7919u % blocks;
    auto gcd = [](unsigned x, unsigned y) { while (y) { unsigned t = x % y; x = y; y = t; } return x; };
    while (stride == 0 || gcd(stride, blocks) != 1) stride++;          // a bijection on [0, blocks)
    hipLaunchKernelGGL(k_hexplane_bwd_agg<C>, dim3(blocks), dim3(HEX_AGG_THREADS), 0, st, *a, *g, stride);
}

int check_hex(const EmdHexArgs* a, const char* who) {
    if (!a) { emd_set_error("%s: null args", who); return EMD_ERR_INVALID; }
    const int C = a->channels;
    if (a->num_points < 0 || a->num_scales < 1 || a->num_scales > EMD_HEX_MAX_SCALES) { emd_set_error("%s: bad sizes", who); return EMD_ERR_INVALID; }
    if (C < 1 || C > 64 || (C & (C - 1))) { emd_set_error("%s: channels must be a power of two <= 64, got %d", who, C); return EMD_ERR_INVALID; }
    if (a->num_points > 0 && (!a->pts || !a->times)) { emd_set_error("%s: pts / times must not be null", who); return EMD_ERR_INVALID; }
    for (int s = 0; s < a->num_scales; s++)
        for (int p = 0; p < 6; p++) {
            if (!a->planes[s][p]) { emd_set_error("%s: plane %d of scale %d is null", who, p, s); return EMD_ERR_INVALID; }
            if (a->res[s][0] < 1 || a->res[s][1] < 1 || a->res[s][2] < 1 || a->res[s][3] < 1) { emd_set_error("%s: bad resolution", who); return EMD_ERR_INVALID; }
            const int A_[6] = {0, 0, 0, 1, 1, 2}, B_[6] = {1, 2, 3, 2, 3, 3};
            if ((int64_t)a->res[s][A_[p]] * a->res[s][B_[p]] * C >= ((int64_t)1 << 31)) {       // the kernels index a plane with 32 bits
                emd_set_error("%s: plane %d of scale %d holds 2^31 floats or more", who, p, s); return EMD_ERR_INVALID;
            }
        }
    return EMD_OK;
}

}  // namespace

extern "C" int emd_hexplane_forward(const EmdHexArgs* a, void* hip_stream) {
    int rc = check_hex(a, "hexplane_forward");
    if (rc) return rc;
    if (!a->out) { emd_set_error("hexplane_forward: null output"); return EMD_ERR_INVALID; }
    if (a->num_points == 0) return EMD_OK;
    hipLaunchKernelGGL(k_hexplane_fwd2, dim3((unsigned)((a->num_points + HEX_F2_POINTS - 1) / HEX_F2_POINTS)), dim3(EMD_BLOCK), 0,
                       (hipStream_t)hip_stream, *a);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_hexplane_backward(const EmdHexArgs* a, const EmdHexGrads* g, void* hip_stream) {
    int rc = check_hex(a, "hexplane_backward");
    if (rc) return rc;
    if (!g || !g->dL_dout) { emd_set_error("hexplane_backward: null gradient"); return EMD_ERR_INVALID; }
    if (a->num_points == 0) return EMD_OK;
    // a visiting order promises spatial coherence: aggregate in LDS (windows are sized for C <= 32)
    if (a->order && a->channels == 32) launch_bwd_agg<32>(a, g, (hipStream_t)hip_stream);
    else if (a->order && a->channels == 16) launch_bwd_agg<16>(a, g, (hipStream_t)hip_stream);
    else {
        const int per_block = EMD_BLOCK / a->channels;
        hipLaunchKernelGGL(k_hexplane_bwd, dim3((unsigned)((a->num_points + per_block - 1) / per_block)), dim3(EMD_BLOCK), 0,
                           (hipStream_t)hip_stream, *a, *g);
    }
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
