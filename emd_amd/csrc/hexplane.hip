// hexplane.hip -- fused multi-scale HexPlane feature lookup, forward and backward.
//
// Replaces HexPlaneField.get_density / interpolate_ms_features (S3Gaussian/scene/hexplane.py:18-110,150-183): per scale six
// F.grid_sample launches (bilinear, align_corners, border padding) + five products + a concat, and their backward
// (six grid_sampler_2d_backward launches that scatter channel-FIRST: one float atomic per channel per tap, each to a
// different cache line).  Here the planes are handed over CHANNEL-LAST ([res_h][res_w][C]) and a point is owned by C
// consecutive lanes (lane = channel): every tap is one coalesced C x 4-byte read, the product over the six planes and
// the concat over scales happen in registers, and the backward scatters C consecutive floats per tap (the row shape
// float atomics like).  One launch forward, one backward, for all scales.
// Kernels: k_hexplane_fwd4 / fwd2 (two-phase forward, lane = four channels / one channel, at the L2 -> CU gather rate), k_hexplane_bwd_agg (backward that aggregates
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
// (arithmetic, not a table: with a lane-varying p -- the staging phases, item = (point, plane) -- a table is a load from constant memory, and
// the a.res[s][axis] behind it a second, dependent one from the kernel arguments: two to three HBM-latency round trips per staging call)
__device__ __forceinline__ void pair_axes(int p, int& a, int& b) {
    a = (p >= 3) + (p >= 5);
    b = p < 3 ? p + 1 : (p == 3 ? 2 : 3);
}

__device__ __forceinline__ int sel4i(int v0, int v1, int v2, int v3, int k) { return k == 0 ? v0 : (k == 1 ? v1 : (k == 2 ? v2 : v3)); }
__device__ __forceinline__ float sel4f(float v0, float v1, float v2, float v3, int k) { return k == 0 ? v0 : (k == 1 ? v1 : (k == 2 ? v2 : v3)); }

// element offset of tap (x, y), channel c, in a channel-last plane (32-bit: a plane holds < 2^30 floats, checked on the host)
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
    q[3] = a.times[a.times_broadcast ? 0 : n];
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_dq = g.dL_dpts || g.dL_dtimes || g.dL_dtime_sum;
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
            // one broadcast timestamp: its gradient is the sum over ALL points -- summed over the wave first (the points' channel-0 lanes
            // hold the per-point values), one atomic per wave instead of one same-address atomic per point (2 M serialised atomics)
            if (k == 3 && !g.dL_dtimes && g.dL_dtime_sum) {
                if (__ballot(true) == ~0ull) {                        // (every lane of the wave carries a point: all but the last wave of the grid)
                    float w = c == 0 ? v : 0.f;
                    for (int off = 32; off; off >>= 1) w += __shfl_xor(w, off);
                    if ((threadIdx.x & 63) == 0) atomicAdd(g.dL_dtime_sum, w);
                } else if (c == 0) atomicAdd(g.dL_dtime_sum, v);
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
            const float qy = ay < 3 ? (a.pts[3 * n + ay] - a.aabb[ay]) * (2.f / (a.aabb[3 + ay] - a.aabb[ay])) - 1.f : a.times[a.times_broadcast ? 0 : n];
            const int W = sel4i(a.res[s][0], a.res[s][1], a.res[s][2], a.res[s][3], ax);
            const Tap1 tx = tap1(qx, W), ty = tap1(qy, sel4i(a.res[s][0], a.res[s][1], a.res[s][2], a.res[s][3], ay));
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
                // uniform plane base + 32-bit byte offsets (the saddr form: no 64-bit address arithmetic per tap; planes of 2^30 floats or more
                // are refused by check_hex)
                const char* __restrict__ pl = (const char*)a.planes[s][p];
                const uint32_t cb = (uint32_t)c << 2;
                prod = prod * (*(const float*)(pl + ((o.x << 2) + cb)) * w.x + *(const float*)(pl + ((o.y << 2) + cb)) * w.y +
                               *(const float*)(pl + ((o.z << 2) + cb)) * w.z + *(const float*)(pl + ((o.w << 2) + cb)) * w.w);
            }
            a.out[(size_t)s_n[j] * (S * C) + s * C + c] = prod;
        }
        __syncthreads();
    }
}


// ---- forward, lane = FOUR channels ---------------------------------------------------------------------------------------------
// The same two phases with C / 4 lanes per point: a tap row is read as 16 bytes per lane (a quarter of the vector-memory instructions and of
// the broadcast LDS reads per point), the output row written the same way.  A workgroup takes 128 points (768 (point, plane) items = three
// full rounds of phase A); the normalised coordinates are formed once per point, not once per scale and plane.
// Chunk -> XCD: workgroups are dealt round-robin over the 8 XCDs, so workgroup b takes chunk (b % 8) * ceil(chunks / 8) + b / 8: every XCD
// walks ONE contiguous eighth of the visiting order, and with a spatially coherent order its L2 holds an eighth of the box's taps instead
// of every XCD streaming every plane.
#define HEX_F4_POINTS 128
__device__ __forceinline__ long xcd_chunk(unsigned b, unsigned chunks) {
    const unsigned per = (chunks + 7u) / 8u, chunk = (b % 8u) * per + b / 8u;
    return (b / 8u < per && chunk < chunks) ? (long)chunk : -1;
}
// 1-D tables of the three time planes for a call whose points share one time (EmdHexArgs.time_tables): row ix of (scale, axis) =
// (1 - ft) plane[it0][ix] + ft plane[it1][ix].  Element offset of that table: C * (sum of res_x + res_y + res_z over the scales before + the axes before).
__device__ __forceinline__ uint32_t time_table_offset(const EmdHexArgs& a, int s, int axis) {
    uint32_t off = 0;
    for (int s2 = 0; s2 < s; s2++) off += (uint32_t)(a.res[s2][0] + a.res[s2][1] + a.res[s2][2]);
    for (int k = 0; k < axis; k++) off += (uint32_t)a.res[s][k];
    return off * (uint32_t)a.channels;
}
__global__ void __launch_bounds__(EMD_BLOCK) k_hexplane_time_tables(EmdHexArgs a) {
    const int C = a.channels, S = a.num_scales;
    long item = (long)blockIdx.x * EMD_BLOCK + threadIdx.x;            // (row of all tables, channel)
    const int c = (int)(item % C);
    long row = item / C;
    for (int s = 0; s < S; s++)
        for (int axis = 0; axis < 3; axis++) {
            const int W = a.res[s][axis];
            if (row < W) {
                const Tap1 tt = tap1(a.times[0], a.res[s][3]);
                const float* pl = a.planes[s][axis == 0 ? 2 : (axis == 1 ? 4 : 5)];     // the planes that pair x, y, z with the time
                const float v0 = pl[tap_at((int)row, tt.i0, W, C, c)], v1 = pl[tap_at((int)row, tt.i1, W, C, c)];
                a.time_tables[time_table_offset(a, s, axis) + (uint32_t)row * C + c] = v0 * (1.f - tt.f) + v1 * tt.f;
                return;
            }
            row -= W;
        }
}

#ifndef HEX_NT
#define HEX_NT 1          /* round 5: the [N, 128] feature / gradient streams bypass the caches, the planes stay (fine-stage step 13.87 -> 13.72 ms; the same hint on the
                             MLP kernels' tile loads costs 1.1 ms: 16-byte pieces of rows) */
#endif
typedef float hex_v4f __attribute__((ext_vector_type(4)));
template <int C>
__global__ void __launch_bounds__(EMD_BLOCK) k_hexplane_fwd4(EmdHexArgs a, unsigned chunks) {
    constexpr int LPP = C / 4, GROUPS = EMD_BLOCK / LPP;
    __shared__ uint4 s_off[HEX_F4_POINTS * 6];          // BYTE offsets of the 2 x 2 taps (without the channel)
    __shared__ float4 s_w[HEX_F4_POINTS * 6];           // their bilinear weights
    __shared__ float4 s_q[HEX_F4_POINTS];               // normalised x, y, z and the time
    __shared__ int s_n[HEX_F4_POINTS];
    const int S = a.num_scales, tid = threadIdx.x;
    const int group = tid / LPP, c4 = (tid % LPP) * 4;
    const long chunk = xcd_chunk(blockIdx.x, chunks);
    if (chunk < 0) return;
    const long first = chunk * HEX_F4_POINTS;
    const int count = (int)min((long)HEX_F4_POINTS, (long)a.num_points - first);
    const bool uni = a.time_tables != nullptr;          // one time for all points: the time planes are read from their 1-D tables, two taps
    if (tid < HEX_F4_POINTS) {
        const long n = tid < count ? (a.order ? (long)a.order[first + tid] : first + tid) : 0;
        s_n[tid] = (int)n;
        float4 q;
        q.x = (a.pts[3 * n] - a.aabb[0]) * (2.f / (a.aabb[3] - a.aabb[0])) - 1.f;
        q.y = (a.pts[3 * n + 1] - a.aabb[1]) * (2.f / (a.aabb[4] - a.aabb[1])) - 1.f;
        q.z = (a.pts[3 * n + 2] - a.aabb[2]) * (2.f / (a.aabb[5] - a.aabb[2])) - 1.f;
        q.w = a.times[a.times_broadcast ? 0 : n];
        s_q[tid] = q;
    }
    __syncthreads();
    for (int s = 0; s < S; s++) {
        // phase A: item = (point, plane).  The scale's four resolutions as scalars, picked per lane by selects (no lane-indexed argument reads)
        const int r0 = a.res[s][0], r1 = a.res[s][1], r2 = a.res[s][2], r3 = a.res[s][3];
        const uint32_t tb0 = uni ? time_table_offset(a, s, 0) : 0u;
        for (int item = tid; item < HEX_F4_POINTS * 6; item += EMD_BLOCK) {
            const int j = item / 6, p = item - 6 * j;
            if (j >= count) continue;
            int ax, ay;
            pair_axes(p, ax, ay);
            const float4 q = s_q[j];
            const int W = sel4i(r0, r1, r2, r3, ax);
            const Tap1 tx = tap1(sel4f(q.x, q.y, q.z, q.w, ax), W);
            if (uni && ay == 3) {                       // two taps of the (scale, axis) table
                const uint32_t tb = tb0 + (uint32_t)((ax > 0 ? r0 : 0) + (ax > 1 ? r1 : 0)) * (uint32_t)a.channels;
                s_off[item] = make_uint4((tb + (uint32_t)tx.i0 * C) << 2, (tb + (uint32_t)tx.i1 * C) << 2, 0u, 0u);
                s_w[item] = make_float4(1.f - tx.f, tx.f, 0.f, 0.f);
                continue;
            }
            const Tap1 ty = tap1(sel4f(q.x, q.y, q.z, q.w, ay), sel4i(r0, r1, r2, r3, ay));
            s_off[item] = make_uint4(tap_at(tx.i0, ty.i0, W, C, 0) << 2, tap_at(tx.i1, ty.i0, W, C, 0) << 2, tap_at(tx.i0, ty.i1, W, C, 0) << 2,
                                     tap_at(tx.i1, ty.i1, W, C, 0) << 2);
            // grid_sampler_2d: nw * (1-fx)(1-fy) + ne * fx (1-fy) + sw * (1-fx) fy + se * fx fy
            s_w[item] = make_float4((1.f - tx.f) * (1.f - ty.f), tx.f * (1.f - ty.f), (1.f - tx.f) * ty.f, tx.f * ty.f);
        }
        __syncthreads();
        // phase B: lane = four channels
        for (int j = group; j < count; j += GROUPS) {
            float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
            const uint32_t cb = (uint32_t)c4 << 2;
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const uint4 o = s_off[j * 6 + p];
                const float4 w = s_w[j * 6 + p];
                if (uni && (p == 2 || p >= 4)) {            // (uniform branch; p is a constant of the unrolled loop)
                    const char* __restrict__ tb = (const char*)a.time_tables;
                    const float4 t0 = *(const float4*)(tb + (o.x + cb)), t1 = *(const float4*)(tb + (o.y + cb));
                    prod.x = prod.x * (t0.x * w.x + t1.x * w.y);
                    prod.y = prod.y * (t0.y * w.x + t1.y * w.y);
                    prod.z = prod.z * (t0.z * w.x + t1.z * w.y);
                    prod.w = prod.w * (t0.w * w.x + t1.w * w.y);
                    continue;
                }
                const char* __restrict__ pl = (const char*)a.planes[s][p];
                const float4 nw = *(const float4*)(pl + (o.x + cb)), ne = *(const float4*)(pl + (o.y + cb)), sw = *(const float4*)(pl + (o.z + cb)),
                             se = *(const float4*)(pl + (o.w + cb));
                // (the evaluation order of k_hexplane_fwd2, per channel)
                prod.x = prod.x * (nw.x * w.x + ne.x * w.y + sw.x * w.z + se.x * w.w);
                prod.y = prod.y * (nw.y * w.x + ne.y * w.y + sw.y * w.z + se.y * w.w);
                prod.z = prod.z * (nw.z * w.x + ne.z * w.y + sw.z * w.z + se.z * w.w);
                prod.w = prod.w * (nw.w * w.x + ne.w * w.y + sw.w * w.z + se.w * w.w);
            }
#if HEX_NT
            __builtin_nontemporal_store((hex_v4f){prod.x, prod.y, prod.z, prod.w}, reinterpret_cast<hex_v4f*>(a.out + ((size_t)s_n[j] * (S * C) + s * C + c4)));
#else
            *(float4*)(a.out + ((size_t)s_n[j] * (S * C) + s * C + c4)) = prod;
#endif
        }
        __syncthreads();
    }
}

// ---- backward with in-LDS aggregation -----------------------------------------------------------------------------------
// The plain backward issues one row of float atomics per tap (N x 24 planes x 4 taps).  A CU issues a 64-lane global float atomic
// in ~117 clocks whatever the rows, their locality or the memory scope (profiles/r03_global_atomic_microbench.txt: 10.4 G rows of
// 128 bytes per second chip-wide, against 57 G rows/s of plain stores), and while it does, the gathers of its other waves queue
// behind the atomics in the same vector-memory pipe: at N = 2 M that is ~20 ms no matter how the rows are spread.  So the lever is
// issuing fewer rows.  When the caller hands the points in a spatially coherent order (`order`: e.g. Morton order of the
// positions), 256 consecutive points fall into a small box, and their taps into a few cells of every plane -- above all on the
// coarse scales and on the three time planes.  A block therefore accumulates its 256 points into LDS windows (12 x 12 cells per
// spatial plane, 32 cells per time plane, anchored at the block's smallest tap) and flushes each touched cell row to HBM once.
// Taps outside a window take the direct global atomic, so any order is correct.
//
// Round 3 (the round-2 kernel: 512 threads, fp32 windows added to with compare-and-swap loops, every channel lane repeating the
// per-point scalar work, 11.1 ms at 2 M uniform points; this one 7.7 ms):
//  * the windows are fp64 and added to with ds_add_f64 -- the one native LDS float add that runs at rate on gfx950 (8.9 clocks per
//    wave64 instruction against 16.9 for the fp32 compare-and-swap loop and 193 for ds_add_f32: profiles/r03_lds_atomic_microbench.txt),
//    returns nothing, so no wave waits on it, and resolves same-cell collisions in hardware.  One 1024-thread block per CU owns 152 KB;
//  * phase 0 normalises the block's points once into LDS (s_q) and finds the bounding box of the block;
//  * the scalar part of a tap -- un-normalise, clip, floor, offset, strides, window address -- is done by one LANE per (point,
//    plane) pair and left in LDS (32 bytes per pair); the lane = channel part reads it back as broadcast loads: per plane 2 LDS
//    reads, 4 gathers (uniform base + 32-bit byte offset), the sample and its two slopes, 4 adds;
//  * the waves of a block run on their own between the flushes, as a software pipeline (see the loop);
//  * when every point of the block carries the same time (one frame per step: always, in training) the three time planes are
//    accumulated as x-MARGINALS: two adds per point instead of four, into 32 cells per plane, and the flush spreads a cell over the
//    two time rows with the block's (1 - ft, ft).  Blocks with mixed times use the same memory as 16 x 2 windows;
//  * dL/dpts and dL/dtimes are summed over the scales in LDS and written once.
// What bounds it (round 4, 2 M uniform points, builds with one part compiled out): without the tap gathers 4.04 -> 3.80 ms, without the
// deferred-row stores 3.84, without the LDS adds and the flush atomics 3.11: the gathers are hidden, the atomics cost 0.9 ms, the rest is
// instruction issue (520 vector + 336 scalar + 81 LDS instructions per iteration of two points).  Two channels per lane (packed fp32, 45 %
// fewer vector instructions per point) was built and measured SLOWER, 5.2 ms: 48 tap registers in flight per iteration, spills at the
// 128-VGPR cap or three waves per SIMD without them -- four un-spilled waves per SIMD are worth more than the instruction count (DESIGN.md
// section 8, round 4).  Built without SLP vectorisation (Makefile): the packer cost 129 v_mov per iteration and 8 spilled registers.
//
// Round 5, late (3.67 -> 2.93 ms without / 3.05 ms with the time-gradient sum; DESIGN.md section 8 "HexPlane, round 5 late" has the A/B of every step):
//  * no lane-indexed reads of the kernel arguments (pair_axes as arithmetic, the scale's resolutions / anchors / gradient pointers as selects of
//    scalars): `a.res[s][axis]` with a lane-varying axis is a global load, and the staging call had two of them, dependent, behind the gathers' wait;
//  * 16 iterations per wave between two flushes (first as 512 points per 1024-thread workgroup), four iterations per staging call into a ring of five iteration slots,
//    one staged row per (point, plane): (fx, fy, slope x, slope y) -- the sample as nested interpolations, whose differences are the slopes;
//  * the next scale's first rows are staged, and its first gathers requested, before the block meets at the flush of the current scale;
//  * a point's three plane positions and its index share one LDS row (s_pn): one read in front of the three deferred-row stores.
//  * TWO workgroups of 512 threads per CU (256 points, 7 x 7 spatial windows, 80 KB each) instead of one of 1024: one computes while the other meets at
//    its flush (3.69 -> 3.49 ms for the whole backward at 2 M points).
//  The numbers in the paragraphs above (12 x 12 / 32-cell windows, two iterations per staging call, one 1024-thread block owning 152 KB) describe the
//  rounds they are dated with; the geometry now is HEX_AGG_THREADS / HEX_AGG_POINTS / HEX_SW / HEX_TW below.
#ifndef HEX_STAGE_BYTES
#define HEX_STAGE_BYTES 1  /* the staged tap offset, row stride and dx in BYTES: four address instructions per plane in the gathers instead of seven (4.31 -> 4.25 ms) */
#endif
#ifndef HEX_NT_ROWS
#define HEX_NT_ROWS 1     /* the deferred rows (2.3 GB written by the main kernel, read once by the per-plane pass) bypass the caches: they evicted the planes the gathers
                              and the cell-row atomics work on -- backward 3.55 -> 3.39 ms at 2 M points, fine-stage step 11.08 -> 10.99 ms */
#endif
#ifndef HEX_STAGE_ITERS
#define HEX_STAGE_ITERS 0  /* iterations staged per stage() call: 0 = as many as fit the wave's lanes (4 at 32 channels, 2 at 16) */
#endif
#ifndef HEX_NEXT_SCALE_EARLY
#define HEX_NEXT_SCALE_EARLY 1
#endif
#ifndef HEX_AGG_THREADS
#define HEX_AGG_THREADS 512              /* round 5 (end): TWO workgroups of 512 threads per CU (80 KB of LDS each at 32 channels; four waves per SIMD as before) instead of one of
                                            1024 -- while one meets at its flush (barrier, atomics, the next scale's first round trip) the other one computes, and a barrier holds
                                            8 waves instead of 16: backward 3.69 -> 3.49 ms at 2 M points, fine-stage step 11.50 -> 11.25 ms */
#define HEX_AGG_POINTS 256               /* points per workgroup: 16 iterations per wave between two flushes (one workgroup of 1024 threads: 512 points were worth 0.16 ms over 256) */
#define HEX_SW 7                         /* spatial window: HEX_SW x HEX_SW cells (finer scales go through the per-plane pass) */
#endif
#ifndef HEX_TW
#define HEX_TW 19                        /* time-plane window of a scale with spatial windows: HEX_TW marginal cells, or (HEX_TW / 2) x 2 cells; a deferred scale gives each time
                                            plane a third of all cells (68: a run of 256 points spans 35 - 50 cells of a 512-cell axis) */
#endif
#define HEX_SCELLS (HEX_SW * HEX_SW)
#define HEX_WIN_CELLS (3 * HEX_SCELLS + 3 * HEX_TW)

__device__ __forceinline__ int order_key(float v) { const int b = __float_as_int(v); return b >= 0 ? b : b ^ 0x7fffffff; }   // monotone int image
__device__ __forceinline__ float key_value(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// DT: 0 = no time gradient (the time planes' slope along t, a fourth lane sum and its adds drop out); 1 = dL/dtimes per point;
// 2 = only its SUM over the points (EmdHexGrads.dL_dtime_sum: one broadcast timestamp, S3Gaussian's time_offset parameter) -- every lane keeps
// its channel's partial sum over all its points and scales in a register, one add per iteration instead of a lane sum, and a workgroup adds
// one float atomically at the end
template <int C, int DT>
__global__ void __launch_bounds__(HEX_AGG_THREADS, 4) k_hexplane_bwd_agg(EmdHexArgs a, EmdHexGrads g, unsigned chunk_stride) {
    constexpr int WAVES = HEX_AGG_THREADS / 64, GW = 64 / C, PER_WAVE = HEX_AGG_POINTS / WAVES, ITERS = PER_WAVE / GW, GROUPS = HEX_AGG_THREADS / C;
    // staging: one stage() call prepares the rows of SK iterations (lane = (iteration, point, plane): 48 of 64 lanes busy at 32 channels) into a ring of
    // SK + 1 iteration slots -- when the call for the next SK iterations runs, at the top of the last iteration of the current group, that
    // iteration's rows are the only live ones
    constexpr int IROWS = GW * 6, SK = HEX_STAGE_ITERS ? HEX_STAGE_ITERS : (64 / IROWS >= 4 ? 4 : 2), RING = SK + 1, SROWS = RING * IROWS;
    static_assert(SK * IROWS <= 64 && C <= 32 && PER_WAVE % GW == 0 && ITERS % SK == 0, "staging geometry");
    __shared__ double win[HEX_WIN_CELLS * C];           // fp64 cells: ds_add_f64 is the one native LDS float add that runs at rate on gfx950
    __shared__ uint4 s_a[WAVES * SROWS];                // per wave, a ring of (point, plane) rows: BYTE offset of tap (x0, y0), row stride to y1 in bytes, window address, dx * 4 | sy << 8
    __shared__ float4 s_w[WAVES * SROWS];               //   the fractions fx, fy and the slopes d(ix)/d(coord), d(iy)/d(coord) (0 where the coordinate was clipped)
    __shared__ float4 s_q[HEX_AGG_POINTS];              // box-normalised x, y, z and the time of every point of the block
    __shared__ float4 s_dq[HEX_AGG_POINTS];             // dL/d(those), summed over the scales
    __shared__ int4 s_pn[HEX_AGG_POINTS];               // its position in the visiting orders of the planes xy, xz, yz (deferred scales: ONE read in front of the three row stores) and its index (-1 past the end)
    __shared__ int qmin[4], qmax[4];
    const int tid = threadIdx.x, group = tid / C, c = tid % C, S = a.num_scales, lane = tid & 63, wave = tid >> 6, gw = lane / C;
    // blocks that run side by side take chunks far apart along the curve (stride coprime with the grid): neighbouring chunks
    // flush to the same plane rows, and float atomics to one cache line from many CUs queue up in a single L2 channel
    const long first = (long)(((unsigned long long)blockIdx.x * chunk_stride) % gridDim.x) * HEX_AGG_POINTS;
    for (int i = tid; i < HEX_WIN_CELLS * C; i += HEX_AGG_THREADS) win[i] = 0.0;
    if (tid < 4) { qmin[tid] = INT_MAX; qmax[tid] = INT_MIN; }
    __syncthreads();
    // phase 0: item = (point, axis).  Un-normalise, clip and floor are monotone, so on every scale and plane the smallest tap cell
    // of the block is the tap cell of its smallest coordinates (order-preserving integer image of the floats)
    for (int item = tid; item < HEX_AGG_POINTS * 4; item += HEX_AGG_THREADS) {
        const int j = item >> 2, k = item & 3;
        const long slot = first + j;
        const bool live = slot < a.num_points;
        const int n = live ? (a.order ? a.order[slot] : (int)slot) : -1, ns = live ? n : 0;
        // one workgroup per CU: nothing else hides this prologue.  The coordinate and the plane position travel together, unconditionally (point 0
        // stands in past the end), from addresses picked by selects: a lane-indexed read of the kernel arguments (a.aabb[k], g.pos2d[k - 1])
        // is itself a load, with the real one waiting behind it
        const float raw = *(k < 3 ? a.pts + (3 * (long)ns + k) : a.times + (a.times_broadcast ? 0 : ns));
        int pos = 0;
        if (g.defer_mask) pos = (k <= 1 ? g.pos2d[0] : (k == 2 ? g.pos2d[1] : g.pos2d[2]))[ns];
        const float lo = sel4f(a.aabb[0], a.aabb[1], a.aabb[2], 0.f, k), hi = sel4f(a.aabb[3], a.aabb[4], a.aabb[5], 0.f, k);
        float qv = 0.f;
        if (live) {
            qv = k < 3 ? (raw - lo) * (2.f / (hi - lo)) - 1.f : raw;
            const int key = order_key(qv);
            atomicMin(&qmin[k], key);
            atomicMax(&qmax[k], key);
        }
        ((float*)&s_q[j])[k] = qv;
        ((float*)&s_dq[j])[k] = 0.f;
        if (k == 0) s_pn[j].w = n;
        else if (g.defer_mask && live) ((int*)&s_pn[j])[k - 1] = pos;
    }
    __syncthreads();
    if (g.defer_mask) for (int t = tid; t < HEX_AGG_POINTS * 3; t += HEX_AGG_THREADS) {
        // the per-plane pass reads the two normalised coordinates of a point at its position in the plane's order (behind the rows)
        const int j = t / 3, pidx = t - 3 * j;
        if (s_pn[j].w >= 0) {
            const float4 q = s_q[j];
            float2* defer_q = (float2*)(g.defer_rows + (size_t)__builtin_popcount(g.defer_mask) * 3 * (size_t)a.num_points * C) + (size_t)pidx * a.num_points;
            defer_q[((const int*)&s_pn[j])[pidx]] = pidx == 0 ? make_float2(q.x, q.y) : (pidx == 1 ? make_float2(q.x, q.z) : make_float2(q.y, q.z));
        }
    }
    float qlo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) qlo[k] = key_value(qmin[k]);
    const bool tuni = qmin[3] == qmax[3];               // one time for the whole block: time planes as marginals
    const bool want_dq = g.dL_dpts || g.dL_dtimes || g.dL_dtime_sum;
    float t_acc = 0.f;                                  // (DT == 2)
    const int twy = tuni ? 1 : 2;
    const int sb = wave * SROWS;                        // the wave's staging rows
    // ---- the scale the waves are staging and gathering for (`cs`): its resolutions, the block's anchor cells, its window layout.  It runs ONE
    // stage() + gather() ahead of the scale `s` being scattered and flushed: the first gathers of scale s + 1 are requested before the block
    // meets at the flush of scale s (HEX_NEXT_SCALE_EARLY), so their latency passes under the barrier and the flush instead of in front of an
    // idle CU -- after a flush all sixteen waves sit in the same phase and nothing hides a round trip
    int cs = 0, anc[4], rs[4], tw = 0, tbase = 0, twx = 0;
    bool deferred = false;
    size_t defer_base = 0;
    auto enter_scale = [&](int ss) {
        cs = ss;
#pragma unroll
        for (int k = 0; k < 4; k++) { rs[k] = a.res[ss][k]; anc[k] = tap1(qlo[k], rs[k]).i0; }
        // deferred scale: the rows of the spatial planes go to the per-plane pass (k_hexplane_bwd_plane) instead of the windows
        deferred = (g.defer_mask >> ss) & 1u;
        defer_base = (size_t)__builtin_popcount(g.defer_mask & ((1u << ss) - 1u)) * 3u;
        // a deferred scale does not use the spatial windows: its three time windows take the whole memory (148 cells each instead of 48 -- on
        // the fine scales, the deferred ones, a run spans 35 - 50 cells of an axis and 18 % of its x taps left the 48-cell window, each such
        // tap four partly filled atomic instructions)
        tw = deferred ? HEX_WIN_CELLS / 3 : HEX_TW; tbase = deferred ? 0 : 3 * HEX_SCELLS; twx = tuni ? tw : tw / 2;
    };
#define HEX_PL(p) ((const char*)a.planes[cs][p])
#define HEX_GP(p) (g.dL_dplanes[s][p])
        // The waves of the block run on their own from here to the flush, GW points (one per C-lane group) per iteration, as a
        // software pipeline:  stage(it + 1) | wait for the taps of `it` | sample, slopes, product rule | issue the gathers of it + 1 |
        // scatter the rows of `it`.  No block barrier in the loop, so the waves drift apart and one wave's gathers overlap another's
        // arithmetic and a third's adds; and within a wave the next gathers are in flight while the adds and atomics are issued
        // (with block-wide batches every wave sat in the same phase at the same time: 70 % of a wave's life in s_waitcnt).
        // ---- staging, item = (point, plane), lanes 0 .. 6 GW - 1 of the wave: un-normalise, clip, floor, offsets, window address
        auto stage = [&](int it) {                      // `it` a multiple of SK: rows of iterations it .. it + SK - 1 into their ring slots
            if (lane < SK * IROWS) {
                const int k2 = lane / IROWS, rem = lane - k2 * IROWS, jj = rem / 6, p = rem - 6 * jj;
                const int pt = wave * PER_WAVE + (it + k2) * GW + jj, n = s_pn[pt].w;
                if (n >= 0) {
                    int ax, ay;
                    pair_axes(p, ax, ay);
                    const float4 q = s_q[pt];
                    const int W = sel4i(rs[0], rs[1], rs[2], rs[3], ax), H = sel4i(rs[0], rs[1], rs[2], rs[3], ay);   // (selects: no lane-indexed argument reads)
                    const Tap1 tx = tap1(sel4f(q.x, q.y, q.z, q.w, ax), W), ty = tap1(sel4f(q.x, q.y, q.z, q.w, ay), H);
                    const bool time_plane = ay == 3, marg = time_plane && tuni;
                    const int wx = time_plane ? twx : HEX_SW, wy = time_plane ? twy : HEX_SW;
                    const int ancx = sel4i(anc[0], anc[1], anc[2], anc[3], ax), ancy = sel4i(anc[0], anc[1], anc[2], anc[3], ay);
                    const int cx0 = tx.i0 - ancx, cx1 = tx.i1 - ancx, cy0 = marg ? 0 : ty.i0 - ancy, cy1 = marg ? 0 : ty.i1 - ancy;
                    // window base of the plane: spatial planes 0 (xy), 1 (xz), 3 (yz), then the time planes 2 (xt), 4 (yt), 5 (zt)
                    const int wbase = time_plane ? tbase + (p == 2 ? 0 : (p == 4 ? tw : 2 * tw)) : (p == 3 ? 2 : p) * HEX_SCELLS;
                    // the whole 2 x 2 footprint in the window (the anchor is the block's smallest tap, but a NaN coordinate maps to
                    // cell 0): one address, two strides (0 where the neighbour was clamped onto the same cell at the border)
                    const bool inside = cx0 >= 0 && cy0 >= 0 && cx1 < wx && cy1 < wy;
                    const uint32_t lds = inside ? (uint32_t)((wbase + cy0 * wx + cx0) * C) : 0xffffffffu;
                    const uint32_t dx = (uint32_t)((tx.i1 - tx.i0) * C), sy = (uint32_t)((cy1 - cy0) * wx * C);
                    int slot = it % RING + k2;
                    slot -= slot >= RING ? RING : 0;
                    const int row = sb + slot * IROWS + rem;
                    constexpr int SH = HEX_STAGE_BYTES ? 2 : 0;
                    s_a[row] = make_uint4(tap_at(tx.i0, ty.i0, W, C, 0) << SH, ((uint32_t)(ty.i1 - ty.i0) * (uint32_t)W * (uint32_t)C) << SH, lds, (dx << SH) | (sy << 8));
                    // grid_sampler_2d: nw * (1-fx)(1-fy) + ne * fx (1-fy) + sw * (1-fx) fy + se * fx fy; the clamped neighbour
                    // (x1 == x0 at the border) contributes no slope: its weight is 0 and the clip mask is 0
                    s_w[row] = make_float4(tx.f, ty.f, tx.ds, ty.ds);
                }
            }
        };
        float nw[6], ne[6], sw[6], se[6], go = 0.f;
        int n = -1;
        // ---- the gathers of iteration `it`: uniform plane base + 32-bit byte offsets (the saddr form: no 64-bit address arithmetic)
        auto gather = [&](int it) {
            n = s_pn[wave * PER_WAVE + it * GW + gw].w;
            if (n >= 0) {
                go = HEX_NT ? __builtin_nontemporal_load(g.dL_dout + ((size_t)n * (S * C) + cs * C + c)) : g.dL_dout[(size_t)n * (S * C) + cs * C + c];
#pragma unroll
                for (int p = 0; p < 6; p++) {
                    const uint4 A = s_a[sb + (it % RING) * IROWS + gw * 6 + p];
                    const char* __restrict__ pl = HEX_PL(p);
#if HEX_STAGE_BYTES
                    const uint32_t dx = A.w & 0xffu, o00 = A.x + ((uint32_t)c << 2), dy = A.y;          // (bytes, as staged)
#else
                    const uint32_t dx = (A.w & 0xffu) << 2, o00 = (A.x + c) << 2, dy = A.y << 2;
#endif
                    nw[p] = *(const float*)(pl + o00); ne[p] = *(const float*)(pl + (o00 + dx));
                    sw[p] = *(const float*)(pl + (o00 + dy)); se[p] = *(const float*)(pl + (o00 + dy + dx));
                }
            }
        };
    auto first_rows = [&](int ss) {
        enter_scale(ss);
        stage(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // the wave's own LDS writes, read back by its other lanes
        __builtin_amdgcn_wave_barrier();
        gather(0);
    };
    first_rows(0);
    for (int s = 0; s < S; s++) {
#pragma unroll 1
        for (int it = 0; it < ITERS; it++) {
            if (it % SK == SK - 1 && it + 1 < ITERS) stage(it + 1);      // (the rows of the next SK iterations)
            const int rowb = sb + (it % RING) * IROWS + gw * 6, pt = wave * PER_WAVE + it * GW + gw, n_cur = n;
            float gi[6];
            if (n_cur >= 0) {
                float f[6], dix[6], diy[6];
#pragma unroll
                for (int p = 0; p < 6; p++) {
                    // the sample as nested interpolations: the differences it forms ARE the slopes (ten instructions for value and both slopes where
                    // the weight form took twelve, and ONE staged row per plane, (fx, fy, slope x, slope y), instead of two)
                    const float4 F = s_w[rowb + p];
                    const float d0 = ne[p] - nw[p], d1 = se[p] - sw[p];
                    const float top = nw[p] + F.x * d0, bot = sw[p] + F.x * d1, dv = bot - top;
                    f[p] = top + F.y * dv;
                    dix[p] = (d0 + F.y * (d1 - d0)) * F.z;
                    diy[p] = dv * F.w;
                }
                float pre[7], suf[7], dq[4] = {0.f, 0.f, 0.f, 0.f};
                pre[0] = 1.f; suf[6] = 1.f;
#pragma unroll
                for (int p = 0; p < 6; p++) pre[p + 1] = pre[p] * f[p];
#pragma unroll
                for (int p = 5; p >= 0; p--) suf[p] = suf[p + 1] * f[p];
#pragma unroll
                for (int p = 0; p < 6; p++) {
                    int ax, ay;
                    pair_axes(p, ax, ay);
                    gi[p] = go * (pre[p] * suf[p + 1]);                      // dL / d interp of plane p, channel c
                    dq[ax] += gi[p] * dix[p];
                    if (DT || ay < 3) dq[ay] += gi[p] * diy[p];
                }
                if (want_dq) {
                    // summed over the C channel lanes; a point belongs to one lane group of one wave: plain adds across the scales
                    if (DT == 2) t_acc += dq[3];
#pragma unroll
                    for (int k = 0; k < (DT == 1 ? 4 : 3); k++) {          // DPP row sums; the last lane of the group holds the total
                        float v = dq[k];
                        v = dpp_add_f32<DPP_ROW_SHR(1), 0xf>(v);
                        v = dpp_add_f32<DPP_ROW_SHR(2), 0xf>(v);
                        v = dpp_add_f32<DPP_ROW_SHR(4), 0xf>(v);
                        v = dpp_add_f32<DPP_ROW_SHR(8), 0xf>(v);
                        if (C == 32) v = dpp_add_f32<DPP_ROW_BCAST15, 0xa>(v);
                        dq[k] = v;
                    }
                    if (c == C - 1) {
                        float4 acc = s_dq[pt];
                        acc.x += dq[0]; acc.y += dq[1]; acc.z += dq[2];
                        if (DT == 1) acc.w += dq[3];
                        s_dq[pt] = acc;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // stage(it + 1) is visible to the wave
            __builtin_amdgcn_wave_barrier();
            // Round 5 (late): the rows of a deferred scale's spatial planes leave BEFORE the next gathers are requested.  gfx9 counts loads and
            // stores on ONE counter, in order: with the stores issued after the gathers, the wait for the gathers at the top of the next
            // iteration (vmcnt(0): the number of stores in between is not a compile-time constant) also waited for the stores' acknowledgements,
            // which are the youngest operations in flight; now the gathers are, and the LDS adds below (a different counter) run under them.
#ifndef HEX_ABL_NO_DEFER_STORE      /* ablation build (profiles/r06_hexplane_ablations.txt): the rows of the deferred scales are computed but never stored */
            if (n_cur >= 0 && deferred) {
                const int4 P = s_pn[pt];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    if (p == 2 || !HEX_GP(p)) continue;
                    const int pidx = p == 3 ? 2 : p;                      // the row itself, at the point's position in that plane's order (plain store)
                    char* rows = (char*)(g.defer_rows + (defer_base + pidx) * (size_t)a.num_points * C);
#if HEX_NT_ROWS
                    __builtin_nontemporal_store(gi[p], (float*)(rows + (((uint32_t)(pidx == 0 ? P.x : (pidx == 1 ? P.y : P.z)) * C + c) << 2)));
#else
                    *(float*)(rows + (((uint32_t)(pidx == 0 ? P.x : (pidx == 1 ? P.y : P.z)) * C + c) << 2)) = gi[p];
#endif
                }
            }
#else
            if (n_cur >= 0 && deferred) {          // (the rows stay computed: their sum reaches memory only for a value that never occurs)
                float sink = 0.f;
#pragma unroll
                for (int p = 0; p < 4; p++) if (p != 2 && HEX_GP(p)) sink += gi[p];
                if (sink == 12345.678f) g.defer_rows[0] = sink;
            }
#endif
            if (it + 1 < ITERS) gather(it + 1);                          // in flight while the rows of `it` are scattered
            // ---- the 24 tap rows of the point: rows inside the windows are native fp64 LDS adds, rows outside leave as global float
            // atomics; both are fire and forget
#ifdef HEX_ABL_NO_ADDS              /* ablation build: no LDS window adds and no fallback atomics (hence empty flushes) */
            if (false) {
#else
            if (n_cur >= 0) {
#endif
#pragma unroll
                for (int p = 0; p < 6; p++) {
                    const bool marg = (p == 2 || p >= 4) && tuni;         // marginal over the block's time: two rows
                    float* gp = HEX_GP(p);
                    if (!gp) continue;
                    if (deferred && p != 2 && p < 4) continue;            // spatial plane of a deferred scale: its row left above
                    if (gi[p] == 0.f) continue;
                    const uint4 A = s_a[rowb + p];                       // (read again: cheaper than live registers)
                    const float4 F = s_w[rowb + p];
                    const float gx1 = gi[p] * F.x, gx0 = gi[p] - gx1;      // gi (1 - fx), gi fx: the x-marginals of the four weights
                    const uint32_t dx = (A.w & 0xffu) >> (HEX_STAGE_BYTES ? 2 : 0), sy = A.w >> 8;
                    if (A.z != 0xffffffffu) {
                        double* w0 = &win[A.z + c];
                        if (marg) {                                       // the x-marginals: the two time rows summed
                            lds_add_f64(w0, gx0);
                            lds_add_f64(w0 + dx, gx1);
                        } else {
                            const float s0 = gx0 * F.y, s1 = gx1 * F.y;
                            lds_add_f64(w0, gx0 - s0);
                            lds_add_f64(w0 + dx, gx1 - s1);
                            lds_add_f64(w0 + sy, s0);
                            lds_add_f64(w0 + sy + dx, s1);
                        }
                    } else {
                        char* g0 = (char*)gp;
                        constexpr int USH = HEX_STAGE_BYTES ? 0 : 2;
                        const uint32_t b00 = (A.x << USH) + ((uint32_t)c << 2), bdx = (A.w & 0xffu) << USH, bdy = A.y << USH;
                        const float s0 = gx0 * F.y, s1 = gx1 * F.y;
                        atomicAdd((float*)(g0 + b00), gx0 - s0);
                        atomicAdd((float*)(g0 + (b00 + bdx)), gx1 - s1);
                        atomicAdd((float*)(g0 + (b00 + bdy)), s0);
                        atomicAdd((float*)(g0 + (b00 + bdy + bdx)), s1);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (stage(it + 2) overwrites the rows read here)
            __builtin_amdgcn_wave_barrier();
        }
        // the flush of scale s by its own values (the waves' state moves on to scale s + 1 first)
        int ancf[4], rsf[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { ancf[k] = anc[k]; rsf[k] = rs[k]; }
        const int twf = tw, tbasef = tbase;
        const Tap1 tt = tap1(qlo[3], rs[3]);            // the block's time tap (meaningful when tuni)
        float* gpl[6];
#pragma unroll
        for (int p = 0; p < 6; p++) gpl[p] = g.dL_dplanes[s][p];
#if HEX_NEXT_SCALE_EARLY
        if (s + 1 < S) first_rows(s + 1);
#endif
        __syncthreads();
        // ---- flush: every touched cell row goes to HBM once; the windows are left clean for the next scale.  The plane of a cell varies
        // over the lane groups: its gradient pointer and its width are SELECTED from scalars (indexing the kernel arguments by a lane value
        // is a load from them, and the atomic's address then waits for two dependent round trips per cell row)
        for (int cell = group; cell < HEX_WIN_CELLS; cell += GROUPS) {
            const float v = (float)win[cell * C + c];
            if (v == 0.f) continue;                       // (only cells a tap reached are non-zero)
            win[cell * C + c] = 0.0;
            if (cell < tbasef) {
                const int idx = cell / HEX_SCELLS, local = cell - idx * HEX_SCELLS, p = idx == 2 ? 3 : idx;
                int ax, ay;
                pair_axes(p, ax, ay);
                const int x = sel4i(ancf[0], ancf[1], ancf[2], ancf[3], ax) + local % HEX_SW, y = sel4i(ancf[0], ancf[1], ancf[2], ancf[3], ay) + local / HEX_SW;
                atomicAdd((idx == 0 ? gpl[0] : (idx == 1 ? gpl[1] : gpl[3])) + tap_at(x, y, sel4i(rsf[0], rsf[1], rsf[2], rsf[3], ax), C, c), v);
            } else {
                const int idx = (cell - tbasef) / twf, local = cell - tbasef - idx * twf;
                const int ax = idx, W = sel4i(rsf[0], rsf[1], rsf[2], rsf[3], ax);       // planes 2, 4, 5 pair x, y, z with the time
                float* gp = idx == 0 ? gpl[2] : (idx == 1 ? gpl[4] : gpl[5]);
                if (tuni) {
                    const int x = sel4i(ancf[0], ancf[1], ancf[2], ancf[3], ax) + local;
                    atomicAdd(gp + tap_at(x, tt.i0, W, C, c), v * (1.f - tt.f));
                    if (tt.f != 0.f) atomicAdd(gp + tap_at(x, tt.i1, W, C, c), v * tt.f);
                } else {
                    const int x = sel4i(ancf[0], ancf[1], ancf[2], ancf[3], ax) + local % (twf / 2), y = ancf[3] + local / (twf / 2);
                    atomicAdd(gp + tap_at(x, y, W, C, c), v);
                }
            }
        }
        __syncthreads();
#if !HEX_NEXT_SCALE_EARLY
        if (s + 1 < S) first_rows(s + 1);
#endif
    }
    // dL/dpts through normalize_aabb (the time coordinate is used as given)
    if (want_dq)
        for (int item = tid; item < HEX_AGG_POINTS * 4; item += HEX_AGG_THREADS) {
            const int j = item >> 2, k = item & 3, n = s_pn[j].w;
            if (n < 0) continue;
            const float v = ((const float*)&s_dq[j])[k];
            if (k < 3) { if (g.dL_dpts) g.dL_dpts[3 * (long)n + k] = v * (2.f / (sel4f(a.aabb[3], a.aabb[4], a.aabb[5], 0.f, k) - sel4f(a.aabb[0], a.aabb[1], a.aabb[2], 0.f, k))); }
            else if (DT == 1 && g.dL_dtimes) g.dL_dtimes[n] = v;
        }
    if (DT == 2) {                                      // the workgroup's share of sum_n dL/dtimes[n]: wave sums, then one atomic
        const float wsum = wave_reduce_to_lane63(t_acc);
        __syncthreads();                                // (qmin / qmax are no longer read)
        if (lane == 63) ((float*)qmax)[0] = 0.f;
        __syncthreads();
        if (lane == 63) lds_add_f32((float*)qmax, wsum);
        __syncthreads();
        if (tid == 0) atomicAdd(g.dL_dtime_sum, ((float*)qmax)[0]);
    }
}

// ---- per-plane pass of the deferred scales -----------------------------------------------------------------------------------
// grid (runs of 256 positions, 3 planes, deferred scales).  The block walks 256 consecutive points of ITS plane's order -- a Hilbert curve of
// the plane's two coordinates (emd_amd/hexplane.py plane_order) -- : on a fine scale they cover a few cells of that plane (2 M points over
// 512 x 512 cells: 7.6 points per cell; a run touches ~60 cells inside a 12 x 12 box: tests/analysis/plane_order_sim.py: 1 % of the taps
// of uniform points leave a 12 x 12 window at resolution 512, none at 256; along a Z-order curve 7 % left even a 16 x 16 one), so a cell
// row reaches HBM once per block instead of once per tap.  One thread per point computes the tap (from the plane's two normalised
// coordinates, which the main kernel left beside the rows -- no dependent gather) and the block's anchor, then lane = channel: the point's
// row (sequential in defer_rows, all loads of a lane group in flight at once), four LDS adds, then the flush.  Round 4: 1.30 -> 0.74 ms
// at 2 M points (Hilbert orders 1.08; a workgroup per scale instead of a loop over the scales 1.05; 12 x 12 windows, i.e. 3 -> 4 resident
// workgroups per CU: the chain load -> taps -> adds -> flush is latency, and what hides it is the number of chains in flight).
#ifndef HEX_PL_THREADS
#define HEX_PL_THREADS 512
#endif
#ifndef HEX_PL_POINTS
#define HEX_PL_POINTS 256
#endif
#ifndef HEX_PW
#define HEX_PW 12                        /* 12 x 12 fp64 cells + staging = 39.9 KB: four workgroups (32 waves) per CU */
#endif
template <int C>
__global__ void __launch_bounds__(HEX_PL_THREADS) k_hexplane_bwd_plane(EmdHexArgs a, EmdHexGrads g) {
    constexpr int GROUPS = HEX_PL_THREADS / C, PER = HEX_PL_POINTS / GROUPS, WCELLS = HEX_PW * HEX_PW;
    __shared__ double win[WCELLS * C];
    __shared__ uint32_t s_tap[HEX_PL_POINTS];                            // tap cell x0 | y0 << 14 | (x1 - x0) << 28 | (y1 - y0) << 29 (resolutions < 2^14, checked on the host)
    __shared__ float2 s_f[HEX_PL_POINTS];
    __shared__ int cmin[2];
    const int tid = threadIdx.x, group = tid / C, c = tid % C, pidx = blockIdx.y, p = pidx == 2 ? 3 : pidx;
    int ax, ay;
    pair_axes(p, ax, ay);
    const long first = (long)blockIdx.x * HEX_PL_POINTS;
    const int count = (int)min((long)HEX_PL_POINTS, (long)a.num_points - first);
    const size_t NC = (size_t)a.num_points * C;
    const float2* defer_q = (const float2*)(g.defer_rows + (size_t)__builtin_popcount(g.defer_mask) * 3 * NC) + (size_t)pidx * a.num_points;
    for (int i = tid; i < WCELLS * C; i += HEX_PL_THREADS) win[i] = 0.0;
    // (unconditional, from a clamped position: a guarded load is waited for where it is issued -- DESIGN.md section 6 -- and the sixteen rows below would be
    // requested one round trip later; the value is used by the threads with a point only)
    const float2 q = defer_q[min(first + tid, (long)a.num_points - 1)];
    // one workgroup per (run, plane, deferred scale): nothing is carried from scale to scale, and the chain load -> taps -> adds -> flush of one
    // scale no longer waits for the flush of the one before
    const int sidx = blockIdx.z;
    {
        uint32_t rest = g.defer_mask;
        for (int k = 0; k < sidx; k++) rest &= rest - 1;
        const int s = __builtin_ctz(rest);
        float* gp = g.dL_dplanes[s][p];
        if (!gp) return;                                                   // (uniform)
        const float* rows = g.defer_rows + ((size_t)sidx * 3 + pidx) * NC;
        const int W = a.res[s][ax], H = a.res[s][ay];
        // the group's rows, all in flight at once (while the taps are staged)
        float gis[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int j = group + GROUPS * k;
#if HEX_NT_ROWS
            gis[k] = __builtin_nontemporal_load(rows + ((size_t)min(first + j, (long)a.num_points - 1) * C + c));
#else
            gis[k] = rows[(size_t)min(first + j, (long)a.num_points - 1) * C + c];     // (clamped, not guarded: rows past the end are skipped below)
#endif
        }
        if (tid < 2) cmin[tid] = INT_MAX;
        __syncthreads();
        if (tid < count) {
            const Tap1 tx = tap1(q.x, W), ty = tap1(q.y, H);
            s_tap[tid] = (uint32_t)tx.i0 | ((uint32_t)ty.i0 << 14) | ((uint32_t)(tx.i1 - tx.i0) << 28) | ((uint32_t)(ty.i1 - ty.i0) << 29);
            s_f[tid] = make_float2(tx.f, ty.f);
            atomicMin(&cmin[0], tx.i0);
            atomicMin(&cmin[1], ty.i0);
        }
        __syncthreads();
        const int ancx = cmin[0], ancy = cmin[1];
        // Round 5 (late): the kernel is bound by its vector instructions (4.9 x 10^8 per launch = 0.79 of its 1.05 ms), and every one of a row's 32 channel
        // lanes decoded the tap, placed it in the window and formed the four weights.  The point's own thread now rewrites its tap in window form once the
        // anchor is known -- byte address of the (x0, y0) cell row | dx C << 16 | dy PW << 24 | 1 << 31 -- (one more barrier), and the weights come out of the
        // row as nested differences: 54 -> ~35 vector instructions per row; backward 3.84 -> 3.74 ms at 2 M points
        static_assert(WCELLS * C * 8 <= 0x10000 && HEX_PW < 128 && C <= 32, "window tap encoding");
        if (tid < count) {
            const uint32_t tap = s_tap[tid];
            const int x0 = tap & 0x3fffu, y0 = (tap >> 14) & 0x3fffu, dx = (tap >> 28) & 1u, dy = (tap >> 29) & 1u;
            const int cx0 = x0 - ancx, cy0 = y0 - ancy;
            if (cx0 + dx < HEX_PW && cy0 + dy < HEX_PW)                    // (cx0, cy0 >= 0: the anchor is the block's smallest tap)
                s_tap[tid] = (uint32_t)((cy0 * HEX_PW + cx0) * C * 8) | ((uint32_t)(dx * C) << 16) | ((uint32_t)(dy * HEX_PW) << 24) | 0x80000000u;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int j = group + GROUPS * k;
            const float gi = gis[k];
            if (j >= count || gi == 0.f) continue;
            const uint32_t tap = s_tap[j];
            const float fx = s_f[j].x, fy = s_f[j].y;
            const float gx1 = gi * fx, gx0 = gi - gx1, s0 = gx0 * fy, s1 = gx1 * fy;
            const float w00 = gx0 - s0, w10 = gx1 - s1, w01 = s0, w11 = s1;
            const int x0 = tap & 0x3fffu, y0 = (tap >> 14) & 0x3fffu, dx = (tap >> 28) & 1u, dy = (tap >> 29) & 1u;      // (meaningful outside the window)
            if ((int)tap < 0) {
                char* w0 = (char*)win + ((tap & 0xffffu) + (uint32_t)c * 8u);
                const uint32_t ox = ((tap >> 16) & 0xffu) << 3, oy = ((tap >> 24) & 0x7fu) * (uint32_t)(C * 8);
                lds_add_f64((double*)w0, w00);
                lds_add_f64((double*)(w0 + ox), w10);
                lds_add_f64((double*)(w0 + oy), w01);
                lds_add_f64((double*)(w0 + oy + ox), w11);
            } else {
                atomicAdd(gp + tap_at(x0, y0, W, C, c), w00);
                atomicAdd(gp + tap_at(x0 + dx, y0, W, C, c), w10);
                atomicAdd(gp + tap_at(x0, y0 + dy, W, C, c), w01);
                atomicAdd(gp + tap_at(x0 + dx, y0 + dy, W, C, c), w11);
            }
        }
        __syncthreads();
        for (int cell = group; cell < WCELLS; cell += GROUPS) {
            const float v = (float)win[cell * C + c];
            if (v == 0.f) continue;
            win[cell * C + c] = 0.0;
            atomicAdd(gp + tap_at(ancx + cell % HEX_PW, ancy + cell / HEX_PW, W, C, c), v);
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
    if (g->dL_dtimes) hipLaunchKernelGGL((k_hexplane_bwd_agg<C, 1>), dim3(blocks), dim3(HEX_AGG_THREADS), 0, st, *a, *g, stride);
    else if (g->dL_dtime_sum) hipLaunchKernelGGL((k_hexplane_bwd_agg<C, 2>), dim3(blocks), dim3(HEX_AGG_THREADS), 0, st, *a, *g, stride);
    else hipLaunchKernelGGL((k_hexplane_bwd_agg<C, 0>), dim3(blocks), dim3(HEX_AGG_THREADS), 0, st, *a, *g, stride);
    if (g->defer_mask)
        hipLaunchKernelGGL(k_hexplane_bwd_plane<C>, dim3((unsigned)((a->num_points + HEX_PL_POINTS - 1) / HEX_PL_POINTS), 3, (unsigned)__builtin_popcount(g->defer_mask)),
                           dim3(HEX_PL_THREADS), 0, st, *a, *g);
}

// the per-plane pass packs a tap cell into 14 + 14 bits
bool defer_res_ok(const EmdHexArgs* a, unsigned mask) {
    for (int s = 0; s < a->num_scales; s++)
        if ((mask >> s) & 1u)
            for (int k = 0; k < 3; k++) if (a->res[s][k] >= (1 << 14)) return false;
    return true;
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
            if ((int64_t)a->res[s][A_[p]] * a->res[s][B_[p]] * C >= ((int64_t)1 << 30)) {       // the kernels address a plane with 32-bit BYTE offsets
                emd_set_error("%s: plane %d of scale %d holds 2^30 floats or more", who, p, s); return EMD_ERR_INVALID;
            }
        }
    return EMD_OK;
}

// ---- sort keys of the visiting orders ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread3(uint32_t v) {     // abcdefghij -> a00b00c00d00e00f00g00h00i00j
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    return (v | (v << 2)) & 0x09249249u;
}
__device__ __forceinline__ uint32_t hilbert2(uint32_t x, uint32_t y, int bits) {       // the classic xy -> d walk
    uint32_t d = 0;
    for (uint32_t s = 1u << (bits - 1); s; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (!ry) {
            if (rx) { x = s - 1u - x; y = s - 1u - y; }
            const uint32_t t = x; x = y; y = t;
        }
    }
    return d;
}
__global__ void __launch_bounds__(EMD_BLOCK) k_hexplane_order_keys(const float* __restrict__ pts, const float* __restrict__ aabb, int64_t N, int32_t* __restrict__ keys) {
    const int64_t n = (int64_t)blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (n >= N) return;
    uint32_t q10[3], q12[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float u = (pts[3 * n + k] - aabb[k]) / (aabb[3 + k] - aabb[k]);
        u = fminf(fmaxf(u, 0.f), 1.f);                    // (NaN -> 0: fmaxf returns the other operand)
        q10[k] = (uint32_t)(u * 1023.f);
        q12[k] = (uint32_t)(u * 4095.f);
    }
    keys[n] = (int32_t)(spread3(q10[0]) | (spread3(q10[1]) << 1) | (spread3(q10[2]) << 2));
    keys[N + n] = (int32_t)hilbert2(q12[0], q12[1], 12);
    keys[2 * N + n] = (int32_t)hilbert2(q12[0], q12[2], 12);
    keys[3 * N + n] = (int32_t)hilbert2(q12[1], q12[2], 12);
}

}  // namespace

extern "C" int emd_hexplane_order_keys(const float* pts, const float* aabb, int64_t num_points, int32_t* keys, void* hip_stream) {
    if (num_points < 0 || (num_points > 0 && (!pts || !aabb || !keys))) { emd_set_error("hexplane_order_keys: null pointer"); return EMD_ERR_INVALID; }
    if (num_points == 0) return EMD_OK;
    hipLaunchKernelGGL(k_hexplane_order_keys, dim3((unsigned)((num_points + EMD_BLOCK - 1) / EMD_BLOCK)), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream,
                       pts, aabb, num_points, keys);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}


extern "C" int emd_hexplane_forward(const EmdHexArgs* a, void* hip_stream) {
    int rc = check_hex(a, "hexplane_forward");
    if (rc) return rc;
    if (!a->out) { emd_set_error("hexplane_forward: null output"); return EMD_ERR_INVALID; }
    if (a->num_points == 0) return EMD_OK;
    const unsigned chunks = (unsigned)((a->num_points + HEX_F4_POINTS - 1) / HEX_F4_POINTS), grid4 = (chunks + 7u) / 8u * 8u;
    if (a->time_tables) {
        if (a->channels != 32 && a->channels != 16) { emd_set_error("hexplane_forward: time_tables are served for 16 or 32 channels"); return EMD_ERR_INVALID; }
        long rows = 0;
        for (int s = 0; s < a->num_scales; s++) rows += (long)a->res[s][0] + a->res[s][1] + a->res[s][2];
        if (rows * a->channels >= ((long)1 << 28)) { emd_set_error("hexplane_forward: time_tables too large"); return EMD_ERR_INVALID; }
        hipLaunchKernelGGL(k_hexplane_time_tables, dim3((unsigned)((rows * a->channels + EMD_BLOCK - 1) / EMD_BLOCK)), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, *a);
    }
    if (a->channels == 32) hipLaunchKernelGGL(k_hexplane_fwd4<32>, dim3(grid4), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, *a, chunks);
    else if (a->channels == 16) hipLaunchKernelGGL(k_hexplane_fwd4<16>, dim3(grid4), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, *a, chunks);
    else          // other channel counts: lane = channel
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
    if (g->defer_mask) {
        const bool agg = a->order && (a->channels == 32 || a->channels == 16);
        if (!agg || !g->defer_rows || !g->order2d[0] || !g->order2d[1] || !g->order2d[2] || !g->pos2d[0] || !g->pos2d[1] || !g->pos2d[2] ||
            (g->defer_mask >> a->num_scales) || (int64_t)a->num_points * a->channels * 4 >= ((int64_t)1 << 32) || !defer_res_ok(a, g->defer_mask)) {
            emd_set_error("hexplane_backward: defer_mask needs a visiting order, 16 or 32 channels, order2d / pos2d / defer_rows, bits below num_scales, deferred resolutions < 2^14 and N * C < 2^30");
            return EMD_ERR_INVALID;
        }
    }
    // a visiting order promises spatial coherence: aggregate in LDS (windows are sized for C <= 32; the kernel addresses a plane
    // with 32-bit BYTE offsets, so planes of 2^30 floats or more take the direct kernel)
    bool small_planes = true;
    for (int s = 0; s < a->num_scales; s++)
        for (int p = 0; p < 6; p++) {
            const int A_[6] = {0, 0, 0, 1, 1, 2}, B_[6] = {1, 2, 3, 2, 3, 3};
            if ((int64_t)a->res[s][A_[p]] * a->res[s][B_[p]] * a->channels >= ((int64_t)1 << 30)) small_planes = false;
        }
    if (a->order && small_planes && a->channels == 32) launch_bwd_agg<32>(a, g, (hipStream_t)hip_stream);
    else if (a->order && small_planes && a->channels == 16) launch_bwd_agg<16>(a, g, (hipStream_t)hip_stream);
    else {
        const int per_block = EMD_BLOCK / a->channels;
        hipLaunchKernelGGL(k_hexplane_bwd, dim3((unsigned)((a->num_points + per_block - 1) / per_block)), dim3(EMD_BLOCK), 0,
                           (hipStream_t)hip_stream, *a, *g);
    }
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
