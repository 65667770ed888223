// hexplane.hip -- fused multi-scale HexPlane feature lookup, forward and backward.
//
// Replaces HexPlaneField.get_density / interpolate_ms_features (S3Gaussian/scene/hexplane.py:18-110,150-183): per scale six
// F.grid_sample launches (bilinear, align_corners, border padding) + five products + a concat, and their backward
// (six grid_sampler_2d_backward launches that scatter channel-FIRST: one float atomic per channel per tap, each to a
// different cache line).  Here the planes are handed over CHANNEL-LAST ([res_h][res_w][C]) and a point is owned by C
// consecutive lanes (lane = channel): every tap is one coalesced C x 4-byte read, the product over the six planes and
// the concat over scales happen in registers, and the backward scatters C consecutive floats per tap (the row shape
// float atomics like).  One launch forward, one backward, for all scales.
// HBM / L2-bound gather-scatter; no reuse to exploit beyond the caches (points arrive unordered).
#include <string.h>

#include "common.h"

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

// plane pair p of (0,1),(0,2),(0,3),(1,2),(1,3),(2,3): first index -> width axis, second -> height axis
__device__ __forceinline__ void pair_axes(int p, int& a, int& b) {
    const int A[6] = {0, 0, 0, 1, 1, 2}, B[6] = {1, 2, 3, 2, 3, 3};
    a = A[p]; b = B[p];
}

__device__ __forceinline__ float sample(const float* __restrict__ pl, const Bilin& t, int W, int C, int c) {
    const float nw = pl[((size_t)t.y0 * W + t.x0) * C + c], ne = pl[((size_t)t.y0 * W + t.x1) * C + c];
    const float sw = pl[((size_t)t.y1 * W + t.x0) * C + c], se = pl[((size_t)t.y1 * W + t.x1) * C + c];
    // grid_sampler_2d: nw * (1-fx)(1-fy) + ne * fx (1-fy) + sw * (1-fx) fy + se * fx fy
    return nw * ((1.f - t.fx) * (1.f - t.fy)) + ne * (t.fx * (1.f - t.fy)) + sw * ((1.f - t.fx) * t.fy) + se * (t.fx * t.fy);
}

template <bool BWD>
__global__ void __launch_bounds__(EMD_BLOCK) k_hexplane(EmdHexArgs a, EmdHexGrads g) {
    const int C = a.channels, S = a.num_scales;
    const int group = threadIdx.x / C, c = threadIdx.x % C, per_block = EMD_BLOCK / C;
    const long n = (long)blockIdx.x * per_block + group;
    if (n >= a.num_points) return;
    float q[4];
#pragma unroll
    for (int k = 0; k < 3; k++) q[k] = (a.pts[3 * n + k] - a.aabb[k]) * (2.f / (a.aabb[3 + k] - a.aabb[k])) - 1.f;
    q[3] = a.times[n];
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_dq = BWD && (g.dL_dpts || g.dL_dtimes);
    for (int s = 0; s < S; s++) {
        float f[6];
        Bilin t[6];
#pragma unroll
        for (int p = 0; p < 6; p++) {
            int ax, ay;
            pair_axes(p, ax, ay);
            const int W = a.res[s][ax], H = a.res[s][ay];
            t[p] = bilin(q[ax], q[ay], W, H);
            f[p] = sample(a.planes[s][p], t[p], W, C, c);
        }
        if (!BWD) {
            float prod = 1.f;
#pragma unroll
            for (int p = 0; p < 6; p++) prod = prod * f[p];
            a.out[(size_t)n * (S * C) + s * C + c] = prod;
        } else {
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
                    atomicAdd(gp + ((size_t)b.y0 * W + b.x0) * C + c, gi * ((1.f - b.fx) * (1.f - b.fy)));
                    atomicAdd(gp + ((size_t)b.y0 * W + b.x1) * C + c, gi * (b.fx * (1.f - b.fy)));
                    atomicAdd(gp + ((size_t)b.y1 * W + b.x0) * C + c, gi * ((1.f - b.fx) * b.fy));
                    atomicAdd(gp + ((size_t)b.y1 * W + b.x1) * C + c, gi * (b.fx * b.fy));
                }
                if (want_dq) {
                    const float* pl = a.planes[s][p];
                    const float nw = pl[((size_t)b.y0 * W + b.x0) * C + c], ne = pl[((size_t)b.y0 * W + b.x1) * C + c];
                    const float sw = pl[((size_t)b.y1 * W + b.x0) * C + c], se = pl[((size_t)b.y1 * W + b.x1) * C + c];
                    // the clamped neighbour (x1 == x0 at the border) contributes no slope there: its weight is 0 and cx = 0
                    const float dix = (ne - nw) * (1.f - b.fy) + (se - sw) * b.fy;
                    const float diy = (sw - nw) * (1.f - b.fx) + (se - ne) * b.fx;
                    dq[ax] += gi * dix * b.cx;
                    dq[ay] += gi * diy * b.cy;
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
        }
    return EMD_OK;
}

}  // namespace

extern "C" int emd_hexplane_forward(const EmdHexArgs* a, void* hip_stream) {
    int rc = check_hex(a, "hexplane_forward");
    if (rc) return rc;
    if (!a->out) { emd_set_error("hexplane_forward: null output"); return EMD_ERR_INVALID; }
    if (a->num_points == 0) return EMD_OK;
    const int per_block = EMD_BLOCK / a->channels;
    EmdHexGrads g;
    memset((void*)&g, 0, sizeof(g));
    hipLaunchKernelGGL(k_hexplane<false>, dim3((unsigned)((a->num_points + per_block - 1) / per_block)), dim3(EMD_BLOCK), 0,
                       (hipStream_t)hip_stream, *a, g);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_hexplane_backward(const EmdHexArgs* a, const EmdHexGrads* g, void* hip_stream) {
    int rc = check_hex(a, "hexplane_backward");
    if (rc) return rc;
    if (!g || !g->dL_dout) { emd_set_error("hexplane_backward: null gradient"); return EMD_ERR_INVALID; }
    if (a->num_points == 0) return EMD_OK;
    const int per_block = EMD_BLOCK / a->channels;
    hipLaunchKernelGGL(k_hexplane<true>, dim3((unsigned)((a->num_points + per_block - 1) / per_block)), dim3(EMD_BLOCK), 0,
                       (hipStream_t)hip_stream, *a, *g);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
