// footprint.h -- exact-conservative culling of a projected Gaussian against a rectangle of pixel centres (gfx950).
// Shared by the duplicate kernel (binning.hip: which (tile, Gaussian) pairs exist at all, and which quadrants of the tile each reaches)
// and the render forward (render.hip: the 4x4 sub-blocks of a quadrant).
//
// A pixel receives alpha >= 1/255 from a Gaussian only if q(d) = A dx^2 + 2 B dx dy + C dy^2 <= 2 ln(255 o); q is convex, so its
// minimum over the rectangle is 0 when the mean lies inside and otherwise sits on one of the four edges.  On the edge dx = const
//     q = C (dy - dy*)^2 + (det / C) dx^2,   dy* = -B dx / C,
// a sum of two non-negative terms (no cancellation), minimised by clamping dy* into the edge.  The threshold carries a 1 % + 0.05
// margin: det = A C - B^2 loses ~eps A C / det relative, so conics with A C > 3e4 det (footprints hundreds of pixels long and
// under a pixel wide: the margin would no longer cover the cancellation) are not culled at all; everything else is a few ulp.
// NaN means are culled (such an entry fails `power <= 0` at every pixel), degenerate conics are kept.
// [upstream has no such test: its tile list is the 3-sigma square of the largest eigenvalue; a pair
//  dropped here contributes to no pixel of the rectangle there either (the alpha >= 1/255 skip of its render loop)]
#pragma once
#include <hip/hip_runtime.h>

struct EllipseCull {
    float mx, my, A, C, kx, ky, nbc, nba, thr;
    bool none, all;
};
__device__ __forceinline__ EllipseCull ellipse_prepare(const float4& r0, const float4& r1) {
    EllipseCull e;
    const float o = r0.w;
    e.mx = r0.x; e.my = r0.y; e.A = r1.x; e.C = r1.z;
    e.none = !(o >= (1.f / 255.f));              // can never reach alpha >= 1/255 (also catches NaN)
    const float ac = r1.x * r1.z, det = ac - r1.y * r1.y;
    e.all = !(det > 0.f) || !(r1.x > 0.f) || !(r1.z > 0.f) || !(det * 3.0e4f > ac);
    e.thr = 2.f * __logf(255.f * o) * 1.01f + 0.05f;
    const float ia = __builtin_amdgcn_rcpf(r1.x), ic = __builtin_amdgcn_rcpf(r1.z);
    e.kx = det * ic; e.ky = det * ia; e.nbc = -r1.y * ic; e.nba = -r1.y * ia;
    return e;
}
// true unless no pixel centre of [x0, x1] x [y0, y1] can reach alpha >= 1/255
__device__ __forceinline__ bool ellipse_hits_rect(const EllipseCull& e, float x0, float x1, float y0, float y1) {
    const float dxa = e.mx - x0, dxb = e.mx - x1, dya = e.my - y0, dyb = e.my - y1;       // dxb <= dxa, dyb <= dya
    const bool inside = dxa >= 0.f && dxb <= 0.f && dya >= 0.f && dyb <= 0.f;
    float ta = e.nbc * dxa, tb = e.nbc * dxb;
    float ea = __builtin_amdgcn_fmed3f(ta, dyb, dya) - ta, eb = __builtin_amdgcn_fmed3f(tb, dyb, dya) - tb;
    const float qa = __builtin_fmaf(e.C * ea, ea, (e.kx * dxa) * dxa), qb = __builtin_fmaf(e.C * eb, eb, (e.kx * dxb) * dxb);
    ta = e.nba * dya; tb = e.nba * dyb;
    ea = __builtin_amdgcn_fmed3f(ta, dxb, dxa) - ta; eb = __builtin_amdgcn_fmed3f(tb, dxb, dxa) - tb;
    const float qc = __builtin_fmaf(e.A * ea, ea, (e.ky * dya) * dya), qd = __builtin_fmaf(e.A * eb, eb, (e.ky * dyb) * dyb);
    const float q = fminf(fminf(qa, qb), fminf(qc, qd));
    return !e.none && (e.all || inside || q <= e.thr);
}
// bit r (r = sby * 2 + sbx) set when 4x4 sub-block r of the quadrant at (qx0, qy0) can receive alpha >= 1/255
__device__ __forceinline__ uint32_t ellipse_subblock_mask(const float4& r0, const float4& r1, float qx0, float qy0) {
    const EllipseCull e = ellipse_prepare(r0, r1);
    return (ellipse_hits_rect(e, qx0, qx0 + 3.f, qy0, qy0 + 3.f) ? 1u : 0u) | (ellipse_hits_rect(e, qx0 + 4.f, qx0 + 7.f, qy0, qy0 + 3.f) ? 2u : 0u) |
           (ellipse_hits_rect(e, qx0, qx0 + 3.f, qy0 + 4.f, qy0 + 7.f) ? 4u : 0u) |
           (ellipse_hits_rect(e, qx0 + 4.f, qx0 + 7.f, qy0 + 4.f, qy0 + 7.f) ? 8u : 0u);
}
// bit q (q = qy * 2 + qx) set when the 8x8 quadrant q of the 16x16 tile at (tx0, ty0) can receive alpha >= 1/255
__device__ __forceinline__ uint32_t ellipse_quadrant_mask(const EllipseCull& e, float tx0, float ty0) {
    return (ellipse_hits_rect(e, tx0, tx0 + 7.f, ty0, ty0 + 7.f) ? 1u : 0u) | (ellipse_hits_rect(e, tx0 + 8.f, tx0 + 15.f, ty0, ty0 + 7.f) ? 2u : 0u) |
           (ellipse_hits_rect(e, tx0, tx0 + 7.f, ty0 + 8.f, ty0 + 15.f) ? 4u : 0u) |
           (ellipse_hits_rect(e, tx0 + 8.f, tx0 + 15.f, ty0 + 8.f, ty0 + 15.f) ? 8u : 0u);
}

// The list entries of the sorted (tile, Gaussian) pairs carry that mask in their top four bits (Gaussian ids stay below 2^28)
#define EMD_ID_BITS 28
#define EMD_ID_MASK ((1u << EMD_ID_BITS) - 1u)
