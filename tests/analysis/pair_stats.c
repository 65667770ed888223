/* pair_stats.c -- analysis tool (NOT product code): how many (pixel, list entry) pairs the render kernels evaluate under
 * different culling granularities, against the pairs that actually contribute.  Built and driven by pair_stats.py. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float gauss_power(float A, float B, float C, float dx, float dy) {
    const float q = fmaf(A * dx, dx, (C * dy) * dy);
    return fmaf(-0.5f, q, -((B * dx) * dy));
}

/* tight box of the alpha >= 1/255 ellipse: returns 0 if the entry can be dropped entirely */
static int tight_box(const float* m, const float* co, float* lx, float* hx, float* ly, float* hy) {
    const float o = co[3];
    if (!(o >= 1.f / 255.f)) return 0;
    const float det = co[0] * co[2] - co[1] * co[1];
    if (!(det > 0.f)) { *lx = -1e30f; *hx = 1e30f; *ly = -1e30f; *hy = 1e30f; return 1; }
    const float tau2 = 2.f * logf(255.f * o);
    const float inv = 1.f / det;
    const float bx = sqrtf(tau2 * co[2] * inv) * 1.0001f + 0.01f, by = sqrtf(tau2 * co[0] * inv) * 1.0001f + 0.01f;
    *lx = m[0] - bx; *hx = m[0] + bx; *ly = m[1] - by; *hy = m[1] + by;
    return 1;
}


/* minimum of the quadratic form A dx^2 + 2 B dx dy + C dy^2 (d = mean - p) over the rectangle of pixel centres [x0,x1] x [y0,y1] */
static float min_form_rect(float mx, float my, float A, float B, float C, float x0, float x1, float y0, float y1) {
    if (mx >= x0 && mx <= x1 && my >= y0 && my <= y1) return 0.f;
    float best = 1e30f;
    const float xs[2] = {x0, x1}, ys[2] = {y0, y1};
    for (int i = 0; i < 2; i++) {          /* vertical edges x = xs[i] */
        const float dx = mx - xs[i];
        float dy = -B * dx / C;             /* unconstrained optimum of dy */
        float lo = my - y1, hi = my - y0;   /* dy range */
        dy = fminf(fmaxf(dy, lo), hi);
        const float q = A * dx * dx + 2.f * B * dx * dy + C * dy * dy;
        if (q < best) best = q;
    }
    for (int i = 0; i < 2; i++) {          /* horizontal edges y = ys[i] */
        const float dy = my - ys[i];
        float dx = -B * dy / A;
        float lo = mx - x1, hi = mx - x0;
        dx = fminf(fmaxf(dx, lo), hi);
        const float q = A * dx * dx + 2.f * B * dx * dy + C * dy * dy;
        if (q < best) best = q;
    }
    return best;
}

/* out[0] D (list entries), [1] tile-level pairs (entries up to the tile's deepest contributor x 256),
 * [2] quadrant entries, [3] quadrant pairs (x64), [4] sub-block entries (4x4), [5] sub-block pairs (x16),
 * [6] hits, [7] entries with >= 1 hit in a quadrant, [8] row-span pixel-pair evaluations x2 (8-wide rows of the quadrant, pairs
 * overlapping the exact x-interval of the ellipse on that row), [9] quadrant-entries x rows touched by the box (x8 pixels),
 * [10] sub-block entries counted with the per-sub-block deepest contributor, [11] 8x4 half-quadrant entries (x32) */
void pair_stats(int W, int H, const uint32_t* ranges, const uint32_t* ids, const float* means2D, const float* conic_opacity,
                const uint32_t* n_contrib, double* out) {
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    double acc[32];
    memset(acc, 0, sizeof(acc));
#pragma omp parallel
    {
        double a[32];
        memset(a, 0, sizeof(a));
#pragma omp for schedule(dynamic, 8)
        for (int t = 0; t < gx * gy; t++) {
            const int tx0 = (t % gx) * 16, ty0 = (t / gx) * 16;
            const uint32_t s = ranges[2 * t], e = ranges[2 * t + 1];
            a[0] += e - s;
            uint32_t nq[4] = {0, 0, 0, 0}, nsb[16], nt = 0;
            memset(nsb, 0, sizeof(nsb));
            for (int y = 0; y < 16; y++)
                for (int x = 0; x < 16; x++) {
                    const int px = tx0 + x, py = ty0 + y;
                    if (px >= W || py >= H) continue;
                    const uint32_t n = n_contrib[(size_t)py * W + px];
                    const int q = (y >> 3) * 2 + (x >> 3), sb = (y >> 2) * 4 + (x >> 2);
                    if (n > nq[q]) nq[q] = n;
                    if (n > nsb[sb]) nsb[sb] = n;
                    if (n > nt) nt = n;
                }
            a[1] += 256.0 * nt;
            for (uint32_t k = s; k < e; k++) {       /* [24] entries whose alpha >= 1/255 ellipse reaches a pixel centre of the 16 x 16 tile at all */
                const uint32_t g = ids[k];
                const float* m = means2D + 2 * g;
                const float* co = conic_opacity + 4 * g;
                if (!(co[3] >= 1.f / 255.f)) continue;
                const float tau2 = 2.f * logf(255.f * co[3]) * 1.01f + 0.05f;
                const float x1 = (float)(tx0 + 15 < W - 1 ? tx0 + 15 : W - 1), y1 = (float)(ty0 + 15 < H - 1 ? ty0 + 15 : H - 1);
                if (min_form_rect(m[0], m[1], co[0], co[1], co[2], (float)tx0, x1, (float)ty0, y1) <= tau2) { a[24] += 1; if (k - s < nt) a[25] += 1; }
            }
            for (uint32_t k = s; k < e; k++) {
                const uint32_t pos = k - s;          /* 0-based list position; contributes to a pixel if pos < n_contrib */
                if (pos >= nt) break;
                const uint32_t g = ids[k];
                const float* m = means2D + 2 * g;
                const float* co = conic_opacity + 4 * g;
                float lx, hx, ly, hy;
                if (!tight_box(m, co, &lx, &hx, &ly, &hy)) continue;
                const float tau2 = 2.f * logf(255.f * co[3]);
                for (int q = 0; q < 4; q++) {
                    if (pos >= nq[q]) continue;
                    const float qx0 = (float)(tx0 + (q & 1) * 8), qy0 = (float)(ty0 + (q >> 1) * 8);
                    if (!(lx <= qx0 + 7.f && hx >= qx0 && ly <= qy0 + 7.f && hy >= qy0)) continue;
                    a[2] += 1; a[3] += 64;

                    {   /* exact ellipse-vs-region tests; region sizes (w,h): 8x8, 8x4, 4x4, 8x2, 4x2, 8x1, 2x2 -> a[16..22] pairs */
                        static const int RW[7] = {8, 8, 4, 8, 4, 8, 2}, RH[7] = {8, 4, 4, 2, 2, 1, 2};
                        const float t2m = tau2 * 1.0002f + 0.02f;
                        for (int r = 0; r < 7; r++)
                            for (int ry = 0; ry < 8; ry += RH[r])
                                for (int rx = 0; rx < 8; rx += RW[r]) {
                                    const float x0 = qx0 + rx, y0 = qy0 + ry;
                                    if (min_form_rect(m[0], m[1], co[0], co[1], co[2], x0, x0 + RW[r] - 1, y0, y0 + RH[r] - 1) <= t2m)
                                        a[16 + r] += RW[r] * RH[r];
                                }
                    }
                    int any = 0;
                    for (int hh = 0; hh < 2; hh++) {
                        const float hy0 = qy0 + 4.f * hh;
                        if (lx <= qx0 + 7.f && hx >= qx0 && ly <= hy0 + 3.f && hy >= hy0) a[11] += 32;
                    }
                    for (int sy = 0; sy < 2; sy++)
                        for (int sx = 0; sx < 2; sx++) {
                            const float bx0 = qx0 + 4.f * sx, by0 = qy0 + 4.f * sy;
                            if (lx <= bx0 + 3.f && hx >= bx0 && ly <= by0 + 3.f && hy >= by0) {
                                a[4] += 1; a[5] += 16;
                                const int sb = (((q >> 1) * 2 + sy) * 4) + (q & 1) * 2 + sx;
                                if (pos < nsb[sb]) a[10] += 16;
                            }
                        }
                    for (int y = 0; y < 8; y++) {
                        const float pyf = qy0 + (float)y;
                        const int py = (int)pyf;
                        if (py >= H) continue;
                        if (ly <= pyf && hy >= pyf) {
                            a[9] += 8;
                            /* exact interval on this row: A dx^2 + 2 B dx dy + C dy^2 <= tau2, dx = mx - px */
                            const float dy = m[1] - pyf;
                            const float A = co[0], B = co[1], C = co[2];
                            const float disc = (B * dy) * (B * dy) - A * (C * dy * dy - tau2);
                            if (disc >= 0.f && A > 0.f) {
                                const float r = sqrtf(disc) / A, c = -B * dy / A;     /* dx in [c - r, c + r] */
                                const float x_lo = m[0] - (c + r) - 0.01f, x_hi = m[0] - (c - r) + 0.01f;
                                for (int pp = 0; pp < 4; pp++) {
                                    const float p0 = qx0 + 2.f * pp;
                                    if (x_lo <= p0 + 1.f && x_hi >= p0) a[8] += 2;
                                }
                            }
                        }
                        for (int x = 0; x < 8; x++) {
                            const int px = (int)qx0 + x;
                            if (px >= W) continue;
                            if (pos >= n_contrib[(size_t)py * W + px]) continue;
                            const float dx = m[0] - (float)px, dy = m[1] - pyf;
                            const float power = gauss_power(co[0], co[1], co[2], dx, dy);
                            if (power > 0.f) continue;
                            const float alpha = fminf(0.99f, co[3] * expf(power));
                            if (alpha < 1.f / 255.f) continue;
                            a[6] += 1; any = 1;
                        }
                    }
                    a[7] += any;
                }
            }
        }
#pragma omp critical
        for (int i = 0; i < 32; i++) acc[i] += a[i];
    }
    for (int i = 0; i < 32; i++) out[i] = acc[i];
}
