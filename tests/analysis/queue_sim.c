/* queue_sim.c -- analysis tool (NOT product code): simulates the per-sub-block queueing of a K7 wave (one wave per 8x8 quadrant,
 * list walked back to front, 64 entries per scan step) for a ring of R live entries, and counts wave-level work units. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float min_form_rect(float mx, float my, float A, float B, float C, float x0, float x1, float y0, float y1) {
    if (mx >= x0 && mx <= x1 && my >= y0 && my <= y1) return 0.f;
    float best = 1e30f;
    const float xs[2] = {x0, x1}, ys[2] = {y0, y1};
    for (int i = 0; i < 2; i++) {
        const float dx = mx - xs[i];
        float dy = -B * dx / C;
        dy = fminf(fmaxf(dy, my - y1), my - y0);
        const float q = A * dx * dx + 2.f * B * dx * dy + C * dy * dy;
        if (q < best) best = q;
    }
    for (int i = 0; i < 2; i++) {
        const float dy = my - ys[i];
        float dx = -B * dy / A;
        dx = fminf(fmaxf(dx, mx - x1), mx - x0);
        const float q = A * dx * dx + 2.f * B * dx * dy + C * dy * dy;
        if (q < best) best = q;
    }
    return best;
}

#define MAXQ 4096
typedef struct { int ref; } Ent;

/* NS = 4: 4x4 sub-blocks, NS = 2: 8x4 halves, NS = 1: whole quadrant (the present kernel, exact test)
 * out[0] scan steps, [1] batches fired, [2] forced (partial) fires, [3] entries kept (ring inserts), [4] sum of batch fill,
 * [5] final partial fires */
void queue_sim(int W, int H, const uint32_t* ranges, const uint32_t* ids, const float* means2D, const float* conic_opacity,
               const uint32_t* n_contrib, int R, int NS, double* out) {
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    double acc[8];
    memset(acc, 0, sizeof(acc));
#pragma omp parallel
    {
        double a[8];
        memset(a, 0, sizeof(a));
        int* refc = (int*)malloc(sizeof(int) * 1 << 20);
        int* qbuf[4];
        for (int s = 0; s < 4; s++) qbuf[s] = (int*)malloc(sizeof(int) * (1 << 20));
#pragma omp for schedule(dynamic, 8)
        for (int t = 0; t < gx * gy; t++) {
            const int tx0 = (t % gx) * 16, ty0 = (t / gx) * 16;
            const uint32_t s0 = ranges[2 * t], e0 = ranges[2 * t + 1];
            for (int q = 0; q < 4; q++) {
                const int qx0 = tx0 + (q & 1) * 8, qy0 = ty0 + (q >> 1) * 8;
                uint32_t nsb[4] = {0, 0, 0, 0}, nq = 0;
                for (int y = 0; y < 8; y++)
                    for (int x = 0; x < 8; x++) {
                        const int px = qx0 + x, py = qy0 + y;
                        if (px >= W || py >= H) continue;
                        const uint32_t n = n_contrib[(size_t)py * W + px];
                        int sb = NS == 4 ? (y >> 2) * 2 + (x >> 2) : NS == 2 ? (y >> 2) : 0;
                        if (n > nsb[sb]) nsb[sb] = n;
                        if (n > nq) nq = n;
                    }
                if (nq > e0 - s0) nq = e0 - s0;
                if (!nq) continue;
                int head = 0, tail = 0;          /* ring positions (monotone counters) */
                int qh[4] = {0, 0, 0, 0}, qt[4] = {0, 0, 0, 0};
                const int steps = (nq + 63) / 64;
                for (int st = 0; st < steps; st++) {
                    a[0] += 1;
                    for (int l = 63; l >= 0; l--) {             /* descending list position */
                        const int pos = (steps - 1 - st) * 64 + l;
                        if (pos >= (int)nq) continue;
                        const uint32_t g = ids[s0 + pos];
                        const float* m = means2D + 2 * g;
                        const float* co = conic_opacity + 4 * g;
                        if (!(co[3] >= 1.f / 255.f)) continue;
                        const float tau2 = 2.f * logf(255.f * co[3]) * 1.01f + 0.05f;
                        if (!(min_form_rect(m[0], m[1], co[0], co[1], co[2], qx0, qx0 + 7, qy0, qy0 + 7) <= tau2)) continue;
                        int mask = 0;
                        for (int s = 0; s < NS; s++) {
                            float x0, x1, y0, y1;
                            if (NS == 4) { x0 = qx0 + 4 * (s & 1); x1 = x0 + 3; y0 = qy0 + 4 * (s >> 1); y1 = y0 + 3; }
                            else if (NS == 2) { x0 = qx0; x1 = x0 + 7; y0 = qy0 + 4 * s; y1 = y0 + 3; }
                            else { x0 = qx0; x1 = x0 + 7; y0 = qy0; y1 = y0 + 7; }
                            if ((uint32_t)pos < nsb[s] && min_form_rect(m[0], m[1], co[0], co[1], co[2], x0, x1, y0, y1) <= tau2) mask |= 1 << s;
                        }
                        if (!mask) continue;
                        a[3] += 1;
                        refc[head & ((1 << 20) - 1)] = __builtin_popcount(mask);
                        for (int s = 0; s < NS; s++) if (mask >> s & 1) qbuf[s][qh[s]++ & ((1 << 20) - 1)] = head;
                        head++;
                    }
                    for (;;) {
                        int fired = 0;
                        for (int s = 0; s < NS; s++)
                            while (qh[s] - qt[s] >= 64) {
                                for (int k = 0; k < 64; k++) refc[qbuf[s][qt[s]++ & ((1 << 20) - 1)] & ((1 << 20) - 1)]--;
                                a[1] += 1; a[4] += 64; fired = 1;
                            }
                        while (tail < head && refc[tail & ((1 << 20) - 1)] == 0) tail++;
                        if (st + 1 < steps && R - (head - tail) < 64) {
                            /* no room for the next step: fire (partially) the queue that holds the oldest live entry */
                            int best = -1;
                            for (int s = 0; s < NS; s++)
                                if (qh[s] > qt[s] && (best < 0 || qbuf[s][qt[s] & ((1 << 20) - 1)] < qbuf[best][qt[best] & ((1 << 20) - 1)])) best = s;
                            if (best < 0) break;
                            int n = qh[best] - qt[best];
                            if (n > 64) n = 64;
                            for (int k = 0; k < n; k++) refc[qbuf[best][qt[best]++ & ((1 << 20) - 1)] & ((1 << 20) - 1)]--;
                            a[1] += 1; a[2] += 1; a[4] += n;
                            continue;
                        }
                        if (!fired) break;
                    }
                }
                for (int s = 0; s < NS; s++)
                    while (qh[s] > qt[s]) {
                        int n = qh[s] - qt[s];
                        if (n > 64) n = 64;
                        qt[s] += n;
                        a[1] += 1; a[5] += 1; a[4] += n;
                    }
            }
        }
        free(refc);
        for (int s = 0; s < 4; s++) free(qbuf[s]);
#pragma omp critical
        for (int i = 0; i < 8; i++) acc[i] += a[i];
    }
    for (int i = 0; i < 8; i++) out[i] = acc[i];
}
