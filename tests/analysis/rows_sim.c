/* rows_sim.c -- analysis tool (NOT product code): K7 "rows" formulation: one wave per 8x8 quadrant, DPP row r <-> 4x4 sub-block r,
 * every row consumes its own queue 16 entries per round (8 pixel-pair iterations per round, all four rows in lockstep). */
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

/* out: [0] scan steps, [1] rounds, [2] row-batches (active rows summed over rounds), [3] entries kept, [4] sum of entries processed (row fill),
 * [5] forced rounds (ring full), [6] quadrant waves */
void rows_sim(int W, int H, const uint32_t* ranges, const uint32_t* ids, const float* means2D, const float* conic_opacity,
              const uint32_t* n_contrib, int R, int STEP, double* out) {
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    double acc[8];
    memset(acc, 0, sizeof(acc));
#pragma omp parallel
    {
        double a[8];
        memset(a, 0, sizeof(a));
        const int CAP = 1 << 16;
        int* refc = (int*)malloc(sizeof(int) * CAP);
        int* qbuf[4];
        for (int s = 0; s < 4; s++) qbuf[s] = (int*)malloc(sizeof(int) * CAP);
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
                        const int sb = (y >> 2) * 2 + (x >> 2);
                        if (n > nsb[sb]) nsb[sb] = n;
                        if (n > nq) nq = n;
                    }
                if (nq > e0 - s0) nq = e0 - s0;
                if (!nq) continue;
                a[6] += 1;
                int head = 0, tail = 0;
                int qh[4] = {0, 0, 0, 0}, qt[4] = {0, 0, 0, 0};
                const int steps = (nq + STEP - 1) / STEP;
                for (int st = 0; st < steps; st++) {
                    a[0] += 1;
                    for (int l = STEP - 1; l >= 0; l--) {
                        const int pos = (steps - 1 - st) * STEP + l;
                        if (pos >= (int)nq) continue;
                        const uint32_t g = ids[s0 + pos];
                        const float* m = means2D + 2 * g;
                        const float* co = conic_opacity + 4 * g;
                        if (!(co[3] >= 1.f / 255.f)) continue;
                        const float tau2 = 2.f * logf(255.f * co[3]) * 1.01f + 0.05f;
                        if (!(min_form_rect(m[0], m[1], co[0], co[1], co[2], qx0, qx0 + 7, qy0, qy0 + 7) <= tau2)) continue;
                        int mask = 0;
                        for (int s = 0; s < 4; s++) {
                            const float x0 = qx0 + 4 * (s & 1), y0 = qy0 + 4 * (s >> 1);
                            if ((uint32_t)pos < nsb[s] && min_form_rect(m[0], m[1], co[0], co[1], co[2], x0, x0 + 3, y0, y0 + 3) <= tau2) mask |= 1 << s;
                        }
                        if (!mask) continue;
                        a[3] += 1;
                        refc[head & (CAP - 1)] = __builtin_popcount(mask);
                        for (int s = 0; s < 4; s++) if (mask >> s & 1) qbuf[s][qh[s]++ & (CAP - 1)] = head;
                        head++;
                    }
                    for (;;) {
                        /* rounds while some row has a full batch */
                        int any = 0;
                        for (int s = 0; s < 4; s++) if (qh[s] - qt[s] >= 16) any = 1;
                        const int last = st + 1 == steps;
                        int force = 0;
                        while (tail < head && refc[tail & (CAP - 1)] == 0) tail++;
                        if (!any && !last && R - (head - tail) < STEP) force = 1;
                        if (!any && !force && !(last && (qh[0] > qt[0] || qh[1] > qt[1] || qh[2] > qt[2] || qh[3] > qt[3]))) break;
                        a[1] += 1;
                        if (force) a[5] += 1;
                        for (int s = 0; s < 4; s++) {
                            int n = qh[s] - qt[s];
                            if (n >= 16) n = 16;
                            else if (!(force || last)) n = 0;
                            if (force && !any) {
                                /* only rows that hold the oldest entries need to run; model: all rows flush what they have */
                            }
                            if (n > 0) { a[2] += 1; a[4] += n; }
                            for (int k = 0; k < n; k++) refc[qbuf[s][qt[s]++ & (CAP - 1)] & (CAP - 1)]--;
                        }
                    }
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
