/*
 * raster_oracle.c -- CPU restatement of the EMD street-Gaussian hot path.   TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this
 * file.  The product path (emd_amd/) never touches it and fails loudly without its HIP extension.
 *
 * PARITY STATUS
 *   - Everything the reference holds in Python for this path (camera matrices, cov3D, SH colour,
 *     projection, rigid-actor transform, quaternion algebra) is restated here and pinned by golden
 *     vectors generated from the imported reference (tests/golden/, made by tests/gen_golden.py):
 *       cov3D        S3Gaussian/utils/general_utils.py:245-277, scene/gaussian_model.py:34-38
 *       SH colour    S3Gaussian/utils/sh_utils.py:57-112, gaussian_renderer/__init__.py:19-25
 *       projection   S3Gaussian/utils/graphics_utils.py:42-49 (row-vector matrices, +1e-7 on w)
 *       rigid motion OmniRe/models/nodes/rigid.py:478-568, models/gaussians/basics.py:30-49,100-110
 *   - The tile rasterizer itself (cull, EWA cov2D, radius/rect, key layout, compositing, backward) is
 *     NOT in /root/reference: it is the un-vendored, un-pinned third-party `diff_gauss`
 *     (github.com/slothfulxtx/diff-gaussian-rasterization, imported at
 *     S3Gaussian/gaussian_renderer/__init__.py:14) / `gsplat` (OmniRe/models/gaussians/basics.py:12).
 *     The reference has no tests, golden vectors or fixtures for it.  PARITY UNPINNED for that part:
 *     this file restates the published 3D Gaussian Splatting rasterization algorithm (Kerbl et al.
 *     2023) anchored on the reference's call sites and consumers
 *     (gaussian_renderer/__init__.py:49-62,145-168; train.py:348-368; gaussian_model.py:728-730).
 *
 * All arithmetic is fp32 in a pinned evaluation order (compile with -ffp-contract=off): the
 * sort keys (tile_id << 32 | float bits of view depth), radii and tile rectangles are a bit-exact
 * contract between this file and the HIP kernels, and so is the FORWARD IMAGE (pinned exp and FMA placement, see
 * "Pinned compositing arithmetic" below); gradients are a tolerance contract.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define ACTOR_STRIDE 12

enum { F_NORMAL = 1, F_MOTION = 2, F_ABSGRAD = 4, F_CLAMP01 = 16 };

typedef struct {
    int32_t H, W;
    float tanfovx, tanfovy;
    float bg[3];
    float scale_modifier;
    float view[16];
    float proj[16];
    int32_t sh_degree;
    float campos[3];
    float near_plane;
} OrcSettings;

/* SH constants, S3Gaussian/utils/sh_utils.py:26-43 (rounded to fp32) */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

static void quat_to_R(const float q[4], float R[9]) {
    /* general_utils.py:245-266 / basics.py:30-49 (q already unit; no renormalisation here) */
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z);
    R[1] = 2.f * (x * y - r * z);
    R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);
    R[4] = 1.f - 2.f * (x * x + z * z);
    R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);
    R[7] = 2.f * (y * z + r * x);
    R[8] = 1.f - 2.f * (x * x + y * y);
}

static void quat_mul(const float a[4], const float b[4], float o[4]) {
    /* basics.py:100-110 */
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

static float quat_norm(const float q[4]) { return sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]); }

/* ------------------------------------------------------------------------------------------------
 * Explicit motion (EMD): world mean / quaternion / opacity of Gaussian i.
 * rigid.py:478-538 (means), :540-568 (quats), :589-591 (opacity x valid), deformable.py:57-69 (residual)
 * ---------------------------------------------------------------------------------------------- */
static void motion_point(int i, const float* means, const float* quats, const float* opac, const int32_t* actor_id,
                         const float* pose, const float* rdx, const float* rdq, float wm[3], float wq[4], float* wo) {
    float m[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    if (rdx) { m[0] += rdx[3 * i]; m[1] += rdx[3 * i + 1]; m[2] += rdx[3 * i + 2]; }
    int a = actor_id ? actor_id[i] : -1;
    if (a < 0) {
        wm[0] = m[0]; wm[1] = m[1]; wm[2] = m[2];
        if (quats) for (int k = 0; k < 4; k++) wq[k] = quats[4 * i + k];
        if (opac) *wo = opac[i];
        return;
    }
    const float* P = pose + (size_t)a * ACTOR_STRIDE;
    float R[9];
    quat_to_R(P, R);
    wm[0] = ((R[0] * m[0] + R[1] * m[1]) + R[2] * m[2]) + P[4];
    wm[1] = ((R[3] * m[0] + R[4] * m[1]) + R[5] * m[2]) + P[5];
    wm[2] = ((R[6] * m[0] + R[7] * m[1]) + R[8] * m[2]) + P[6];
    if (quats) {
        float ql[4], qn[4], p[4];
        for (int k = 0; k < 4; k++) ql[k] = quats[4 * i + k] + (rdq ? rdq[4 * i + k] : 0.f);
        float n = fmaxf(quat_norm(ql), 1e-12f); /* F.normalize eps, vanilla.py:145-146 */
        for (int k = 0; k < 4; k++) qn[k] = ql[k] / n;
        quat_mul(P + 8, qn, p);
        float n2 = fmaxf(quat_norm(p), 1e-12f);
        for (int k = 0; k < 4; k++) wq[k] = p[k] / n2;
    }
    if (opac) *wo = opac[i] * P[7];
}

int orc_motion_forward(int N, const float* means, const float* quats, const float* opac, const int32_t* actor_id,
                       const float* pose, const float* rdx, const float* rdq, float* wmeans, float* wquats,
                       float* wopac) {
    for (int i = 0; i < N; i++) {
        float wm[3], wq[4], wo = 0.f;
        motion_point(i, means, quats, opac, actor_id, pose, rdx, rdq, wm, wq, &wo);
        if (wmeans) memcpy(wmeans + 3 * i, wm, 12);
        if (wquats && quats) memcpy(wquats + 4 * i, wq, 16);
        if (wopac && opac) wopac[i] = wo;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * SH colour (no +0.5 / clamp).  sh_utils.py:57-112, evaluated per channel in the reference's order.
 * coeffs: [K][3] for this Gaussian (layout of pc.get_features, gaussian_renderer/__init__.py:20)
 * ---------------------------------------------------------------------------------------------- */
static void sh_basis(int deg, const float d[3], float b[16]) {
    float x = d[0], y = d[1], z = d[2];
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.f * zz - xx - yy);
            b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (3.f * xx - yy);
                b[10] = SH_C3[1] * xy * z;
                b[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                b[14] = SH_C3[5] * z * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}

static void sh_eval(int deg, const float d[3], const float* sh /*[K][3]*/, float out[3]) {
    float b[16];
    sh_basis(deg, d, b);
    int K = (deg + 1) * (deg + 1);
    for (int c = 0; c < 3; c++) {
        float r = 0.f;
        for (int k = 0; k < K; k++) r += b[k] * sh[3 * k + c];
        out[c] = r;
    }
}

int orc_sh_forward(int N, int deg, int M, const float* dirs, const float* coeffs, float* rgb) {
    for (int i = 0; i < N; i++) {
        float d[3] = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
        float n = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        d[0] /= n; d[1] /= n; d[2] /= n;
        sh_eval(deg, d, coeffs + (size_t)i * M * 3, rgb + 3 * i);
    }
    return 0;
}

/* cov3D from activated scale and unit quaternion: Sigma = L L^T, L = R diag(mod * s)
 * gaussian_model.py:34-38, general_utils.py:231-277; six upper-triangular values xx,xy,xz,yy,yz,zz */
static void cov3d_from_sr(const float s[3], float mod, const float q[4], float c[6]) {
    float R[9], L[9];
    quat_to_R(q, R);
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) L[3 * r + k] = R[3 * r + k] * (mod * s[k]);
    int o = 0;
    for (int r = 0; r < 3; r++)
        for (int k = r; k < 3; k++)
            c[o++] = (L[3 * r] * L[3 * k] + L[3 * r + 1] * L[3 * k + 1]) + L[3 * r + 2] * L[3 * k + 2];
}

int orc_cov3d(int N, const float* scales, float mod, const float* rots, float* cov) {
    for (int i = 0; i < N; i++) cov3d_from_sr(scales + 3 * i, mod, rots + 4 * i, cov + 6 * i);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Pinned compositing arithmetic (bit-exact contract with render.hip for the FORWARD image).
 * A pixel's colour depends discontinuously on alpha >= 1/255 and T (1 - alpha) >= 1e-4: with two different exp
 * implementations a handful of the 1.7 M pixels of a 1066x1600 image flip and move by ~1/255.  Hence exp and every
 * rounding of the compositing loop are specified, not left to libm / the compiler:
 *   power = fma(-0.5, fma(A dx, dx, (C dy) dy), -((B dx) dy))
 *   exp(x) = ldexp(P6(t - n), n), t = x * log2(e), n = rint(t), P6 = degree-6 Taylor of 2^f in Horner/FMA form
 *   w = alpha T; acc = fma(feature, w, acc); out = fma(T_final, bg, acc)
 * ---------------------------------------------------------------------------------------------- */
static float gauss_power(float A, float B, float C, float dx, float dy) {
    const float q = fmaf(A * dx, dx, (C * dy) * dy);
    return fmaf(-0.5f, q, -((B * dx) * dy));
}

static float pinned_exp(float x) {
    const float t = x * 1.44269504088896341f;
    const float n = rintf(t);
    const float f = t - n;
    float p = 1.54035304e-4f;
    p = fmaf(p, f, 1.33335581e-3f);
    p = fmaf(p, f, 9.61812911e-3f);
    p = fmaf(p, f, 5.55041087e-2f);
    p = fmaf(p, f, 2.40226507e-1f);
    p = fmaf(p, f, 6.93147181e-1f);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}

/* clamp-then-truncate of a tile coordinate (identical to min(grid, max(0, (int)f)) for finite f) */
static int tile_clamp(float f, int grid) {
    float g = (float)grid;
    if (!(f > 0.f)) return 0;
    if (f > g) return grid;
    return (int)f;
}

/* ------------------------------------------------------------------------------------------------
 * K1: per-Gaussian preprocess.  [UPSTREAM 3DGS forward preprocess, SURVEY appendix A steps 1-8]
 * outputs (all per Gaussian): means2D[2] pixel coords, depth, conic_opacity[4], rgb[3], normal[3] (view space),
 * radii, tiles_touched, rect[4] (xmin,ymin,xmax,ymax), clamped[3] (SH colour clamp mask)
 * ---------------------------------------------------------------------------------------------- */
int orc_preprocess(const OrcSettings* S, int N, int M, int flags, const float* means3D, const float* shs,
                   const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                   const float* cov3D_precomp, const int32_t* actor_id, const float* pose, const float* rdx,
                   const float* rdq, float* means2D, float* depths, float* conic_opacity, float* rgb, float* normal,
                   int32_t* radii, uint32_t* tiles_touched, int32_t* rect, uint8_t* clamped, float* cov3D_out) {
    const float* V = S->view;
    const float* P = S->proj;
    const int W = S->W, H = S->H;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const float fx = (float)W / (2.f * S->tanfovx), fy = (float)H / (2.f * S->tanfovy);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        radii[i] = 0; tiles_touched[i] = 0;
        if (rect) { rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0; }
        if (clamped) clamped[3 * i] = clamped[3 * i + 1] = clamped[3 * i + 2] = 0;
        float m[3], q[4] = {1, 0, 0, 0}, op = opacities[i];
        if (flags & F_MOTION) {
            motion_point(i, means3D, rotations, opacities, actor_id, pose, rdx, rdq, m, q, &op);
        } else {
            m[0] = means3D[3 * i]; m[1] = means3D[3 * i + 1]; m[2] = means3D[3 * i + 2];
            if (rotations) memcpy(q, rotations + 4 * i, 16);
        }
        /* 1. view space, row-vector convention (cameras.py:61): t = [m 1] @ viewmatrix */
        float tx = ((V[0] * m[0] + V[4] * m[1]) + V[8] * m[2]) + V[12];
        float ty = ((V[1] * m[0] + V[5] * m[1]) + V[9] * m[2]) + V[13];
        float tz = ((V[2] * m[0] + V[6] * m[1]) + V[10] * m[2]) + V[14];
        if (!(tz > S->near_plane)) continue;
        /* 2. clip space (graphics_utils.py:42-49) */
        float hx = ((P[0] * m[0] + P[4] * m[1]) + P[8] * m[2]) + P[12];
        float hy = ((P[1] * m[0] + P[5] * m[1]) + P[9] * m[2]) + P[13];
        float hw = ((P[3] * m[0] + P[7] * m[1]) + P[11] * m[2]) + P[15];
        float pw = 1.f / (hw + 0.0000001f);
        float px = hx * pw, py = hy * pw;
        /* 3. cov3D */
        float c3[6];
        if (cov3D_precomp) memcpy(c3, cov3D_precomp + 6 * i, 24);
        else cov3d_from_sr(scales + 3 * i, S->scale_modifier, q, c3);
        if (cov3D_out) memcpy(cov3D_out + 6 * i, c3, 24);
        /* 4. EWA cov2D = (J Wv) Sigma (J Wv)^T + 0.3 I, with tx/tz, ty/tz clamped to 1.3 tanfov */
        float limx = 1.3f * S->tanfovx, limy = 1.3f * S->tanfovy;
        float txtz = tx / tz, tytz = ty / tz;
        float cx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        float cy = fminf(limy, fmaxf(-limy, tytz)) * tz;
        float J00 = fx / tz, J02 = -(fx * cx) / (tz * tz), J11 = fy / tz, J12 = -(fy * cy) / (tz * tz);
        /* Wv[r][k] = V[4k + r] (world->view rotation); M = J Wv (2x3) */
        float M0[3], M1[3];
        for (int k = 0; k < 3; k++) {
            M0[k] = J00 * V[4 * k + 0] + J02 * V[4 * k + 2];
            M1[k] = J11 * V[4 * k + 1] + J12 * V[4 * k + 2];
        }
        float Sg[9] = {c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]};
        float T0[3], T1[3];
        for (int k = 0; k < 3; k++) {
            T0[k] = (M0[0] * Sg[k] + M0[1] * Sg[3 + k]) + M0[2] * Sg[6 + k];
            T1[k] = (M1[0] * Sg[k] + M1[1] * Sg[3 + k]) + M1[2] * Sg[6 + k];
        }
        float a = ((T0[0] * M0[0] + T0[1] * M0[1]) + T0[2] * M0[2]) + 0.3f;
        float b = (T0[0] * M1[0] + T0[1] * M1[1]) + T0[2] * M1[2];
        float c = ((T1[0] * M1[0] + T1[1] * M1[1]) + T1[2] * M1[2]) + 0.3f;
        /* 5. conic, radius */
        float det = a * c - b * b;
        if (det == 0.f) continue;
        float det_inv = 1.f / det;
        float conA = c * det_inv, conB = -b * det_inv, conC = a * det_inv;
        float mid = 0.5f * (a + c);
        float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        float lam1 = mid + sq, lam2 = mid - sq;
        float rad = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
        /* 6. pixel centre and tile rectangle */
        float ix = ((px + 1.f) * (float)W - 1.f) * 0.5f;
        float iy = ((py + 1.f) * (float)H - 1.f) * 0.5f;
        int x0 = tile_clamp((ix - rad) / (float)TILE, gx);
        int y0 = tile_clamp((iy - rad) / (float)TILE, gy);
        int x1 = tile_clamp((ix + rad + (float)(TILE - 1)) / (float)TILE, gx);
        int y1 = tile_clamp((iy + rad + (float)(TILE - 1)) / (float)TILE, gy);
        int area = (x1 - x0) * (y1 - y0);
        if (area <= 0) continue;
        /* 7. colour */
        float col[3];
        if (colors_precomp) {
            col[0] = colors_precomp[3 * i]; col[1] = colors_precomp[3 * i + 1]; col[2] = colors_precomp[3 * i + 2];
        } else {
            float d[3] = {m[0] - S->campos[0], m[1] - S->campos[1], m[2] - S->campos[2]};
            float n = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            d[0] /= n; d[1] /= n; d[2] /= n;
            sh_eval(S->sh_degree, d, shs + (size_t)i * M * 3, col);
            for (int ch = 0; ch < 3; ch++) {
                col[ch] += 0.5f;
                if (col[ch] < 0.f) { col[ch] = 0.f; if (clamped) clamped[3 * i + ch] = 1; }
                if ((flags & F_CLAMP01) && col[ch] > 1.f) { col[ch] = 1.f; if (clamped) clamped[3 * i + ch] = 1; }
            }
        }
        /* normal: shortest principal axis in view space, flipped to face the camera [UPSTREAM diff_gauss fork;
         * only visualised by the reference, S3Gaussian/utils/scene_utils.py:24-25] */
        float nv[3] = {0, 0, 0};
        if ((flags & F_NORMAL) && scales) {
            const float* s = scales + 3 * i;
            int ax = 0;
            if (s[1] < s[ax]) ax = 1;
            if (s[2] < s[ax]) ax = 2;
            float R[9];
            quat_to_R(q, R);
            float nw[3] = {R[ax], R[3 + ax], R[6 + ax]};
            nv[0] = (V[0] * nw[0] + V[4] * nw[1]) + V[8] * nw[2];
            nv[1] = (V[1] * nw[0] + V[5] * nw[1]) + V[9] * nw[2];
            nv[2] = (V[2] * nw[0] + V[6] * nw[1]) + V[10] * nw[2];
            float dp = (nv[0] * tx + nv[1] * ty) + nv[2] * tz;
            if (dp > 0.f) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
        }
        depths[i] = tz;
        radii[i] = (int32_t)rad;
        means2D[2 * i] = ix; means2D[2 * i + 1] = iy;
        conic_opacity[4 * i] = conA; conic_opacity[4 * i + 1] = conB; conic_opacity[4 * i + 2] = conC;
        conic_opacity[4 * i + 3] = op;
        rgb[3 * i] = col[0]; rgb[3 * i + 1] = col[1]; rgb[3 * i + 2] = col[2];
        if (normal) { normal[3 * i] = nv[0]; normal[3 * i + 1] = nv[1]; normal[3 * i + 2] = nv[2]; }
        tiles_touched[i] = (uint32_t)area;
        if (rect) { rect[4 * i] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1; }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * K2-K5: duplicate with keys, stable sort, tile ranges.  key = tile_id << 32 | float bits of depth.
 * Order contract: ascending (key, Gaussian id) -- what a stable sort of the id-ordered duplicate list gives.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t key; uint32_t id; } KV;
static int kv_cmp(const void* a, const void* b) {
    const KV* x = (const KV*)a; const KV* y = (const KV*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}

int64_t orc_bin(const OrcSettings* S, int N, const float* depths, const uint32_t* tiles_touched, const int32_t* rect,
                int64_t capacity, uint64_t* keys, uint32_t* ids, uint32_t* ranges) {
    const int gx = (S->W + TILE - 1) / TILE, gy = (S->H + TILE - 1) / TILE;
    int64_t D = 0;
    for (int i = 0; i < N; i++) D += tiles_touched[i];
    if (!keys) return D;
    if (D > capacity) return -D;
    KV* kv = (KV*)malloc(sizeof(KV) * (size_t)(D > 0 ? D : 1));
    int64_t o = 0;
    for (int i = 0; i < N; i++) {
        if (!tiles_touched[i]) continue;
        uint32_t db; memcpy(&db, depths + i, 4);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; y++)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; x++) {
                kv[o].key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | db;
                kv[o].id = (uint32_t)i; o++;
            }
    }
    qsort(kv, (size_t)D, sizeof(KV), kv_cmp);
    for (int64_t k = 0; k < D; k++) { keys[k] = kv[k].key; ids[k] = kv[k].id; }
    free(kv);
    for (int t = 0; t < gx * gy; t++) ranges[2 * t] = ranges[2 * t + 1] = 0;
    for (int64_t k = 0; k < D; k++) {
        uint32_t t = (uint32_t)(keys[k] >> 32);
        if (k == 0 || (uint32_t)(keys[k - 1] >> 32) != t) ranges[2 * t] = (uint32_t)k;
        if (k == D - 1 || (uint32_t)(keys[k + 1] >> 32) != t) ranges[2 * t + 1] = (uint32_t)(k + 1);
    }
    return D;
}

/* ------------------------------------------------------------------------------------------------
 * K6: per-pixel front-to-back compositing.  [UPSTREAM render forward, SURVEY appendix A]
 * out_color[3,H,W] (+ T_final * bg), out_depth = sum w z, out_normal = sum w n, out_alpha = 1 - T_final,
 * n_contrib[H,W] = 1 + index of the last list entry that contributed, final_T[H,W]
 * ---------------------------------------------------------------------------------------------- */
int orc_render_forward(const OrcSettings* S, int flags, const uint32_t* ranges, const uint32_t* ids,
                       const float* means2D, const float* conic_opacity, const float* rgb, const float* depths,
                       const float* normal, float* out_color, float* out_depth, float* out_normal, float* out_alpha,
                       uint32_t* n_contrib, float* final_T) {
    const int W = S->W, H = S->H, gx = (W + TILE - 1) / TILE;
    /* pixels are independent: rows are dealt to the OpenMP threads (results do not depend on the thread count) */
#pragma omp parallel for schedule(dynamic, 2)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            int t = (py / TILE) * gx + px / TILE;
            float T = 1.f, C[3] = {0, 0, 0}, Dp = 0.f, Nn[3] = {0, 0, 0};
            uint32_t contributor = 0, last = 0;
            for (uint32_t k = ranges[2 * t]; k < ranges[2 * t + 1]; k++) {
                contributor++;
                uint32_t g = ids[k];
                float dx = means2D[2 * g] - (float)px, dy = means2D[2 * g + 1] - (float)py;
                const float* co = conic_opacity + 4 * g;
                float power = gauss_power(co[0], co[1], co[2], dx, dy);
                if (power > 0.f) continue;
                float alpha = fminf(0.99f, co[3] * pinned_exp(power));
                if (alpha < 1.f / 255.f) continue;
                float test_T = T * (1.f - alpha);
                if (test_T < 0.0001f) break;
                float w = alpha * T;
                for (int ch = 0; ch < 3; ch++) C[ch] = fmaf(rgb[3 * g + ch], w, C[ch]);
                Dp = fmaf(depths[g], w, Dp);
                if (flags & F_NORMAL) for (int ch = 0; ch < 3; ch++) Nn[ch] = fmaf(normal[3 * g + ch], w, Nn[ch]);
                T = test_T;
                last = contributor;
            }
            size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
            for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = fmaf(T, S->bg[ch], C[ch]);
            out_depth[pix] = Dp;
            if ((flags & F_NORMAL) && out_normal) for (int ch = 0; ch < 3; ch++) out_normal[ch * HW + pix] = Nn[ch];
            out_alpha[pix] = 1.f - T;
            n_contrib[pix] = last;
            final_T[pix] = T;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Checker of the product's footprint culling (round 4; upstream has no such step): for every entry of the sorted list, by brute force
 * over the pixels of its tile that lie inside the image, the 4-bit mask of the 8x8 quadrants (bit qy * 2 + qx) in which AT LEAST ONE
 * pixel passes the render loop's own skip tests (power <= 0 and alpha >= 1/255, evaluated exactly as orc_render_forward does).
 * The product may drop an entry from its list only when this mask is 0, and must set every bit that is set here.
 * ---------------------------------------------------------------------------------------------- */
int orc_pair_quadrant_hits(const OrcSettings* S, int64_t D, const uint64_t* keys, const uint32_t* ids, const float* means2D,
                           const float* conic_opacity, uint8_t* out_mask) {
    const int W = S->W, H = S->H, gx = (W + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t k = 0; k < D; k++) {
        const uint32_t t = (uint32_t)(keys[k] >> 32), g = ids[k];
        const int x0 = (int)(t % (uint32_t)gx) * TILE, y0 = (int)(t / (uint32_t)gx) * TILE;
        const float* co = conic_opacity + 4 * g;
        uint8_t m = 0;
        for (int py = y0; py < y0 + TILE && py < H; py++)
            for (int px = x0; px < x0 + TILE && px < W; px++) {
                const uint8_t bit = (uint8_t)(1u << (((py - y0) >> 3) * 2 + ((px - x0) >> 3)));
                if (m & bit) continue;
                float dx = means2D[2 * g] - (float)px, dy = means2D[2 * g + 1] - (float)py;
                float power = gauss_power(co[0], co[1], co[2], dx, dy);
                if (power > 0.f) continue;
                float alpha = fminf(0.99f, co[3] * pinned_exp(power));
                if (alpha < 1.f / 255.f) continue;
                m |= bit;
            }
        out_mask[k] = m;
    }
    return 0;
}

static int g_accumulate_fp32 = 0;
void orc_set_accumulate_fp32(int on) { g_accumulate_fp32 = on; }

/* ------------------------------------------------------------------------------------------------
 * K7: render backward (back-to-front replay).  [UPSTREAM render backward]
 * Per-Gaussian accumulators (zeroed here):
 *   g_mean2D[N,2]  dL/d(pixel-space mean), NOT yet scaled to NDC units
 *   g_abs[N,2]     sum over pixels of |dL/d mean| (gsplat absgrad), or NULL
 *   g_conic[N,3]   dL/d(A,B,C) with power = -0.5 (A dx^2 + C dy^2) - B dx dy
 *   g_opacity[N], g_rgb[N,3], g_depth[N], g_normal[N,3]
 * The alpha clamp min(0.99, .) is passed straight through, as in the upstream implementation.
 * ---------------------------------------------------------------------------------------------- */
int orc_render_backward(const OrcSettings* S, int N, int flags, const uint32_t* ranges, const uint32_t* ids,
                        const float* means2D, const float* conic_opacity, const float* rgb, const float* depths,
                        const float* normal, const uint32_t* n_contrib, const float* final_T, const float* dL_dcolor,
                        const float* dL_ddepth, const float* dL_dalpha, const float* dL_dnormal, float* g_mean2D,
                        float* g_abs, float* g_conic, float* g_opacity, float* g_rgb, float* g_depth,
                        float* g_normal) {
    const int W = S->W, H = S->H, gx = (W + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
    /* The blend weights are the forward pass's fp32 values; the per-pixel recurrences built from them and the sums over
     * pixels (15 accumulators per Gaussian) are evaluated in double, so that the result neither depends on the order in
     * which the OpenMP threads reach a Gaussian nor carries the rounding of one particular fp32 evaluation order: the
     * oracle is the derivative of the forward function as evaluated, rounded once at the end. */
    enum { A_MX = 0, A_MY, A_AX, A_AY, A_CA, A_CB, A_CC, A_OP, A_R, A_G, A_B, A_D, A_N0, A_N1, A_N2, A_STRIDE };
    double* acc = (double*)calloc((size_t)(N > 0 ? N : 1) * A_STRIDE, sizeof(double));
    if (!acc) return -1;
    /* orc_set_accumulate_fp32(1): sum the partials in fp32, sequentially in pixel order -- the arithmetic of an fp32 atomicAdd
     * implementation in ONE particular order (diagnostic: how much of a deviation is summation order). */
    const int acc32 = g_accumulate_fp32;
#define ACC(g, k, v) do { if (acc32) { float* a_ = (float*)acc + (size_t)(g) * A_STRIDE + (k); *a_ += (v); } \
                          else { const double v_ = (double)(v); _Pragma("omp atomic") acc[(size_t)(g) * A_STRIDE + (k)] += v_; } } while (0)
#pragma omp parallel for schedule(dynamic, 2) if (!acc32)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            size_t pix = (size_t)py * W + px;
            int t = (py / TILE) * gx + px / TILE;
            uint32_t start = ranges[2 * t];
            float dC[3] = {0, 0, 0}, dD = 0.f, dA = 0.f, dN[3] = {0, 0, 0};
            if (dL_dcolor) for (int ch = 0; ch < 3; ch++) dC[ch] = dL_dcolor[ch * HW + pix];
            if (dL_ddepth) dD = dL_ddepth[pix];
            if (dL_dalpha) dA = dL_dalpha[pix];
            if (dL_dnormal && (flags & F_NORMAL)) for (int ch = 0; ch < 3; ch++) dN[ch] = dL_dnormal[ch * HW + pix];
            const double bgdot = ((double)S->bg[0] * dC[0] + (double)S->bg[1] * dC[1]) + (double)S->bg[2] * dC[2];
            /* The blend weights alpha_k are the fp32 values the forward pass used (same skip decisions, pinned exp); everything
             * built FROM them -- the transmittance T_k, the colour behind each Gaussian, dL/dalpha -- is evaluated in double.
             * Upstream rebuilds T_k in fp32 by repeated division from T_final; over a list thousands of entries deep that walk
             * drifts by several 1e-6 relative exactly where the weights are largest (the front of the list), which would make
             * this oracle LESS accurate than the kernel it checks (whose prefix product starts at the front).  The oracle is the
             * derivative of the forward function as the forward pass evaluated it, not a replay of one fp32 summation order. */
            double Tf = 1.0;
            for (int64_t k = (int64_t)start; k < (int64_t)start + (int64_t)n_contrib[pix]; k++) {
                uint32_t g = ids[k];
                float dx = means2D[2 * g] - (float)px, dy = means2D[2 * g + 1] - (float)py;
                const float* co = conic_opacity + 4 * g;
                float power = gauss_power(co[0], co[1], co[2], dx, dy);
                if (power > 0.f) continue;
                float alpha = fminf(0.99f, co[3] * pinned_exp(power));
                if (alpha < 1.f / 255.f) continue;
                Tf *= 1.0 - (double)alpha;
            }
            double T = Tf;
            /* colour behind the current Gaussian, per unit of transmittance after it */
            double accC[3] = {0, 0, 0}, accD = 0.0, accN[3] = {0, 0, 0};
            double last_alpha = 0.0, lastC[3] = {0, 0, 0}, lastD = 0.0, lastN[3] = {0, 0, 0};
            for (int64_t k = (int64_t)start + (int64_t)n_contrib[pix] - 1; k >= (int64_t)start; k--) {
                uint32_t g = ids[k];
                float dx = means2D[2 * g] - (float)px, dy = means2D[2 * g + 1] - (float)py;
                const float* co = conic_opacity + 4 * g;
                float power = gauss_power(co[0], co[1], co[2], dx, dy);
                if (power > 0.f) continue;
                float G = pinned_exp(power);
                float alpha = fminf(0.99f, co[3] * G);
                if (alpha < 1.f / 255.f) continue;
                T = T / (1.0 - (double)alpha);
                const double w = (double)alpha * T; /* d out / d feature */
                double dL_da = 0.0;
                for (int ch = 0; ch < 3; ch++) {
                    double c = rgb[3 * g + ch];
                    accC[ch] = last_alpha * lastC[ch] + (1.0 - last_alpha) * accC[ch];
                    lastC[ch] = c;
                    dL_da += (c - accC[ch]) * dC[ch];
                    ACC(g, A_R + ch, w * dC[ch]);
                }
                {
                    double z = depths[g];
                    accD = last_alpha * lastD + (1.0 - last_alpha) * accD;
                    lastD = z;
                    dL_da += (z - accD) * dD;
                    ACC(g, A_D, w * dD);
                }
                if (flags & F_NORMAL) for (int ch = 0; ch < 3; ch++) {
                    double n = normal[3 * g + ch];
                    accN[ch] = last_alpha * lastN[ch] + (1.0 - last_alpha) * accN[ch];
                    lastN[ch] = n;
                    dL_da += (n - accN[ch]) * dN[ch];
                    if (g_normal) ACC(g, A_N0 + ch, w * dN[ch]);
                }
                dL_da *= T;
                last_alpha = alpha;
                /* background and alpha image: both depend on T_final = prod (1 - alpha_i) */
                dL_da += (Tf / (1.0 - (double)alpha)) * ((double)dA - bgdot);
                const double dL_dG = (double)co[3] * dL_da;
                const double gdx = (double)G * dx, gdy = (double)G * dy;
                const double dG_ddx = -gdx * co[0] - gdy * co[1];
                const double dG_ddy = -gdy * co[2] - gdx * co[1];
                const double mx = dL_dG * dG_ddx, my = dL_dG * dG_ddy;
                ACC(g, A_MX, mx); ACC(g, A_MY, my);
                if (g_abs) { ACC(g, A_AX, fabs(mx)); ACC(g, A_AY, fabs(my)); }
                ACC(g, A_CA, -0.5 * gdx * dx * dL_dG);
                ACC(g, A_CB, -gdx * dy * dL_dG);
                ACC(g, A_CC, -0.5 * gdy * dy * dL_dG);
                ACC(g, A_OP, (double)G * dL_da);
            }
        }
#undef ACC
#pragma omp parallel for schedule(static)
    for (int g = 0; g < N; g++) {
        double a[A_STRIDE];
        for (int k = 0; k < A_STRIDE; k++) a[k] = acc32 ? (double)((const float*)acc)[(size_t)g * A_STRIDE + k] : acc[(size_t)g * A_STRIDE + k];
        g_mean2D[2 * g] = (float)a[A_MX]; g_mean2D[2 * g + 1] = (float)a[A_MY];
        if (g_abs) { g_abs[2 * g] = (float)a[A_AX]; g_abs[2 * g + 1] = (float)a[A_AY]; }
        g_conic[3 * g] = (float)a[A_CA]; g_conic[3 * g + 1] = (float)a[A_CB]; g_conic[3 * g + 2] = (float)a[A_CC];
        g_opacity[g] = (float)a[A_OP];
        g_rgb[3 * g] = (float)a[A_R]; g_rgb[3 * g + 1] = (float)a[A_G]; g_rgb[3 * g + 2] = (float)a[A_B];
        g_depth[g] = (float)a[A_D];
        if (g_normal) { g_normal[3 * g] = (float)a[A_N0]; g_normal[3 * g + 1] = (float)a[A_N1]; g_normal[3 * g + 2] = (float)a[A_N2]; }
    }
    free(acc);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * K8: preprocess backward.  Chain (g_mean2D, g_conic, g_opacity, g_rgb, g_depth) to the inputs.
 * dL_dmeans2D_out[N,3] = (0.5 W gx, 0.5 H gy, 0): the NDC-scaled pixel units consumed at
 * S3Gaussian/scene/gaussian_model.py:729.  Motion gradients: dL_dpose[A,12] (zeroed here), dL_drdx, dL_drdq.
 * ---------------------------------------------------------------------------------------------- */
static void dnormalize(const float v_unit[4], float n, const float g[4], float out[4]) {
    float dot = ((v_unit[0] * g[0] + v_unit[1] * g[1]) + v_unit[2] * g[2]) + v_unit[3] * g[3];
    for (int k = 0; k < 4; k++) out[k] = (g[k] - v_unit[k] * dot) / n;
}

static void dR_to_dq(const float q[4], const float dR[9], float dq[4]) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    dq[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    dq[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
    dq[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
    dq[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
}

int orc_preprocess_backward(const OrcSettings* S, int N, int M, int flags, const float* means3D, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales,
                            const float* rotations, const float* cov3D_precomp, const int32_t* actor_id,
                            const float* pose, int A, const float* rdx, const float* rdq, const int32_t* radii,
                            const uint8_t* clamped, const float* g_mean2D, const float* g_conic,
                            const float* g_opacity, const float* g_rgb, const float* g_depth, float* dL_dmeans3D,
                            float* dL_dmeans2D_out, float* dL_dshs, float* dL_dcolors, float* dL_dopacities,
                            float* dL_dscales, float* dL_drotations, float* dL_dcov3D, float* dL_dpose,
                            float* dL_drdx, float* dL_drdq) {
    const float* V = S->view;
    const float* P = S->proj;
    const int W = S->W, H = S->H;
    const float fx = (float)W / (2.f * S->tanfovx), fy = (float)H / (2.f * S->tanfovy);
    /* per-actor pose gradients: segmented sums over up to thousands of points, accumulated in double (order-free) */
    double* pose_acc = dL_dpose ? (double*)calloc((size_t)(A > 0 ? A : 1) * ACTOR_STRIDE, sizeof(double)) : NULL;
#define PACC(idx, v) do { const double v_ = (double)(v); _Pragma("omp atomic") pose_acc[idx] += v_; } while (0)
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        float dm[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0}, ds[3] = {0, 0, 0}, dop = 0.f, dc6[6] = {0, 0, 0, 0, 0, 0};
        if (dL_dmeans2D_out) dL_dmeans2D_out[3 * i] = dL_dmeans2D_out[3 * i + 1] = dL_dmeans2D_out[3 * i + 2] = 0.f;
        if (dL_dshs) memset(dL_dshs + (size_t)i * M * 3, 0, sizeof(float) * 3 * M);
        if (dL_dcolors) dL_dcolors[3 * i] = dL_dcolors[3 * i + 1] = dL_dcolors[3 * i + 2] = 0.f;
        float m[3], q[4] = {1, 0, 0, 0}, op = opacities[i];
        if (flags & F_MOTION) motion_point(i, means3D, rotations, opacities, actor_id, pose, rdx, rdq, m, q, &op);
        else {
            m[0] = means3D[3 * i]; m[1] = means3D[3 * i + 1]; m[2] = means3D[3 * i + 2];
            if (rotations) memcpy(q, rotations + 4 * i, 16);
        }
        if (radii[i] > 0) {
            float tx = ((V[0] * m[0] + V[4] * m[1]) + V[8] * m[2]) + V[12];
            float ty = ((V[1] * m[0] + V[5] * m[1]) + V[9] * m[2]) + V[13];
            float tz = ((V[2] * m[0] + V[6] * m[1]) + V[10] * m[2]) + V[14];
            /* (e) colour */
            if (colors_precomp) {
                if (dL_dcolors) for (int ch = 0; ch < 3; ch++) dL_dcolors[3 * i + ch] = g_rgb[3 * i + ch];
            } else {
                float d0[3] = {m[0] - S->campos[0], m[1] - S->campos[1], m[2] - S->campos[2]};
                float n = sqrtf((d0[0] * d0[0] + d0[1] * d0[1]) + d0[2] * d0[2]);
                float d[3] = {d0[0] / n, d0[1] / n, d0[2] / n};
                float gc[3];
                for (int ch = 0; ch < 3; ch++) gc[ch] = clamped[3 * i + ch] ? 0.f : g_rgb[3 * i + ch];
                float bs[16];
                sh_basis(S->sh_degree, d, bs);
                int K = (S->sh_degree + 1) * (S->sh_degree + 1);
                const float* sh = shs + (size_t)i * M * 3;
                if (dL_dshs) for (int k = 0; k < K; k++)
                    for (int ch = 0; ch < 3; ch++) dL_dshs[((size_t)i * M + k) * 3 + ch] = bs[k] * gc[ch];
                /* d colour / d dir by differentiating the basis polynomials */
                float x = d[0], y = d[1], z = d[2];
                float dbx[16] = {0}, dby[16] = {0}, dbz[16] = {0};
                if (S->sh_degree > 0) {
                    dby[1] = -SH_C1; dbz[2] = SH_C1; dbx[3] = -SH_C1;
                    if (S->sh_degree > 1) {
                        float xx = x * x, yy = y * y, zz = z * z;
                        dbx[4] = SH_C2[0] * y; dby[4] = SH_C2[0] * x;
                        dby[5] = SH_C2[1] * z; dbz[5] = SH_C2[1] * y;
                        dbx[6] = SH_C2[2] * -2.f * x; dby[6] = SH_C2[2] * -2.f * y; dbz[6] = SH_C2[2] * 4.f * z;
                        dbx[7] = SH_C2[3] * z; dbz[7] = SH_C2[3] * x;
                        dbx[8] = SH_C2[4] * 2.f * x; dby[8] = SH_C2[4] * -2.f * y;
                        if (S->sh_degree > 2) {
                            dbx[9] = SH_C3[0] * 6.f * x * y; dby[9] = SH_C3[0] * (3.f * xx - 3.f * yy);
                            dbx[10] = SH_C3[1] * y * z; dby[10] = SH_C3[1] * x * z; dbz[10] = SH_C3[1] * x * y;
                            dbx[11] = SH_C3[2] * -2.f * x * y; dby[11] = SH_C3[2] * (4.f * zz - xx - 3.f * yy);
                            dbz[11] = SH_C3[2] * 8.f * y * z;
                            dbx[12] = SH_C3[3] * -6.f * x * z; dby[12] = SH_C3[3] * -6.f * y * z;
                            dbz[12] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                            dbx[13] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); dby[13] = SH_C3[4] * -2.f * x * y;
                            dbz[13] = SH_C3[4] * 8.f * x * z;
                            dbx[14] = SH_C3[5] * 2.f * x * z; dby[14] = SH_C3[5] * -2.f * y * z;
                            dbz[14] = SH_C3[5] * (xx - yy);
                            dbx[15] = SH_C3[6] * (3.f * xx - 3.f * yy); dby[15] = SH_C3[6] * -6.f * x * y;
                        }
                    }
                }
                float gd[3] = {0, 0, 0};
                for (int k = 0; k < K; k++) {
                    float s = (sh[3 * k] * gc[0] + sh[3 * k + 1] * gc[1]) + sh[3 * k + 2] * gc[2];
                    gd[0] += dbx[k] * s; gd[1] += dby[k] * s; gd[2] += dbz[k] * s;
                }
                /* d (v/|v|) */
                float dot = (d[0] * gd[0] + d[1] * gd[1]) + d[2] * gd[2];
                for (int k = 0; k < 3; k++) dm[k] += (gd[k] - d[k] * dot) / n;
            }
            /* (a) conic -> cov2D */
            float c3[6];
            if (cov3D_precomp) memcpy(c3, cov3D_precomp + 6 * i, 24);
            else cov3d_from_sr(scales + 3 * i, S->scale_modifier, q, c3);
            float limx = 1.3f * S->tanfovx, limy = 1.3f * S->tanfovy;
            float txtz = tx / tz, tytz = ty / tz;
            int clx = (txtz < -limx) || (txtz > limx), cly = (tytz < -limy) || (tytz > limy);
            float cx = fminf(limx, fmaxf(-limx, txtz)) * tz;
            float cy = fminf(limy, fmaxf(-limy, tytz)) * tz;
            float J00 = fx / tz, J02 = -(fx * cx) / (tz * tz), J11 = fy / tz, J12 = -(fy * cy) / (tz * tz);
            float M0[3], M1[3];
            for (int k = 0; k < 3; k++) {
                M0[k] = J00 * V[4 * k + 0] + J02 * V[4 * k + 2];
                M1[k] = J11 * V[4 * k + 1] + J12 * V[4 * k + 2];
            }
            float Sg[9] = {c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]};
            float T0[3], T1[3]; /* Sigma M0^T, Sigma M1^T */
            for (int k = 0; k < 3; k++) {
                T0[k] = (M0[0] * Sg[k] + M0[1] * Sg[3 + k]) + M0[2] * Sg[6 + k];
                T1[k] = (M1[0] * Sg[k] + M1[1] * Sg[3 + k]) + M1[2] * Sg[6 + k];
            }
            float a = ((T0[0] * M0[0] + T0[1] * M0[1]) + T0[2] * M0[2]) + 0.3f;
            float b = (T0[0] * M1[0] + T0[1] * M1[1]) + T0[2] * M1[2];
            float c = ((T1[0] * M1[0] + T1[1] * M1[1]) + T1[2] * M1[2]) + 0.3f;
            float det = a * c - b * b;
            float gA = g_conic[3 * i], gB = g_conic[3 * i + 1], gC = g_conic[3 * i + 2];
            float da = 0, db = 0, dc = 0;
            if (det != 0.f) {
                float i2 = 1.f / (det * det);
                da = (-c * c * gA + b * c * gB - b * b * gC) * i2;
                db = (2.f * b * c * gA - (a * c + b * b) * gB + 2.f * a * b * gC) * i2;
                dc = (-b * b * gA + a * b * gB - a * a * gC) * i2;
            }
            /* (b) cov2D -> Sigma (6 unique entries) and -> M */
            dc6[0] = da * M0[0] * M0[0] + db * M0[0] * M1[0] + dc * M1[0] * M1[0];
            dc6[3] = da * M0[1] * M0[1] + db * M0[1] * M1[1] + dc * M1[1] * M1[1];
            dc6[5] = da * M0[2] * M0[2] + db * M0[2] * M1[2] + dc * M1[2] * M1[2];
            dc6[1] = 2.f * da * M0[0] * M0[1] + db * (M0[0] * M1[1] + M0[1] * M1[0]) + 2.f * dc * M1[0] * M1[1];
            dc6[2] = 2.f * da * M0[0] * M0[2] + db * (M0[0] * M1[2] + M0[2] * M1[0]) + 2.f * dc * M1[0] * M1[2];
            dc6[4] = 2.f * da * M0[1] * M0[2] + db * (M0[1] * M1[2] + M0[2] * M1[1]) + 2.f * dc * M1[1] * M1[2];
            float dM0[3], dM1[3];
            for (int k = 0; k < 3; k++) {
                dM0[k] = 2.f * da * T0[k] + db * T1[k];
                dM1[k] = 2.f * dc * T1[k] + db * T0[k];
            }
            float dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
            for (int k = 0; k < 3; k++) {
                dJ00 += dM0[k] * V[4 * k + 0]; dJ02 += dM0[k] * V[4 * k + 2];
                dJ11 += dM1[k] * V[4 * k + 1]; dJ12 += dM1[k] * V[4 * k + 2];
            }
            float tz2 = 1.f / (tz * tz), tz3 = tz2 / tz;
            float dtx = clx ? 0.f : -fx * tz2 * dJ02;
            float dty = cly ? 0.f : -fy * tz2 * dJ12;
            float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + 2.f * fx * cx * tz3 * dJ02 + 2.f * fy * cy * tz3 * dJ12;
            /* (d) depth */
            dtz += g_depth[i];
            for (int k = 0; k < 3; k++) dm[k] += V[4 * k + 0] * dtx + V[4 * k + 1] * dty + V[4 * k + 2] * dtz;
            /* (c) pixel mean -> clip -> world */
            float gxn = 0.5f * (float)W * g_mean2D[2 * i], gyn = 0.5f * (float)H * g_mean2D[2 * i + 1];
            if (dL_dmeans2D_out) { dL_dmeans2D_out[3 * i] = gxn; dL_dmeans2D_out[3 * i + 1] = gyn; }
            float hx = ((P[0] * m[0] + P[4] * m[1]) + P[8] * m[2]) + P[12];
            float hy = ((P[1] * m[0] + P[5] * m[1]) + P[9] * m[2]) + P[13];
            float hw = ((P[3] * m[0] + P[7] * m[1]) + P[11] * m[2]) + P[15];
            float pw = 1.f / (hw + 0.0000001f);
            float mul1 = hx * pw * pw, mul2 = hy * pw * pw;
            for (int k = 0; k < 3; k++)
                dm[k] += (P[4 * k] * pw - P[4 * k + 3] * mul1) * gxn + (P[4 * k + 1] * pw - P[4 * k + 3] * mul2) * gyn;
            /* (f) Sigma -> scale, quaternion */
            if (!cov3D_precomp) {
                float R[9], L[9];
                quat_to_R(q, R);
                const float* s = scales + 3 * i;
                float mod = S->scale_modifier;
                for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) L[3 * r + k] = R[3 * r + k] * (mod * s[k]);
                float Gf[9] = {dc6[0], 0.5f * dc6[1], 0.5f * dc6[2], 0.5f * dc6[1], dc6[3], 0.5f * dc6[4],
                               0.5f * dc6[2], 0.5f * dc6[4], dc6[5]};
                float dL[9], dR[9];
                for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++)
                    dL[3 * r + k] = 2.f * ((Gf[3 * r] * L[k] + Gf[3 * r + 1] * L[3 + k]) + Gf[3 * r + 2] * L[6 + k]);
                for (int k = 0; k < 3; k++) {
                    ds[k] = mod * ((dL[k] * R[k] + dL[3 + k] * R[3 + k]) + dL[6 + k] * R[6 + k]);
                    for (int r = 0; r < 3; r++) dR[3 * r + k] = dL[3 * r + k] * (mod * s[k]);
                }
                dR_to_dq(q, dR, dq);
            }
            dop = g_opacity[i];
        }
        if (dL_dcov3D) memcpy(dL_dcov3D + 6 * i, dc6, 24);
        if (dL_dscales) memcpy(dL_dscales + 3 * i, ds, 12);
        /* (h) explicit motion */
        int a_id = (flags & F_MOTION) && actor_id ? actor_id[i] : -1;
        if ((flags & F_MOTION) && a_id >= 0) {
            const float* Pp = pose + (size_t)a_id * ACTOR_STRIDE;
            float R[9];
            quat_to_R(Pp, R);
            float ml[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
            if (rdx) { ml[0] += rdx[3 * i]; ml[1] += rdx[3 * i + 1]; ml[2] += rdx[3 * i + 2]; }
            float dl[3];
            for (int k = 0; k < 3; k++) dl[k] = (R[k] * dm[0] + R[3 + k] * dm[1]) + R[6 + k] * dm[2];
            float dRm[9];
            for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) dRm[3 * r + k] = dm[r] * ml[k];
            float dqm[4];
            dR_to_dq(Pp, dRm, dqm);
            if (dL_dpose) {
                const size_t gp = (size_t)a_id * ACTOR_STRIDE;
                for (int k = 0; k < 4; k++) PACC(gp + k, dqm[k]);
                for (int k = 0; k < 3; k++) PACC(gp + 4 + k, dm[k]);
                PACC(gp + 7, dop * opacities[i]);
            }
            float dql[4] = {0, 0, 0, 0};
            if (rotations) {
                float ql[4], qn[4], p[4];
                for (int k = 0; k < 4; k++) ql[k] = rotations[4 * i + k] + (rdq ? rdq[4 * i + k] : 0.f);
                float n = fmaxf(quat_norm(ql), 1e-12f);
                for (int k = 0; k < 4; k++) qn[k] = ql[k] / n;
                quat_mul(Pp + 8, qn, p);
                float n2 = fmaxf(quat_norm(p), 1e-12f);
                float pu[4] = {p[0] / n2, p[1] / n2, p[2] / n2, p[3] / n2};
                float dp[4];
                dnormalize(pu, n2, dq, dp);
                /* p = a (x) b : dL/da = g (x) conj(b), dL/db = conj(a) (x) g */
                float bc[4] = {qn[0], -qn[1], -qn[2], -qn[3]}, ac[4] = {Pp[8], -Pp[9], -Pp[10], -Pp[11]};
                float dqa[4], dqb[4];
                quat_mul(dp, bc, dqa);
                quat_mul(ac, dp, dqb);
                if (dL_dpose) for (int k = 0; k < 4; k++) PACC((size_t)a_id * ACTOR_STRIDE + 8 + k, dqa[k]);
                dnormalize(qn, n, dqb, dql);
            }
            if (dL_dmeans3D) memcpy(dL_dmeans3D + 3 * i, dl, 12);
            if (dL_drdx) memcpy(dL_drdx + 3 * i, dl, 12);
            if (dL_drotations) memcpy(dL_drotations + 4 * i, dql, 16);
            if (dL_drdq) memcpy(dL_drdq + 4 * i, dql, 16);
            if (dL_dopacities) dL_dopacities[i] = dop * Pp[7];
        } else {
            if (dL_dmeans3D) memcpy(dL_dmeans3D + 3 * i, dm, 12);
            if (dL_drdx) memcpy(dL_drdx + 3 * i, dm, 12);
            if (dL_drotations) memcpy(dL_drotations + 4 * i, dq, 16);
            if (dL_drdq) memset(dL_drdq + 4 * i, 0, 16);
            if (dL_dopacities) dL_dopacities[i] = dop;
        }
    }
#undef PACC
    if (dL_dpose) {
        for (size_t k = 0; k < (size_t)A * ACTOR_STRIDE; k++) dL_dpose[k] = (float)pose_acc[k];
        free(pose_acc);
    }
    return 0;
}
