// preprocess.hip -- K1 (fused explicit motion + projection + EWA covariance + SH colour) and K8 (its backward),
// plus the stand-alone motion and SH entry points.  gfx950, one Gaussian per lane.
//
// Built with -ffp-contract=off: the view depth (sort key), radius and tile rectangle are a bit-exact
// contract with oracle/raster_oracle.c, so every operation here is an individually rounded fp32 op in
// the documented order ("pinned evaluation order").
//
// Reference behaviour this replaces (file:line):
//   activations + rasterizer preprocess   S3Gaussian/gaussian_renderer/__init__.py:99-101,145-155 [UPSTREAM K1]
//   cov3D                                 S3Gaussian/utils/general_utils.py:245-277, scene/gaussian_model.py:34-38
//   SH colour                             S3Gaussian/utils/sh_utils.py:57-112, gaussian_renderer/__init__.py:19-25
//   projection                            S3Gaussian/utils/graphics_utils.py:42-49
//   rigid actor motion + residual         OmniRe/models/nodes/rigid.py:478-568, deformable.py:57-69
#include "common.h"
#include "device_utils.h"

#pragma clang fp contract(off)

// Cache policy and traversal order (round 5; each a compile-time knob, A/B in profiles/r05_cache_policy_variants.txt).  A step moves ~2 GB
// through a 256 MiB Infinity Cache; what decides the two projection kernels' time is which of it is still there when they ask:
//   * streams that are touched once per step bypass the caches: the SH rows K1 reads (192 B per visible Gaussian, EMD_K1_NT_SH), the dense
//     dL/dshs rows K8 writes (384 MB, EMD_K8_NT_SH), the colour Jacobian K1 leaves for K8 a whole step's traffic later (EMD_K1_NT_JAC);
//   * K1 walks the Gaussians from the LAST block to the first (EMD_K1_REVERSE): the parameters K8 read last, at the end of the step
//     before, are the ones K1 asks for first;
//   * K8 takes the colour clamp bits from the Jacobian row it reads anyway (EMD_K8_BITS_IN_JAC) instead of fetching a 64-byte record for them.
// Measured together on one box: K1 0.159 -> 0.121 ms, K8 0.229 -> 0.214 ms, 748 -> 782 it/s.  Not kept: nontemporal loads of the Jacobian in
// K8 (+10 us there), nontemporal scalar stores of K8's small gradients (+8 us in K8 for -5 in K1), nontemporal stores of the projected
// records (K1 0.11 -> 0.19 ms: 48-byte pieces; and the render kernels do not care where the records come from: +2 us).
//   * K8 compacts the visible Gaussians of a 256-block onto its first lanes (EMD_K8_COMPACT) and sends the five small gradients through
//     the idle SH tile as whole 16-byte nontemporal stores (EMD_K8_STAGE_NT): K8 0.201 -> 0.193 ms, 799 -> 806 it/s.
#ifndef EMD_K1_NT_SH
#define EMD_K1_NT_SH 1
#endif
#ifndef EMD_K8_NT_SH
#define EMD_K8_NT_SH 1
#endif
#ifndef EMD_K8_NT_ALL
#define EMD_K8_NT_ALL 0          /* the small gradient outputs of K8 (means, scales, rotations, opacity, mean2D) */
#endif
#ifndef EMD_K8_COMPACT
#define EMD_K8_COMPACT 1         /* K8: the visible Gaussians of a 256-block are compacted onto its first lanes (full waves do the work, the others only write zero rows) */
#endif
#ifndef EMD_K8_STAGE_NT
#define EMD_K8_STAGE_NT 1        /* the staged small gradients leave as nontemporal stores */
#endif
#ifndef EMD_K8_STAGE_SMALL
#define EMD_K8_STAGE_SMALL 0     /* K8's five small gradients leave through LDS as whole 16-byte nontemporal stores */
#endif
#ifndef EMD_K8_NT_JAC
#define EMD_K8_NT_JAC 0          /* K8's read of the colour Jacobian */
#endif
#ifndef EMD_K8_BITS_IN_JAC
#define EMD_K8_BITS_IN_JAC 1     /* K8 takes the colour clamp bits from the Jacobian row (K1 stores them in its spare word) instead of fetching a 64-byte record for 4 bytes */
#endif
#ifndef EMD_K1_NT_JAC
#define EMD_K1_NT_JAC 1          /* K1's store of the colour Jacobian (read once, by K8, a whole step's traffic later) */
#endif
#ifndef EMD_K1_HOIST
#define EMD_K1_HOIST 1           /* round 5 (late): K1 issues every index-addressed load of a Gaussian together, before the first use */
#endif
#ifndef EMD_K1_REVERSE
#define EMD_K1_REVERSE 1         /* K1 walks the Gaussians from the last block to the first: what K8 touched last is what K1 reads first */
#endif
typedef float emd_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_f(float* p, float v) { if (EMD_K8_NT_ALL) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ float4 load_f4_nt(const float4* p) {
    const emd_v4f v = __builtin_nontemporal_load(reinterpret_cast<const emd_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_f4_nt(float4* p, float4 v) {
    __builtin_nontemporal_store((emd_v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<emd_v4f*>(p));
}

namespace {

__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                   -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                   0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                   -0.5900435899266435f};

__device__ __forceinline__ void quat_to_R(const float q[4], float R[9]) {
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

__device__ __forceinline__ void quat_mul(const float a[4], const float b[4], float o[4]) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

__device__ __forceinline__ float quat_norm(const float q[4]) {
    return sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
}

// World-space mean / quaternion / opacity of Gaussian i under the explicit-motion model.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// raw = EMD_FLAG_RAW_PARAMS: opacities are logits (sigmoid here), static quaternions are un-normalised (normalised
// here, F.normalize eps 1e-12) -- the activations of S3Gaussian/gaussian_renderer/__init__.py:99-101 fused in.
// The arithmetic of motion_point on values already in registers (round 5: K8 issues every load of a Gaussian together before any of them is used):
// m = local mean (residual_dx applied), a = actor id or -1, q = the stored quaternion, dq_res = OmniRe's quaternion residual (dynamic points),
// op_in = the stored opacity, p0 / p1 / p2 = the actor's pose rows (read when a >= 0 only).
__device__ __forceinline__ void motion_apply(const float m[3], int a, bool has_q, float4 q, bool has_dq, float4 dq_res, bool has_op, float op_in, float4 p0,
                                             float4 p1, float4 p2, bool raw, float wm[3], float wq[4], float* wo) {
    if (a < 0) {
        wm[0] = m[0]; wm[1] = m[1]; wm[2] = m[2];
        if (has_q) {
            wq[0] = q.x; wq[1] = q.y; wq[2] = q.z; wq[3] = q.w;
            if (raw) { const float n = fmaxf(quat_norm(wq), 1e-12f); wq[0] /= n; wq[1] /= n; wq[2] /= n; wq[3] /= n; }
        }
        if (has_op) *wo = raw ? sigmoidf_(op_in) : op_in;
        return;
    }
    const float qm[4] = {p0.x, p0.y, p0.z, p0.w};
    float R[9];
    quat_to_R(qm, R);
    wm[0] = ((R[0] * m[0] + R[1] * m[1]) + R[2] * m[2]) + p1.x;
    wm[1] = ((R[3] * m[0] + R[4] * m[1]) + R[5] * m[2]) + p1.y;
    wm[2] = ((R[6] * m[0] + R[7] * m[1]) + R[8] * m[2]) + p1.z;
    if (has_q) {
        float ql[4] = {q.x, q.y, q.z, q.w};
        if (has_dq) { ql[0] += dq_res.x; ql[1] += dq_res.y; ql[2] += dq_res.z; ql[3] += dq_res.w; }
        float n = fmaxf(quat_norm(ql), 1e-12f);
        float qn[4] = {ql[0] / n, ql[1] / n, ql[2] / n, ql[3] / n};
        const float qr[4] = {p2.x, p2.y, p2.z, p2.w};
        float p[4];
        quat_mul(qr, qn, p);
        float n2 = fmaxf(quat_norm(p), 1e-12f);
        wq[0] = p[0] / n2; wq[1] = p[1] / n2; wq[2] = p[2] / n2; wq[3] = p[3] / n2;
    }
    if (has_op) *wo = (raw ? sigmoidf_(op_in) : op_in) * p1.w;
}

__device__ __forceinline__ void motion_point(int i, const float* __restrict__ means, const float* __restrict__ quats,
                                             const float* __restrict__ opac, const EmdMotion& mo, float wm[3],
                                             float wq[4], float* wo, bool raw = false) {
    float m[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    if (mo.residual_dx) {
        m[0] += mo.residual_dx[3 * i]; m[1] += mo.residual_dx[3 * i + 1]; m[2] += mo.residual_dx[3 * i + 2];
    }
    int a = mo.actor_id ? mo.actor_id[i] : -1;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 q = z4, dqr = z4, p0 = z4, p1 = z4, p2 = z4;
    if (quats) q = *(const float4*)(quats + 4 * i);
    if (a >= 0) {
        const float4* Pp = (const float4*)(mo.actor_pose + (size_t)a * EMD_ACTOR_STRIDE);
        p0 = Pp[0]; p1 = Pp[1]; p2 = Pp[2];
        if (quats && mo.residual_dq) dqr = *(const float4*)(mo.residual_dq + 4 * i);
    }
    motion_apply(m, a, quats != nullptr, q, mo.residual_dq != nullptr, dqr, opac != nullptr, opac ? opac[i] : 0.f, p0, p1, p2, raw, wm, wq, wo);
}

__device__ __forceinline__ void sh_basis(int deg, const float d[3], float b[16]) {
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

__device__ __forceinline__ void cov3d_from_sr(const float s[3], float mod, const float q[4], float c[6]) {
    float R[9], L[9];
    quat_to_R(q, R);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) L[3 * r + k] = R[3 * r + k] * (mod * s[k]);
    c[0] = (L[0] * L[0] + L[1] * L[1]) + L[2] * L[2];
    c[1] = (L[0] * L[3] + L[1] * L[4]) + L[2] * L[5];
    c[2] = (L[0] * L[6] + L[1] * L[7]) + L[2] * L[8];
    c[3] = (L[3] * L[3] + L[4] * L[4]) + L[5] * L[5];
    c[4] = (L[3] * L[6] + L[4] * L[7]) + L[5] * L[8];
    c[5] = (L[6] * L[6] + L[7] * L[7]) + L[8] * L[8];
}

__device__ __forceinline__ int tile_clamp(float f, int grid) {
    float g = (float)grid;
    if (!(f > 0.f)) return 0;
    if (f > g) return grid;
    return (int)f;
}

// Shared by K1 and K8: everything up to cov2D for one visible Gaussian.
struct Proj {
    float tx, ty, tz, cx, cy;
    bool clx, cly;
    float M0[3], M1[3], T0[3], T1[3];
    float a, b, c, det;
};

__device__ __forceinline__ void project_cov2d(const EmdSettings& S, const float m[3], const float c3[6], float fx,
                                              float fy, Proj& p) {
    const float* V = S.viewmatrix;
    float limx = 1.3f * S.tanfovx, limy = 1.3f * S.tanfovy;
    float txtz = p.tx / p.tz, tytz = p.ty / p.tz;
    p.clx = (txtz < -limx) || (txtz > limx);
    p.cly = (tytz < -limy) || (tytz > limy);
    p.cx = fminf(limx, fmaxf(-limx, txtz)) * p.tz;
    p.cy = fminf(limy, fmaxf(-limy, tytz)) * p.tz;
    float J00 = fx / p.tz, J02 = -(fx * p.cx) / (p.tz * p.tz), J11 = fy / p.tz, J12 = -(fy * p.cy) / (p.tz * p.tz);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        p.M0[k] = J00 * V[4 * k + 0] + J02 * V[4 * k + 2];
        p.M1[k] = J11 * V[4 * k + 1] + J12 * V[4 * k + 2];
    }
    const float Sg[9] = {c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        p.T0[k] = (p.M0[0] * Sg[k] + p.M0[1] * Sg[3 + k]) + p.M0[2] * Sg[6 + k];
        p.T1[k] = (p.M1[0] * Sg[k] + p.M1[1] * Sg[3 + k]) + p.M1[2] * Sg[6 + k];
    }
    p.a = ((p.T0[0] * p.M0[0] + p.T0[1] * p.M0[1]) + p.T0[2] * p.M0[2]) + 0.3f;
    p.b = (p.T0[0] * p.M1[0] + p.T0[1] * p.M1[1]) + p.T0[2] * p.M1[2];
    p.c = ((p.T1[0] * p.M1[0] + p.T1[1] * p.M1[1]) + p.T1[2] * p.M1[2]) + 0.3f;
    p.det = p.a * p.c - p.b * p.b;
}

// d colour / d (unit) direction contracted with the colour gradient gc: gd = sum_k d basis_k/d dir * (sh[k] . gc)
__device__ __forceinline__ void sh_dir_backward(int deg, const float d[3], const float* __restrict__ sh,
                                                const float gc[3], float gd[3]) {
    const float x = d[0], y = d[1], z = d[2];
    gd[0] = gd[1] = gd[2] = 0.f;
#define SDOT(k) ((sh[3 * (k)] * gc[0] + sh[3 * (k) + 1] * gc[1]) + sh[3 * (k) + 2] * gc[2])
    if (deg > 0) {
        gd[1] += -SH_C1 * SDOT(1); gd[2] += SH_C1 * SDOT(2); gd[0] += -SH_C1 * SDOT(3);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float s4 = SDOT(4), s5 = SDOT(5), s6 = SDOT(6), s7 = SDOT(7), s8 = SDOT(8);
            gd[0] += SH_C2[0] * y * s4 + SH_C2[2] * -2.f * x * s6 + SH_C2[3] * z * s7 + SH_C2[4] * 2.f * x * s8;
            gd[1] += SH_C2[0] * x * s4 + SH_C2[1] * z * s5 + SH_C2[2] * -2.f * y * s6 + SH_C2[4] * -2.f * y * s8;
            gd[2] += SH_C2[1] * y * s5 + SH_C2[2] * 4.f * z * s6 + SH_C2[3] * x * s7;
            if (deg > 2) {
                float s9 = SDOT(9), s10 = SDOT(10), s11 = SDOT(11), s12 = SDOT(12), s13 = SDOT(13),
                      s14 = SDOT(14), s15 = SDOT(15);
                gd[0] += SH_C3[0] * 6.f * x * y * s9 + SH_C3[1] * y * z * s10 + SH_C3[2] * -2.f * x * y * s11 +
                         SH_C3[3] * -6.f * x * z * s12 + SH_C3[4] * (4.f * zz - 3.f * xx - yy) * s13 +
                         SH_C3[5] * 2.f * x * z * s14 + SH_C3[6] * (3.f * xx - 3.f * yy) * s15;
                gd[1] += SH_C3[0] * (3.f * xx - 3.f * yy) * s9 + SH_C3[1] * x * z * s10 +
                         SH_C3[2] * (4.f * zz - xx - 3.f * yy) * s11 + SH_C3[3] * -6.f * y * z * s12 +
                         SH_C3[4] * -2.f * x * y * s13 + SH_C3[5] * -2.f * y * z * s14 +
                         SH_C3[6] * -6.f * x * y * s15;
                gd[2] += SH_C3[1] * x * y * s10 + SH_C3[2] * 8.f * y * z * s11 +
                         SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy) * s12 + SH_C3[4] * 8.f * x * z * s13 +
                         SH_C3[5] * (xx - yy) * s14;
            }
        }
    }
#undef SDOT
}

// d colour_c / d (unit direction) for the three channels: J[3 c + axis] = sum_k d basis_k / d axis * sh[k][c].
// K1 stores it (36 B) so that K8 gets d L / d dir = J^T gc without touching the SH coefficients again.
__device__ __forceinline__ void sh_dir_jacobian(int deg, const float d[3], const float* __restrict__ sh, float J[9]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float gc[3] = {0.f, 0.f, 0.f};
        gc[c] = 1.f;
        sh_dir_backward(deg, d, sh, gc, J + 3 * c);
    }
}

// ---------------------------------------------------------------------------------------------------
// SH rows through LDS.  shs is [N,16,3]: 192 contiguous bytes per Gaussian, so one-Gaussian-per-lane loads/stores
// touch 64 different cache lines per instruction.  Instead the block moves its rows (contiguous in HBM) with fully
// coalesced dwordx4 accesses and each lane reads / writes its own row in LDS -- HALF of the block's rows at a time, so that the
// staging tile (6.5 KB per wave in K1, 26 KB per 256 threads in K8 and the factor rebuild) does not cap the resident waves: with
// whole-block tiles K1 ran 3 waves per SIMD and 0.196 ms, with half tiles 4 and 0.154 ms.  Rows are padded to
// 13 float4 (52 dwords): 52 t mod 64 takes 16 distinct multiples of 4, so a ds_read_b128 lane group is conflict-free.
// ---------------------------------------------------------------------------------------------------
#define SH_ROW4 13
// this lane's 48 coefficients [k][c] into registers straight from HBM (the M != 16 path; M == 16 goes through the staging tile)
__device__ __forceinline__ void sh_row_load(const float* __restrict__ shs, const float* __restrict__ r0, const float* __restrict__ r1, int i, int M,
                                            int K, float v[48]) {
    const size_t o = (size_t)i * M * 3;
#pragma unroll
    for (int k = 0; k < 48; k++) {
        float x = (k < 3 * K) ? shs[o + k] : 0.f;
        if (r0 && k < 3 * K) x = x + r0[o + k];          // (shs + r0) + r1, the order of the reference's two adds
        if (r1 && k < 3 * K) x = x + r1[o + k];
        v[k] = x;
    }
}

// ---------------------------------------------------------------------------------------------------
// K1.  One wave per workgroup.  Order of work: (1) geometry of the lane's Gaussian up to the visibility decision, (2) the
// wave stages the 192-byte SH rows of its VISIBLE Gaussians only (coalesced dwordx4 pieces, predicated per row by the
// visibility ballot: a quarter of the bench scene is culled and its coefficients are never read), (3) colour, the
// colour Jacobian and the record stores.  Twelve independent 13 KB workgroups per CU overlap these phases.
// ---------------------------------------------------------------------------------------------------
#define PRE_BLOCK 64
#ifndef EMD_K1_WAVES
#define EMD_K1_WAVES 4
#endif
// PART 0: the whole kernel.  PART 1 / PART 2 (round 3): its geometry half (everything the binning needs, and rows 0, 1, 3 of the record)
// and its colour half (SH colour, clamp bits, colour Jacobian: row 2 of the record and shjac) as two launches -- the colour half is
// needed by K6 only, so with an auxiliary stream (EmdFwdArgs.aux_stream) it runs BESIDE the twenty launch-bound kernels of the binning
// stage instead of in front of them.  PART 2 recomputes the world mean of its Gaussian (one gather for an actor's point) and takes
// the visibility from the radii PART 1 wrote.
// RES: residuals of the SH coefficients are added while the rows are staged (EmdFwdArgs.shs_residual); a separate instantiation, so
// that the usual kernel keeps its registers (with the residual loads in the same code it spilled 24)
template <int PART, bool RES = false>
__global__ void __launch_bounds__(PRE_BLOCK) __attribute__((amdgpu_waves_per_eu(PART == 1 ? 6 : (RES ? 4 : EMD_K1_WAVES)))) k_preprocess(PreArgs a) {
    __shared__ float4 s_sh[PART == 1 ? 1 : (PRE_BLOCK / 2) * SH_ROW4];       // half of the wave's rows at a time: 6.5 KB keeps four waves per SIMD
    EmdSettings S = a.s;
    emd_settings_from_device(S, a.sdev, a.flags);
    const bool sh_staged = a.shs && a.M == 16;
    const uint32_t blk = EMD_K1_REVERSE ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
    const int i = blk * PRE_BLOCK + threadIdx.x;
    // the call's four status words are cleared here (the binning kernels behind this launch raise bits in them): no launch of its own
    if (PART != 2 && blockIdx.x == 0 && threadIdx.x < 4 && a.status) reinterpret_cast<uint32_t*>(a.status)[threadIdx.x] = 0u;
    uint32_t touched = 0, rect = 0, rect_h = 0, dkey = 0xFFFFFFFFu, skip = 0;
    int radius_out = 0;
    const float* V = S.viewmatrix;
    float m[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, op = 0.f, sc[3] = {1.f, 1.f, 1.f};
    float ix = 0.f, iy = 0.f, conA = 0.f, conB = 0.f, conC = 0.f;
    Proj p;
    p.tx = p.ty = p.tz = 0.f;
    // the scales are requested with the other parameters, not behind the near-plane test they used to wait for (one HBM round trip less
    // on the way to the visibility decision; a culled Gaussian's 12 bytes are read in vain: measured 0.180 -> 0.172 ms)
    float sc_raw[3] = {0.f, 0.f, 0.f};
#if !EMD_K1_HOIST
    if (PART != 2 && i < a.N && !a.cov3D_precomp) { sc_raw[0] = a.scales[3 * i]; sc_raw[1] = a.scales[3 * i + 1]; sc_raw[2] = a.scales[3 * i + 2]; }
#endif
    if (PART == 2) {
        // colour half: visibility from the geometry half's radii; the world mean again (static point: the parameter itself)
        if (i < a.N && a.radii[i] > 0) {
            touched = 1u;
            if (a.flags & EMD_FLAG_MOTION) motion_point(i, a.means3D, a.rotations, a.opacities, a.motion, m, q, &op, (a.flags & EMD_FLAG_RAW_PARAMS) != 0);
            else { m[0] = a.means3D[3 * i]; m[1] = a.means3D[3 * i + 1]; m[2] = a.means3D[3 * i + 2]; }
        }
    } else if (i < a.N) {
        const float* P = S.projmatrix;
        const int W = S.image_width, H = S.image_height;
        const int gx = (W + EMD_TILE_X - 1) / EMD_TILE_X, gy = (H + EMD_TILE_Y - 1) / EMD_TILE_Y;
        const float fx = (float)W / (2.f * S.tanfovx), fy = (float)H / (2.f * S.tanfovy);
        const bool raw = (a.flags & EMD_FLAG_RAW_PARAMS) != 0;
#if EMD_K1_HOIST
        {
            // Round 5 (late), as in K8: every load whose address depends on the index alone is issued here, back to back, before any is used (optional
            // inputs through a pointer that is valid either way -- the Gaussian's own mean stands in --, selected where they are used); then the actor's pose
            // rows.  Before: mean -> actor id -> quaternion / pose rows -> opacity, each behind the wait of the one before.
            const bool motion = (a.flags & EMD_FLAG_MOTION) != 0;
            const bool has_ids = motion && a.motion.actor_id != nullptr, has_rot = a.rotations != nullptr;
            const bool has_rdx = motion && a.motion.residual_dx != nullptr, has_rdq = motion && has_rot && a.motion.residual_dq != nullptr;
            const float* mp = a.means3D + 3 * (size_t)i;
            const float* xp = has_rdx ? a.motion.residual_dx + 3 * (size_t)i : mp;
            const float* sp = a.cov3D_precomp ? mp : a.scales + 3 * (size_t)i;
            const int aid_raw = *(has_ids ? a.motion.actor_id + i : (const int*)mp);
            const float m0 = mp[0], m1 = mp[1], m2 = mp[2];
            const float x0 = xp[0], x1 = xp[1], x2 = xp[2];
            const float s0 = sp[0], s1 = sp[1], s2 = sp[2];
            const float opv = a.opacities[i];
            // (a quaternion row is 16 bytes at a 16-byte stride; without rotations the stand-in is read as three dwords + one)
            float4 qq = make_float4(0.f, 0.f, 0.f, 0.f), dqq = qq;
            if (has_rot) qq = *(const float4*)(a.rotations + 4 * (size_t)i);                 // (kernel-uniform; the loads above are already in flight)
            if (has_rdq) dqq = *(const float4*)(a.motion.residual_dq + 4 * (size_t)i);
            const int aid = has_ids ? aid_raw : -1;
            float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
            if (aid >= 0) {
                const float4* Pp = (const float4*)(a.motion.actor_pose + (size_t)aid * EMD_ACTOR_STRIDE);
                p0 = Pp[0]; p1 = Pp[1]; p2 = Pp[2];
            }
            if (!a.cov3D_precomp) { sc_raw[0] = s0; sc_raw[1] = s1; sc_raw[2] = s2; }
            if (motion) {
                const float ml[3] = {has_rdx ? m0 + x0 : m0, has_rdx ? m1 + x1 : m1, has_rdx ? m2 + x2 : m2};
                motion_apply(ml, aid, has_rot, qq, has_rdq, dqq, true, opv, p0, p1, p2, raw, m, q, &op);
            } else {
                op = raw ? sigmoidf_(opv) : opv;
                m[0] = m0; m[1] = m1; m[2] = m2;
                if (has_rot) {
                    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
                    if (raw) { const float n = fmaxf(quat_norm(q), 1e-12f); q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
                }
            }
        }
#else
        if (a.flags & EMD_FLAG_MOTION) {
            motion_point(i, a.means3D, a.rotations, a.opacities, a.motion, m, q, &op, raw);
        } else {
            op = raw ? sigmoidf_(a.opacities[i]) : a.opacities[i];
            m[0] = a.means3D[3 * i]; m[1] = a.means3D[3 * i + 1]; m[2] = a.means3D[3 * i + 2];
            if (a.rotations) {
                const float4 qq = *(const float4*)(a.rotations + 4 * i); q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
                if (raw) { const float n = fmaxf(quat_norm(q), 1e-12f); q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
            }
        }
#endif
        p.tx = ((V[0] * m[0] + V[4] * m[1]) + V[8] * m[2]) + V[12];
        p.ty = ((V[1] * m[0] + V[5] * m[1]) + V[9] * m[2]) + V[13];
        p.tz = ((V[2] * m[0] + V[6] * m[1]) + V[10] * m[2]) + V[14];
        if (p.tz > S.near_plane) {
            float hx = ((P[0] * m[0] + P[4] * m[1]) + P[8] * m[2]) + P[12];
            float hy = ((P[1] * m[0] + P[5] * m[1]) + P[9] * m[2]) + P[13];
            float hw = ((P[3] * m[0] + P[7] * m[1]) + P[11] * m[2]) + P[15];
            float pw = 1.f / (hw + 0.0000001f);
            float px = hx * pw, py = hy * pw;
            float c3[6];
            if (a.cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) c3[k] = a.cov3D_precomp[6 * i + k];
            } else {
                sc[0] = sc_raw[0]; sc[1] = sc_raw[1]; sc[2] = sc_raw[2];
                if (raw) { sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]); }
                cov3d_from_sr(sc, S.scale_modifier, q, c3);
            }
            project_cov2d(S, m, c3, fx, fy, p);
            if (p.det != 0.f) {
                float det_inv = 1.f / p.det;
                conA = p.c * det_inv; conB = -p.b * det_inv; conC = p.a * det_inv;
                float mid = 0.5f * (p.a + p.c);
                float sq = sqrtf(fmaxf(0.1f, mid * mid - p.det));
                float lam1 = mid + sq, lam2 = mid - sq;
                float rad = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
                ix = ((px + 1.f) * (float)W - 1.f) * 0.5f;
                iy = ((py + 1.f) * (float)H - 1.f) * 0.5f;
                int x0 = tile_clamp((ix - rad) / (float)EMD_TILE_X, gx);
                int y0 = tile_clamp((iy - rad) / (float)EMD_TILE_Y, gy);
                int x1 = tile_clamp((ix + rad + (float)(EMD_TILE_X - 1)) / (float)EMD_TILE_X, gx);
                int y1 = tile_clamp((iy + rad + (float)(EMD_TILE_Y - 1)) / (float)EMD_TILE_Y, gy);
                int area = (x1 - x0) * (y1 - y0);
                if (area > 0) {
                    touched = (uint32_t)area;
                    dkey = __float_as_uint(p.tz);
                    radius_out = (int)rad;
                    // Round 4: the rectangle the binning ENUMERATES is upstream's 3-sigma square cut down to the tiles the alpha >= 1/255
                    // ellipse of this Gaussian can reach: a pixel contributes only where d^T Conic d <= 2 ln(255 o), whose bounding box
                    // has the half extents sqrt(2 ln(255 o) cov_xx), sqrt(. cov_yy) -- straight from the 2-D covariance, no division.
                    // Pixel centres are integers, so the box is kept as the range of 8-pixel quadrant columns / rows that hold a centre
                    // inside it (11 bits each, clamped to the image): the duplicate kernel derives every pair's quadrant mask from it.
                    const float thr = 2.f * __logf(255.f * op) * 1.01f + 0.05f;           // (same margin as footprint.h)
                    float bx = __builtin_sqrtf(thr * p.a) * 1.0001f + 0.01f, by = __builtin_sqrtf(thr * p.c) * 1.0001f + 0.01f;
                    const bool never = !(op >= (1.f / 255.f));                            // cannot reach 1/255 anywhere (also NaN)
                    if (!(bx == bx) || !(by == by)) bx = by = 3.0e38f;                    // (NaN covariance entries: keep everything)
                    //  the render kernels evaluate alpha with the float conic adj(cov) / fl(det), whose relative error is
                    // ~1.2e-7 a c / det: for needle footprints (a c > 3e4 det -- hundreds of pixels long, under a pixel wide) the 1 % + 0.05
                    // margin above no longer covers it, so such a Gaussian keeps upstream's rectangle and all four quadrants, exactly as
                    // footprint.h refuses to cull it (`e.all`)
                    if (!(p.det * 3.0e4f > p.a * p.c)) bx = by = 3.0e38f;
                    const float QMAX = 2047.f;
                    const float lqxf = floorf(ceilf(ix - bx) * 0.125f), hqxf = floorf(floorf(ix + bx) * 0.125f);
                    const float lqyf = floorf(ceilf(iy - by) * 0.125f), hqyf = floorf(floorf(iy + by) * 0.125f);
                    const bool off = never || hqxf < 0.f || hqyf < 0.f || lqxf > QMAX || lqyf > QMAX || !(lqxf == lqxf) || !(lqyf == lqyf);
                    const int lqx = (int)fminf(fmaxf(lqxf, 0.f), QMAX), hqx = (int)fminf(fmaxf(hqxf, 0.f), QMAX);
                    const int lqy = (int)fminf(fmaxf(lqyf, 0.f), QMAX), hqy = (int)fminf(fmaxf(hqyf, 0.f), QMAX);
                    int cx0 = x0, cy0 = y0, cx1 = x1, cy1 = y1;
                    if (!(a.flags & EMD_FLAG_KEEP_ALL_PAIRS)) {
                        cx0 = max(x0, lqx >> 1); cx1 = min(x1, (hqx >> 1) + 1);
                        cy0 = max(y0, lqy >> 1); cy1 = min(y1, (hqy >> 1) + 1);
                        if (off || cx1 <= cx0 || cy1 <= cy0) { cx1 = cx0; cy1 = cy0; }
                        // Inside the cut rectangle every quadrant lies in the box except, possibly, the outer quadrant column / row of its
                        // first and last tile: where the rectangle ends at the box (not at upstream's square) and the box starts in the
                        // right (odd) / ends in the left (even) quadrant column -- rows likewise.  Four bits give every pair its mask.
                        skip = ((cx0 == (lqx >> 1) && (lqx & 1)) ? 1u : 0u) | ((cx1 - 1 == (hqx >> 1) && !(hqx & 1)) ? 2u : 0u) |
                               ((cy0 == (lqy >> 1) && (lqy & 1)) ? 4u : 0u) | ((cy1 - 1 == (hqy >> 1) && !(hqy & 1)) ? 8u : 0u);
                    }       // (with every pair kept the masks are all ones: the render forward's own sub-block test decides)
                    rect = (uint32_t)cx0 | ((uint32_t)cy0 << 10) | ((uint32_t)(cx1 - cx0) << 20) | ((skip & 3u) << 30);
                    rect_h = (uint32_t)(cy1 - cy0) | ((skip >> 2) << 10);
                }
            }
        }
        a.radii[i] = radius_out;
        // binning input (by Gaussian id), 8 bytes: x = ENUMERATED tile rectangle x0 | y0 << 10 | width << 20 | skip-left/right bits << 30,
        // y = its height | skip-top/bottom bits << 10 | upstream's tiles touched << 12
        a.g.binrec[i] = make_uint2(rect, rect_h | (touched << 12));
        a.g.depth_key[i] = dkey;                         // key of the depth sort (invisible: sorts last)
    }
    const bool vis = touched != 0u;
    float sh[48];
    if (PART != 1 && sh_staged) {
        // rows of the VISIBLE Gaussians only, in two halves of 32 rows: coalesced dwordx4 pieces into LDS, then each lane of the half
        // takes its own row into registers
        const unsigned long long vmask = __ballot(vis);
        const size_t base4 = (size_t)blk * PRE_BLOCK * 12;
        const float4* src = (const float4*)a.shs;
        // Round 4: all twelve pieces of the wave's rows are requested at once -- both halves in flight together, one HBM round trip instead
        // of two -- and then staged half by half (48 registers more while they fly: four waves per SIMD instead of five; measured
        // 738 -> 745 it/s, and 730 at five waves, which spills).  The instantiation with residuals keeps the half-by-half loads.
        constexpr bool SH_ALL = !RES;
        float4 pv[SH_ALL ? 12 : 1];
        if (SH_ALL) {
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const uint32_t idx = threadIdx.x + PRE_BLOCK * j, row = idx / 12;
                pv[SH_ALL ? j : 0] = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((vmask >> row) & 1ull) pv[SH_ALL ? j : 0] = EMD_K1_NT_SH ? load_f4_nt(src + base4 + idx) : src[base4 + idx];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const uint32_t idx = threadIdx.x + PRE_BLOCK * (6 * h + j), row = idx / 12;
                if ((vmask >> row) & 1ull) {
                    float4 v;
                    if (SH_ALL) v = pv[SH_ALL ? 6 * h + j : 0];
                    else {
                        v = src[base4 + idx];
                        if (RES && a.shs_res0) {                          // (shs + r0) + r1, the order of the reference's two adds
                            const float4 r = ((const float4*)a.shs_res0)[base4 + idx];
                            v.x = v.x + r.x; v.y = v.y + r.y; v.z = v.z + r.z; v.w = v.w + r.w;
                        }
                        if (RES && a.shs_res1) {
                            const float4 r = ((const float4*)a.shs_res1)[base4 + idx];
                            v.x = v.x + r.x; v.y = v.y + r.y; v.z = v.z + r.z; v.w = v.w + r.w;
                        }
                    }
                    s_sh[(row - 32 * h) * SH_ROW4 + (idx % 12)] = v;
                }
            }
            __syncthreads();
            if ((int)(threadIdx.x >> 5) == h) {
#pragma unroll
                for (int j = 0; j < 12; j++) {
                    const float4 t = s_sh[(threadIdx.x & 31) * SH_ROW4 + j];
                    sh[4 * j] = t.x; sh[4 * j + 1] = t.y; sh[4 * j + 2] = t.z; sh[4 * j + 3] = t.w;
                }
            }
            __syncthreads();
        }
    }
    if (vis) {
        float col[3] = {0.f, 0.f, 0.f};
        uint32_t bits = 0;
        if (PART == 1) {
            // (geometry half: no colour)
        } else if (a.colors_precomp) {
            col[0] = a.colors_precomp[3 * i]; col[1] = a.colors_precomp[3 * i + 1];
            col[2] = a.colors_precomp[3 * i + 2];
        } else {
            float d[3] = {m[0] - S.campos[0], m[1] - S.campos[1], m[2] - S.campos[2]};
            float n = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            d[0] /= n; d[1] /= n; d[2] /= n;
            float bs[16];
            sh_basis(S.sh_degree, d, bs);
            const int K = (S.sh_degree + 1) * (S.sh_degree + 1);
            if (!sh_staged) sh_row_load(a.shs, RES ? a.shs_res0 : nullptr, RES ? a.shs_res1 : nullptr, i, a.M, K, sh);
            col[0] = col[1] = col[2] = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < K) { col[0] += bs[k] * sh[3 * k]; col[1] += bs[k] * sh[3 * k + 1]; col[2] += bs[k] * sh[3 * k + 2]; }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                col[ch] += 0.5f;
                if (col[ch] < 0.f) { col[ch] = 0.f; bits |= 1u << ch; }
                if ((a.flags & EMD_FLAG_CLAMP_RGB01) && col[ch] > 1.f) { col[ch] = 1.f; bits |= 1u << ch; }
            }
            float J[9];
            sh_dir_jacobian(S.sh_degree, d, sh, J);
            float4* jr = a.g.shjac + (size_t)i * 3;
            // (the spare word of row 0 carries the clamp bits: the projection backward reads them here, with the Jacobian it needs anyway)
            const float bw = __uint_as_float(bits);
            if (EMD_K1_NT_JAC) {
                store_f4_nt(jr, make_float4(J[0], J[1], J[2], bw)); store_f4_nt(jr + 1, make_float4(J[3], J[4], J[5], 0.f));
                store_f4_nt(jr + 2, make_float4(J[6], J[7], J[8], 0.f));
            } else {
                jr[0] = make_float4(J[0], J[1], J[2], bw);
                jr[1] = make_float4(J[3], J[4], J[5], 0.f);
                jr[2] = make_float4(J[6], J[7], J[8], 0.f);
            }
        }
        float4* rec = a.g.rec + (size_t)i * EMD_REC_F4;
        // the clamp bits of the colour ride in row 2 (with the colour they belong to), so that the two halves write disjoint rows
#ifdef EMD_K1_NT_REC          /* experiment: do the render kernels care whether the records are cache-resident? */
        if (PART != 2) { store_f4_nt(rec, make_float4(ix, iy, p.tz, op)); store_f4_nt(rec + 1, make_float4(conA, conB, conC, 0.f)); }
        if (PART != 1) store_f4_nt(rec + 2, make_float4(col[0], col[1], col[2], __uint_as_float(bits)));
#else
        if (PART != 2) { rec[0] = make_float4(ix, iy, p.tz, op); rec[1] = make_float4(conA, conB, conC, 0.f); }
        if (PART != 1) rec[2] = make_float4(col[0], col[1], col[2], __uint_as_float(bits));
#endif
        if (PART != 2 && (a.flags & EMD_FLAG_NORMAL)) {
            float nv[3] = {0.f, 0.f, 0.f};
            if (a.scales) {
                int ax = 0;
                if (sc[1] < sc[ax]) ax = 1;
                if (sc[2] < sc[ax]) ax = 2;
                float R[9];
                quat_to_R(q, R);
                float nw[3] = {R[ax], R[3 + ax], R[6 + ax]};
                nv[0] = (V[0] * nw[0] + V[4] * nw[1]) + V[8] * nw[2];
                nv[1] = (V[1] * nw[0] + V[5] * nw[1]) + V[9] * nw[2];
                nv[2] = (V[2] * nw[0] + V[6] * nw[1]) + V[10] * nw[2];
                float dp = (nv[0] * p.tx + nv[1] * p.ty) + nv[2] * p.tz;
                if (dp > 0.f) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
            }
            rec[3] = make_float4(nv[0], nv[1], nv[2], 0.f);
        }
    }
}

__device__ __forceinline__ void dR_to_dq(const float q[4], const float dR[9], float dq[4]) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    dq[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    dq[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
    dq[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
    dq[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
}

__device__ __forceinline__ void dnormalize4(const float vu[4], float n, const float g[4], float out[4]) {
    float dot = ((vu[0] * g[0] + vu[1] * g[1]) + vu[2] * g[2]) + vu[3] * g[3];
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = (g[k] - vu[k] * dot) / n;
}

// Backward of motion_point for an actor point (a_id >= 0): world-space gradients (dm, dq, dop) -> local-space
// gradients (dl, dql, dopl) and this point's contribution to its actor's pose row (pose_g[12]).
// (the arithmetic on values in registers; motion_point_backward below loads them)
__device__ __forceinline__ void motion_backward_apply(const float ml[3], bool has_q, float4 qq, bool has_dq, float4 dq_res, bool has_op, float op_in,
                                                      float4 p0, float4 p1, float4 p2, const float dm[3], const float dq[4], float dop, float dl[3],
                                                      float dql[4], float* dopl, float pose_g[12], bool raw) {
    const float qm[4] = {p0.x, p0.y, p0.z, p0.w};
    float R[9];
    quat_to_R(qm, R);
#pragma unroll
    for (int k = 0; k < 3; k++) dl[k] = (R[k] * dm[0] + R[3 + k] * dm[1]) + R[6 + k] * dm[2];
    float dRm[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) dRm[3 * r + k] = dm[r] * ml[k];
    float dqm[4];
    dR_to_dq(qm, dRm, dqm);
    pose_g[0] = dqm[0]; pose_g[1] = dqm[1]; pose_g[2] = dqm[2]; pose_g[3] = dqm[3];
    pose_g[4] = dm[0]; pose_g[5] = dm[1]; pose_g[6] = dm[2];
    pose_g[7] = has_op ? dop * (raw ? sigmoidf_(op_in) : op_in) : 0.f;
    pose_g[8] = pose_g[9] = pose_g[10] = pose_g[11] = 0.f;
    dql[0] = dql[1] = dql[2] = dql[3] = 0.f;
    if (has_q) {
        float ql[4] = {qq.x, qq.y, qq.z, qq.w};
        if (has_dq) { ql[0] += dq_res.x; ql[1] += dq_res.y; ql[2] += dq_res.z; ql[3] += dq_res.w; }
        float n = fmaxf(quat_norm(ql), 1e-12f);
        float qn[4] = {ql[0] / n, ql[1] / n, ql[2] / n, ql[3] / n};
        const float qr[4] = {p2.x, p2.y, p2.z, p2.w};
        float pp[4];
        quat_mul(qr, qn, pp);
        float n2 = fmaxf(quat_norm(pp), 1e-12f);
        float pu[4] = {pp[0] / n2, pp[1] / n2, pp[2] / n2, pp[3] / n2};
        float dp[4];
        dnormalize4(pu, n2, dq, dp);
        // p = a (x) b : dL/da = g (x) conj(b), dL/db = conj(a) (x) g
        const float bc[4] = {qn[0], -qn[1], -qn[2], -qn[3]}, ac[4] = {qr[0], -qr[1], -qr[2], -qr[3]};
        float dqa[4], dqb[4];
        quat_mul(dp, bc, dqa);
        quat_mul(ac, dp, dqb);
        pose_g[8] = dqa[0]; pose_g[9] = dqa[1]; pose_g[10] = dqa[2]; pose_g[11] = dqa[3];
        dnormalize4(qn, n, dqb, dql);
    }
    *dopl = dop * p1.w;
}

__device__ __forceinline__ void motion_point_backward(int i, int a_id, const float* __restrict__ means,
                                                      const float* __restrict__ quats, const float* __restrict__ opac,
                                                      const EmdMotion& mo, const float dm[3], const float dq[4],
                                                      float dop, float dl[3], float dql[4], float* dopl,
                                                      float pose_g[12], bool raw = false) {
    const float4* Pp = (const float4*)(mo.actor_pose + (size_t)a_id * EMD_ACTOR_STRIDE);
    const float4 p0 = Pp[0], p1 = Pp[1], p2 = Pp[2];
    float ml[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    if (mo.residual_dx) { ml[0] += mo.residual_dx[3 * i]; ml[1] += mo.residual_dx[3 * i + 1]; ml[2] += mo.residual_dx[3 * i + 2]; }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 qq = z4, dqr = z4;
    if (quats) {
        qq = *(const float4*)(quats + 4 * i);
        if (mo.residual_dq) dqr = *(const float4*)(mo.residual_dq + 4 * i);
    }
    motion_backward_apply(ml, quats != nullptr, qq, mo.residual_dq != nullptr, dqr, opac != nullptr, opac ? opac[i] : 0.f, p0, p1, p2, dm, dq, dop, dl, dql,
                          dopl, pose_g, raw);
}

// Segmented reduction of per-point pose gradients into dL_dactor_pose.  Actor points are stored contiguously
// per instance (rigid.py:53-145), so most waves hold one actor id: DPP wave sum, one atomic row per wave.
__device__ __forceinline__ void reduce_pose_grad(int a_id, const float pose_g[12], float* __restrict__ dL_dpose) {
    const unsigned long long has = __ballot(a_id >= 0);
    if (!has) return;
    const int first = __ffsll((long long)has) - 1;
    const int a0 = __builtin_amdgcn_readlane(a_id, first);
    const bool uniform = __ballot(a_id >= 0 && a_id != a0) == 0ull;
    if (uniform) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            float v = wave_reduce_to_lane63(a_id >= 0 ? pose_g[k] : 0.f);
            if ((threadIdx.x & 63) == 63) atomicAdd(dL_dpose + (size_t)a0 * EMD_ACTOR_STRIDE + k, v);
        }
    } else if (a_id >= 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) atomicAdd(dL_dpose + (size_t)a_id * EMD_ACTOR_STRIDE + k, pose_g[k]);
    }
}

// ---------------------------------------------------------------------------------------------------
// K8
// ---------------------------------------------------------------------------------------------------
#ifndef K8_BLOCK
#define K8_BLOCK 256          // threads per workgroup of K8 (A/B: 64 = one wave per workgroup, no cross-wave barrier coupling)
#endif
#define K8_HALF (K8_BLOCK / 2)
#ifndef EMD_K8_WAVES
#define EMD_K8_WAVES 4        // round 3: with the SH rows stored first the live state across the staging barriers needs 112 VGPRs; at 5 waves it spills
#endif
__global__ void __launch_bounds__(K8_BLOCK) __attribute__((amdgpu_waves_per_eu(EMD_K8_WAVES))) k_preprocess_backward(PreBwdArgs a) {
    EmdSettings S = a.s;
    emd_settings_from_device(S, a.sdev, a.flags);
    // staging of the dL/dshs rows (coalesced copy-out), half of the block's rows at a time: 26 KB instead of 52 keeps five
    // workgroup-waves per SIMD resident instead of three.  A row is the outer product basis[k] x gc[c]: the lane keeps the 19 factors
    // and multiplies them out when its half is staged.
    __shared__ float4 s_sh[K8_HALF * SH_ROW4];
    float sh_b[16], sh_g[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; k++) sh_b[k] = 0.f;
    const bool sh_staged = a.shs && a.M == 16;
#if EMD_K8_COMPACT
    // Round 5: 47 % of the Gaussians of a view are invisible, so a lane-per-Gaussian wave works with half of its lanes.  The visible Gaussians
    // of the workgroup's 256 are compacted (in index order) onto its first lanes: ~2.1 full waves run the loads and the chain rule, the rest of
    // the workgroup only writes the zero rows of the invisible ones ("natural" duties: thread t for Gaussian t of the block).
    __shared__ uint32_t s_cscan[K8_BLOCK / 64];
    __shared__ uint16_t s_list[K8_BLOCK];
    const int inat = blockIdx.x * K8_BLOCK + threadIdx.x;
    const bool nat_in = inat < a.N;
    const bool nat_vis = nat_in && a.radii[inat] > 0;
    uint32_t nv;
    {
        // inclusive scan of the visibility flags over the workgroup (any number of waves)
        const uint32_t winc = wave_scan_add_u32(nat_vis ? 1u : 0u), wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 63) s_cscan[wv] = winc;
        __syncthreads();
        uint32_t wbase = 0;
        nv = 0;
#pragma unroll
        for (uint32_t w = 0; w < K8_BLOCK / 64; w++) { const uint32_t c = s_cscan[w]; if (w < wv) wbase += c; nv += c; }
        const uint32_t incl = wbase + winc;
        if (nat_vis) s_list[incl - 1u] = (uint16_t)threadIdx.x;
        __syncthreads();
    }
    const bool in_range = threadIdx.x < nv;                        // this thread carries a visible Gaussian
    const int lrow = in_range ? (int)s_list[threadIdx.x] : 0;      // its row inside the block
    const int i = blockIdx.x * K8_BLOCK + lrow;
    const bool nat_zero = nat_in && !nat_vis;                      // this thread also owns the zero row of an invisible Gaussian
#else
    const int i = blockIdx.x * K8_BLOCK + threadIdx.x;
    const bool in_range = i < a.N;
    const int lrow = threadIdx.x, inat = i;
    const bool nat_zero = false, nat_vis = true;
#endif
    const float* V = S.viewmatrix;
    const float* P = S.projmatrix;
    const int W = S.image_width, H = S.image_height;
    const float fx = (float)W / (2.f * S.tanfovx), fy = (float)H / (2.f * S.tanfovy);
    float dm[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f}, ds[3] = {0.f, 0.f, 0.f}, dop = 0.f;
    float dc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gm2[2] = {0.f, 0.f}, gabs[2] = {0.f, 0.f};
    int a_id = -1;
    float pose_g[12];
#pragma unroll
    for (int k = 0; k < 12; k++) pose_g[k] = 0.f;
    // Part 1: loads, the Gaussian's world pose, and the SH colour part (whose dense 192-byte rows leave FIRST, below: their stores
    // then drain while the wave works through the geometry chain of part 2 instead of at the very end of its life)
    bool visible = false;
    const bool raw = (a.flags & EMD_FLAG_RAW_PARAMS) != 0;
    float m[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, op = 0.f, q_norm = 1.f;
    float gcol[3] = {0.f, 0.f, 0.f}, sh_gc[3] = {0.f, 0.f, 0.f};
    float g_depth = 0.f, gA = 0.f, gB = 0.f, gC = 0.f;
    float sc_in[3] = {0.f, 0.f, 0.f};       // (loaded in part 1: no global load waits behind the staging barriers)
    Proj p;
    p.tx = p.ty = p.tz = 0.f;
    // Round 5 (late): EVERY load of a visible Gaussian whose address depends on its index alone is issued here, back to back, before any of them is
    // used -- its parameters, its accumulated gradient row, its colour Jacobian --, then the actor's pose rows (the one dependent address).  Before, the
    // loads sat in the branches that use them (motion / raw quaternion / scales / gradient row / Jacobian, each kernel-uniform or per-lane): seven
    // load -> s_waitcnt -> use rounds one behind the other, each a trip to the Infinity Cache or HBM, in a kernel whose compacted blocks keep ~2 waves
    // busy.  Optional inputs are read through a pointer that is valid either way (the Gaussian's own gradient row stands in for an absent array) and
    // selected where they are used: a branch around a load would put it back behind a wait.
    const bool motion = (a.flags & EMD_FLAG_MOTION) != 0;
    const bool has_ids = motion && a.motion.actor_id != nullptr, has_rot = a.rotations != nullptr;
    const bool has_rdx = motion && a.motion.residual_dx != nullptr, has_rdq = motion && has_rot && a.motion.residual_dq != nullptr;
    const float4 z4c = make_float4(0.f, 0.f, 0.f, 0.f);
    float mloc[3] = {0.f, 0.f, 0.f}, op_raw = 0.f;
    float4 q_raw = z4c, rdq = z4c, pr0 = z4c, pr1 = z4c, pr2 = z4c;
    if (in_range) {
        visible = EMD_K8_COMPACT ? true : a.radii[i] > 0;
        // (the actor id of every Gaussian in range: the pose-gradient reduction at the end looks at whole waves)
        const int aid_raw = *(has_ids ? a.motion.actor_id + i : a.radii + i);
        a_id = has_ids ? aid_raw : -1;
        if (visible) {
            float4* gr = (float4*)(a.grad_rec + (size_t)i * a.bwd_stride);
            const float* safe = (const float*)gr;                                          // 48 readable, 16-byte aligned bytes
            const float* mp = a.means3D + 3 * (size_t)i;
            const float* xp = has_rdx ? a.motion.residual_dx + 3 * (size_t)i : safe;
            const float* sp = a.cov3D_precomp ? safe : a.scales + 3 * (size_t)i;
            const float4* jr = a.colors_precomp ? (const float4*)gr : a.g.shjac + (size_t)i * 3;
            const float m0 = mp[0], m1 = mp[1], m2 = mp[2];
            const float x0 = xp[0], x1 = xp[1], x2 = xp[2];
            const float4 qq = *(const float4*)(has_rot ? a.rotations + 4 * (size_t)i : safe);
            const float4 dqq = *(const float4*)(has_rdq ? a.motion.residual_dq + 4 * (size_t)i : safe);
            const float opv = a.opacities[i];
            const float s0 = sp[0], s1 = sp[1], s2 = sp[2];
            const float4 g0 = gr[0], g1 = gr[1], g2 = gr[2];
            const float4 j0 = EMD_K8_NT_JAC ? load_f4_nt(jr) : jr[0], j1 = EMD_K8_NT_JAC ? load_f4_nt(jr + 1) : jr[1], j2 = EMD_K8_NT_JAC ? load_f4_nt(jr + 2) : jr[2];
            if (a_id >= 0) {                              // the one dependent address
                const float4* Pp = (const float4*)(a.motion.actor_pose + (size_t)a_id * EMD_ACTOR_STRIDE);
                pr0 = Pp[0]; pr1 = Pp[1]; pr2 = Pp[2];
            }
            if (a.flags & EMD_FLAG_BWD_WS_CLEAN) {          // the row is handed back clean: the next backward needs no zero fill
                gr[0] = z4c; gr[1] = z4c; gr[2] = z4c;
            }
            // ---- the Gaussian's world pose (motion_point's arithmetic on the values above)
            mloc[0] = has_rdx ? m0 + x0 : m0; mloc[1] = has_rdx ? m1 + x1 : m1; mloc[2] = has_rdx ? m2 + x2 : m2;
            q_raw = qq; rdq = dqq; op_raw = opv;
            if (motion) motion_apply(mloc, a_id, has_rot, q_raw, has_rdq, rdq, true, op_raw, pr0, pr1, pr2, raw, m, q, &op);
            else {
                op = raw ? sigmoidf_(op_raw) : op_raw;
                m[0] = m0; m[1] = m1; m[2] = m2;
                if (has_rot) { q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w; }
            }
            if (raw && a_id < 0 && has_rot) {   // static point: q is the raw quaternion (no-motion path) or already unit (motion path)
                const float qr[4] = {qq.x, qq.y, qq.z, qq.w};
                q_norm = fmaxf(quat_norm(qr), 1e-12f);
                q[0] = qr[0] / q_norm; q[1] = qr[1] / q_norm; q[2] = qr[2] / q_norm; q[3] = qr[3] / q_norm;
            }
            if (!a.cov3D_precomp) { sc_in[0] = s0; sc_in[1] = s1; sc_in[2] = s2; }
            uint32_t bits = 0u;
            if (!EMD_K8_BITS_IN_JAC) bits = __float_as_uint(a.g.rec[(size_t)i * EMD_REC_F4 + 2].w);   // (only the SH branch below reads them)
            gm2[0] = g0.x; gm2[1] = g0.y;
            if (a.flags & EMD_FLAG_ABSGRAD) { gabs[0] = g2.z; gabs[1] = g2.w; }
            g_depth = g0.z;
            dop = g0.w;
            gA = g1.x; gB = g1.y; gC = g1.z;
            gcol[0] = g1.w; gcol[1] = g2.x; gcol[2] = g2.y;
            p.tx = ((V[0] * m[0] + V[4] * m[1]) + V[8] * m[2]) + V[12];
            p.ty = ((V[1] * m[0] + V[5] * m[1]) + V[9] * m[2]) + V[13];
            p.tz = ((V[2] * m[0] + V[6] * m[1]) + V[10] * m[2]) + V[14];
            // (e) colour
            if (!a.colors_precomp) {
                float d0[3] = {m[0] - S.campos[0], m[1] - S.campos[1], m[2] - S.campos[2]};
                float n = sqrtf((d0[0] * d0[0] + d0[1] * d0[1]) + d0[2] * d0[2]);
                float d[3] = {d0[0] / n, d0[1] / n, d0[2] / n};
                // d L / d dir = J^T gc with the 3x3 Jacobian K1 stored: no second pass over the 192 B of coefficients
                if (EMD_K8_BITS_IN_JAC) bits = __float_as_uint(j0.w);
                float gc[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) gc[ch] = ((bits >> ch) & 1u) ? 0.f : gcol[ch];
                sh_gc[0] = gc[0]; sh_gc[1] = gc[1]; sh_gc[2] = gc[2];
                const int deg = S.sh_degree;
                const int K = (deg + 1) * (deg + 1);
                float bs[16];
                sh_basis(deg, d, bs);
                float gd[3];
                gd[0] = (j0.x * gc[0] + j1.x * gc[1]) + j2.x * gc[2];
                gd[1] = (j0.y * gc[0] + j1.y * gc[1]) + j2.y * gc[2];
                gd[2] = (j0.z * gc[0] + j1.z * gc[1]) + j2.z * gc[2];
                if (a.dL_dshs) {
                    if (sh_staged) {          // the factors of this Gaussian's row; multiplied out at staging time
#pragma unroll
                        for (int k = 0; k < 16; k++) sh_b[k] = k < K ? bs[k] : 0.f;
                        sh_g[0] = gc[0]; sh_g[1] = gc[1]; sh_g[2] = gc[2];
                    } else {
                        float* o = a.dL_dshs + (size_t)i * a.M * 3;
                        for (int k = 0; k < a.M; k++) {
                            float bk = k < K ? bs[k] : 0.f;
                            o[3 * k] = bk * gc[0]; o[3 * k + 1] = bk * gc[1]; o[3 * k + 2] = bk * gc[2];
                        }
                    }
                }
                float dot = (d[0] * gd[0] + d[1] * gd[1]) + d[2] * gd[2];
#pragma unroll
                for (int k = 0; k < 3; k++) dm[k] += (gd[k] - d[k] * dot) / n;
            }
        } else if (a.dL_dshs) {
            if (sh_staged) {
                // (factors stay zero: a zero row)
            } else {
                float* o = a.dL_dshs + (size_t)i * a.M * 3;
                for (int k = 0; k < 3 * a.M; k++) o[k] = 0.f;
            }
        }
    }
    if (EMD_K8_COMPACT && nat_zero && a.dL_dshs && !sh_staged) {
        float* o = a.dL_dshs + (size_t)inat * a.M * 3;
        for (int k = 0; k < 3 * a.M; k++) o[k] = 0.f;
    }
    if (sh_staged && a.dL_dshs) {
        const size_t lim4 = (size_t)a.N * 12;
        float4* out = (float4*)a.dL_dshs;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if ((EMD_K8_COMPACT ? in_range : true) && (int)(lrow / K8_HALF) == h) {
                float g48[48];
#pragma unroll
                for (int k = 0; k < 16; k++) { g48[3 * k] = sh_b[k] * sh_g[0]; g48[3 * k + 1] = sh_b[k] * sh_g[1]; g48[3 * k + 2] = sh_b[k] * sh_g[2]; }
#pragma unroll
                for (int j = 0; j < 12; j++)
                    s_sh[(lrow % K8_HALF) * SH_ROW4 + j] = make_float4(g48[4 * j], g48[4 * j + 1], g48[4 * j + 2], g48[4 * j + 3]);
            }
            if (EMD_K8_COMPACT && !nat_vis && (int)(threadIdx.x / K8_HALF) == h) {          // the zero row of an invisible (or absent) Gaussian
#pragma unroll
                for (int j = 0; j < 12; j++) s_sh[(threadIdx.x % K8_HALF) * SH_ROW4 + j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();
            const size_t base4 = ((size_t)blockIdx.x * K8_BLOCK + K8_HALF * h) * 12;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const uint32_t idx = threadIdx.x + K8_BLOCK * j;
                if (base4 + idx < lim4) {
                    if (EMD_K8_NT_SH) store_f4_nt(out + base4 + idx, s_sh[(idx / 12) * SH_ROW4 + (idx % 12)]);
                    else out[base4 + idx] = s_sh[(idx / 12) * SH_ROW4 + (idx % 12)];
                }
            }
            __syncthreads();
        }
    }
    // Part 2: the geometry chain and the remaining (small) stores
#if EMD_K8_STAGE_SMALL || EMD_K8_COMPACT
    // Round 5: the five small gradients every training step writes -- means3D [N,3], scales [N,3], rotations [N,4], opacities [N], means2D
    // [N,3]: 56 bytes per Gaussian as fourteen dword stores at strides of 12 / 16 / 4 bytes -- leave through the (now idle) SH staging tile
    // as whole 16-byte NONTEMPORAL stores: nothing reads them before the optimiser, and kept out of the Infinity Cache they stop evicting
    // the parameters this kernel has just read and the next step's projection kernel asks for first.
    // (compacted: the tile also is where the values of the compacted lanes and the zeros of the invisible Gaussians meet in index order)
    const bool staged5 = a.dL_dmeans3D && a.dL_dscales && a.dL_drotations && a.dL_dopacities && a.dL_dmeans2D && !EMD_K8_NT_ALL &&
                         (((uintptr_t)a.dL_dmeans3D | (uintptr_t)a.dL_dscales | (uintptr_t)a.dL_drotations | (uintptr_t)a.dL_dopacities |
                           (uintptr_t)a.dL_dmeans2D) & 15) == 0 && (size_t)(blockIdx.x + 1) * K8_BLOCK <= (size_t)a.N;     // (uniform per workgroup; a ragged last block stores directly)
    float o_dl[3] = {0.f, 0.f, 0.f}, o_ds[3] = {0.f, 0.f, 0.f}, o_dq[4] = {0.f, 0.f, 0.f, 0.f}, o_dop = 0.f;
#else
    const bool staged5 = false;
#endif
    if (in_range) {
        if (visible) {
            // (a) conic -> cov2D, (b) cov2D -> Sigma and J, t
            float c3[6];
            float sc[3] = {1.f, 1.f, 1.f};
            if (a.cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) c3[k] = a.cov3D_precomp[6 * i + k];
            } else {
                sc[0] = sc_in[0]; sc[1] = sc_in[1]; sc[2] = sc_in[2];
                if (raw) { sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]); }
                cov3d_from_sr(sc, S.scale_modifier, q, c3);
            }
            project_cov2d(S, m, c3, fx, fy, p);
            float da = 0.f, db = 0.f, dc = 0.f;
            if (p.det != 0.f) {
                float i2 = 1.f / (p.det * p.det);
                da = (-p.c * p.c * gA + p.b * p.c * gB - p.b * p.b * gC) * i2;
                db = (2.f * p.b * p.c * gA - (p.a * p.c + p.b * p.b) * gB + 2.f * p.a * p.b * gC) * i2;
                dc = (-p.b * p.b * gA + p.a * p.b * gB - p.a * p.a * gC) * i2;
            }
            const float* M0 = p.M0; const float* M1 = p.M1;
            dc6[0] = da * M0[0] * M0[0] + db * M0[0] * M1[0] + dc * M1[0] * M1[0];
            dc6[3] = da * M0[1] * M0[1] + db * M0[1] * M1[1] + dc * M1[1] * M1[1];
            dc6[5] = da * M0[2] * M0[2] + db * M0[2] * M1[2] + dc * M1[2] * M1[2];
            dc6[1] = 2.f * da * M0[0] * M0[1] + db * (M0[0] * M1[1] + M0[1] * M1[0]) + 2.f * dc * M1[0] * M1[1];
            dc6[2] = 2.f * da * M0[0] * M0[2] + db * (M0[0] * M1[2] + M0[2] * M1[0]) + 2.f * dc * M1[0] * M1[2];
            dc6[4] = 2.f * da * M0[1] * M0[2] + db * (M0[1] * M1[2] + M0[2] * M1[1]) + 2.f * dc * M1[1] * M1[2];
            float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float dM0 = 2.f * da * p.T0[k] + db * p.T1[k];
                float dM1 = 2.f * dc * p.T1[k] + db * p.T0[k];
                dJ00 += dM0 * V[4 * k + 0]; dJ02 += dM0 * V[4 * k + 2];
                dJ11 += dM1 * V[4 * k + 1]; dJ12 += dM1 * V[4 * k + 2];
            }
            float tz2 = 1.f / (p.tz * p.tz), tz3 = tz2 / p.tz;
            float dtx = p.clx ? 0.f : -fx * tz2 * dJ02;
            float dty = p.cly ? 0.f : -fy * tz2 * dJ12;
            float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + 2.f * fx * p.cx * tz3 * dJ02 + 2.f * fy * p.cy * tz3 * dJ12;
            dtz += g_depth;  // (d)
#pragma unroll
            for (int k = 0; k < 3; k++) dm[k] += V[4 * k + 0] * dtx + V[4 * k + 1] * dty + V[4 * k + 2] * dtz;
            // (c) pixel mean -> clip -> world
            float gxn = 0.5f * (float)W * gm2[0], gyn = 0.5f * (float)H * gm2[1];
            gm2[0] = gxn; gm2[1] = gyn;
            gabs[0] *= 0.5f * (float)W; gabs[1] *= 0.5f * (float)H;
            float hx = ((P[0] * m[0] + P[4] * m[1]) + P[8] * m[2]) + P[12];
            float hy = ((P[1] * m[0] + P[5] * m[1]) + P[9] * m[2]) + P[13];
            float hw = ((P[3] * m[0] + P[7] * m[1]) + P[11] * m[2]) + P[15];
            float pw = 1.f / (hw + 0.0000001f);
            float mul1 = hx * pw * pw, mul2 = hy * pw * pw;
#pragma unroll
            for (int k = 0; k < 3; k++)
                dm[k] += (P[4 * k] * pw - P[4 * k + 3] * mul1) * gxn + (P[4 * k + 1] * pw - P[4 * k + 3] * mul2) * gyn;
            // (f) Sigma -> scale, quaternion
            if (!a.cov3D_precomp) {
                float R[9], L[9];
                quat_to_R(q, R);
                const float mod = S.scale_modifier;
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int k = 0; k < 3; k++) L[3 * r + k] = R[3 * r + k] * (mod * sc[k]);
                const float Gf[9] = {dc6[0], 0.5f * dc6[1], 0.5f * dc6[2], 0.5f * dc6[1], dc6[3], 0.5f * dc6[4],
                                     0.5f * dc6[2], 0.5f * dc6[4], dc6[5]};
                float dL[9], dR[9];
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        dL[3 * r + k] = 2.f * ((Gf[3 * r] * L[k] + Gf[3 * r + 1] * L[3 + k]) + Gf[3 * r + 2] * L[6 + k]);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    ds[k] = mod * ((dL[k] * R[k] + dL[3 + k] * R[3 + k]) + dL[6 + k] * R[6 + k]);
#pragma unroll
                    for (int r = 0; r < 3; r++) dR[3 * r + k] = dL[3 * r + k] * (mod * sc[k]);
                }
                dR_to_dq(q, dR, dq);
                if (raw) { ds[0] *= sc[0]; ds[1] *= sc[1]; ds[2] *= sc[2]; }   // d exp(x) = exp(x)
            }
        }
        if (a.dL_dmeans2D && !staged5) { st_f(a.dL_dmeans2D + 3 * i, gm2[0]); st_f(a.dL_dmeans2D + 3 * i + 1, gm2[1]); st_f(a.dL_dmeans2D + 3 * i + 2, 0.f); }
        if (a.dL_dmeans2D_abs) { a.dL_dmeans2D_abs[2 * i] = gabs[0]; a.dL_dmeans2D_abs[2 * i + 1] = gabs[1]; }
        if (a.dL_dsh_color) { a.dL_dsh_color[3 * i] = sh_gc[0]; a.dL_dsh_color[3 * i + 1] = sh_gc[1]; a.dL_dsh_color[3 * i + 2] = sh_gc[2]; }
        if (a.dL_dcolors) { a.dL_dcolors[3 * i] = gcol[0]; a.dL_dcolors[3 * i + 1] = gcol[1]; a.dL_dcolors[3 * i + 2] = gcol[2]; }
        for (int k = 0; k < a.num_extra; k++) {           // extra colour sets: the accumulated dL/d colour is the gradient of the input itself
            float4* gxp = (float4*)(a.grad_rec + (size_t)i * a.bwd_stride + EMD_BWD_STRIDE + 4 * k);
            const float4 gx = visible ? *gxp : make_float4(0.f, 0.f, 0.f, 0.f);
            if (visible && (a.flags & EMD_FLAG_BWD_WS_CLEAN)) *gxp = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!a.dL_dextra[k]) continue;
            a.dL_dextra[k][3 * i] = gx.x; a.dL_dextra[k][3 * i + 1] = gx.y; a.dL_dextra[k][3 * i + 2] = gx.z;
        }
        if (a.dL_dcov3D) {
#pragma unroll
            for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * i + k] = dc6[k];
        }
        if (a.dL_dscales && !staged5) { st_f(a.dL_dscales + 3 * i, ds[0]); st_f(a.dL_dscales + 3 * i + 1, ds[1]); st_f(a.dL_dscales + 3 * i + 2, ds[2]); }
        // (h) explicit motion
        float dl[3] = {dm[0], dm[1], dm[2]}, dql[4] = {dq[0], dq[1], dq[2], dq[3]}, dopl = dop;
        if (a_id >= 0 && visible)           // (an invisible Gaussian -- uncompacted build only -- has zero gradients: nothing to transform)
            motion_backward_apply(mloc, has_rot, q_raw, has_rdq, rdq, true, op_raw, pr0, pr1, pr2, dm, dq, dop, dl, dql, &dopl, pose_g, raw);
        if (raw && visible) {
            if (a_id < 0 && has_rot) dnormalize4(q, q_norm, dq, dql);              // through F.normalize of the raw quaternion
            const float o = sigmoidf_(op_raw);
            dopl *= o * (1.f - o);                                                 // through the sigmoid
        }
#if EMD_K8_STAGE_SMALL || EMD_K8_COMPACT
        if (staged5) {
            o_dl[0] = dl[0]; o_dl[1] = dl[1]; o_dl[2] = dl[2]; o_ds[0] = ds[0]; o_ds[1] = ds[1]; o_ds[2] = ds[2];
            o_dq[0] = dql[0]; o_dq[1] = dql[1]; o_dq[2] = dql[2]; o_dq[3] = dql[3]; o_dop = dopl;
        }
#endif
        if (a.dL_dmeans3D && !staged5) { st_f(a.dL_dmeans3D + 3 * i, dl[0]); st_f(a.dL_dmeans3D + 3 * i + 1, dl[1]); st_f(a.dL_dmeans3D + 3 * i + 2, dl[2]); }
        if (a.dL_dresidual_dx) { a.dL_dresidual_dx[3 * i] = dl[0]; a.dL_dresidual_dx[3 * i + 1] = dl[1]; a.dL_dresidual_dx[3 * i + 2] = dl[2]; }
        if (a.dL_drotations && !staged5) {
            if (EMD_K8_NT_ALL) store_f4_nt((float4*)(a.dL_drotations + 4 * i), make_float4(dql[0], dql[1], dql[2], dql[3]));
            else *(float4*)(a.dL_drotations + 4 * i) = make_float4(dql[0], dql[1], dql[2], dql[3]);
        }
        if (a.dL_dresidual_dq) {
            *(float4*)(a.dL_dresidual_dq + 4 * i) = a_id >= 0 ? make_float4(dql[0], dql[1], dql[2], dql[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (a.dL_dopacities && !staged5) st_f(a.dL_dopacities + i, dopl);
    }
#if EMD_K8_COMPACT
    if (nat_zero) {
        // the rows no compacted lane owns: zeros (the five main arrays only when they do not leave through the tile below)
        const size_t z = (size_t)inat;
        if (!staged5) {
            if (a.dL_dmeans2D) { a.dL_dmeans2D[3 * z] = 0.f; a.dL_dmeans2D[3 * z + 1] = 0.f; a.dL_dmeans2D[3 * z + 2] = 0.f; }
            if (a.dL_dscales) { a.dL_dscales[3 * z] = 0.f; a.dL_dscales[3 * z + 1] = 0.f; a.dL_dscales[3 * z + 2] = 0.f; }
            if (a.dL_dmeans3D) { a.dL_dmeans3D[3 * z] = 0.f; a.dL_dmeans3D[3 * z + 1] = 0.f; a.dL_dmeans3D[3 * z + 2] = 0.f; }
            if (a.dL_drotations) *(float4*)(a.dL_drotations + 4 * z) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.dL_dopacities) a.dL_dopacities[z] = 0.f;
        }
        if (a.dL_dmeans2D_abs) { a.dL_dmeans2D_abs[2 * z] = 0.f; a.dL_dmeans2D_abs[2 * z + 1] = 0.f; }
        if (a.dL_dsh_color) { a.dL_dsh_color[3 * z] = 0.f; a.dL_dsh_color[3 * z + 1] = 0.f; a.dL_dsh_color[3 * z + 2] = 0.f; }
        if (a.dL_dcolors) { a.dL_dcolors[3 * z] = 0.f; a.dL_dcolors[3 * z + 1] = 0.f; a.dL_dcolors[3 * z + 2] = 0.f; }
        for (int k = 0; k < a.num_extra; k++)
            if (a.dL_dextra[k]) { a.dL_dextra[k][3 * z] = 0.f; a.dL_dextra[k][3 * z + 1] = 0.f; a.dL_dextra[k][3 * z + 2] = 0.f; }
        if (a.dL_dcov3D) {
#pragma unroll
            for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * z + k] = 0.f;
        }
        if (a.dL_dresidual_dx) { a.dL_dresidual_dx[3 * z] = 0.f; a.dL_dresidual_dx[3 * z + 1] = 0.f; a.dL_dresidual_dx[3 * z + 2] = 0.f; }
        if (a.dL_dresidual_dq) *(float4*)(a.dL_dresidual_dq + 4 * z) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#endif
#if EMD_K8_STAGE_SMALL || EMD_K8_COMPACT
    if (staged5) {
        // tile: [B x 3 means3D | B x 3 scales | B x 4 rotations | B opacities | B x 3 means2D] floats (B = K8_BLOCK): 14 of the SH tile's 26 KB at B = 256
        constexpr int B = K8_BLOCK, O_SC = 3 * B, O_ROT = 6 * B, O_OP = 10 * B, O_M2 = 11 * B;
        static_assert(14 * B * 4 <= (int)sizeof(float4) * K8_HALF * SH_ROW4, "the small-gradient tile must fit the SH staging tile");
        float* sf = reinterpret_cast<float*>(s_sh);
        const int t = threadIdx.x;
        if (in_range) {
            const int u = lrow;
            sf[3 * u] = o_dl[0]; sf[3 * u + 1] = o_dl[1]; sf[3 * u + 2] = o_dl[2];
            sf[O_SC + 3 * u] = o_ds[0]; sf[O_SC + 3 * u + 1] = o_ds[1]; sf[O_SC + 3 * u + 2] = o_ds[2];
            *reinterpret_cast<float4*>(sf + O_ROT + 4 * u) = make_float4(o_dq[0], o_dq[1], o_dq[2], o_dq[3]);
            sf[O_OP + u] = o_dop;
            sf[O_M2 + 3 * u] = gm2[0]; sf[O_M2 + 3 * u + 1] = gm2[1]; sf[O_M2 + 3 * u + 2] = 0.f;
        }
        if (EMD_K8_COMPACT && !nat_vis) {
            sf[3 * t] = 0.f; sf[3 * t + 1] = 0.f; sf[3 * t + 2] = 0.f;
            sf[O_SC + 3 * t] = 0.f; sf[O_SC + 3 * t + 1] = 0.f; sf[O_SC + 3 * t + 2] = 0.f;
            *reinterpret_cast<float4*>(sf + O_ROT + 4 * t) = make_float4(0.f, 0.f, 0.f, 0.f);
            sf[O_OP + t] = 0.f;
            sf[O_M2 + 3 * t] = 0.f; sf[O_M2 + 3 * t + 1] = 0.f; sf[O_M2 + 3 * t + 2] = 0.f;
        }
        __syncthreads();
        const size_t b = (size_t)blockIdx.x * K8_BLOCK;
        const float4* s4 = reinterpret_cast<const float4*>(sf);
        auto st4 = [](float4* p, float4 v) { if (EMD_K8_STAGE_NT) store_f4_nt(p, v); else *p = v; };
        if (t < 3 * B / 4) {
            st4(reinterpret_cast<float4*>(a.dL_dmeans3D + 3 * b) + t, s4[t]);
            st4(reinterpret_cast<float4*>(a.dL_dscales + 3 * b) + t, s4[O_SC / 4 + t]);
            st4(reinterpret_cast<float4*>(a.dL_dmeans2D + 3 * b) + t, s4[O_M2 / 4 + t]);
        }
        st4(reinterpret_cast<float4*>(a.dL_drotations + 4 * b) + t, s4[O_ROT / 4 + t]);
        if (t < B / 4) st4(reinterpret_cast<float4*>(a.dL_dopacities + b) + t, s4[O_OP / 4 + t]);
    }
#endif
    if ((a.flags & EMD_FLAG_MOTION) && a.dL_dactor_pose) reduce_pose_grad(a_id, pose_g, a.dL_dactor_pose);
}

// ---------------------------------------------------------------------------------------------------
// stand-alone motion / SH kernels
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(EMD_BLOCK) k_motion_forward(int n, const float* means, const float* quats,
                                                              const float* opac, EmdMotion mo, float* wm, float* wq,
                                                              float* wo) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (i >= n) return;
    float m[3], q[4] = {1.f, 0.f, 0.f, 0.f}, o = 0.f;
    motion_point(i, means, quats, opac, mo, m, q, &o);
    if (wm) { wm[3 * i] = m[0]; wm[3 * i + 1] = m[1]; wm[3 * i + 2] = m[2]; }
    if (wq && quats) *(float4*)(wq + 4 * i) = make_float4(q[0], q[1], q[2], q[3]);
    if (wo && opac) wo[i] = o;
}

__global__ void __launch_bounds__(EMD_BLOCK) k_sh_forward(int n, int deg, int M, const float* dirs,
                                                          const float* coeffs, float* rgb) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (i >= n) return;
    float d[3] = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
    float nn = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    d[0] /= nn; d[1] /= nn; d[2] /= nn;
    float bs[16];
    sh_basis(deg, d, bs);
    const int K = (deg + 1) * (deg + 1);
    const float* sh = coeffs + (size_t)i * M * 3;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int k = 0; k < K; k++) { c0 += bs[k] * sh[3 * k]; c1 += bs[k] * sh[3 * k + 1]; c2 += bs[k] * sh[3 * k + 2]; }
    rgb[3 * i] = c0; rgb[3 * i + 1] = c1; rgb[3 * i + 2] = c2;
}

__global__ void __launch_bounds__(EMD_BLOCK) k_motion_backward(int n, const float* means, const float* quats,
                                                               const float* opac, EmdMotion mo, const float* g_wm,
                                                               const float* g_wq, const float* g_wo, float* d_means,
                                                               float* d_quats, float* d_opac, float* d_pose,
                                                               float* d_rdx, float* d_rdq) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    int a_id = -1;
    float pose_g[12];
#pragma unroll
    for (int k = 0; k < 12; k++) pose_g[k] = 0.f;
    if (i < n) {
        float dm[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f}, dop = 0.f;
        if (g_wm) { dm[0] = g_wm[3 * i]; dm[1] = g_wm[3 * i + 1]; dm[2] = g_wm[3 * i + 2]; }
        if (g_wq) { const float4 t = *(const float4*)(g_wq + 4 * i); dq[0] = t.x; dq[1] = t.y; dq[2] = t.z; dq[3] = t.w; }
        if (g_wo) dop = g_wo[i];
        a_id = mo.actor_id ? mo.actor_id[i] : -1;
        float dl[3] = {dm[0], dm[1], dm[2]}, dql[4] = {dq[0], dq[1], dq[2], dq[3]}, dopl = dop;
        if (a_id >= 0) motion_point_backward(i, a_id, means, quats, opac, mo, dm, dq, dop, dl, dql, &dopl, pose_g);
        if (d_means) { d_means[3 * i] = dl[0]; d_means[3 * i + 1] = dl[1]; d_means[3 * i + 2] = dl[2]; }
        if (d_rdx) { d_rdx[3 * i] = dl[0]; d_rdx[3 * i + 1] = dl[1]; d_rdx[3 * i + 2] = dl[2]; }
        if (d_quats) *(float4*)(d_quats + 4 * i) = make_float4(dql[0], dql[1], dql[2], dql[3]);
        if (d_rdq) *(float4*)(d_rdq + 4 * i) = a_id >= 0 ? make_float4(dql[0], dql[1], dql[2], dql[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (d_opac) d_opac[i] = dopl;
    }
    if (d_pose) reduce_pose_grad(a_id, pose_g, d_pose);
}

__global__ void __launch_bounds__(EMD_BLOCK) k_sh_backward(int n, int deg, int M, const float* dirs,
                                                           const float* coeffs, const float* g_rgb, float* d_coeffs,
                                                           float* d_dirs) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (i >= n) return;
    float d0[3] = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
    float nn = sqrtf((d0[0] * d0[0] + d0[1] * d0[1]) + d0[2] * d0[2]);
    float d[3] = {d0[0] / nn, d0[1] / nn, d0[2] / nn};
    const float gc[3] = {g_rgb[3 * i], g_rgb[3 * i + 1], g_rgb[3 * i + 2]};
    const int K = (deg + 1) * (deg + 1);
    if (d_coeffs) {
        float bs[16];
        sh_basis(deg, d, bs);
        float* o = d_coeffs + (size_t)i * M * 3;
        for (int k = 0; k < M; k++) {
            float bk = k < K ? bs[k] : 0.f;
            o[3 * k] = bk * gc[0]; o[3 * k + 1] = bk * gc[1]; o[3 * k + 2] = bk * gc[2];
        }
    }
    if (d_dirs) {
        float gd[3];
        sh_dir_backward(deg, d, coeffs + (size_t)i * M * 3, gc, gd);
        float dot = (d[0] * gd[0] + d[1] * gd[1]) + d[2] * gd[2];
        d_dirs[3 * i] = (gd[0] - d[0] * dot) / nn;
        d_dirs[3 * i + 1] = (gd[1] - d[1] * dot) / nn;
        d_dirs[3 * i + 2] = (gd[2] - d[2] * dot) / nn;
    }
}

// Dense, view-averaged SH gradient from the per-view rank-one factors (emd_sh_grad_from_factors): one Gaussian per lane,
// rows leave through LDS as coalesced dwordx4 stores like K8's.
__global__ void __launch_bounds__(EMD_BLOCK) k_sh_grad_from_factors(int n, int V, int deg, const float* __restrict__ means,
                                                                    EmdMotion mo, int pose_per_view, const float* __restrict__ campos,
                                                                    const float* __restrict__ gc, float scale,
                                                                    float* __restrict__ d_shs) {
    __shared__ float4 s_sh[(EMD_BLOCK / 2) * SH_ROW4];    // half of the block's rows at a time (26 KB: more resident waves)
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; k++) acc[k] = 0.f;
    if (i < n) {
        float m[3], qd[4], od;
        if (mo.actor_id || mo.residual_dx) motion_point(i, means, nullptr, nullptr, mo, m, qd, &od, false);
        else { m[0] = means[3 * i]; m[1] = means[3 * i + 1]; m[2] = means[3 * i + 2]; }
        const int K = (deg + 1) * (deg + 1);
        // views of different timestamps (6 cameras on 8 ranks): an actor's Gaussians sit at a different world position in every
        // view, so the pose table is per view ([V][A][12]); static Gaussians keep the position computed above
        const bool moving = pose_per_view && mo.actor_id && mo.actor_id[i] >= 0;
        for (int v = 0; v < V; v++) {
            const float* g = gc + ((size_t)v * n + i) * 3;
            const float g0 = g[0], g1 = g[1], g2 = g[2];
            if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;          // not visible in view v
            if (moving && v > 0) {
                EmdMotion mv = mo;
                mv.actor_pose = mo.actor_pose + (size_t)v * mo.num_actors * EMD_ACTOR_STRIDE;
                motion_point(i, means, nullptr, nullptr, mv, m, qd, &od, false);
            }
            float d[3] = {m[0] - campos[3 * v], m[1] - campos[3 * v + 1], m[2] - campos[3 * v + 2]};
            const float nn = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            d[0] /= nn; d[1] /= nn; d[2] /= nn;
            float bs[16];
            sh_basis(deg, d, bs);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < K) { acc[3 * k] += bs[k] * g0; acc[3 * k + 1] += bs[k] * g1; acc[3 * k + 2] += bs[k] * g2; }
            }
        }
    }
    const size_t lim4 = (size_t)n * 12;
    float4* out = (float4*)d_shs;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if ((int)(threadIdx.x >> 7) == h) {
#pragma unroll
            for (int j = 0; j < 12; j++)
                s_sh[(threadIdx.x & 127) * SH_ROW4 + j] = make_float4(acc[4 * j] * scale, acc[4 * j + 1] * scale, acc[4 * j + 2] * scale, acc[4 * j + 3] * scale);
        }
        __syncthreads();
        const size_t base4 = ((size_t)blockIdx.x * EMD_BLOCK + 128 * h) * 12;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const uint32_t idx = threadIdx.x + EMD_BLOCK * j;
            if (base4 + idx < lim4) out[base4 + idx] = s_sh[(idx / 12) * SH_ROW4 + (idx % 12)];
        }
        __syncthreads();
    }
}

// Densification statistics of one view, in place and without the boolean-mask indexing (= a device-to-host sync) of the
// reference: for every visible Gaussian  accum += |dL/dmean2D.xy|, denom += 1, max_radii = max(max_radii, radius)
// (S3Gaussian/scene/gaussian_model.py:728-730, train.py:403-406).
__global__ void __launch_bounds__(EMD_BLOCK) k_densification_stats(int n, const int32_t* __restrict__ radii,
                                                                   const float* __restrict__ g2d /*[N,3]*/,
                                                                   float* __restrict__ accum, float* __restrict__ denom,
                                                                   float* __restrict__ max_radii) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = g2d[3 * i], gy = g2d[3 * i + 1];
    if (accum) accum[i] += sqrtf(gx * gx + gy * gy);
    if (denom) denom[i] += 1.f;
    if (max_radii) max_radii[i] = fmaxf(max_radii[i], (float)r);
}

__global__ void __launch_bounds__(EMD_BLOCK) k_export_geometry(int N, const float4* rec, const uint2* binrec,
                                                               float* means2D, float* depths, float* conic_opacity,
                                                               float* rgb, float* normal, uint32_t* tiles_touched) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (i >= N) return;
    const uint32_t tt = binrec[i].y >> 12;               // upstream's tiles touched (the enumerated rectangle may be smaller)
    const bool vis = tt != 0;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 r0 = vis ? rec[(size_t)i * EMD_REC_F4] : z, r1 = vis ? rec[(size_t)i * EMD_REC_F4 + 1] : z,
                 r2 = vis ? rec[(size_t)i * EMD_REC_F4 + 2] : z;
    if (means2D) { means2D[2 * i] = r0.x; means2D[2 * i + 1] = r0.y; }
    if (depths) depths[i] = r0.z;
    if (conic_opacity) *(float4*)(conic_opacity + 4 * i) = make_float4(r1.x, r1.y, r1.z, r0.w);
    if (rgb) { rgb[3 * i] = r2.x; rgb[3 * i + 1] = r2.y; rgb[3 * i + 2] = r2.z; }
    if (normal) { const float4 r3 = vis ? rec[(size_t)i * EMD_REC_F4 + 3] : z; normal[3 * i] = r3.x; normal[3 * i + 1] = r3.y; normal[3 * i + 2] = r3.z; }
    if (tiles_touched) tiles_touched[i] = tt;
}

__global__ void __launch_bounds__(EMD_BLOCK) k_activations(int n, const float* ls, float* sc, const float* rq, float* q,
                                                           const float* lo, float* o) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (ls && sc) { sc[3 * i] = expf(ls[3 * i]); sc[3 * i + 1] = expf(ls[3 * i + 1]); sc[3 * i + 2] = expf(ls[3 * i + 2]); }
    if (rq && q) {
        const float4 t = *(const float4*)(rq + 4 * i);
        float v[4] = {t.x, t.y, t.z, t.w};
        const float nn = fmaxf(quat_norm(v), 1e-12f);
        *(float4*)(q + 4 * i) = make_float4(v[0] / nn, v[1] / nn, v[2] / nn, v[3] / nn);
    }
    if (lo && o) o[i] = sigmoidf_(lo[i]);
}

// ---------------------------------------------------------------------------------------------------
// Per-frame actor pose table (training branch of rigid.py:478-568): one lane per actor.
//   q_mean = normalize(q_f)                       rotation applied to local means          (rigid.py:499-503)
//   trans  = t_f + dt      (dt skipped when NaN)                                            (rigid.py:519-532)
//   q_rot  = normalize(q_f (x) dq)  (dq skipped when NaN) composed onto local quaternions   (rigid.py:547-566)
// Replaces ~25 launch-bound torch kernels (normalize / cat / index and their backward) per step.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool any_nan(const float* v, int n) {
    bool b = false;
    for (int k = 0; k < n; k++) b |= !(v[k] == v[k]);
    return b;
}

__global__ void k_actor_pose_forward(int A, const float* __restrict__ q_f, const float* __restrict__ t_f,
                                     const uint8_t* __restrict__ valid, const float* __restrict__ dt,
                                     const float* __restrict__ dq, float* __restrict__ pose, const int32_t* __restrict__ frame_dev) {
    if (frame_dev) {        // q_f / t_f / valid are the whole [F, A, .] tables and the frame index lives on the device (hipGraph replay)
        const size_t f = (size_t)frame_dev[0];
        q_f += f * A * 4; t_f += f * A * 3;
        if (valid) valid += f * A;
    }
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    const float q[4] = {q_f[4 * a], q_f[4 * a + 1], q_f[4 * a + 2], q_f[4 * a + 3]};
    const float n = fmaxf(quat_norm(q), 1e-12f);
    float* P = pose + (size_t)a * EMD_ACTOR_STRIDE;
    for (int k = 0; k < 4; k++) P[k] = q[k] / n;
    float t[3] = {t_f[3 * a], t_f[3 * a + 1], t_f[3 * a + 2]};
    if (dt && !any_nan(dt + 3 * a, 3)) { t[0] += dt[3 * a]; t[1] += dt[3 * a + 1]; t[2] += dt[3 * a + 2]; }
    P[4] = t[0]; P[5] = t[1]; P[6] = t[2];
    P[7] = valid ? (valid[a] ? 1.f : 0.f) : 1.f;
    float p[4] = {q[0], q[1], q[2], q[3]};
    if (dq && !any_nan(dq + 4 * a, 4)) { const float r[4] = {dq[4 * a], dq[4 * a + 1], dq[4 * a + 2], dq[4 * a + 3]}; quat_mul(q, r, p); }
    const float n2 = fmaxf(quat_norm(p), 1e-12f);
    for (int k = 0; k < 4; k++) P[8 + k] = p[k] / n2;
}

__global__ void k_actor_pose_backward(int A, const float* __restrict__ q_f, const float* __restrict__ dt,
                                      const float* __restrict__ dq, const float* __restrict__ g_pose,
                                      float* __restrict__ d_q_f, float* __restrict__ d_t_f, float* __restrict__ d_dt,
                                      float* __restrict__ d_dq, const int32_t* __restrict__ frame_dev) {
    if (frame_dev) { const size_t f = (size_t)frame_dev[0]; q_f += f * A * 4; d_q_f += f * A * 4; d_t_f += f * A * 3; }
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    const float* G = g_pose + (size_t)a * EMD_ACTOR_STRIDE;
    const float q[4] = {q_f[4 * a], q_f[4 * a + 1], q_f[4 * a + 2], q_f[4 * a + 3]};
    const float n = fmaxf(quat_norm(q), 1e-12f);
    const float qu[4] = {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
    const float gm[4] = {G[0], G[1], G[2], G[3]};
    float dqf[4];
    dnormalize4(qu, n, gm, dqf);
    const bool use_r = dq && !any_nan(dq + 4 * a, 4);
    float p[4] = {q[0], q[1], q[2], q[3]}, r[4] = {1.f, 0.f, 0.f, 0.f};
    if (use_r) { r[0] = dq[4 * a]; r[1] = dq[4 * a + 1]; r[2] = dq[4 * a + 2]; r[3] = dq[4 * a + 3]; quat_mul(q, r, p); }
    const float n2 = fmaxf(quat_norm(p), 1e-12f);
    const float pu[4] = {p[0] / n2, p[1] / n2, p[2] / n2, p[3] / n2};
    const float gr[4] = {G[8], G[9], G[10], G[11]};
    float dp[4];
    dnormalize4(pu, n2, gr, dp);
    float dr[4] = {0.f, 0.f, 0.f, 0.f};
    if (use_r) {   // p = q (x) r : dL/dq = dp (x) conj(r), dL/dr = conj(q) (x) dp
        const float rc[4] = {r[0], -r[1], -r[2], -r[3]}, qc[4] = {q[0], -q[1], -q[2], -q[3]};
        float t1[4];
        quat_mul(dp, rc, t1);
        quat_mul(qc, dp, dr);
        for (int k = 0; k < 4; k++) dqf[k] += t1[k];
    } else {
        for (int k = 0; k < 4; k++) dqf[k] += dp[k];
    }
    for (int k = 0; k < 4; k++) d_q_f[4 * a + k] = dqf[k];
    for (int k = 0; k < 3; k++) d_t_f[3 * a + k] = G[4 + k];
    if (d_dt) { const bool ok = dt && !any_nan(dt + 3 * a, 3); for (int k = 0; k < 3; k++) d_dt[3 * a + k] = ok ? G[4 + k] : 0.f; }
    if (d_dq) for (int k = 0; k < 4; k++) d_dq[4 * a + k] = dr[k];
}

// ---------------------------------------------------------------------------------------------------
// L1 photometric loss (S3Gaussian/utils/loss_utils.py:21-22, train.py:226): mean |a - b| and its gradient
// sign(a - b) / n in one pass (the reference spends ~9 element-wise launches on it per step).
// ---------------------------------------------------------------------------------------------------
#define L1_THREADS 1024      // (the block count is capped by the same-address atomics below: wide blocks keep enough bytes in flight)
// TICKET (round 5): `loss` needs no zero fill in front of the kernel -- that fill was a launch of its own (4.6 us for 4 bytes in the replayed
// step).  Every workgroup publishes its partial sum as ONE aligned 8-byte {value, tag = 1} granule (a single device-scope store: no fence, no
// wait -- MI355X_MICROARCH.md, "R2's granule needs no ordering at all") in a caller-kept scratch table that is zero between calls; workgroup 0
// polls the table with device-scope loads, adds the partials in workgroup order (a deterministic sum, unlike the float atomics it replaces),
// writes the loss and clears the tags for the next call.  Nobody but workgroup 0 waits for anything, so the scheme cannot deadlock however
// few workgroups are resident.  (First built with a returning atomic add + a ticket per workgroup: two serialised memory round trips at the
// end of EVERY workgroup made the kernel 5.8 us longer than the 4.6 us fill it replaced.)
template <bool TICKET>
__global__ void __launch_bounds__(L1_THREADS) k_l1_loss(size_t n, const float* __restrict__ a, const float* __restrict__ b,
                                                        float inv_n, float* __restrict__ loss, float* __restrict__ grad, uint32_t* __restrict__ scratch) {
    __shared__ float s_part[L1_THREADS / 64];
    float acc = 0.f;
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * L1_THREADS;
    for (size_t i = (size_t)blockIdx.x * L1_THREADS + threadIdx.x; i < n4; i += stride) {
        const float4 x = ((const float4*)a)[i], y = b ? ((const float4*)b)[i] : make_float4(0.f, 0.f, 0.f, 0.f);   // b == NULL: mean |a|
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
        if (grad) {
            auto sg = [inv_n](float d) { return d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f); };
            ((float4*)grad)[i] = make_float4(sg(d0), sg(d1), sg(d2), sg(d3));
        }
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * L1_THREADS + threadIdx.x; i < n; i += stride) {
        const float d = a[i] - (b ? b[i] : 0.f);
        acc += fabsf(d);
        if (grad) grad[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
    }
    acc = wave_reduce_to_lane63(acc);
    if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < L1_THREADS / 64; w++) t += s_part[w];
        if (!TICKET) { atomicAdd(loss, t * inv_n); return; }
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(scratch);
        const unsigned long long mine = ((unsigned long long)__float_as_uint(t * inv_n) << 32) | 1ull;
        if (blockIdx.x != 0) __hip_atomic_store(tab + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else s_part[0] = t * inv_n;
    }
    if (TICKET && blockIdx.x == 0) {
        __syncthreads();
        // one poller per other workgroup (gridDim.x <= 512 <= L1_THREADS): spin on ITS granule, hand the value to thread 0 through LDS
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(scratch);
        float v = threadIdx.x == 0 ? s_part[0] : 0.f;
        if (threadIdx.x > 0 && threadIdx.x < gridDim.x) {
            unsigned long long g = 0ull;
            for (;;) {
                g = __hip_atomic_load(tab + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g & 1ull) break;
                __builtin_amdgcn_s_sleep(2);
            }
            v = __uint_as_float((uint32_t)(g >> 32));
            __hip_atomic_store(tab + threadIdx.x, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the table is zero again for the next call
        }
        __syncthreads();                                      // (thread 0 has read s_part[0])
        v = wave_reduce_to_lane63(v);                         // a fixed tree: the same sum for the same partials
        if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < L1_THREADS / 64; w++) tot += s_part[w];
            loss[0] = tot;
        }
    }
}

__global__ void __launch_bounds__(EMD_BLOCK) k_abs_mean_backward(size_t n, const float* __restrict__ x, const float* __restrict__ g, float inv_n,
                                                                 float* __restrict__ out) {
    const float s = g[0] * inv_n;
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * EMD_BLOCK;
    auto sg = [s](float d) { return d > 0.f ? s : (d < 0.f ? -s : 0.f); };
    for (size_t i = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x; i < n4; i += stride) {
        const float4 v = ((const float4*)x)[i];
        ((float4*)out)[i] = make_float4(sg(v.x), sg(v.y), sg(v.z), sg(v.w));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x; i < n; i += stride) out[i] = sg(x[i]);
}

// The same for a PAIR of residuals that also carry an upstream gradient (the fine stage's dshs_coarse / dshs_fine: both receive the
// rasterizer's dL/dshs -- usually the very same tensor -- plus the gradient of their L1 regulariser):
//   out_a[i] = up_a[i] + sign(x_a[i]) g_a[0] / n,   out_b[i] = up_b[i] + sign(x_b[i]) g_b[0] / n
// in one pass that reads the shared upstream gradient once (instead of two sign passes and two adds over [N,16,3]).
__global__ void __launch_bounds__(EMD_BLOCK) k_residual_l1_backward(size_t n, const float* __restrict__ up_a, const float* __restrict__ up_b,
                                                                    const float* __restrict__ x_a, const float* __restrict__ x_b,
                                                                    const float* __restrict__ g_a, const float* __restrict__ g_b, float inv_n,
                                                                    float* __restrict__ out_a, float* __restrict__ out_b) {
    const float sa = g_a ? g_a[0] * inv_n : 0.f, sb = g_b ? g_b[0] * inv_n : 0.f;
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * EMD_BLOCK;
    auto sg = [](float d, float s) { return d > 0.f ? s : (d < 0.f ? -s : 0.f); };
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x; i < n4; i += stride) {
        const float4 ua = up_a ? ((const float4*)up_a)[i] : z;
        const float4 ub = (up_b == up_a) ? ua : (up_b ? ((const float4*)up_b)[i] : z);
        const float4 a = ((const float4*)x_a)[i], b = ((const float4*)x_b)[i];
        ((float4*)out_a)[i] = make_float4(ua.x + sg(a.x, sa), ua.y + sg(a.y, sa), ua.z + sg(a.z, sa), ua.w + sg(a.w, sa));
        ((float4*)out_b)[i] = make_float4(ub.x + sg(b.x, sb), ub.y + sg(b.y, sb), ub.z + sg(b.z, sb), ub.w + sg(b.w, sb));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x; i < n; i += stride) {
        out_a[i] = (up_a ? up_a[i] : 0.f) + sg(x_a[i], sa);
        out_b[i] = (up_b ? up_b[i] : 0.f) + sg(x_b[i], sb);
    }
}

}  // namespace

int emd_launch_residual_l1_backward(size_t n, const float* up_a, const float* up_b, const float* x_a, const float* x_b, const float* g_a,
                                    const float* g_b, float* out_a, float* out_b, hipStream_t st) {
    if (n == 0) return EMD_OK;
    size_t blocks = (n / 4 + EMD_BLOCK - 1) / EMD_BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_residual_l1_backward, dim3((unsigned)blocks), dim3(EMD_BLOCK), 0, st, n, up_a, up_b, x_a, x_b, g_a, g_b, 1.f / (float)n,
                       out_a, out_b);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_abs_mean_backward(size_t n, const float* x, const float* g, float* out, hipStream_t st) {
    if (n == 0) return EMD_OK;
    size_t blocks = (n / 4 + EMD_BLOCK - 1) / EMD_BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_abs_mean_backward, dim3((unsigned)blocks), dim3(EMD_BLOCK), 0, st, n, x, g, 1.f / (float)n, out);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_preprocess(const PreArgs& a, int part, hipStream_t st) {
    if (a.N <= 0) return EMD_OK;
    const int nb = (a.N + PRE_BLOCK - 1) / PRE_BLOCK;
    const bool res = a.shs_res0 != nullptr;
    if (part == 1) hipLaunchKernelGGL(k_preprocess<1>, dim3(nb), dim3(PRE_BLOCK), 0, st, a);
    else if (part == 2 && res) hipLaunchKernelGGL((k_preprocess<2, true>), dim3(nb), dim3(PRE_BLOCK), 0, st, a);
    else if (part == 2) hipLaunchKernelGGL(k_preprocess<2>, dim3(nb), dim3(PRE_BLOCK), 0, st, a);
    else if (res) hipLaunchKernelGGL((k_preprocess<0, true>), dim3(nb), dim3(PRE_BLOCK), 0, st, a);
    else hipLaunchKernelGGL(k_preprocess<0>, dim3(nb), dim3(PRE_BLOCK), 0, st, a);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_preprocess_backward(const PreBwdArgs& a, hipStream_t st) {
    if (a.N <= 0) return EMD_OK;
    const int nb = (a.N + K8_BLOCK - 1) / K8_BLOCK;
    hipLaunchKernelGGL(k_preprocess_backward, dim3(nb), dim3(K8_BLOCK), 0, st, a);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_motion_forward(int n, const float* means, const float* quats, const float* opac, const EmdMotion& mo,
                              float* wm, float* wq, float* wo, hipStream_t st) {
    if (n <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_motion_forward, dim3((n + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, n, means, quats,
                       opac, mo, wm, wq, wo);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_sh_forward(int n, int deg, int M, const float* dirs, const float* coeffs, float* rgb, hipStream_t st) {
    if (n <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_sh_forward, dim3((n + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, n, deg, M, dirs,
                       coeffs, rgb);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_motion_backward(int n, const float* means, const float* quats, const float* opac, const EmdMotion& mo,
                               const float* g_wm, const float* g_wq, const float* g_wo, float* d_means, float* d_quats,
                               float* d_opac, float* d_pose, float* d_rdx, float* d_rdq, hipStream_t st) {
    if (n <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_motion_backward, dim3((n + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, n, means, quats,
                       opac, mo, g_wm, g_wq, g_wo, d_means, d_quats, d_opac, d_pose, d_rdx, d_rdq);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_sh_backward(int n, int deg, int M, const float* dirs, const float* coeffs, const float* g_rgb,
                           float* d_coeffs, float* d_dirs, hipStream_t st) {
    if (n <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_sh_backward, dim3((n + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, n, deg, M, dirs,
                       coeffs, g_rgb, d_coeffs, d_dirs);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_sh_grad_from_factors(int n, int V, int deg, int M, const float* means, const EmdMotion& mo, int pose_per_view,
                                    const float* campos, const float* gc, float scale, float* d_shs, hipStream_t st) {
    if (n <= 0) return EMD_OK;
    if (M != 16) { emd_set_error("sh_grad_from_factors: the staged row store needs sh_coeffs == 16"); return EMD_ERR_INVALID; }
    hipLaunchKernelGGL(k_sh_grad_from_factors, dim3((n + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, n, V, deg, means, mo, pose_per_view, campos,
                       gc, scale, d_shs);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_densification_stats(int n, const int32_t* radii, const float* g2d, float* accum, float* denom, float* max_radii,
                                   hipStream_t st) {
    if (n <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_densification_stats, dim3((n + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, n, radii, g2d, accum, denom,
                       max_radii);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

// dL/d(SH colour) of every Gaussian, clamp-masked (0 when not visible): what K8 writes as dL_dsh_color, available as soon as the render
// backward has run -- accumulator row columns 7..9 and the clamp bits K1 left in row 2 of the record
__global__ void __launch_bounds__(EMD_BLOCK) k_sh_factor(int N, const int32_t* __restrict__ radii, const float4* __restrict__ rec,
                                                         const float* __restrict__ grad_rec, int stride, float* __restrict__ out) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (i >= N) return;
    float g[3] = {0.f, 0.f, 0.f};
    if (radii[i] > 0) {
        const float* r = grad_rec + (size_t)i * stride;
        const uint32_t bits = __float_as_uint(rec[(size_t)i * EMD_REC_F4 + 2].w);
        g[0] = (bits & 1u) ? 0.f : r[7]; g[1] = (bits & 2u) ? 0.f : r[8]; g[2] = (bits & 4u) ? 0.f : r[9];
    }
    out[3 * i] = g[0]; out[3 * i + 1] = g[1]; out[3 * i + 2] = g[2];
}

int emd_launch_sh_factor(int N, const int32_t* radii, const GeomWs& g, const float* grad_rec, int bwd_stride, float* dL_dsh_color, hipStream_t st) {
    if (N <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_sh_factor, dim3((N + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, N, radii, g.rec, grad_rec, bwd_stride, dL_dsh_color);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_export_geometry(int N, const GeomWs& g, float* means2D, float* depths, float* conic_opacity, float* rgb,
                               float* normal, uint32_t* tiles_touched, hipStream_t st) {
    if (N <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_export_geometry, dim3((N + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, N, g.rec,
                       g.binrec, means2D, depths, conic_opacity, rgb, normal, tiles_touched);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_activations(int n, const float* ls, float* sc, const float* rq, float* q, const float* lo, float* o, hipStream_t st) {
    if (n <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_activations, dim3((n + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, n, ls, sc, rq, q, lo, o);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_actor_pose_forward(int A, const float* q, const float* t, const uint8_t* valid, const float* dt, const float* dq,
                                  float* pose, const int32_t* frame_dev, hipStream_t st) {
    if (A <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_actor_pose_forward, dim3((A + 63) / 64), dim3(64), 0, st, A, q, t, valid, dt, dq, pose, frame_dev);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_actor_pose_backward(int A, const float* q, const float* dt, const float* dq, const float* g_pose, float* d_q,
                                   float* d_t, float* d_dt, float* d_dq, const int32_t* frame_dev, hipStream_t st) {
    if (A <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_actor_pose_backward, dim3((A + 63) / 64), dim3(64), 0, st, A, q, dt, dq, g_pose, d_q, d_t, d_dt, d_dq, frame_dev);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

int emd_launch_l1_loss(size_t n, const float* a, const float* b, float* loss, float* grad, uint32_t* scratch, hipStream_t st) {
    if (!scratch || n == 0) { int zrc = emd_zero_async(loss, sizeof(float), st); if (zrc) return zrc; }
    if (n == 0) return EMD_OK;
    size_t blocks = (n / 4 + L1_THREADS - 1) / L1_THREADS;
    if (blocks > 512) blocks = 512;     // one same-address float atomic per block: 2048 of them serialised for ~20 us
    if (blocks == 0) blocks = 1;
    if (scratch) hipLaunchKernelGGL((k_l1_loss<true>), dim3((unsigned)blocks), dim3(L1_THREADS), 0, st, n, a, b, 1.0f / (float)n, loss, grad, scratch);
    else hipLaunchKernelGGL((k_l1_loss<false>), dim3((unsigned)blocks), dim3(L1_THREADS), 0, st, n, a, b, 1.0f / (float)n, loss, grad, scratch);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
