// sky.hip -- sky cube-map lookup + final blend, forward and backward.
//
// Replaces, for the two call sites of the reference,
//   S3Gaussian/scene/sky_cubemap.py:41-87 + gaussian_renderer/__init__.py:299-301   (rays, mask, dr.texture, clamp, blend)
//   OmniRe/models/modules.py:174-208 + models/trainers/base.py:491-497              (dr.texture on view directions, blend)
// the chain get_rays_torch (7 element-wise / matmul launches over H*W*3) -> boolean-mask gather -> nvdiffrast
// `dr.texture(..., filter_mode='linear', boundary_mode='cube')` (CUDA-only) -> masked scatter -> permute -> clamp -> blend
// by ONE kernel per direction of the pass: a lane owns a pixel, builds its ray from the pinhole parameters in registers,
// selects the cube face (OpenGL convention), gathers the four bilinear taps (taps that leave the face are fetched from
// the adjacent face; at a cube corner the missing fourth tap is dropped and the other three renormalised), clamps and
// blends.  The backward recomputes the taps (cheaper than storing 4 indices + 4 weights per pixel) and scatters
// dL/dtexel with float atomics; dL/d(foreground) and dL/d(opacity) of the blend come out of the same kernel.
// HBM-bound by construction: 3 planar image reads/writes per pixel + 4 x 12 B texel gathers that hit L2.
// The lookup restates the published nvdiffrast algorithm (oracle/sky_oracle.py; PARITY UNPINNED for that part).
#include "common.h"
#include "device_utils.h"

namespace {

struct Tap { uint32_t idx[4]; float w[4]; };

// direction -> face, (u, v) in [0,1]
__device__ __forceinline__ int index_cube(float x, float y, float z, float& u, float& v) {
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    int face;
    float c, sc, tc;
    if (az > fmaxf(ax, ay)) { c = z; face = 4; sc = c > 0.f ? x : -x; tc = -y; }
    else if (ay > ax)       { c = y; face = 2; sc = x; tc = c > 0.f ? z : -z; }
    else                    { c = x; face = 0; sc = c > 0.f ? -z : z; tc = -y; }
    if (c < 0.f) face += 1;
    const float m = 0.5f / fabsf(c);
    u = fminf(fmaxf(sc * m + 0.5f, 0.f), 1.f);
    v = fminf(fmaxf(tc * m + 0.5f, 0.f), 1.f);
    return face;
}

__device__ __forceinline__ void face_uv_to_dir(int face, float u, float v, float& x, float& y, float& z) {
    const float s = 2.f * u - 1.f, t = 2.f * v - 1.f;
    switch (face) {
        case 0: x = 1.f;  y = -t;  z = -s;  break;
        case 1: x = -1.f; y = -t;  z = s;   break;
        case 2: x = s;    y = 1.f; z = t;   break;
        case 3: x = s;    y = -1.f; z = -t; break;
        case 4: x = s;    y = -t;  z = 1.f; break;
        default: x = -s;  y = -t;  z = -1.f; break;
    }
}

__device__ __forceinline__ void cube_taps(float dx, float dy, float dz, int res, Tap& tp) {
    float u, v;
    const int face = index_cube(dx, dy, dz, u, v);
    const float uu = u * (float)res - 0.5f, vv = v * (float)res - 0.5f;
    const float fu0 = floorf(uu), fv0 = floorf(vv);
    const float fu = uu - fu0, fv = vv - fv0;
    const int iu0 = (int)fu0, iv0 = (int)fv0;
    const float eps = 0.25f / (float)res;
    float wsum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int iu = iu0 + (k & 1), iv = iv0 + (k >> 1);
        float w = ((k & 1) ? fu : 1.f - fu) * ((k >> 1) ? fv : 1.f - fv);
        const bool out_u = iu < 0 || iu >= res, out_v = iv < 0 || iv >= res;
        int ff = face, ju = iu, jv = iv;
        if (out_u && out_v) { w = 0.f; ff = 0; ju = 0; jv = 0; }        // cube corner: this tap does not exist
        else if (out_u || out_v) {
            // texel centre of the tap with the coordinate that left the face pushed just beyond the edge, re-indexed
            const float tu = iu < 0 ? -eps : (iu >= res ? 1.f + eps : ((float)iu + 0.5f) / (float)res);
            const float tv = iv < 0 ? -eps : (iv >= res ? 1.f + eps : ((float)iv + 0.5f) / (float)res);
            float x, y, z, u2, v2;
            face_uv_to_dir(face, tu, tv, x, y, z);
            ff = index_cube(x, y, z, u2, v2);
            ju = min(max((int)floorf(u2 * (float)res), 0), res - 1);
            jv = min(max((int)floorf(v2 * (float)res), 0), res - 1);
        }
        tp.idx[k] = (uint32_t)((ff * res + jv) * res + ju);
        tp.w[k] = w;
        wsum += w;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) tp.w[k] = tp.w[k] / wsum;
}

// ray of pixel (px, py): get_rays_torch (S3Gaussian/utils/graphics_utils.py:220-241), same operation order
__device__ __forceinline__ void pixel_ray(const EmdSkyArgs& a, int px, int py, size_t p, float& dx, float& dy, float& dz) {
    if (a.dirs) { dx = a.dirs[3 * p]; dy = a.dirs[3 * p + 1]; dz = a.dirs[3 * p + 2]; return; }
    const float ox = a.jitter ? a.jitter[2 * p] : 0.5f, oy = a.jitter ? a.jitter[2 * p + 1] : 0.5f;
    const float X = (float)px + ox, Y = (float)py + oy;
    const float* Ki = a.camera_dev ? a.camera_dev : a.Kinv;           // (device copy of the 21 floats: a recorded pass serves every camera)
    const float* R = a.camera_dev ? a.camera_dev + 9 : a.R; const float* T = a.camera_dev ? a.camera_dev + 18 : a.T;
    // pixel_camera = (X, Y, 1) Kinv^T ; pixel_world = (pixel_camera - T) R ; rays_o = -(R^T T)
    float pc[3], ro[3], pw[3];
#pragma unroll
    for (int k = 0; k < 3; k++) pc[k] = (X * Ki[3 * k] + Y * Ki[3 * k + 1]) + Ki[3 * k + 2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        pw[k] = ((pc[0] - T[0]) * R[k] + (pc[1] - T[1]) * R[3 + k]) + (pc[2] - T[2]) * R[6 + k];
        ro[k] = -((R[k] * T[0] + R[3 + k] * T[1]) + R[6 + k] * T[2]);
    }
    const float rx = pw[0] - ro[0], ry = pw[1] - ro[1], rz = pw[2] - ro[2];
    const float n = sqrtf((rx * rx + ry * ry) + rz * rz);
    dx = rx / n; dy = ry / n; dz = rz / n;
}

__global__ void __launch_bounds__(EMD_BLOCK) k_sky_forward(EmdSkyArgs a) {
    const size_t P = (size_t)a.height * a.width;
    const size_t p = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (p >= P) return;
    const float acc = a.acc ? a.acc[p] : 0.f;
    const bool sampled = !(a.acc && a.mask_threshold >= 0.f) || (1.f - acc) > a.mask_threshold;
    float s[3] = {a.fill, a.fill, a.fill};
    if (sampled) {
        float dx, dy, dz;
        pixel_ray(a, (int)(p % (size_t)a.width), (int)(p / (size_t)a.width), p, dx, dy, dz);
        Tap tp;
        cube_taps(dx, dy, dz, a.resolution, tp);
        s[0] = s[1] = s[2] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float* t = a.cube + (size_t)tp.idx[k] * 3;
            s[0] += tp.w[k] * t[0]; s[1] += tp.w[k] * t[1]; s[2] += tp.w[k] * t[2];
        }
    }
    if (a.flags & EMD_SKY_CLAMP01) {
#pragma unroll
        for (int c = 0; c < 3; c++) s[c] = fminf(fmaxf(s[c], 0.f), 1.f);
    }
    if (a.sky) {
        if (a.flags & EMD_SKY_INTERLEAVED) { a.sky[3 * p] = s[0]; a.sky[3 * p + 1] = s[1]; a.sky[3 * p + 2] = s[2]; }
        else { a.sky[p] = s[0]; a.sky[P + p] = s[1]; a.sky[2 * P + p] = s[2]; }
    }
    if (a.out && a.fg) {
        const bool il = (a.flags & EMD_SKY_INTERLEAVED) != 0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t q = il ? 3 * p + c : (size_t)c * P + p;
            const float f = a.fg[q];
            a.out[q] = (a.flags & EMD_SKY_BLEND_S3G) ? f * acc + s[c] * (1.f - acc) : f + s[c] * (1.f - acc);
        }
    }
}

// Texel gradients are combined per workgroup before they touch memory.  A workgroup owns a 16 x 16 pixel tile, whose
// bilinear taps fall on a few dozen texels when the map is magnified (1024^2 faces under a 1700-px focal length: ~7
// pixels per texel); issued one by one, that many float atomics to the same address serialise in L2 (measured
// 1.17 ms for a 1066 x 1600 view).  The tile's taps lie in a small window of one face, so the workgroup accumulates into
// an LDS window of SKY_WIN x SKY_WIN texels anchored at the tap of its first pixel (plain ds_add_f32, no hashing); taps
// outside the window or on another face (tiles that straddle a cube edge, heavily minified maps) go straight to the
// global atomic.  One global atomic per touched texel and channel leaves the workgroup.
#define SKY_WIN 24
__global__ void __launch_bounds__(EMD_BLOCK) k_sky_backward(EmdSkyBwdArgs b) {
    // fp64 cells: ds_add_f64 is the native LDS float add that runs at rate on gfx950 (8.9 clocks per wave64 instruction; ds_add_f32, which a plain
    // atomicAdd on a __shared__ float compiles to, takes 193: profiles/r03_lds_atomic_microbench.txt -- twelve of them per pixel were this kernel)
    __shared__ double s_val[SKY_WIN * SKY_WIN * 3];
    __shared__ int s_org[3];     // face, u0, v0 of the window
    const EmdSkyArgs& a = b.f;
    for (int i = threadIdx.x; i < SKY_WIN * SKY_WIN * 3; i += EMD_BLOCK) s_val[i] = 0.0;
    const size_t P = (size_t)a.height * a.width;
    const int res = a.resolution;
    size_t p;
    int px, py;
    bool valid;
    if (a.height > 1) {        // 16 x 16 pixel tiles: neighbouring pixels share texels
        const int tiles_x = (a.width + 15) / 16;
        px = (int)(blockIdx.x % (unsigned)tiles_x) * 16 + (int)(threadIdx.x & 15);
        py = (int)(blockIdx.x / (unsigned)tiles_x) * 16 + (int)(threadIdx.x >> 4);
        valid = px < a.width && py < a.height;
        p = (size_t)py * a.width + px;
    } else {
        p = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x;
        valid = p < P;
        px = (int)p; py = 0;
    }
    float acc = 0.f, gs[3] = {0.f, 0.f, 0.f};
    bool sampled = false;
    Tap tp;
    const bool il = (a.flags & EMD_SKY_INTERLEAVED) != 0;
    // the pixel's inputs first, unconditionally (an absent input reads the cube map's first texel and is not used; a pixel outside the image reads
    // pixel 0): with the loads inside the branches that use them the thread made a trip to memory per input, one after the other
    const size_t pcl = valid ? p : 0;
    const bool blend = b.dL_dout && a.fg;
    const float acc_raw = *(a.acc ? a.acc + pcl : a.cube);
    float gsky_raw[3], gout_raw[3], fg_raw[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const size_t q = il ? 3 * pcl + c : (size_t)c * P + pcl;
        gsky_raw[c] = *(b.dL_dsky ? b.dL_dsky + q : a.cube);
        gout_raw[c] = *(blend ? b.dL_dout + q : a.cube);
        fg_raw[c] = *(blend ? a.fg + q : a.cube);
    }
    if (valid) {
        acc = a.acc ? acc_raw : 0.f;
        sampled = !(a.acc && a.mask_threshold >= 0.f) || (1.f - acc) > a.mask_threshold;
        float s[3] = {a.fill, a.fill, a.fill};
        bool pass[3] = {false, false, false};
        if (sampled) {
            float dx, dy, dz;
            pixel_ray(a, px, py, p, dx, dy, dz);
            cube_taps(dx, dy, dz, res, tp);
            s[0] = s[1] = s[2] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float* t = a.cube + (size_t)tp.idx[k] * 3;
                s[0] += tp.w[k] * t[0]; s[1] += tp.w[k] * t[1]; s[2] += tp.w[k] * t[2];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) pass[c] = !(a.flags & EMD_SKY_CLAMP01) || (s[c] >= 0.f && s[c] <= 1.f);   // torch.clamp backward
        }
        if (a.flags & EMD_SKY_CLAMP01) {
#pragma unroll
            for (int c = 0; c < 3; c++) s[c] = fminf(fmaxf(s[c], 0.f), 1.f);
        }
        float dacc = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t q = il ? 3 * p + c : (size_t)c * P + p;
            gs[c] = b.dL_dsky ? gsky_raw[c] : 0.f;
            if (blend) {
                const float g = gout_raw[c], f = fg_raw[c];
                gs[c] += g * (1.f - acc);
                if (a.flags & EMD_SKY_BLEND_S3G) { dacc += g * (f - s[c]); if (b.dL_dfg) b.dL_dfg[q] = g * acc; }
                else { dacc -= g * s[c]; if (b.dL_dfg) b.dL_dfg[q] = g; }
            }
            if (!pass[c]) gs[c] = 0.f;
        }
        if (b.dL_dacc) b.dL_dacc[p] = dacc;
    }
    if (!b.dL_dcube) return;
    // window origin: centred on the first tap of the lowest contributing lane of the workgroup
    __shared__ int s_leader;
    if (threadIdx.x == 0) s_leader = EMD_BLOCK;
    __syncthreads();                                  // also: s_val is cleared
    const bool contributes = valid && sampled && (gs[0] != 0.f || gs[1] != 0.f || gs[2] != 0.f);
    if (contributes) atomicMin(&s_leader, (int)threadIdx.x);
    __syncthreads();
    if (s_leader == EMD_BLOCK) return;                // nothing to scatter (uniform)
    if ((int)threadIdx.x == s_leader) {
        const uint32_t t0 = tp.idx[0];
        const int f = (int)(t0 / (uint32_t)(res * res)), r = (int)(t0 % (uint32_t)(res * res));
        s_org[0] = f; s_org[1] = r % res - SKY_WIN / 2; s_org[2] = r / res - SKY_WIN / 2;
    }
    __syncthreads();
    const int of = s_org[0], ou = s_org[1], ov = s_org[2];
    if (contributes) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (tp.w[k] == 0.f) continue;
            const uint32_t key = tp.idx[k];
            const int f = (int)(key / (uint32_t)(res * res)), r = (int)(key % (uint32_t)(res * res));
            const int wu = r % res - ou, wv = r / res - ov;
            const bool in_win = f == of && wu >= 0 && wu < SKY_WIN && wv >= 0 && wv < SKY_WIN;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (gs[c] == 0.f) continue;
                if (in_win) lds_add_f64(&s_val[(wv * SKY_WIN + wu) * 3 + c], tp.w[k] * gs[c]);
                else atomicAdd(b.dL_dcube + (size_t)key * 3 + c, tp.w[k] * gs[c]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SKY_WIN * SKY_WIN * 3; i += EMD_BLOCK) {
        const float v = (float)s_val[i];
        if (v == 0.f) continue;
        const int t = i / 3, wu = t % SKY_WIN + ou, wv = t / SKY_WIN + ov;
        if (wu < 0 || wu >= res || wv < 0 || wv >= res) continue;     // never written: in_win implies a real texel
        atomicAdd(b.dL_dcube + ((size_t)(of * res + wv) * res + wu) * 3 + (i % 3), v);
    }
}

int check_sky(const EmdSkyArgs* a, const char* who) {
    if (!a) { emd_set_error("%s: null args", who); return EMD_ERR_INVALID; }
    if (a->height <= 0 || a->width <= 0 || a->resolution <= 0) { emd_set_error("%s: bad sizes H=%d W=%d res=%d", who, a->height, a->width, a->resolution); return EMD_ERR_INVALID; }
    if (!a->cube) { emd_set_error("%s: cube must not be null", who); return EMD_ERR_INVALID; }
    if ((a->flags & EMD_SKY_BLEND_S3G) && (a->flags & EMD_SKY_BLEND_ADD)) { emd_set_error("%s: choose one blend mode", who); return EMD_ERR_INVALID; }
    if (a->out && (!a->fg || !a->acc || !(a->flags & (EMD_SKY_BLEND_S3G | EMD_SKY_BLEND_ADD)))) { emd_set_error("%s: blend output needs fg, acc and a blend mode", who); return EMD_ERR_INVALID; }
    return EMD_OK;
}

}  // namespace

extern "C" int emd_sky_forward(const EmdSkyArgs* a, void* hip_stream) {
    int rc = check_sky(a, "sky_forward");
    if (rc) return rc;
    if (!a->sky && !a->out) { emd_set_error("sky_forward: no output requested"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t P = (size_t)a->height * a->width;
    hipLaunchKernelGGL(k_sky_forward, dim3((unsigned)((P + EMD_BLOCK - 1) / EMD_BLOCK)), dim3(EMD_BLOCK), 0, st, *a);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_sky_backward(const EmdSkyBwdArgs* b, void* hip_stream) {
    if (!b) { emd_set_error("sky_backward: null args"); return EMD_ERR_INVALID; }
    int rc = check_sky(&b->f, "sky_backward");
    if (rc) return rc;
    if (!b->dL_dout && !b->dL_dsky) { emd_set_error("sky_backward: no incoming gradient"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t P = (size_t)b->f.height * b->f.width;
    if (b->dL_dcube)
        { int zrc = emd_zero_async(b->dL_dcube, (size_t)6 * b->f.resolution * b->f.resolution * 3 * sizeof(float), st); if (zrc) return zrc; }
    const unsigned grid = b->f.height > 1 ? (unsigned)(((b->f.width + 15) / 16) * ((b->f.height + 15) / 16))
                                          : (unsigned)((P + EMD_BLOCK - 1) / EMD_BLOCK);
    hipLaunchKernelGGL(k_sky_backward, dim3(grid), dim3(EMD_BLOCK), 0, st, *b);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
