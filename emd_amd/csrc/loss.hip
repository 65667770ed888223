// loss.hip -- the image-loss tail of the training step, fused.
//
//   loss = L1(image, gt) + lambda_depth * L2(norm. depth on valid lidar pixels) + lambda_dssim * (1 - SSIM_11x11(image, gt))
//        + lambda_sky * BCE(weight, sky mask)                      S3Gaussian/train.py:226-363, utils/loss_utils.py:21-98
// and its gradients dL/dimage [3,H,W], dL/ddepth [1,H,W], dL/dweight [1,H,W] in the planar layouts the render backward
// (K7) consumes.  The reference spends 5 grouped 11x11 convolutions + ~40 element-wise launches forward and as many
// backward, with a dozen image-sized temporaries; here:
//   k_loss_pointwise   one pass over the pixels: L1 term and its gradient, depth term (valid mask, clamp, squared error,
//                      valid count; gradient left un-normalised), sky BCE term and gradient; block sums -> 5 atomics
//   k_ssim_forward     32x32 output tile per workgroup and channel: 42x42 halo of both images in LDS (zero padding),
//                      separable 11-tap Gaussian (horizontal then vertical, in LDS) of x, y, x^2, y^2, xy; SSIM map value
//                      summed; the three partial derivatives d map / d mu1, d sigma1^2, d sigma12 stored for the backward
//   k_ssim_backward    same tiling: dL/dx = conv(dmu1) + 2 x conv(dsigma1^2) + y conv(dsigma12), scaled by
//                      -lambda_dssim / (3 H W) and ADDED to the L1 gradient; also normalises the depth gradient by the
//                      valid count that is only known after the first pass
//   k_loss_finalize    the five scalars
// HBM-bound: ~12 image-sized reads / writes in total.  fp32 throughout; the separable window equals the reference's outer
// product window up to rounding (tests: loss terms 1e-6, gradients 1e-4 of the largest entry).
#include "common.h"
#include "device_utils.h"

namespace {

// Block sums are spread over 64 slots per quantity: thousands of float atomics to ONE address serialise in L2
// (measured 344 us for the 6.7 k workgroups of the point-wise pass alone); k_loss_finalize / the consumers add the slots up.
#define SUM_SLOTS 64
#define SUM_AT(q) ((q) * SUM_SLOTS)
#define SS_R 5                       // window radius (11 taps)
#define SS_T 32                      // output tile: 32 x 32 pixels per 256-thread workgroup, 4 vertically adjacent outputs per thread
#define SS_H (SS_T + 2 * SS_R)       // 42
#define SS_V 4                       // (the vertical pass shares 14 of its 4 x 11 window reads between the four outputs)

struct Win { float w[11]; };

__device__ __forceinline__ float block_sum(float v, float* s4) {
    v = wave_scan_add_f32(v);
    if ((threadIdx.x & 63) == 63) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    return t;
}

// sums: [0] sum |x - y|   [1] sum ssim map   [2] sum depth sq. err   [3] valid depth count   [4] sum sky bce   [5] sky pixel count
#define NUM_SUMS 6
__global__ void __launch_bounds__(EMD_BLOCK) k_loss_pointwise(EmdLossArgs a, float* __restrict__ sums) {
    const size_t HW = (size_t)a.height * a.width;
    const size_t p = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x;
    float l1 = 0.f, dsq = 0.f, dcnt = 0.f, sky = 0.f, skyn = 0.f;
    // every load of the pixel first, unconditionally (round 5: a pixel past the end reads pixel 0, an absent input reads the image instead and
    // is not used): with the loads inside the branches that use them the thread made five trips to memory one after the other
    const bool use_depth = a.depth && a.gt_depth && a.lambda_depth != 0.f, use_sky = a.weight && a.sky_mask && a.lambda_sky > 0.f;
    const size_t pc = p < HW ? p : 0;
    float im[3], gt[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { im[c] = a.image[c * HW + pc]; gt[c] = a.gt[c * HW + pc]; }
    const float depth_v = (use_depth ? a.depth : a.image)[pc], gt_depth_v = (use_depth ? a.gt_depth : a.image)[pc];
    const float mask_v = ((use_depth && a.mask) ? a.mask : a.image)[pc];
    const float weight_v = (use_sky ? a.weight : a.image)[pc];
    const uint8_t sky_v = (use_sky ? a.sky_mask : reinterpret_cast<const uint8_t*>(a.image))[pc];
    if (p < HW) {
        const float inv_n = 1.f / (float)(3 * HW);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d = im[c] - gt[c];
            l1 += fabsf(d);
            if (a.dL_dimage) a.dL_dimage[c * HW + p] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
        }
        if (use_depth) {
            const float m = a.mask ? mask_v : 1.f;
            const float pd = depth_v * m, gd = gt_depth_v * m;
            float g = 0.f;
            if (gd > 0.01f && gd < a.max_depth) {
                const float pn = pd / a.max_depth, gn = gd / a.max_depth;
                const float pcl = fminf(fmaxf(pn, 0.f), 1.f), gc = fminf(fmaxf(gn, 0.f), 1.f);
                const float e = pcl - gc;
                dsq = e * e;
                dcnt = 1.f;
                if (pn >= 0.f && pn <= 1.f) g = 2.f * e * (m / a.max_depth);     // un-normalised: / count in k_ssim_backward
            }
            if (a.dL_ddepth) a.dL_ddepth[p] = g;
        }
        if (use_sky) {
            const float w0 = weight_v;
            const float w = fminf(fmaxf(w0, 1e-6f), 1.f - 1e-6f);
            const bool is_sky = sky_v != 0;
            sky = is_sky ? -logf(1.f - w) : -logf(w);
            skyn = is_sky ? 1.f : 0.f;
            // (the reference applies the term only when the mask holds at least one sky pixel, train.py:360: the gradient is
            // zeroed by the second pass when the count turns out to be 0)
            if (a.dL_dweight) {
                const bool pass = w0 >= 1e-6f && w0 <= 1.f - 1e-6f;
                a.dL_dweight[p] = pass ? a.lambda_sky * (is_sky ? 1.f / (1.f - w) : -1.f / w) / (float)HW : 0.f;
            }
        }
    }
    {   // the five block sums through ONE barrier (five block_sum calls were ten)
        __shared__ float s5[5][4];
        const float w0 = wave_scan_add_f32(l1), w1 = wave_scan_add_f32(dsq), w2 = wave_scan_add_f32(dcnt), w3 = wave_scan_add_f32(sky), w4 = wave_scan_add_f32(skyn);
        if ((threadIdx.x & 63) == 63) { const int w = threadIdx.x >> 6; s5[0][w] = w0; s5[1][w] = w1; s5[2][w] = w2; s5[3][w] = w3; s5[4][w] = w4; }
        __syncthreads();
        if (threadIdx.x == 0) {
            l1 = (s5[0][0] + s5[0][1]) + (s5[0][2] + s5[0][3]); dsq = (s5[1][0] + s5[1][1]) + (s5[1][2] + s5[1][3]);
            dcnt = (s5[2][0] + s5[2][1]) + (s5[2][2] + s5[2][3]); sky = (s5[3][0] + s5[3][1]) + (s5[3][2] + s5[3][3]);
            skyn = (s5[4][0] + s5[4][1]) + (s5[4][2] + s5[4][3]);
        }
    }
    if (threadIdx.x == 0) {
        const int slot = blockIdx.x & (SUM_SLOTS - 1);
        atomicAdd(sums + SUM_AT(0) + slot, l1);
        if (dcnt != 0.f) { atomicAdd(sums + SUM_AT(2) + slot, dsq); atomicAdd(sums + SUM_AT(3) + slot, dcnt); }
        if (sky != 0.f) atomicAdd(sums + SUM_AT(4) + slot, sky);
        if (skyn != 0.f) atomicAdd(sums + SUM_AT(5) + slot, skyn);
    }
}

// stage the 42 x 42 halos of NIMG planes (zero outside the image).  All of a thread's 7 x NIMG pixels are requested before the first is written to
// LDS (round 5): as a plain loop -- load, wait, LDS store, next -- a workgroup made 14 - 21 trips to memory one after the other and the two SSIM
// kernels, which move ~100 MB each, took 80 us.
template <int NIMG>
__device__ __forceinline__ void load_halos(const float* __restrict__ i0, const float* __restrict__ i1, const float* __restrict__ i2, int H, int W, int x0, int y0,
                                           float (*s0)[SS_H + 1], float (*s1)[SS_H + 1], float (*s2)[SS_H + 1]) {
    constexpr int IT = (SS_H * SS_H + EMD_BLOCK - 1) / EMD_BLOCK;
    float v[3][IT];
#pragma unroll
    for (int k = 0; k < IT; k++) {
        const int i = threadIdx.x + k * EMD_BLOCK, ly = i / SS_H, lx = i % SS_H;
        const int x = x0 + lx - SS_R, y = y0 + ly - SS_R;
        const bool in = i < SS_H * SS_H && x >= 0 && x < W && y >= 0 && y < H;
        const size_t off = in ? (size_t)y * W + x : 0;
        v[0][k] = i0[off];
        if (NIMG > 1) v[1][k] = i1[off];
        if (NIMG > 2) v[2][k] = i2[off];
    }
#pragma unroll
    for (int k = 0; k < IT; k++) {
        const int i = threadIdx.x + k * EMD_BLOCK, ly = i / SS_H, lx = i % SS_H;
        const int x = x0 + lx - SS_R, y = y0 + ly - SS_R;
        const bool in = x >= 0 && x < W && y >= 0 && y < H;
        if (i < SS_H * SS_H) {
            s0[ly][lx] = in ? v[0][k] : 0.f;
            if (NIMG > 1) s1[ly][lx] = in ? v[1][k] : 0.f;
            if (NIMG > 2) s2[ly][lx] = in ? v[2][k] : 0.f;
        }
    }
}

__device__ __forceinline__ float slot_sum(const float* __restrict__ sums, int q) {
    float t = 0.f;
    for (int i = 0; i < SUM_SLOTS; i++) t += sums[SUM_AT(q) + i];
    return t;
}

__global__ void __launch_bounds__(EMD_BLOCK) k_ssim_forward(EmdLossArgs a, Win win, float* __restrict__ sums,
                                                            float* __restrict__ dmaps /*[3 maps][3 ch][HW]*/) {
    __shared__ float sx[SS_H][SS_H + 1], sy[SS_H][SS_H + 1];
    __shared__ float hz[5][SS_H][SS_T + 1];        // horizontally filtered x, y, xx, yy, xy
    __shared__ float s4[4];
    const int H = a.height, W = a.width;
    const size_t HW = (size_t)H * W;
    const int tiles_x = (W + SS_T - 1) / SS_T;
    const int c = blockIdx.y;
    const int x0 = (int)(blockIdx.x % (unsigned)tiles_x) * SS_T, y0 = (int)(blockIdx.x / (unsigned)tiles_x) * SS_T;
    load_halos<2>(a.image + c * HW, a.gt + c * HW, nullptr, H, W, x0, y0, sx, sy, nullptr);
    __syncthreads();
    for (int i = threadIdx.x; i < SS_H * SS_T; i += EMD_BLOCK) {
        const int ly = i / SS_T, lx = i % SS_T;
        float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float u = sx[ly][lx + k], v = sy[ly][lx + k], w = win.w[k];
            m1 += w * u; m2 += w * v; xx += w * (u * u); yy += w * (v * v); xy += w * (u * v);
        }
        hz[0][ly][lx] = m1; hz[1][ly][lx] = m2; hz[2][ly][lx] = xx; hz[3][ly][lx] = yy; hz[4][ly][lx] = xy;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, lyb = (threadIdx.x >> 5) * SS_V;
    const int x = x0 + lx;
    float val = 0.f;
    float acc[SS_V][5];
#pragma unroll
    for (int j = 0; j < SS_V; j++)
#pragma unroll
        for (int m = 0; m < 5; m++) acc[j][m] = 0.f;
#pragma unroll
    for (int r = 0; r < 10 + SS_V; r++) {          // input row lyb + r feeds output j with tap r - j
        float v[5];
#pragma unroll
        for (int m = 0; m < 5; m++) v[m] = hz[m][lyb + r][lx];
#pragma unroll
        for (int j = 0; j < SS_V; j++) {
            if (r - j >= 0 && r - j < 11) {
                const float w = win.w[r - j];
#pragma unroll
                for (int m = 0; m < 5; m++) acc[j][m] += w * v[m];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < SS_V; j++) {
        const int y = y0 + lyb + j;
        if (x < W && y < H) {
            const float mu1 = acc[j][0], mu2 = acc[j][1], xx = acc[j][2], yy = acc[j][3], xy = acc[j][4];
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = xx - mu1s, s2 = yy - mu2s, s12 = xy - mu12;
            const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1s + mu2s + C1, D = s1 + s2 + C2;
            val += (A * B) / (Cc * D);
            if (dmaps) {
                // d map / d mu1 (through mu1^2, mu1 mu2 everywhere they occur), d map / d sigma1^2, d map / d sigma12
                const float dm_dmu1 = (mu2 * 2.f * B) / (Cc * D) - (mu2 * 2.f * A) / (Cc * D) - (mu1 * 2.f * A * B) / (Cc * Cc * D) +
                                      (mu1 * 2.f * A * B) / (Cc * D * D);
                const float dm_ds1 = -(A * B) / (Cc * D * D);
                const float dm_ds12 = (2.f * A) / (Cc * D);
                const size_t q = (size_t)y * W + x;
                dmaps[(0 * 3 + c) * HW + q] = dm_dmu1;
                dmaps[(1 * 3 + c) * HW + q] = dm_ds1;
                dmaps[(2 * 3 + c) * HW + q] = dm_ds12;
            }
        }
    }
    val = block_sum(val, s4);
    if (threadIdx.x == 0) atomicAdd(sums + SUM_AT(1) + ((blockIdx.x + 21 * blockIdx.y) & (SUM_SLOTS - 1)), val);
}

__global__ void __launch_bounds__(EMD_BLOCK) k_ssim_backward(EmdLossArgs a, Win win, const float* __restrict__ sums,
                                                             const float* __restrict__ dmaps) {
    __shared__ float s0[SS_H][SS_H + 1], s1[SS_H][SS_H + 1], s2[SS_H][SS_H + 1];
    __shared__ float hz[3][SS_H][SS_T + 1];
    __shared__ float s_cnt, s_sky;
    const int H = a.height, W = a.width;
    const size_t HW = (size_t)H * W;
    const int tiles_x = (W + SS_T - 1) / SS_T;
    const int c = blockIdx.y;
    const int x0 = (int)(blockIdx.x % (unsigned)tiles_x) * SS_T, y0 = (int)(blockIdx.x / (unsigned)tiles_x) * SS_T;
    if (threadIdx.x == 0) { s_cnt = slot_sum(sums, 3); s_sky = slot_sum(sums, 5); }
    load_halos<3>(dmaps + (0 * 3 + c) * HW, dmaps + (1 * 3 + c) * HW, dmaps + (2 * 3 + c) * HW, H, W, x0, y0, s0, s1, s2);
    __syncthreads();
    for (int i = threadIdx.x; i < SS_H * SS_T; i += EMD_BLOCK) {
        const int ly = i / SS_T, lx = i % SS_T;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) { const float w = win.w[k]; t0 += w * s0[ly][lx + k]; t1 += w * s1[ly][lx + k]; t2 += w * s2[ly][lx + k]; }
        hz[0][ly][lx] = t0; hz[1][ly][lx] = t1; hz[2][ly][lx] = t2;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, lyb = (threadIdx.x >> 5) * SS_V;
    const int x = x0 + lx;
    float acc[SS_V][3];
#pragma unroll
    for (int j = 0; j < SS_V; j++) { acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; }
#pragma unroll
    for (int r = 0; r < 10 + SS_V; r++) {
        const float v0 = hz[0][lyb + r][lx], v1 = hz[1][lyb + r][lx], v2 = hz[2][lyb + r][lx];
#pragma unroll
        for (int j = 0; j < SS_V; j++) {
            if (r - j >= 0 && r - j < 11) { const float w = win.w[r - j]; acc[j][0] += w * v0; acc[j][1] += w * v1; acc[j][2] += w * v2; }
        }
    }
    const float scale = -a.lambda_dssim / (float)(3 * HW);          // d (lambda (1 - mean map)) / d map
    const float cnt = s_cnt, skyn = s_sky;
    // the four outputs' inputs first (a pixel outside the image reads pixel 0 and is not stored), then the arithmetic and the stores
    const bool fix_depth = c == 0 && a.dL_ddepth && a.depth && a.gt_depth && a.lambda_depth != 0.f;
    size_t q[SS_V];
    bool ok[SS_V];
    float xv[SS_V], yv[SS_V], gi[SS_V], gd[SS_V];
#pragma unroll
    for (int j = 0; j < SS_V; j++) {
        const int y = y0 + lyb + j;
        ok[j] = x < W && y < H;
        q[j] = ok[j] ? (size_t)y * W + x : 0;
        xv[j] = a.image[c * HW + q[j]]; yv[j] = a.gt[c * HW + q[j]];
        gi[j] = (a.dL_dimage ? a.dL_dimage : a.image)[c * HW + q[j]];
        gd[j] = (fix_depth ? a.dL_ddepth : a.image)[q[j]];
    }
#pragma unroll
    for (int j = 0; j < SS_V; j++) {
        if (ok[j]) {
            if (a.dL_dimage) a.dL_dimage[c * HW + q[j]] = gi[j] + scale * (acc[j][0] + 2.f * xv[j] * acc[j][1] + yv[j] * acc[j][2]);
            if (fix_depth) a.dL_ddepth[q[j]] = cnt > 0.f ? gd[j] * (a.lambda_depth / cnt) : 0.f;
            if (c == 0 && a.dL_dweight && a.weight && a.sky_mask && a.lambda_sky > 0.f && !(skyn > 0.f)) a.dL_dweight[q[j]] = 0.f;
        }
    }
}

// no SSIM term: the depth gradient still needs its 1 / count, the sky gradient its "any sky pixel at all" gate
__global__ void __launch_bounds__(EMD_BLOCK) k_depth_normalize(EmdLossArgs a, const float* __restrict__ sums) {
    const size_t HW = (size_t)a.height * a.width;
    const size_t p = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (p >= HW) return;
    if (a.dL_ddepth && a.depth && a.lambda_depth != 0.f) {
        const float cnt = slot_sum(sums, 3);
        a.dL_ddepth[p] = cnt > 0.f ? a.dL_ddepth[p] * (a.lambda_depth / cnt) : 0.f;
    }
    if (a.dL_dweight && a.weight && a.sky_mask && a.lambda_sky > 0.f && !(slot_sum(sums, 5) > 0.f)) a.dL_dweight[p] = 0.f;
}

// losses: [0] total  [1] l1  [2] ssim  [3] depth  [4] sky
__global__ void k_loss_finalize(EmdLossArgs a, const float* __restrict__ sums, float* __restrict__ losses) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float HW = (float)((size_t)a.height * a.width);
    const float cnt = slot_sum(sums, 3);
    const float l1 = slot_sum(sums, 0) / (3.f * HW);
    const float ss = a.lambda_dssim != 0.f ? slot_sum(sums, 1) / (3.f * HW) : 0.f;
    const float dp = cnt > 0.f ? slot_sum(sums, 2) / cnt : 0.f;
    const float sk = slot_sum(sums, 5) > 0.f ? slot_sum(sums, 4) / HW : 0.f;      // train.py:360: skipped without sky pixels
    losses[1] = l1; losses[2] = ss; losses[3] = dp; losses[4] = sk;
    losses[0] = l1 + (a.lambda_dssim != 0.f ? a.lambda_dssim * (1.f - ss) : 0.f) + a.lambda_depth * dp + a.lambda_sky * sk;
}

}  // namespace

extern "C" size_t emd_image_loss_workspace(int height, int width) {
    return (size_t)9 * height * width * sizeof(float) + NUM_SUMS * SUM_SLOTS * sizeof(float) + 256;   // 3 derivative maps x 3 channels + the sums
}

extern "C" int emd_image_loss(const EmdLossArgs* a, void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (!a || !a->image || !a->gt || !a->losses) { emd_set_error("image_loss: null image / gt / losses"); return EMD_ERR_INVALID; }
    if (a->height <= 0 || a->width <= 0) { emd_set_error("image_loss: bad size %d x %d", a->height, a->width); return EMD_ERR_INVALID; }
    if ((a->depth != nullptr) != (a->gt_depth != nullptr)) { emd_set_error("image_loss: depth and gt_depth go together"); return EMD_ERR_INVALID; }
    if ((a->weight != nullptr) != (a->sky_mask != nullptr)) { emd_set_error("image_loss: weight and sky_mask go together"); return EMD_ERR_INVALID; }
    if (!workspace || workspace_bytes < emd_image_loss_workspace(a->height, a->width)) { emd_set_error("image_loss: workspace too small"); return EMD_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t HW = (size_t)a->height * a->width;
    float* sums = (float*)workspace;
    float* dmaps = sums + NUM_SUMS * SUM_SLOTS;
    { int zrc = emd_zero_async(sums, NUM_SUMS * SUM_SLOTS * sizeof(float), st); if (zrc) return zrc; }
    Win win;
    {   // loss_utils.py:56-58: exp(-(x - 5)^2 / (2 * 1.5^2)) normalised, built in double like Python, stored in float
        double g[11], s = 0.0;
        for (int k = 0; k < 11; k++) { g[k] = exp(-(double)((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5)); s += (double)(float)g[k]; }
        float sf = 0.f;
        for (int k = 0; k < 11; k++) sf += (float)g[k];
        for (int k = 0; k < 11; k++) win.w[k] = (float)g[k] / sf;
        (void)s;
    }
    const unsigned pb = (unsigned)((HW + EMD_BLOCK - 1) / EMD_BLOCK);
    hipLaunchKernelGGL(k_loss_pointwise, dim3(pb), dim3(EMD_BLOCK), 0, st, *a, sums);
    EMD_LAUNCH_CHECK();
    const unsigned tiles = (unsigned)(((a->width + SS_T - 1) / SS_T) * ((a->height + SS_T - 1) / SS_T));
    if (a->lambda_dssim != 0.f) {
        hipLaunchKernelGGL(k_ssim_forward, dim3(tiles, 3), dim3(EMD_BLOCK), 0, st, *a, win, sums, a->dL_dimage ? dmaps : nullptr);
        EMD_LAUNCH_CHECK();
        if (a->dL_dimage || a->dL_ddepth) {
            hipLaunchKernelGGL(k_ssim_backward, dim3(tiles, 3), dim3(EMD_BLOCK), 0, st, *a, win, sums, dmaps);
            EMD_LAUNCH_CHECK();
        }
    } else if ((a->dL_ddepth && a->depth && a->lambda_depth != 0.f) || (a->dL_dweight && a->weight && a->lambda_sky > 0.f)) {
        hipLaunchKernelGGL(k_depth_normalize, dim3(pb), dim3(EMD_BLOCK), 0, st, *a, sums);
        EMD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, st, *a, sums, a->losses);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
