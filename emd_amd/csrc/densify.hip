// densify.hip -- adaptive density control on the SoA parameter + Adam-state buffers, on the device.
//
//   densify  = densify_and_clone + densify_and_split        S3Gaussian/scene/gaussian_model.py:515-603,696-701
//   prune    = prune / prune_points                          S3Gaussian/scene/gaussian_model.py:441-479,683-695
//   (OmniRe's per-class version of the same logic: models/gaussians/vanilla.py:206-376)
//
// The reference performs each of them with boolean-mask indexing (a device-to-host sync per mask), `torch.cat` / `repeat` on
// every parameter and both Adam moments (~60 launches and as many temporaries), and draws the split samples from the global
// CUDA generator (`torch.normal`, gaussian_model.py:543) -- a different stream on every rank.  Here one event is
//   k_densify_decide   one pass over the N Gaussians: per point a code (keep / clone / split / drop) from the accumulated
//                      statistics, the activated scale and opacity; per 256-point block the number of survivors / clones / splits
//   k_densify_scan     exclusive prefix sums of the block counts (one workgroup per column) + the column totals
//                      [ONE host read: the totals = the new point count.  Round 6: the scan used to be torch.cumsum over [3, N] -- three rows
//                       of 3 M elements, i.e. three workgroups: 7.3 ms of the event's 8.2 at 3 M Gaussians, profiles/r06_density_event.txt]
//   k_densify_index    for every OUTPUT row its source row and kind (position = block offset + ballot prefix), in the reference's order:
//                      survivors (in order), clones (in order), split samples replica 0 (in order), replica 1 (in order)
//   k_densify_gather   one launch for ALL tensors (7 parameters + 14 Adam moments + statistics): output row <- source row;
//                      the moments and statistics of new rows are zero; split samples get xyz = R(q) (std * n) + xyz with
//                      n ~ N(0, 1) from Philox4x32-10 keyed by (seed, SOURCE Gaussian index, replica) -- identical on every
//                      rank without communication -- and scaling = log(exp(s) / (0.8 * 2)).
// HBM-bound: every surviving byte is read and written once.
#include <string.h>

#include "common.h"
#include "device_utils.h"

namespace {

enum { CODE_KEEP = 1, CODE_CLONE = 2, CODE_SPLIT = 4 };

// ---- 0/1 columns of a 256-thread block: counts per block (decide), exclusive position inside the block (index) --------------------------------
// Every thread of the block calls these (threads past the end with all flags false).
template <int NC>
__device__ __forceinline__ void block_flag_counts(const bool (&f)[NC], int32_t* __restrict__ counts, int nblocks) {
    __shared__ uint32_t s_cnt[NC][EMD_BLOCK / EMD_WAVE];
    const int wave = threadIdx.x / EMD_WAVE, lane = threadIdx.x % EMD_WAVE;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const unsigned long long b = __ballot(f[c]);
        if (lane == 0) s_cnt[c][wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    if (threadIdx.x < NC) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < EMD_BLOCK / EMD_WAVE; w++) t += s_cnt[threadIdx.x][w];
        counts[(size_t)threadIdx.x * nblocks + blockIdx.x] = (int32_t)t;
    }
}
// position of the thread's 1 among the 1s of its column = block offset + 1s of the waves before + 1s of the lanes before
template <int NC>
__device__ __forceinline__ void block_flag_positions(const bool (&f)[NC], const int32_t* __restrict__ offsets, int nblocks, int (&pos)[NC]) {
    __shared__ uint32_t s_cnt[NC][EMD_BLOCK / EMD_WAVE];
    const int wave = threadIdx.x / EMD_WAVE, lane = threadIdx.x % EMD_WAVE;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int in_wave[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const unsigned long long b = __ballot(f[c]);
        in_wave[c] = __popcll(b & lt);
        if (lane == 0) s_cnt[c][wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; c++) {
        int before = offsets[(size_t)c * nblocks + blockIdx.x];
        for (int w = 0; w < wave; w++) before += (int)s_cnt[c][w];
        pos[c] = before + in_wave[c];
    }
}

// exclusive prefix sums over the blocks, in place; one 1024-thread workgroup per column; totals[c] = the column's sum
__global__ void __launch_bounds__(1024) k_densify_scan(int nblocks, int32_t* __restrict__ counts, int32_t* __restrict__ totals) {
    __shared__ int32_t s_wave[16];
    __shared__ int32_t s_carry;
    int32_t* col = counts + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x % EMD_WAVE, wave = threadIdx.x / EMD_WAVE;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const int32_t v = i < nblocks ? col[i] : 0;
        int32_t x = v;                                  // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < EMD_WAVE; off <<= 1) { const int32_t y = __shfl_up(x, off); if (lane >= off) x += y; }
        if (lane == EMD_WAVE - 1) s_wave[wave] = x;
        __syncthreads();
        int32_t before = s_carry;
        for (int w = 0; w < wave; w++) before += s_wave[w];
        if (i < nblocks) col[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = s_carry;
}

__global__ void __launch_bounds__(EMD_BLOCK) k_densify_decide(EmdDensifyArgs a, int32_t* __restrict__ code, int32_t* __restrict__ counts /*[3][nblocks]*/) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    const bool live = i < a.num_points;
    int c = 0;
    if (live) {
    const float s0 = expf(a.scaling[3 * i]), s1 = expf(a.scaling[3 * i + 1]), s2 = expf(a.scaling[3 * i + 2]);
    const float smax = fmaxf(s0, fmaxf(s1, s2));
    c = CODE_KEEP;
    if (a.mode == EMD_DENSIFY_MODE_DENSIFY) {
        // grads = xyz_gradient_accum / denom, NaN -> 0 (gaussian_model.py:697-698); clone: small Gaussians, split: large ones
        float g = a.grad_accum[i] / a.denom[i];
        if (g != g) g = 0.f;
        const bool hot = g >= a.grad_threshold;                       // (torch.norm of the [N,1] column = |g|; g >= 0)
        const bool small_ = smax <= a.percent_dense * a.scene_extent;
        if (hot && small_) c |= CODE_CLONE;
        if (hot && !small_) c = CODE_SPLIT;                           // the original is removed after its two samples are appended
    } else {
        // prune (gaussian_model.py:683-692): opacity below the threshold, or -- once max_screen_size is set -- too large on screen or in
        // the world
        const float op = 1.f / (1.f + expf(-a.opacity[i]));
        bool drop = op < a.min_opacity;
        if (a.max_screen_size > 0.f) drop = drop || a.max_radii2D[i] > a.max_screen_size || smax > 0.1f * a.scene_extent;
        if (a.extra_drop && a.extra_drop[i]) drop = true;
        if (drop) c = 0;
    }
    code[i] = c;
    }
    const bool f[3] = {(c & CODE_KEEP) != 0, (c & CODE_CLONE) != 0, (c & CODE_SPLIT) != 0};
    block_flag_counts<3>(f, counts, (int)gridDim.x);
}

// offsets: exclusive prefix sums of the block counts (k_densify_scan); totals: the three column sums
__global__ void __launch_bounds__(EMD_BLOCK) k_densify_index(int n, const int32_t* __restrict__ code, const int32_t* __restrict__ offsets,
                                                             const int32_t* __restrict__ totals, int32_t* __restrict__ src, int32_t* __restrict__ kind) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    const int c = i < n ? code[i] : 0;
    const bool f[3] = {(c & CODE_KEEP) != 0, (c & CODE_CLONE) != 0, (c & CODE_SPLIT) != 0};
    int pos[3];
    block_flag_positions<3>(f, offsets, (int)gridDim.x, pos);
    const int n_keep = totals[0], n_clone = totals[1], n_split = totals[2];
    if (c & CODE_KEEP) { const int j = pos[0]; src[j] = i; kind[j] = 0; }
    if (c & CODE_CLONE) { const int j = n_keep + pos[1]; src[j] = i; kind[j] = 1; }
    if (c & CODE_SPLIT) {
        const int j0 = n_keep + n_clone + pos[2], j1 = j0 + n_split;
        src[j0] = i; kind[j0] = 2;
        src[j1] = i; kind[j1] = 3;
    }
}

// Philox4x32-10 (Salmon et al., SC'11): counter = (gaussian index, replica, 0, 0), key = (seed lo, seed hi)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// three standard normals of (seed, index, replica): Box-Muller on uniforms in (0, 1]
__device__ __forceinline__ void normal3(uint64_t seed, uint32_t index, uint32_t rep, float n[3]) {
    uint32_t r[4];
    philox4x32_10(index, rep, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u0 = ((float)(r[0] >> 8) + 1.f) * (1.f / 16777216.f), u1 = (float)(r[1] >> 8) * (1.f / 16777216.f);
    const float u2 = ((float)(r[2] >> 8) + 1.f) * (1.f / 16777216.f), u3 = (float)(r[3] >> 8) * (1.f / 16777216.f);
    const float ra = sqrtf(-2.f * logf(u0)), rb = sqrtf(-2.f * logf(u2));
    n[0] = ra * cosf(6.28318530717958648f * u1);
    n[1] = ra * sinf(6.28318530717958648f * u1);
    n[2] = rb * cosf(6.28318530717958648f * u3);
}

// One launch for all tensors: blockIdx.y = tensor, grid-stride over its output elements.
__global__ void __launch_bounds__(EMD_BLOCK) k_densify_gather(EmdDensifyGather g) {
    const EmdDensifyTensor t = g.tensors[blockIdx.y];
    const size_t total = (size_t)g.num_out * t.width;
    for (size_t idx = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * EMD_BLOCK) {
        const int j = (int)(idx / t.width), c = (int)(idx % t.width);
        const int i = g.src[j], kd = g.kind[j];
        float v;
        if (g.mode == EMD_DENSIFY_MODE_REFINE) {
            // OmniRe (vanilla.py:256-263,328-349; basics.py:219-242): kind 0 original, 1 duplicate, 2 + r sample of replica r, + 16 = the source was split
            const int base = kd & 15;
            v = t.src[(size_t)i * t.width + c];
            if (t.role == EMD_DENSIFY_ROLE_STATE || t.role == EMD_DENSIFY_ROLE_ZERO) { if (base != 0) v = 0.f; }
            else if (t.role == EMD_DENSIFY_ROLE_SCALING) { if (kd & 16) v = logf(expf(v) / 1.6f); }      // original, samples and duplicate of a split source alike
            else if (t.role == EMD_DENSIFY_ROLE_XYZ && base >= 2) {
                float n[3];
                if (g.samples) { const float* s_ = g.samples + ((size_t)(base - 2) * g.num_split + g.split_rank[j]) * 3; n[0] = s_[0]; n[1] = s_[1]; n[2] = s_[2]; }
                else normal3(g.seed, (uint32_t)i, (uint32_t)(base - 2), n);
                const float* sc = g.scaling + 3 * (size_t)i;          // the scale BEFORE the reduction (vanilla.py:337-340)
                const float* qq = g.rotation + 4 * (size_t)i;
                const float e0 = expf(sc[0]) * n[0], e1 = expf(sc[1]) * n[1], e2 = expf(sc[2]) * n[2];
                const float qn = sqrtf(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
                const float r = qq[0] / qn, x = qq[1] / qn, y = qq[2] / qn, z = qq[3] / qn;
                float R0, R1, R2;
                if (c == 0) { R0 = 1.f - 2.f * (y * y + z * z); R1 = 2.f * (x * y - r * z); R2 = 2.f * (x * z + r * y); }
                else if (c == 1) { R0 = 2.f * (x * y + r * z); R1 = 1.f - 2.f * (x * x + z * z); R2 = 2.f * (y * z - r * x); }
                else { R0 = 2.f * (x * z - r * y); R1 = 2.f * (y * z + r * x); R2 = 1.f - 2.f * (x * x + y * y); }
                v = ((R0 * e0 + R1 * e1) + R2 * e2) + v;
            }
            t.dst[idx] = v;
            continue;
        }
        if (t.role == EMD_DENSIFY_ROLE_STATE) v = kd == 0 ? t.src[(size_t)i * t.width + c] : 0.f;             // Adam moments of new rows: zero
        else if (t.role == EMD_DENSIFY_ROLE_ZERO) v = (g.mode == EMD_DENSIFY_MODE_DENSIFY) ? 0.f : t.src[(size_t)i * t.width + c];   // statistics: reset by a densification
        else {
            v = t.src[(size_t)i * t.width + c];
            if (kd >= 2) {
                if (t.role == EMD_DENSIFY_ROLE_SCALING) v = logf(expf(v) / (0.8f * 2.f));
                else if (t.role == EMD_DENSIFY_ROLE_XYZ) {
                    // new_xyz = R(q) (std * n) + xyz, q normalised as build_rotation does (general_utils.py:245-266)
                    float n[3];
                    if (g.samples) { const float* s = g.samples + ((size_t)(kd - 2) * g.num_split + g.split_rank[j]) * 3; n[0] = s[0]; n[1] = s[1]; n[2] = s[2]; }
                    else normal3(g.seed, (uint32_t)i, (uint32_t)(kd - 2), n);
                    const float* sc = g.scaling + 3 * (size_t)i;
                    const float* qq = g.rotation + 4 * (size_t)i;
                    const float e0 = expf(sc[0]) * n[0], e1 = expf(sc[1]) * n[1], e2 = expf(sc[2]) * n[2];
                    const float qn = sqrtf(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
                    const float r = qq[0] / qn, x = qq[1] / qn, y = qq[2] / qn, z = qq[3] / qn;
                    float R0, R1, R2;
                    if (c == 0) { R0 = 1.f - 2.f * (y * y + z * z); R1 = 2.f * (x * y - r * z); R2 = 2.f * (x * z + r * y); }
                    else if (c == 1) { R0 = 2.f * (x * y + r * z); R1 = 1.f - 2.f * (x * x + z * z); R2 = 2.f * (y * z - r * x); }
                    else { R0 = 2.f * (x * z - r * y); R1 = 2.f * (y * z + r * x); R2 = 1.f - 2.f * (x * x + y * y); }
                    v = ((R0 * e0 + R1 * e1) + R2 * e2) + v;
                }
            }
        }
        t.dst[idx] = v;
    }
}

// rank of every split output row among the split rows of its replica (needed only when the samples are supplied by the caller)
__global__ void __launch_bounds__(EMD_BLOCK) k_densify_split_rank(int num_out, int n_keep, int n_clone, int n_split, int32_t* __restrict__ rank) {
    const int j = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (j >= num_out) return;
    const int base = n_keep + n_clone;
    rank[j] = j < base ? 0 : (j - base) % (n_split > 0 ? n_split : 1);
}

// ---- OmniRe refinement (EmdRefineArgs in include/emd_raster.h restates the semantics) ------------------------------------------------------------
enum { RCODE_KEEP = 1, RCODE_DUP = 2, RCODE_SAMPLES = 4, RCODE_SPLIT = 8 };

__global__ void __launch_bounds__(EMD_BLOCK) k_refine_decide(EmdRefineArgs a, int32_t* __restrict__ code, int32_t* __restrict__ counts /*[4][nblocks]*/) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    bool keep = false, keep_samples = false, keep_dup = false, split = false;
    if (i < a.num_points) {
    const float l0 = a.scaling[3 * i], l1 = a.scaling[3 * i + 1], l2 = a.scaling[3 * i + 2];
    float smax = fmaxf(expf(l0), fmaxf(expf(l1), expf(l2)));
    const float m2d = a.max_2Dsize ? a.max_2Dsize[i] : 0.f;
    bool dup = false;
    float smax_new = smax;
    if (a.do_densify) {
        const bool high = a.grad_norm[i] / a.vis_counts[i] > a.grad_threshold;
        split = (smax > a.size_threshold || (a.use_split_screen && m2d > a.split_screen)) && high;
        if (split) {
            // the original takes log(exp(s) / 1.6) in place; what the later tests see is exp() of THAT (vanilla.py:344-345,302-309)
            smax_new = fmaxf(expf(logf(expf(l0) / 1.6f)), fmaxf(expf(logf(expf(l1) / 1.6f)), expf(logf(expf(l2) / 1.6f))));
            smax = smax_new;
        }
        dup = smax <= a.size_threshold && high;
    }
    keep = true; keep_samples = split; keep_dup = dup;
    if (a.do_cull) {
        const float op = 1.f / (1.f + expf(-a.opacity[i]));
        const bool faint = op < a.cull_alpha;
        const bool big = a.cull_big && smax > a.cull_size;                        // the same (possibly reduced) scale on the original, its samples, its duplicate
        const bool wide_old = a.cull_big && a.use_cull_screen && m2d > a.cull_screen;
        const bool wide_new = a.cull_big && a.use_cull_screen && 0.f > a.cull_screen;      // new rows carry max_2Dsize 0 (vanilla.py:266-269)
        keep = !(faint || big || wide_old);
        keep_samples = split && !(faint || big || wide_new);
        keep_dup = dup && !(faint || big || wide_new);
    }
    code[i] = (keep ? RCODE_KEEP : 0) | (keep_dup ? RCODE_DUP : 0) | (keep_samples ? RCODE_SAMPLES : 0) | (split ? RCODE_SPLIT : 0);
    }
    const bool f[4] = {keep, keep_dup, keep_samples, split};
    block_flag_counts<4>(f, counts, (int)gridDim.x);
}

__global__ void __launch_bounds__(EMD_BLOCK) k_refine_index(int n, int num_samples, const int32_t* __restrict__ code, const int32_t* __restrict__ offsets,
                                                            const int32_t* __restrict__ totals, int32_t* __restrict__ src, int32_t* __restrict__ kind,
                                                            int32_t* __restrict__ split_rank) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    const int c = i < n ? code[i] : 0;
    const bool f[4] = {(c & RCODE_KEEP) != 0, (c & RCODE_DUP) != 0, (c & RCODE_SAMPLES) != 0, (c & RCODE_SPLIT) != 0};
    int pos[4];
    block_flag_positions<4>(f, offsets, (int)gridDim.x, pos);
    const int n_keep = totals[0], n_samp = totals[2];
    const int sp = (c & RCODE_SPLIT) ? 16 : 0;
    if (c & RCODE_KEEP) { const int j = pos[0]; src[j] = i; kind[j] = sp; if (split_rank) split_rank[j] = 0; }
    if (c & RCODE_SAMPLES) {
        for (int rep = 0; rep < num_samples; rep++) {
            const int j = n_keep + rep * n_samp + pos[2];
            src[j] = i; kind[j] = (2 + rep) | sp;
            if (split_rank) split_rank[j] = pos[3];          // rank among ALL split sources (the row of a caller-supplied draw)
        }
    }
    if (c & RCODE_DUP) { const int j = n_keep + num_samples * n_samp + pos[1]; src[j] = i; kind[j] = 1 | sp; if (split_rank) split_rank[j] = 0; }
}

// VanillaGaussians.after_train (vanilla.py:163-191) for one view, in place
__global__ void __launch_bounds__(EMD_BLOCK) k_after_train_stats(int n, const int32_t* __restrict__ radii, const float* __restrict__ g, int stride,
                                                                 float* __restrict__ grad_norm, float* __restrict__ vis, float* __restrict__ m2d, float last_size) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = g[(size_t)stride * i], gy = g[(size_t)stride * i + 1];
    if (grad_norm) grad_norm[i] += sqrtf(gx * gx + gy * gy);
    if (vis) vis[i] += 1.f;
    if (m2d) m2d[i] = fmaxf(m2d[i], (float)r / last_size);
}

}  // namespace

extern "C" int emd_refine_decide(const EmdRefineArgs* a, int32_t* code, int32_t* columns, void* hip_stream) {
    if (!a || !code || !columns) { emd_set_error("refine_decide: null argument"); return EMD_ERR_INVALID; }
    if (a->num_points < 0) { emd_set_error("refine_decide: bad size"); return EMD_ERR_INVALID; }
    if (a->num_points == 0) return EMD_OK;
    if (!a->scaling || (a->do_densify && (!a->grad_norm || !a->vis_counts)) || (a->do_cull && !a->opacity) ||
        (((a->do_densify && a->use_split_screen) || (a->do_cull && a->cull_big && a->use_cull_screen)) && !a->max_2Dsize)) {
        emd_set_error("refine_decide: null input"); return EMD_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_refine_decide, dim3((a->num_points + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, *a, code, columns);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_refine_index(int32_t num_points, int32_t num_out, int32_t num_samples, const int32_t* code, const int32_t* block_offsets, const int32_t* totals,
                                int32_t* src, int32_t* kind, int32_t* split_rank, void* hip_stream) {
    if (num_points < 0 || num_out < 0 || num_samples < 1 || num_samples > 13 || (num_points > 0 && (!code || !block_offsets || !totals)) || (num_out > 0 && (!src || !kind))) {
        emd_set_error("refine_index: bad argument"); return EMD_ERR_INVALID;
    }
    if (num_points == 0) return EMD_OK;
    hipLaunchKernelGGL(k_refine_index, dim3((num_points + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, num_points, num_samples, code,
                       block_offsets, totals, src, kind, split_rank);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_after_train_stats(int32_t n, const int32_t* radii, const float* xys_grad, int32_t grad_stride, float* grad_norm, float* vis_counts,
                                     float* max_2Dsize, float last_size, void* hip_stream) {
    if (n < 0 || grad_stride < 2 || !(last_size > 0.f) || (n > 0 && (!radii || !xys_grad))) { emd_set_error("after_train_stats: bad argument"); return EMD_ERR_INVALID; }
    if (n == 0) return EMD_OK;
    hipLaunchKernelGGL(k_after_train_stats, dim3((n + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, n, radii, xys_grad, grad_stride, grad_norm,
                       vis_counts, max_2Dsize, last_size);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_densify_decide(const EmdDensifyArgs* a, int32_t* code, int32_t* columns, void* hip_stream) {
    if (!a || !code || !columns) { emd_set_error("densify_decide: null argument"); return EMD_ERR_INVALID; }
    if (a->num_points < 0 || (a->mode != EMD_DENSIFY_MODE_DENSIFY && a->mode != EMD_DENSIFY_MODE_PRUNE)) { emd_set_error("densify_decide: bad mode / size"); return EMD_ERR_INVALID; }
    if (a->num_points == 0) return EMD_OK;
    if (!a->scaling || (a->mode == EMD_DENSIFY_MODE_DENSIFY && (!a->grad_accum || !a->denom)) ||
        (a->mode == EMD_DENSIFY_MODE_PRUNE && (!a->opacity || (a->max_screen_size > 0.f && !a->max_radii2D)))) {
        emd_set_error("densify_decide: null input for mode %d", a->mode); return EMD_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_densify_decide, dim3((a->num_points + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, *a, code, columns);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_densify_scan(int32_t num_points, int32_t num_columns, int32_t* block_counts, int32_t* totals, void* hip_stream) {
    if (num_points < 0 || num_columns < 1 || num_columns > 8 || !totals || (num_points > 0 && !block_counts)) { emd_set_error("densify_scan: bad argument"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    if (num_points == 0) return emd_zero_async(totals, sizeof(int32_t) * (size_t)num_columns, st);
    hipLaunchKernelGGL(k_densify_scan, dim3((unsigned)num_columns), dim3(1024), 0, st, (num_points + EMD_BLOCK - 1) / EMD_BLOCK, block_counts, totals);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_densify_index(int32_t num_points, int32_t num_out, const int32_t* code, const int32_t* block_offsets, const int32_t* totals, int32_t* src,
                                 int32_t* kind, void* hip_stream) {
    if (num_points < 0 || num_out < 0 || (num_points > 0 && (!code || !block_offsets || !totals)) || (num_out > 0 && (!src || !kind))) {
        emd_set_error("densify_index: bad argument"); return EMD_ERR_INVALID;
    }
    if (num_points == 0) return EMD_OK;
    hipLaunchKernelGGL(k_densify_index, dim3((num_points + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, num_points, code,
                       block_offsets, totals, src, kind);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_densify_gather(const EmdDensifyGather* g, void* hip_stream) {
    if (!g) { emd_set_error("densify_gather: null args"); return EMD_ERR_INVALID; }
    if (g->num_tensors < 0 || g->num_tensors > EMD_DENSIFY_MAX_TENSORS || g->num_out < 0) { emd_set_error("densify_gather: bad sizes"); return EMD_ERR_INVALID; }
    if (g->num_out == 0 || g->num_tensors == 0) return EMD_OK;
    if (!g->src || !g->kind) { emd_set_error("densify_gather: null index"); return EMD_ERR_INVALID; }
    bool has_xyz = false;
    size_t widest = 1;
    for (int k = 0; k < g->num_tensors; k++) {
        const EmdDensifyTensor& t = g->tensors[k];
        if (!t.src || !t.dst || t.width <= 0) { emd_set_error("densify_gather: tensor %d: null pointer / bad width", k); return EMD_ERR_INVALID; }
        if (t.role == EMD_DENSIFY_ROLE_XYZ) { has_xyz = true; if (t.width != 3) { emd_set_error("densify_gather: xyz width must be 3"); return EMD_ERR_INVALID; } }
        if ((size_t)t.width > widest) widest = (size_t)t.width;
    }
    if (has_xyz && g->mode != EMD_DENSIFY_MODE_PRUNE && (!g->scaling || !g->rotation)) { emd_set_error("densify_gather: split needs scaling and rotation"); return EMD_ERR_INVALID; }
    if (g->samples && !g->split_rank) { emd_set_error("densify_gather: caller-supplied samples need split_rank"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t total = (size_t)g->num_out * widest;
    unsigned blocks = (unsigned)((total + EMD_BLOCK - 1) / EMD_BLOCK);
    if (blocks > 16384u) blocks = 16384u;
    hipLaunchKernelGGL(k_densify_gather, dim3(blocks, (unsigned)g->num_tensors), dim3(EMD_BLOCK), 0, st, *g);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_densify_split_rank(int32_t num_out, int32_t n_keep, int32_t n_clone, int32_t n_split, int32_t* rank, void* hip_stream) {
    if (num_out < 0 || (num_out > 0 && !rank)) { emd_set_error("densify_split_rank: bad argument"); return EMD_ERR_INVALID; }
    if (num_out == 0) return EMD_OK;
    hipLaunchKernelGGL(k_densify_split_rank, dim3((num_out + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, num_out, n_keep,
                       n_clone, n_split, rank);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
