// embed.hip -- the small embedding ops in front of the deformation MLPs.
//
//  k_temporal_embed     one row of the coarse-to-fine temporal embedding: the [rows, dim] table resized to k rows (bilinear,
//                       align_corners) and sampled at time t (bilinear, align_corners, reflection padding), forward and
//                       backward (table and t).  S3Gaussian/scene/deformation.py:208-221; the same construction per actor in
//                       OmniRe/models/nodes/rigid.py:150-164.  The reference spends ~10 launches and an [N, dim] repeat on
//                       it; the row is the same for every Gaussian, so it is computed once here and enters the first Linear
//                       as a bias on the host side.
//  k_deform_input       the input matrix of OmniRe's ConditionalDeformNetwork: per point the normalised position
//                       x = mean / height(actor) * 2, its frequency encoding, the encoding of the frame time and the actor's
//                       embedding row, written once as [N, ld] (OmniRe/models/nodes/deformable.py:35-47,
//                       models/modules.py:318-366,430-437).  The reference builds it from 43 tensors and two concats.
//  k_deform_input_bwd   gradient of that matrix back to the actor embedding table (positions and time are detached there).
// Launch-latency / HBM-write bound; nothing here is reused.
#include <string.h>

#include "common.h"

namespace {

// grid_sample coordinate: align_corners un-normalise, reflect over [0, size-1], clip; *mult = d(result)/d(coord)
// (no contraction in the sampling arithmetic of this file -- reflect_coord, resized_row, te_rows, the column coordinate, te_eval: the same source is
//  inlined into several kernels whose results are compared bit for bit, and left to itself the compiler fused e.g. `scale * r - h0` in one copy only)
__device__ __forceinline__ float reflect_coord(float c, int size, float* mult) {
#pragma clang fp contract(off)
    float v = ((c + 1.f) * 0.5f) * (float)(size - 1);
    float m = 0.5f * (float)(size - 1);
    if (size <= 1) { *mult = 0.f; return 0.f; }
    const float span = (float)(size - 1);
    if (v < 0.f) { v = -v; m = -m; }
    const float extra = fmodf(v, span);
    const int flips = (int)floorf(v / span);
    if (flips & 1) { v = span - extra; m = -m; } else v = extra;
    if (!(v > 0.f)) { v = 0.f; m = 0.f; }                       // clip_coordinates_set_grad
    else if (!(v < span)) { v = span; m = 0.f; }
    *mult = m;
    return v;
}

// row r of the table resized to k rows (upsample_bilinear2d, align_corners): two source rows and their weights
__device__ __forceinline__ void resized_row(int r, int k, int rows, int* h0, int* h1, float* l0, float* l1) {
#pragma clang fp contract(off)
    const float scale = k > 1 ? (float)(rows - 1) / (float)(k - 1) : 0.f;
    const float src = scale * (float)r;
    *h0 = (int)src;
    *h1 = *h0 + (*h0 < rows - 1 ? 1 : 0);
    *l1 = src - (float)*h0;
    *l0 = 1.f - *l1;
}

template <bool BWD>
__global__ void __launch_bounds__(EMD_WAVE) k_temporal_embed(const float* __restrict__ weight, int rows, int dim, int k,
                                                              const float* __restrict__ t_ptr, float* __restrict__ out,
                                                              const float* __restrict__ g_out, float* __restrict__ g_weight,
                                                              float* __restrict__ g_t) {
    // one wave per table (S3Gaussian has one; OmniRe keeps one per actor)
    weight += (size_t)blockIdx.x * rows * dim;
    if (out) out += (size_t)blockIdx.x * dim;
    if (g_out) g_out += (size_t)blockIdx.x * dim;
    if (g_weight) g_weight += (size_t)blockIdx.x * rows * dim;
    const float t = t_ptr[0];
    float my, mx;
    const float iy = reflect_coord((t - 0.5f) * 2.f, k, &my);
    const float y0f = floorf(iy);
    const int y0 = (int)y0f, y1 = y0 + 1;
    const float wy1 = iy - y0f, wy0 = 1.f - wy1;
    int ha[2], hb[2];
    float la[2], lb[2];
    resized_row(min(y0, k - 1), k, rows, &ha[0], &hb[0], &la[0], &lb[0]);
    resized_row(min(y1, k - 1), k, rows, &ha[1], &hb[1], &la[1], &lb[1]);
    const bool in1 = y1 <= k - 1;                                // the row past the end contributes nothing
    float dt_acc = 0.f;
    for (int j = threadIdx.x; j < dim; j += EMD_WAVE) {
        // the reference samples column j at x = (j / (dim-1) - 0.5) * 2, which un-normalises to j up to rounding
        const float gx = dim > 1 ? ((float)j / (float)(dim - 1) - 0.5f) * 2.f : -1.f;
        const float ix = reflect_coord(gx, dim, &mx);
        const float x0f = floorf(ix);
        const int x0 = (int)x0f, x1 = x0 + 1;
        const float wx1 = ix - x0f, wx0 = 1.f - wx1;
        const bool inx1 = x1 <= dim - 1;
        float v[2][2];                                           // [row 0/1][col 0/1] of the resized table
#pragma unroll
        for (int r = 0; r < 2; r++) {
            v[r][0] = la[r] * weight[ha[r] * dim + x0] + lb[r] * weight[hb[r] * dim + x0];
            v[r][1] = inx1 ? la[r] * weight[ha[r] * dim + x1] + lb[r] * weight[hb[r] * dim + x1] : 0.f;
        }
        if (!in1) v[1][0] = v[1][1] = 0.f;
        if (!BWD) {
            out[j] = v[0][0] * (wx0 * wy0) + v[0][1] * (wx1 * wy0) + v[1][0] * (wx0 * wy1) + v[1][1] * (wx1 * wy1);
        } else {
            const float g = g_out[j];
            if (g_weight) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    if (r == 1 && !in1) continue;
                    const float wy = r ? wy1 : wy0;
                    atomicAdd(g_weight + ha[r] * dim + x0, g * wx0 * wy * la[r]);
                    atomicAdd(g_weight + hb[r] * dim + x0, g * wx0 * wy * lb[r]);
                    if (inx1) {
                        atomicAdd(g_weight + ha[r] * dim + x1, g * wx1 * wy * la[r]);
                        atomicAdd(g_weight + hb[r] * dim + x1, g * wx1 * wy * lb[r]);
                    }
                }
            }
            // d out / d iy = (row1 - row0) interpolated along x
            dt_acc += g * ((v[1][0] - v[0][0]) * wx0 + (v[1][1] - v[0][1]) * wx1);
        }
    }
    if (BWD && g_t) {
        for (int off = 32; off; off >>= 1) dt_acc += __shfl_xor(dt_acc, off);
        if (threadIdx.x == 0) atomicAdd(g_t, dt_acc * my * 2.f);   // y = (t - 0.5) * 2
    }
}

// column c of one row of the encoder input: [x(3), {sin(x f), cos(x f)}_f (3 each), t, {sin(t f), cos(t f)}_f, embed(E)]
__global__ void __launch_bounds__(EMD_BLOCK) k_deform_input(EmdDeformInArgs a) {
    const int dx = 3 * (1 + 2 * a.num_freqs_x), dt = 1 + 2 * a.num_freqs_t, width = dx + dt + a.embed_dim;
    const size_t total = (size_t)a.num_points * width;
    for (size_t idx = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * EMD_BLOCK) {
        const size_t n = idx / width;
        const int c = (int)(idx - n * width);
        const size_t id = a.point_ids ? (size_t)a.point_ids[n] : n;   // no ids: per-point rows (the network called directly)
        float v;
        if (c < dx) {
            const int comp = c % 3, blk = c / 3;                 // blk 0: identity; 2f+1: sin, 2f+2: cos of frequency 2^f
            float x = a.means[3 * n + comp];
            if (a.inst_size) x = x / a.inst_size[3 * id + 2] * 2.f;
            if (blk == 0) v = x;
            else {
                const float arg = x * (float)(1 << ((blk - 1) >> 1));
                v = ((blk - 1) & 1) ? cosf(arg) : sinf(arg);
            }
        } else if (c < dx + dt) {
            const int blk = c - dx;
            const float t = a.t[0];
            if (blk == 0) v = t;
            else {
                const float arg = t * (float)(1 << ((blk - 1) >> 1));
                v = ((blk - 1) & 1) ? cosf(arg) : sinf(arg);
            }
        } else {
            v = a.inst_embed[id * a.embed_dim + (c - dx - dt)];
        }
        a.out[n * a.ld + c] = v;
    }
}

// d embed[id] += d input[n, col0 : col0+E]; points of one actor are contiguous in the reference (create_from_pcd concatenates per
// instance), so a thread walks a chunk of points and flushes one atomic per run of equal ids
#define DEF_CHUNK 128
__global__ void __launch_bounds__(EMD_WAVE) k_deform_input_bwd(int n, int E, int ld, int col0, const int32_t* __restrict__ ids,
                                                                const float* __restrict__ g_in, float* __restrict__ g_embed) {
    const int e = threadIdx.x;
    if (e >= E) return;
    const int lo = blockIdx.x * DEF_CHUNK, hi = min(n, lo + DEF_CHUNK);
    int cur = -1;
    float acc = 0.f;
    for (int i = lo; i < hi; i++) {
        const int id = ids[i];
        if (id != cur) {
            if (cur >= 0) atomicAdd(g_embed + (size_t)cur * E + e, acc);
            cur = id; acc = 0.f;
        }
        acc += g_in[(size_t)i * ld + col0 + e];
    }
    if (cur >= 0) atomicAdd(g_embed + (size_t)cur * E + e, acc);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Learned per-actor track offsets (OmniRe/models/nodes/rigid.py:150-246), all actors, both levels, one launch each way.
//   h_c = [TE_kc(t) of the actor's table, mean embedding of the actor's Gaussians],  h_f = [TE_kf(t), same mean]
//   track_trans = W_tc h_c + b_tc + W_tf h_f + b_tf;   theta_c = w_rc . h_c + b_rc,  theta_f = w_rf . h_f + b_rf
//   track_rot = (cos theta_c, 0, 0, sin theta_c) (x) (cos theta_f, 0, 0, sin theta_f)                      (theta, not theta / 2)
// The reference spends ~40 launches per actor and loops over the actors in Python (rigid.py:520-562).  One wave per actor:
// lane j < dim holds column j of both temporal rows, lanes dim .. dim+E-1 the mean embedding; the eight head outputs are
// eight DPP wave sums.
// ---------------------------------------------------------------------------------------------------------------------------
struct TeSample { int ha[2], hb[2]; float la[2], lb[2], wy0, wy1; bool in1; };

__device__ __forceinline__ TeSample te_rows(float t, int k, int rows) {
#pragma clang fp contract(off)
    TeSample s;
    float my;
    const float iy = reflect_coord((t - 0.5f) * 2.f, k, &my);
    const float y0f = floorf(iy);
    const int y0 = (int)y0f, y1 = y0 + 1;
    s.wy1 = iy - y0f; s.wy0 = 1.f - s.wy1;
    resized_row(min(y0, k - 1), k, rows, &s.ha[0], &s.hb[0], &s.la[0], &s.lb[0]);
    resized_row(min(y1, k - 1), k, rows, &s.ha[1], &s.hb[1], &s.la[1], &s.lb[1]);
    s.in1 = y1 <= k - 1;
    return s;
}

// the eight taps of a sample, unconditionally (x1c: the column neighbour, clamped onto x0 where the sample has none)
__device__ __forceinline__ void te_taps(const float* __restrict__ weight, int dim, const TeSample& s, int x0, int x1c, float (&wv)[2][4]) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
        wv[r][0] = weight[s.ha[r] * dim + x0]; wv[r][1] = weight[s.hb[r] * dim + x0];
        wv[r][2] = weight[s.ha[r] * dim + x1c]; wv[r][3] = weight[s.hb[r] * dim + x1c];
    }
}

// the bilinear blend of a sample's eight taps.  ONE function without contraction for every caller: the one-launch actor chain (te_column2) and the
// three-launch path (te_column) must produce the same bits (tests/test_motion_sh_gpu.py), and with contraction left to the compiler the two
// inlined copies were fused differently.
__device__ __forceinline__ float te_eval(const TeSample& s, const float (&wv)[2][4], float wx0, float wx1, bool inx1) {
#pragma clang fp contract(off)
    float out = 0.f;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const float wy = r ? s.wy1 : s.wy0;
        const float v0 = s.la[r] * wv[r][0] + s.lb[r] * wv[r][1];
        const float v1 = inx1 ? s.la[r] * wv[r][2] + s.lb[r] * wv[r][3] : 0.f;
        const float term = v0 * (wx0 * wy) + v1 * (wx1 * wy);
        out = (r == 1 && !s.in1) ? out : out + term;
    }
    return out;
}

// column j of the sampled row (same arithmetic as k_temporal_embed); with G != nullptr scatters g into the table gradient instead
__device__ __forceinline__ float te_column(const float* __restrict__ weight, int dim, const TeSample& s, int j, float g, float* __restrict__ G) {
#pragma clang fp contract(off)
    float mx;
    const float gx = dim > 1 ? ((float)j / (float)(dim - 1) - 0.5f) * 2.f : -1.f;
    const float ix = reflect_coord(gx, dim, &mx);
    const float x0f = floorf(ix);
    const int x0 = (int)x0f, x1 = x0 + 1;
    const float wx1 = ix - x0f, wx0 = 1.f - wx1;
    const bool inx1 = x1 <= dim - 1;
    float out = 0.f;
    if (!G) {
        // all eight taps first, unconditionally (a tap the sample does not use -- the row past the table's end, the column past its width -- is read
        // from a valid neighbour and selected away): with the loads inside the row loop's branches each row waited for its own round trip
        float wv[2][4];
        te_taps(weight, dim, s, x0, inx1 ? x1 : x0, wv);
        return te_eval(s, wv, wx0, wx1, inx1);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (r == 1 && !s.in1) continue;
        const float wy = r ? s.wy1 : s.wy0;
        if (!G) {
        } else {
            atomicAdd(G + s.ha[r] * dim + x0, g * wx0 * wy * s.la[r]);
            atomicAdd(G + s.hb[r] * dim + x0, g * wx0 * wy * s.lb[r]);
            if (inx1) {
                atomicAdd(G + s.ha[r] * dim + x1, g * wx1 * wy * s.la[r]);
                atomicAdd(G + s.hb[r] * dim + x1, g * wx1 * wy * s.lb[r]);
            }
        }
    }
    return out;
}

// two samples of the same table column (the coarse and the fine level of an actor's table): all sixteen taps are requested before either is evaluated
__device__ __forceinline__ void te_column2(const float* __restrict__ weight, int dim, const TeSample& sa, const TeSample& sb, int j, float* oa, float* ob) {
#pragma clang fp contract(off)
    float mx;
    const float gx = dim > 1 ? ((float)j / (float)(dim - 1) - 0.5f) * 2.f : -1.f;
    const float ix = reflect_coord(gx, dim, &mx);
    const float x0f = floorf(ix);
    const int x0 = (int)x0f, x1 = x0 + 1;
    const float wx1 = ix - x0f, wx0 = 1.f - wx1;
    const bool inx1 = x1 <= dim - 1;
    float wa[2][4], wb[2][4];
    te_taps(weight, dim, sa, x0, inx1 ? x1 : x0, wa);
    te_taps(weight, dim, sb, x0, inx1 ? x1 : x0, wb);
    *oa = te_eval(sa, wa, wx0, wx1, inx1);
    *ob = te_eval(sb, wb, wx0, wx1, inx1);
}

__device__ __forceinline__ float wave_sum_all(float v) {
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// per-actor sums of the point embeddings.  Points of one actor are contiguous in the reference (create_from_pcd concatenates per
// instance): with the segment starts known (EmdTrackArgs.segment_start) one workgroup sums one actor's points and WRITES the
// result -- no atomics, no zero fill.  Without them: a wave owns 64 consecutive points, one DPP sum + E atomics per uniform wave
// (10 k atomics onto 128 addresses at the bench size: 53 us against 5 us for the segmented form).
#define SEG_THREADS 1024
__global__ void __launch_bounds__(SEG_THREADS) k_track_embed_sum_seg(int E, const float* __restrict__ emb, const int32_t* __restrict__ seg,
                                                                     float* __restrict__ sums) {
    // 1024 threads per actor: the 5000 points of a bench actor are 5 independent loads per thread (20 dependent trips with 256 threads:
    // the kernel is pure latency, 19 us -> 7 us); E == 4 rows are read as one float4
    __shared__ float s_part[SEG_THREADS / 64][8];
    const int a = blockIdx.x, lo = seg[a], hi = seg[a + 1];
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.f;
    if (E == 4 && ((uintptr_t)emb & 15) == 0) {
        const float4* e4 = (const float4*)emb;
        for (int i = lo + threadIdx.x; i < hi; i += SEG_THREADS) {
            const float4 v = e4[i];
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += SEG_THREADS)
#pragma unroll
            for (int e = 0; e < 8; e++) if (e < E) acc[e] += emb[(size_t)i * E + e];
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float w = wave_sum_all(acc[e]);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6][e] = w;
    }
    __syncthreads();
    if ((int)threadIdx.x < E) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < SEG_THREADS / 64; w++) t += s_part[w][threadIdx.x];
        sums[(size_t)a * E + threadIdx.x] = t;
    }
}

__global__ void __launch_bounds__(EMD_BLOCK) k_track_embed_sum(int n, int E, const float* __restrict__ emb, const int32_t* __restrict__ ids,
                                                               float* __restrict__ sums) {
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    const int id = i < n ? ids[i] : -1;
    const unsigned long long has = __ballot(id >= 0);
    if (!has) return;
    const int a0 = __builtin_amdgcn_readlane(id, __ffsll((long long)has) - 1);
    const bool uniform = __ballot(id >= 0 && id != a0) == 0ull;
    for (int e = 0; e < E; e++) {
        const float v = id >= 0 ? emb[(size_t)i * E + e] : 0.f;
        if (uniform) {
            const float sum = wave_sum_all(v);
            if ((threadIdx.x & 63) == 0) atomicAdd(sums + (size_t)a0 * E + e, sum);
        } else if (id >= 0) atomicAdd(sums + (size_t)id * E + e, v);
    }
}

__global__ void __launch_bounds__(EMD_BLOCK) k_track_embed_bwd(int n, int E, const int32_t* __restrict__ ids, const float* __restrict__ count,
                                                               const float* __restrict__ d_mean, float* __restrict__ d_emb) {
    const size_t idx = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (idx >= (size_t)n * E) return;
    const int i = (int)(idx / E), e = (int)(idx % E), id = ids[i];
    d_emb[idx] = id >= 0 ? d_mean[(size_t)id * E + e] / count[id] : 0.f;
}

template <bool BWD>
__global__ void __launch_bounds__(EMD_WAVE) k_track_heads(EmdTrackArgs a, EmdTrackGrads g) {
    const int act = blockIdx.x, lane = threadIdx.x, dim = a.dim, E = a.embed_dim, width = dim + E;
    const float* w = a.weight + (size_t)act * a.rows * dim;
    const float t = a.t_dev ? a.t_dev[0] : a.t;
    const int k_fine = a.k_fine_dev ? min(max(a.k_fine_dev[0], 1), a.rows) : a.k_fine;      // (a device-side level is clamped into the table)
    const TeSample sc = te_rows(t, a.k_coarse, a.rows), sf = te_rows(t, k_fine, a.rows);
    float hc = 0.f, hf = 0.f;
    if (lane < dim) { hc = te_column(w, dim, sc, lane, 0.f, nullptr); hf = te_column(w, dim, sf, lane, 0.f, nullptr); }
    else if (lane < width) hc = hf = a.emb_sum[(size_t)act * E + (lane - dim)] / a.count[act];
    // head rows: 0-2 trans_c, 3-5 trans_f, 6 rot_c, 7 rot_f
    float out[8], wrow[8];
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int hd = o < 3 ? 0 : o < 6 ? 1 : o - 4, r = o < 3 ? o : o < 6 ? o - 3 : 0;
        wrow[o] = lane < width ? a.head_w[hd][r * width + lane] : 0.f;
        out[o] = wave_sum_all(wrow[o] * ((o < 3 || o == 6) ? hc : hf)) + a.head_b[hd][r];
    }
    const float cc = cosf(out[6]), scn = sinf(out[6]), cf = cosf(out[7]), sfn = sinf(out[7]);
    const float ow = cc * cf - scn * sfn, oz = cc * sfn + scn * cf;          // quaternion_raw_multiply of two rotations about z
    if (!BWD) {
        if (lane == 0) {
            a.trans[3 * act] = out[0] + out[3]; a.trans[3 * act + 1] = out[1] + out[4]; a.trans[3 * act + 2] = out[2] + out[5];
            a.rot[4 * act] = ow; a.rot[4 * act + 1] = 0.f; a.rot[4 * act + 2] = 0.f; a.rot[4 * act + 3] = oz;
        }
        return;
    }
    // NaN gradients of an actor that is skipped in the pose table arrive as zeros from the pose kernel
    const float gt0 = g.g_trans[3 * act], gt1 = g.g_trans[3 * act + 1], gt2 = g.g_trans[3 * act + 2];
    const float gw = g.g_rot[4 * act], gz = g.g_rot[4 * act + 3];
    const float dth = gw * (-oz) + gz * ow;                                  // d / d theta_c = d / d theta_f (the product depends on theta_c + theta_f)
    const float go[8] = {gt0, gt1, gt2, gt0, gt1, gt2, dth, dth};
    float dhc = 0.f, dhf = 0.f;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int hd = o < 3 ? 0 : o < 6 ? 1 : o - 4, r = o < 3 ? o : o < 6 ? o - 3 : 0;
        const bool coarse = o < 3 || o == 6;
        if (lane < width) atomicAdd(g.d_head_w[hd] + r * width + lane, go[o] * (coarse ? hc : hf));
        if (lane == 0) atomicAdd(g.d_head_b[hd] + r, go[o]);
        if (coarse) dhc += go[o] * wrow[o]; else dhf += go[o] * wrow[o];
    }
    if (lane < dim) {
        float* G = g.d_weight + (size_t)act * a.rows * dim;
        te_column(w, dim, sc, lane, dhc, G);
        te_column(w, dim, sf, lane, dhf, G);
    } else if (lane < width) g.d_mean[(size_t)act * E + (lane - dim)] = dhc + dhf;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The whole per-actor chain in ONE launch each way (round 3): embedding sums -> track heads -> pose row, and pose row gradient ->
// heads -> temporal table / embeddings / per-frame pose tables.  Every step of the chain is per actor, so one 1024-thread
// workgroup per actor runs it front to back: the forward replaces three launch-bound kernels (k_track_embed_sum_seg,
// k_track_heads, k_actor_pose_forward: ~21 us of a 1.5 ms step), the backward replaces three more plus the two zero fills of
// their dense outputs -- each workgroup clears what it owns (its actor's temporal table gradient and its column of the
// [F, A, .] pose gradients) before accumulating.  Only the head parameters are shared between actors: their gradients are
// float atomics into a 296-float accumulator that the FORWARD launch of the same step cleared (EmdTrackedPoseArgs.head_acc; a
// last-workgroup reduction with release / acquire fences was measured first: 28 us for the backward launch against 9), so no
// output needs a fill launch.  The pose arithmetic is the one of k_actor_pose_*
// (preprocess.hip), evaluated without contraction so that both produce the same bits.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tp_quat_mul(const float a[4], const float b[4], float o[4]) {
#pragma clang fp contract(off)
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
__device__ __forceinline__ float tp_quat_norm(const float q[4]) {
#pragma clang fp contract(off)
    return sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
}
__device__ __forceinline__ void tp_dnormalize4(const float vu[4], float n, const float g[4], float out[4]) {
#pragma clang fp contract(off)
    float dot = ((vu[0] * g[0] + vu[1] * g[1]) + vu[2] * g[2]) + vu[3] * g[3];
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = (g[k] - vu[k] * dot) / n;
}
__device__ __forceinline__ bool tp_any_nan(const float* v, int n) {
    bool b = false;
    for (int k = 0; k < n; k++) b |= !(v[k] == v[k]);
    return b;
}

// per-actor sum of the embeddings of its points [lo, hi) by the whole workgroup -> s_sum[0..E) (E <= 8)
__device__ __forceinline__ void tp_segment_sum(int E, const float* __restrict__ emb, int lo, int hi, float (*s_part)[8], float* s_sum) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.f;
    if (E == 4 && ((uintptr_t)emb & 15) == 0) {
        // eight rows per thread in flight at once, unconditionally (a row past the segment's end reads its last row and counts as zero): the
        // plain loop waited for every row before asking for the next -- five dependent trips to memory for a bench actor's 5 000 points.
        // (The adds run in the same order, row i before row i + 1 024: the same sums.)
        const float4* e4 = (const float4*)emb;
        if (hi > lo) {
            for (int base = lo + (int)threadIdx.x; base < hi; base += 8 * SEG_THREADS) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = e4[min(base + u * SEG_THREADS, hi - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (base + u * SEG_THREADS < hi) { acc[0] += v[u].x; acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w; }
            }
        }
    } else {
        for (int i = lo + (int)threadIdx.x; i < hi; i += SEG_THREADS)
#pragma unroll
            for (int e = 0; e < 8; e++) if (e < E) acc[e] += emb[(size_t)i * E + e];
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float w = wave_sum_all(acc[e]);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6][e] = w;
    }
    __syncthreads();
    if ((int)threadIdx.x < 8) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < SEG_THREADS / 64; w++) t += s_part[w][threadIdx.x];
        s_sum[threadIdx.x] = t;
    }
    __syncthreads();
}

template <bool BWD>
__global__ void __launch_bounds__(SEG_THREADS) k_tracked_pose(EmdTrackedPoseArgs p, EmdTrackedPoseGrads g) {
    __shared__ float s_part[SEG_THREADS / 64][8];
    __shared__ float s_sum[8];
    __shared__ float s_dmean[8];
    const EmdTrackArgs& a = p.track;
    const int act = blockIdx.x, A = a.num_actors, dim = a.dim, E = a.embed_dim, width = dim + E;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = a.segment_start[act], hi = a.segment_start[act + 1];
    // a frame / level that arrives from the DEVICE (a step replayed from a hipGraph) is clamped into the tables it indexes: the host can
    // only validate the values it passes itself (a select table built for another clip length would otherwise read and write out of bounds)
    const int frame = p.frame_dev ? min(max(p.frame_dev[0], 0), p.num_frames - 1) : p.frame;
    // wave 0's loads and arithmetic that do not depend on the embedding sums (temporal rows, head weights, the frame's pose row) are
    // issued FIRST: they overlap the segment sum / the clears of the other waves instead of following them (the kernel is pure latency)
    float hc_t = 0.f, hf_t = 0.f, wrow[8], hbias[8], cnt = 1.f, qf[4] = {1.f, 0.f, 0.f, 0.f}, tf[3] = {0.f, 0.f, 0.f}, vflag = 1.f;
    float lane_in = 0.f, gp_in[12];
    uint32_t valid_in = 1u;
    TeSample sc, sf;
#pragma unroll
    for (int o = 0; o < 8; o++) wrow[o] = hbias[o] = 0.f;
    if (wave == 0) {
        const float* w = a.weight + (size_t)act * a.rows * dim;
        const float t = a.t_dev ? a.t_dev[0] : a.t;
        const int k_fine = a.k_fine_dev ? min(max(a.k_fine_dev[0], 1), a.rows) : a.k_fine;
        sc = te_rows(t, a.k_coarse, a.rows); sf = te_rows(t, k_fine, a.rows);
        // Round 5 (late): the wave's UNIFORM inputs -- eight head biases, the frame's quaternion and translation, the point count, the valid flag and
        // (backward) the twelve floats of the pose row's gradient -- are one value per LANE of a single vector load.  As uniform loads the
        // compiler moved each of them to a scalar register right behind its own load: one memory round trip per value, 20 - 32 of them one after
        // the other in a kernel that is nothing but latency.  They are handed out with v_readlane where they are used (after the segment sum).
        {
            const float* q_f = p.q_all + ((size_t)frame * A + act) * 4;
            const float* t_f = p.t_all + ((size_t)frame * A + act) * 3;
            const float* src = nullptr;
            if (lane < 3) src = a.head_b[0] + lane;
            else if (lane < 6) src = a.head_b[1] + (lane - 3);
            else if (lane == 6) src = a.head_b[2];
            else if (lane == 7) src = a.head_b[3];
            else if (lane < 12) src = q_f + (lane - 8);
            else if (lane < 15) src = t_f + (lane - 12);
            else if (lane == 15) src = a.count + act;
            else if (BWD && lane < 28) src = g.g_pose + (size_t)act * EMD_ACTOR_STRIDE + (lane - 16);
            if (src) lane_in = *src;
            if (p.valid_all && lane == 28) valid_in = p.valid_all[(size_t)frame * A + act];
        }
#pragma unroll
        for (int o = 0; o < 8; o++) {
            const int hd = o < 3 ? 0 : o < 6 ? 1 : o - 4, r = o < 3 ? o : o < 6 ? o - 3 : 0;
            wrow[o] = a.head_w[hd][r * width + (lane < width ? lane : 0)];          // (unconditional; lanes past the width are zeroed below)
        }
        if (lane < dim) te_column2(w, dim, sc, sf, lane, &hc_t, &hf_t);
#pragma unroll
        for (int o = 0; o < 8; o++) wrow[o] = lane < width ? wrow[o] : 0.f;
    }
    if (BWD) {
        // clear what this workgroup owns: its actor's temporal-table gradient and its column of the dense per-frame pose gradients
        float* G = g.d_weight + (size_t)act * a.rows * dim;
        for (int i = threadIdx.x; i < a.rows * dim; i += SEG_THREADS) G[i] = 0.f;
        for (int f = threadIdx.x; f < p.num_frames; f += SEG_THREADS) {
            if (f == frame) continue;                 // (written below)
            float* dq = g.d_q_all + ((size_t)f * A + act) * 4;
            float* dt = g.d_t_all + ((size_t)f * A + act) * 3;
            dq[0] = dq[1] = dq[2] = dq[3] = 0.f; dt[0] = dt[1] = dt[2] = 0.f;
        }
    }
    if (!BWD) {
        // the accumulator the backward launch of this step adds the shared head gradients into (no fill launch, no ticket)
        if (blockIdx.x == 0 && p.head_acc) for (int i = threadIdx.x; i < p.head_acc_floats; i += SEG_THREADS) p.head_acc[i] = 0.f;
        if (E > 0) tp_segment_sum(E, a.embeddings, lo, hi, s_part, s_sum);
        if ((int)threadIdx.x < E) a.emb_sum[(size_t)act * E + threadIdx.x] = s_sum[threadIdx.x];      // kept for the backward
    } else {
        if ((int)threadIdx.x < E) s_sum[threadIdx.x] = a.emb_sum[(size_t)act * E + threadIdx.x];
        __syncthreads();                                                                           // (also: the clears above are done)
    }
    if (wave == 0) {
        const float* w = a.weight + (size_t)act * a.rows * dim;
        {   // the uniform inputs, from the lanes that loaded them
            auto rl = [&](int k) -> float { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_in), k)); };
#pragma unroll
            for (int o = 0; o < 8; o++) hbias[o] = rl(o);
#pragma unroll
            for (int k = 0; k < 4; k++) qf[k] = rl(8 + k);
#pragma unroll
            for (int k = 0; k < 3; k++) tf[k] = rl(12 + k);
            cnt = rl(15);
            if (BWD) {
#pragma unroll
                for (int k = 0; k < 12; k++) gp_in[k] = rl(16 + k);
            }
            vflag = __builtin_amdgcn_readlane((int)valid_in, 28) ? 1.f : 0.f;
        }
        float hc = hc_t, hf = hf_t;
        if (lane >= dim && lane < width) hc = hf = s_sum[lane - dim] / cnt;
        float out[8];
#pragma unroll
        for (int o = 0; o < 8; o++) out[o] = wave_sum_all(wrow[o] * ((o < 3 || o == 6) ? hc : hf)) + hbias[o];
        const float cc = cosf(out[6]), scn = sinf(out[6]), cf = cosf(out[7]), sfn = sinf(out[7]);
        const float ow = cc * cf - scn * sfn, oz = cc * sfn + scn * cf;
        const float dtv[3] = {out[0] + out[3], out[1] + out[4], out[2] + out[5]};
        const float dqv[4] = {ow, 0.f, 0.f, oz};
        // ---- pose row of this actor (k_actor_pose_forward / _backward, every lane computes the same values) ----
        const float q[4] = {qf[0], qf[1], qf[2], qf[3]};
        const float* t_f = tf;
        const float n = fmaxf(tp_quat_norm(q), 1e-12f);
        const bool ok_t = !tp_any_nan(dtv, 3), ok_r = !tp_any_nan(dqv, 4);
        float pq[4] = {q[0], q[1], q[2], q[3]};
        if (ok_r) tp_quat_mul(q, dqv, pq);
        const float n2 = fmaxf(tp_quat_norm(pq), 1e-12f);
        if (!BWD) {
            if (lane == 0) {
                if (a.trans) { a.trans[3 * act] = dtv[0]; a.trans[3 * act + 1] = dtv[1]; a.trans[3 * act + 2] = dtv[2]; }
                if (a.rot) { a.rot[4 * act] = ow; a.rot[4 * act + 1] = 0.f; a.rot[4 * act + 2] = 0.f; a.rot[4 * act + 3] = oz; }
                float* P = p.pose + (size_t)act * EMD_ACTOR_STRIDE;
#pragma unroll
                for (int k = 0; k < 4; k++) P[k] = q[k] / n;
#pragma unroll
                for (int k = 0; k < 3; k++) P[4 + k] = ok_t ? t_f[k] + dtv[k] : t_f[k];
                P[7] = vflag;
#pragma unroll
                for (int k = 0; k < 4; k++) P[8 + k] = pq[k] / n2;
            }
        } else {
            const float* Gp = gp_in;                      // (the pose row's gradient: loaded with the other uniform inputs)
            const float qu[4] = {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
            const float gm[4] = {Gp[0], Gp[1], Gp[2], Gp[3]};
            float dqf[4];
            tp_dnormalize4(qu, n, gm, dqf);
            const float pu[4] = {pq[0] / n2, pq[1] / n2, pq[2] / n2, pq[3] / n2};
            const float gr[4] = {Gp[8], Gp[9], Gp[10], Gp[11]};
            float dp[4], dr[4] = {0.f, 0.f, 0.f, 0.f};
            tp_dnormalize4(pu, n2, gr, dp);
            if (ok_r) {
                const float rc[4] = {dqv[0], -dqv[1], -dqv[2], -dqv[3]}, qc[4] = {q[0], -q[1], -q[2], -q[3]};
                float t1[4];
                tp_quat_mul(dp, rc, t1);
                tp_quat_mul(qc, dp, dr);
#pragma unroll
                for (int k = 0; k < 4; k++) dqf[k] += t1[k];
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) dqf[k] += dp[k];
            }
            if (lane == 0) {
                float* dq = g.d_q_all + ((size_t)frame * A + act) * 4;
                float* dt = g.d_t_all + ((size_t)frame * A + act) * 3;
#pragma unroll
                for (int k = 0; k < 4; k++) dq[k] = dqf[k];
#pragma unroll
                for (int k = 0; k < 3; k++) dt[k] = Gp[4 + k];
            }
            // ---- heads ----
            const float gt0 = ok_t ? Gp[4] : 0.f, gt1 = ok_t ? Gp[5] : 0.f, gt2 = ok_t ? Gp[6] : 0.f;
            const float dth = dr[0] * (-oz) + dr[3] * ow;
            const float go[8] = {gt0, gt1, gt2, gt0, gt1, gt2, dth, dth};
            float dhc = 0.f, dhf = 0.f;
            // head parameters are shared by the actors: float atomics into `head_acc`, which the FORWARD launch of this step cleared
            // (layout = the eight head tensors back to back: w_tc, b_tc, w_tf, b_tf, w_rc, b_rc, w_rf, b_rf)
#pragma unroll
            for (int o = 0; o < 8; o++) {
                const int hd = o < 3 ? 0 : o < 6 ? 1 : o - 4, r = o < 3 ? o : o < 6 ? o - 3 : 0;
                const bool coarse = o < 3 || o == 6;
                if (go[o] != 0.f) {
                    if (lane < width) atomicAdd(g.d_head_w[hd] + r * width + lane, go[o] * (coarse ? hc : hf));
                    if (lane == 0) atomicAdd(g.d_head_b[hd] + r, go[o]);
                }
                if (coarse) dhc += go[o] * wrow[o]; else dhf += go[o] * wrow[o];
            }
            if (lane < dim) {
                float* G = g.d_weight + (size_t)act * a.rows * dim;
                te_column(w, dim, sc, lane, dhc, G);
                te_column(w, dim, sf, lane, dhf, G);
            } else if (lane < width) s_dmean[lane - dim] = (dhc + dhf) / cnt;
        }
    }
    if (!BWD) return;
    __syncthreads();
    if (g.d_embeddings && E > 0) {
        if (E == 4 && ((uintptr_t)g.d_embeddings & 15) == 0) {
            const float4 v = make_float4(s_dmean[0], s_dmean[1], s_dmean[2], s_dmean[3]);
            float4* d4 = (float4*)g.d_embeddings;
            for (int i = lo + (int)threadIdx.x; i < hi; i += SEG_THREADS) d4[i] = v;
        } else {
            for (int i = (lo * E) + (int)threadIdx.x; i < hi * E; i += SEG_THREADS) g.d_embeddings[i] = s_dmean[i % E];
        }
    }
}

int check_track(const EmdTrackArgs* a, const char* who) {
    if (!a) { emd_set_error("%s: null args", who); return EMD_ERR_INVALID; }
    if (a->num_actors < 0 || a->rows < 1 || a->dim < 1 || a->embed_dim < 0 || a->dim + a->embed_dim > EMD_WAVE || a->k_coarse < 1 || a->k_fine < 1 ||
        a->num_points < 0) { emd_set_error("%s: bad sizes (dim + embed_dim <= 64)", who); return EMD_ERR_INVALID; }
    if (a->num_actors == 0) return EMD_OK;
    if (!a->weight || !a->count || !a->emb_sum || (a->embed_dim > 0 && a->num_points > 0 && (!a->embeddings || !a->point_ids))) {
        emd_set_error("%s: null pointer", who); return EMD_ERR_INVALID;
    }
    for (int h = 0; h < 4; h++) if (!a->head_w[h] || !a->head_b[h]) { emd_set_error("%s: null head parameter %d", who, h); return EMD_ERR_INVALID; }
    return EMD_OK;
}

int check_te(const float* weight, int tables, int rows, int dim, int k, const float* t, const char* who) {
    if (!weight || !t) { emd_set_error("%s: null table / time", who); return EMD_ERR_INVALID; }
    if (tables < 1 || rows < 1 || dim < 1 || k < 1) { emd_set_error("%s: bad sizes tables=%d rows=%d dim=%d k=%d", who, tables, rows, dim, k); return EMD_ERR_INVALID; }
    return EMD_OK;
}

}  // namespace

extern "C" int emd_temporal_embed_forward(const float* weight, int tables, int rows, int dim, int k, const float* t, float* out,
                                          void* hip_stream) {
    int rc = check_te(weight, tables, rows, dim, k, t, "temporal_embed_forward");
    if (rc) return rc;
    if (!out) { emd_set_error("temporal_embed_forward: null output"); return EMD_ERR_INVALID; }
    hipLaunchKernelGGL(k_temporal_embed<false>, dim3(tables), dim3(EMD_WAVE), 0, (hipStream_t)hip_stream, weight, rows, dim, k, t, out,
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_temporal_embed_backward(const float* weight, int tables, int rows, int dim, int k, const float* t,
                                           const float* dL_dout, float* dL_dweight, float* dL_dt, void* hip_stream) {
    int rc = check_te(weight, tables, rows, dim, k, t, "temporal_embed_backward");
    if (rc) return rc;
    if (!dL_dout) { emd_set_error("temporal_embed_backward: null gradient"); return EMD_ERR_INVALID; }
    hipLaunchKernelGGL(k_temporal_embed<true>, dim3(tables), dim3(EMD_WAVE), 0, (hipStream_t)hip_stream, weight, rows, dim, k, t,
                       (float*)nullptr, dL_dout, dL_dweight, dL_dt);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_deform_input_width(int num_freqs_x, int num_freqs_t, int embed_dim) {
    return 3 * (1 + 2 * num_freqs_x) + (1 + 2 * num_freqs_t) + embed_dim;
}

extern "C" int emd_deform_input_forward(const EmdDeformInArgs* a, void* hip_stream) {
    if (!a) { emd_set_error("deform_input_forward: null args"); return EMD_ERR_INVALID; }
    const int width = emd_deform_input_width(a->num_freqs_x, a->num_freqs_t, a->embed_dim);
    if (a->num_points < 0 || a->num_freqs_x < 0 || a->num_freqs_x > 24 || a->num_freqs_t < 0 || a->num_freqs_t > 24 || a->embed_dim < 0 ||
        a->ld < width) { emd_set_error("deform_input_forward: bad sizes"); return EMD_ERR_INVALID; }
    if (a->num_points == 0) return EMD_OK;
    if (!a->means || !a->t || !a->out || (a->embed_dim > 0 && !a->inst_embed)) {
        emd_set_error("deform_input_forward: null pointer"); return EMD_ERR_INVALID;
    }
    const size_t total = (size_t)a->num_points * width;
    const unsigned blocks = (unsigned)((total + EMD_BLOCK - 1) / EMD_BLOCK < 65536 ? (total + EMD_BLOCK - 1) / EMD_BLOCK : 65536);
    hipLaunchKernelGGL(k_deform_input, dim3(blocks), dim3(EMD_BLOCK), 0, (hipStream_t)hip_stream, *a);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_deform_input_backward(int num_points, int embed_dim, int ld, int col0, const int32_t* point_ids, const float* dL_din,
                                         float* dL_dembed, void* hip_stream) {
    if (num_points < 0 || embed_dim < 0 || embed_dim > EMD_WAVE || col0 < 0 || ld < col0 + embed_dim) {
        emd_set_error("deform_input_backward: bad sizes (embed_dim <= 64)"); return EMD_ERR_INVALID;
    }
    if (num_points == 0 || embed_dim == 0) return EMD_OK;
    if (!point_ids || !dL_din || !dL_dembed) { emd_set_error("deform_input_backward: null pointer"); return EMD_ERR_INVALID; }
    hipLaunchKernelGGL(k_deform_input_bwd, dim3((unsigned)((num_points + DEF_CHUNK - 1) / DEF_CHUNK)), dim3(EMD_WAVE), 0,
                       (hipStream_t)hip_stream, num_points, embed_dim, ld, col0, point_ids, dL_din, dL_dembed);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_track_heads_forward(const EmdTrackArgs* a, void* hip_stream) {
    int rc = check_track(a, "track_heads_forward");
    if (rc || a->num_actors == 0) return rc;
    if (!a->trans || !a->rot) { emd_set_error("track_heads_forward: null output"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    const int E = a->embed_dim;
    // (emb_sum arrives zero-filled unless segment_start is given: the caller's allocation is its zero fill)
    if (E > 0 && a->num_points > 0) {
        if (a->segment_start && E <= 8)
            hipLaunchKernelGGL(k_track_embed_sum_seg, dim3(a->num_actors), dim3(SEG_THREADS), 0, st, E, a->embeddings, a->segment_start, a->emb_sum);
        else
            hipLaunchKernelGGL(k_track_embed_sum, dim3((a->num_points + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, a->num_points, E,
                               a->embeddings, a->point_ids, a->emb_sum);
        EMD_LAUNCH_CHECK();
    }
    EmdTrackGrads none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL(k_track_heads<false>, dim3(a->num_actors), dim3(EMD_WAVE), 0, st, *a, none);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_track_heads_backward(const EmdTrackArgs* a, const EmdTrackGrads* g, void* hip_stream) {
    int rc = check_track(a, "track_heads_backward");
    if (rc || a->num_actors == 0) return rc;
    if (!g || !g->g_trans || !g->g_rot || !g->d_weight || !g->d_mean) { emd_set_error("track_heads_backward: null gradient pointer"); return EMD_ERR_INVALID; }
    for (int h = 0; h < 4; h++) if (!g->d_head_w[h] || !g->d_head_b[h]) { emd_set_error("track_heads_backward: null head gradient %d", h); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    const int E = a->embed_dim;
    // d_weight, d_head_w[], d_head_b[] are accumulated into and arrive ZERO-FILLED from the caller (carved from one zeroed
    // allocation: one fill launch instead of nine memsets); emb_sum still holds the forward's sums
    hipLaunchKernelGGL(k_track_heads<true>, dim3(a->num_actors), dim3(EMD_WAVE), 0, st, *a, *g);
    EMD_LAUNCH_CHECK();
    if (g->d_embeddings && E > 0 && a->num_points > 0) {
        const size_t total = (size_t)a->num_points * E;
        hipLaunchKernelGGL(k_track_embed_bwd, dim3((unsigned)((total + EMD_BLOCK - 1) / EMD_BLOCK)), dim3(EMD_BLOCK), 0, st, a->num_points, E,
                           a->point_ids, a->count, g->d_mean, g->d_embeddings);
        EMD_LAUNCH_CHECK();
    }
    return EMD_OK;
}

static int check_tracked(const EmdTrackedPoseArgs* p, const char* who) {
    if (!p) { emd_set_error("%s: null args", who); return EMD_ERR_INVALID; }
    int rc = check_track(&p->track, who);
    if (rc) return rc;
    const EmdTrackArgs& a = p->track;
    if (a.num_actors == 0) return EMD_OK;
    if (!a.segment_start || a.embed_dim > 8) { emd_set_error("%s: needs segment_start (points sorted by actor) and embed_dim <= 8", who); return EMD_ERR_INVALID; }
    if (!p->q_all || !p->t_all || p->num_frames < 1 || (!p->frame_dev && (p->frame < 0 || p->frame >= p->num_frames))) {
        emd_set_error("%s: bad pose tables / frame", who); return EMD_ERR_INVALID;
    }
    return EMD_OK;
}

extern "C" int emd_tracked_pose_forward(const EmdTrackedPoseArgs* p, void* hip_stream) {
    int rc = check_tracked(p, "tracked_pose_forward");
    if (rc || p->track.num_actors == 0) return rc;
    if (!p->pose) { emd_set_error("tracked_pose_forward: null output"); return EMD_ERR_INVALID; }
    EmdTrackedPoseGrads none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL(k_tracked_pose<false>, dim3(p->track.num_actors), dim3(SEG_THREADS), 0, (hipStream_t)hip_stream, *p, none);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_tracked_pose_backward(const EmdTrackedPoseArgs* p, const EmdTrackedPoseGrads* g, void* hip_stream) {
    int rc = check_tracked(p, "tracked_pose_backward");
    if (rc || p->track.num_actors == 0) return rc;
    if (!g || !g->g_pose || !g->d_q_all || !g->d_t_all || !g->d_weight) {
        emd_set_error("tracked_pose_backward: null gradient pointer"); return EMD_ERR_INVALID;
    }
    for (int h = 0; h < 4; h++) if (!g->d_head_w[h] || !g->d_head_b[h]) { emd_set_error("tracked_pose_backward: null head gradient %d", h); return EMD_ERR_INVALID; }
    hipLaunchKernelGGL(k_tracked_pose<true>, dim3(p->track.num_actors), dim3(SEG_THREADS), 0, (hipStream_t)hip_stream, *p, *g);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
