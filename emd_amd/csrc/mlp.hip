// mlp.hip -- the width-64 MLPs of the EMD deformation network as fused fp32-MFMA kernels.
//
//   trunk   h = b + W[:, a-block] xa + W[:, b-block] xb                 S3Gaussian/scene/deformation.py:100-112,254-296 (feature_out,
//           (xa = HexPlane features [N,128] or nothing, xb = the per-Gaussian embedding [N,4]; the temporal row is the same for
//           every Gaussian and arrives folded into b)                   defor_depth = 1: one Linear)
//   branch  out = W_o act(W_2 act(W_1 in + b_1) + b_2) + b_o             :113-185,298-337 (pos / scales / rotations / opacity / shs
//           in = relu(h) for the deformation heads (one hidden layer),    heads: nn.Sequential(ReLU, Linear, ReLU, Linear); dino_head:
//           in = h for the feature head (two hidden layers)               Linear, ReLU, Linear, ReLU, Linear on the un-rectified h)
//
// The reference runs these as ~12 cuBLAS GEMMs per level plus one element-wise launch per bias / ReLU / slice, forward and backward;
// over 2 M Gaussians the [N, 64..192] intermediates cross HBM ~40 times per step.  Here a wave owns 32 Gaussians and keeps every
// intermediate in registers:
//   * `v_mfma_f32_32x32x2_f32` (exact fp32: a k-ordered fmaf chain) with the DATA ROWS on the N dimension: A = weights [out feature x k]
//     from LDS (one ds_read_b128 per four MFMAs, row stride 68 = 4 x odd floats: conflict-free), B = activations [k x row].  The
//     accumulator tile of a layer (lane = row, registers = features 8(v/4) + 4(lane/32) + v%4) IS the B operand of the next layer, register v
//     for k-step v -- the contraction order is free, so the weights are simply read in the same permuted k order: no lane movement,
//     no LDS round trip between layers.  The backward's data path (W^T g) reads the same LDS weights column-wise.
//   * weight gradients dW[o][i] = sum_rows g[o][row] act[i][row] contract over the rows, i.e. over the LANE index of both tiles: each
//     tile takes one wave-private trip through LDS (16 ds_write_b32, 4 ds_read_b128) that leaves it with lane = feature and its 16
//     registers = the data rows 16(lane/32) .. +15, which is an MFMA operand again; dW accumulates in registers over all tiles of the
//     wave and is added to HBM once per wave with float atomics.  Bias gradients are the row sums of the same fragments.
//   * the ReLU masks, bias adds and the sum over the heads' contributions to dL/dh ride in registers.
// HBM traffic per Gaussian and level: x (528 B) + h (256 B) + outputs forward; h, g_out, one g_h per branch, g_x backward.
// Bound: fp32 MFMA (157 TFLOP/s); no kernel here waits on HBM.
#include <string.h>

#include <atomic>

#include "common.h"
#include "device_utils.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MLP_W 64          // layer width
#define WS 68             // LDS row stride of a [.][64] weight matrix (floats): 4 x odd -> conflict-free ds_read_b128 across 16 rows
#define TS 36             // LDS row stride of a transposed 32 x 32 tile
#define MLP_THREADS 256
#define MLP_WAVES (MLP_THREADS / 64)
// Two waves per SIMD (<= 256 registers each) where the accumulators allow it: a single wave exposes every LDS / HBM wait to the MFMA
// pipe (measured: SQ_VALU_MFMA_BUSY_CYCLES = 45-59 % of the kernel time with one wave per SIMD).
#ifndef MLP_FWD_WAVES
#define MLP_FWD_WAVES 1
#endif
#ifndef MLP_BWD_WAVES
#define MLP_BWD_WAVES 1
#endif
#define MLP_OCC(n) __attribute__((amdgpu_waves_per_eu(n, n)))
// Kernels whose tiles fit 256 registers run two waves per SIMD (measured round 3: k_mlp_branch_fwd<1,1> 228 -> 219 us, k_mlp_trunk_bwd<0>
// 504 -> 457 us at 2 M rows; the others spill at half the register file and lose).
#define MLP_W_F11 2
#define MLP_W_T0 2
#define MLP_W_TB0 2

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int v = 0; v < 16; v++) z[v] = 0.f;
    return z;
}
__device__ __forceinline__ f32x16 relu16(f32x16 t) {
#pragma unroll
    for (int v = 0; v < 16; v++) t[v] = fmaxf(t[v], 0.f);
    return t;
}
// g where y > 0, else 0
__device__ __forceinline__ f32x16 mask16(f32x16 g, f32x16 y) {
#pragma unroll
    for (int v = 0; v < 16; v++) g[v] = y[v] > 0.f ? g[v] : 0.f;
    return g;
}

// ---- global <-> tile (lane = (r = row, hh), register v = feature c0 + 8 (v / 4) + 4 hh + v % 4) -------------------------------------
// wide: ld and c0 multiples of 4, the whole 32-feature tile inside the row.
// STRAIGHT: rows past the end load row 0 and are zeroed afterwards instead of branching around the load.  A branch makes the compiler
// fall back to s_waitcnt vmcnt(0) at the join, which waits for the prefetch of the NEXT tile in every iteration (SQ_WAIT_ANY 15-25 %
// of the wave time): the forward kernels gain 12-15 % from the straight form.  (The backward kernels lost as much with it when this was written --
// the product behind the load put the wait AT the load -- and kept the branch until round 5: they now use load_tile_raw / mask_tile below.)
// `width` (a multiple of 4): columns >= width do not exist and read as 0 (the last tile of a narrow xa)
// A/B knob (round 5): the [N, 64..128] activation / gradient streams of these kernels are each far larger than the 256 MiB Infinity Cache and
// are touched once or twice per step; nontemporal accesses keep them from evicting what IS reused (the HexPlane planes, the weights).
#ifndef MLP_NT
#define MLP_NT 0
#endif
#ifndef MLP_L1_F11_TWO
#define MLP_L1_F11_TWO 1      /* the narrow heads' forward WITH the folded L1 sum also runs two waves per SIMD (195 -> 182 us at 2 M rows; 152 without the sum) */
#endif
#ifndef MLP_NARROW_OUT
#define MLP_NARROW_OUT 1      /* output layers of <= 4 features run on the vector pipe instead of a padded 32-row MFMA tile (forward) */
#endif
typedef float mlp_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) {
    if (MLP_NT) { const mlp_v4f v = __builtin_nontemporal_load(reinterpret_cast<const mlp_v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *(const float4*)p;
}
__device__ __forceinline__ void st4(float* p, float4 v) {
    if (MLP_NT) __builtin_nontemporal_store((mlp_v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<mlp_v4f*>(p));
    else *(float4*)p = v;
}
template <bool STRAIGHT = false>
__device__ __forceinline__ f32x16 load_tile(const float* __restrict__ src, size_t ld, size_t row, bool ok, int c0, int hh, int width = 1 << 30) {
    f32x16 t;
    if (STRAIGHT) {
        const float* p = src + (ok ? row : 0) * ld;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int c = c0 + 8 * a + 4 * hh;
            const bool in = c < width;
            const float4 q = ld4(p + (in ? c : 0));
            const float keep = (ok && in) ? 1.f : 0.f;
            t[4 * a] = q.x * keep; t[4 * a + 1] = q.y * keep; t[4 * a + 2] = q.z * keep; t[4 * a + 3] = q.w * keep;
        }
    } else {
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int c = c0 + 8 * a + 4 * hh;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && c < width) q = ld4(src + row * ld + c);
            t[4 * a] = q.x; t[4 * a + 1] = q.y; t[4 * a + 2] = q.z; t[4 * a + 3] = q.w;
        }
    }
    return t;
}
// any width: features >= width read as 0
__device__ __forceinline__ f32x16 load_tile_narrow(const float* __restrict__ src, int width, size_t row, bool ok, int c0, int hh) {
    f32x16 t;
#pragma unroll
    for (int v = 0; v < 16; v++) {
        const int f = c0 + 8 * (v >> 2) + 4 * hh + (v & 3);
        t[v] = (ok && f < width) ? src[row * (size_t)width + f] : 0.f;
    }
    return t;
}
__device__ __forceinline__ void store_tile(float* __restrict__ dst, size_t ld, size_t row, bool ok, int c0, int hh, f32x16 t, int width = 1 << 30) {
    if (!ok) return;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int c = c0 + 8 * a + 4 * hh;
        if (c < width) st4(dst + row * ld + c, make_float4(t[4 * a], t[4 * a + 1], t[4 * a + 2], t[4 * a + 3]));
    }
}
__device__ __forceinline__ void store_tile_narrow(float* __restrict__ dst, int width, size_t row, bool ok, int c0, int hh, f32x16 t) {
    if (!ok) return;
#pragma unroll
    for (int v = 0; v < 16; v++) {
        const int f = c0 + 8 * (v >> 2) + 4 * hh + (v & 3);
        if (f < width) dst[row * (size_t)width + f] = t[v];
    }
}

// RAW tile loads for the backward kernels' loop-carried prefetches (round 5).  A guarded load (a branch around it, or a select / product right behind
// it) makes the compiler wait for the load where it is issued -- with `s_waitcnt vmcnt(0)`, i.e. for every other load in flight too: the head
// kernels stalled for a full HBM round trip at the top of every tile.  So: the caller clamps the row, columns >= width read column 0, nothing is
// masked here; the values travel raw across the loop iteration and are masked (mask_tile) where they are used.
__device__ __forceinline__ f32x16 load_tile_raw(const float* __restrict__ src, size_t ld, size_t row_c, int c0, int hh, int width = 1 << 30) {
    f32x16 t;
    const float* p = src + row_c * ld;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int c = c0 + 8 * a + 4 * hh;
        const float4 q = ld4(p + (c < width ? c : 0));
        t[4 * a] = q.x; t[4 * a + 1] = q.y; t[4 * a + 2] = q.z; t[4 * a + 3] = q.w;
    }
    return t;
}
__device__ __forceinline__ f32x16 mask_tile(f32x16 t, bool ok, int c0, int hh, int width) {
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const float keep = (ok && c0 + 8 * a + 4 * hh < width) ? 1.f : 0.f;
        t[4 * a] *= keep; t[4 * a + 1] *= keep; t[4 * a + 2] *= keep; t[4 * a + 3] *= keep;
    }
    return t;
}

// bias as the initial accumulator: register v of lane half hh holds feature c0 + 8 (v / 4) + 4 hh + v % 4
__device__ __forceinline__ f32x16 bias_tile(const float* __restrict__ b_lds, int c0, int hh) {
    f32x16 t;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const float4 q = *(const float4*)(b_lds + c0 + 8 * a + 4 * hh);
        t[4 * a] = q.x; t[4 * a + 1] = q.y; t[4 * a + 2] = q.z; t[4 * a + 3] = q.w;
    }
    return t;
}

// out[to] += W[32 to + r][k] in[k]: KT input tiles, NT output tiles; W in LDS, row stride `stride`, first input column `k0`.
// NA < 4: only the first NA 8-feature chunks of an input tile are non-zero.
// The backward's data path is the same function on the TRANSPOSED weights (a second LDS image, [in][out]): reading W column-wise
// instead needs one ds_read_b32 per MFMA, which a wave with a full register file cannot prefetch -- the matrix pipe then waits for an
// LDS round trip per instruction (SQ_VALU_MFMA_BUSY_CYCLES was 53 % of the kernel time).
template <int KT, int NT, int NA = 4>
__device__ __forceinline__ void layer_fwd(f32x16 (&acc)[NT], const f32x16 (&in)[KT], const float* __restrict__ W, int stride, int k0, int r, int hh) {
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int a = 0; a < NA; a++) {
            float4 w[NT];
#pragma unroll
            for (int to = 0; to < NT; to++) w[to] = *(const float4*)(W + (32 * to + r) * stride + k0 + 32 * kt + 8 * a + 4 * hh);
#pragma unroll
            for (int to = 0; to < NT; to++) {
                acc[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[to].x, in[kt][4 * a], acc[to], 0, 0, 0);
                acc[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[to].y, in[kt][4 * a + 1], acc[to], 0, 0, 0);
                acc[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[to].z, in[kt][4 * a + 2], acc[to], 0, 0, 0);
                acc[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[to].w, in[kt][4 * a + 3], acc[to], 0, 0, 0);
            }
        }
}

// ---- split-bf16 MFMA (round 4) ---------------------------------------------------------------------------------------------------------
// fp32 MFMA runs at 1/16 of the bf16 rate on gfx950.  A product of fp32 values is reproduced to fp32 accuracy from bf16 pieces:
//     x = x1 + x2 + x3 (three bf16 terms = 24 mantissa bits: exact), W x ~ W1 x1 + (W1 x2 + W2 x1) + (W1 x3 + W2 x2 + W3 x1)
// -- the six products above 2^-24 of the result; a bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32.
// profiles/microbench_mfma_split_bf16.hip: the same error against float64 as the fp32 MFMA path (4.1e-7 of O(1) outputs after three
// layers) at 1.8x its rate, conversions included.  The data flow does not change: a tile's 16 registers of a lane are, in pairs of eight,
// exactly the eight consecutive k-positions the bf16 shape wants from that lane (k-block 0 = registers 0..7 of both lane halves = features
// {0-3, 8-11 | 4-7, 12-15} of the tile, k-block 1 = registers 8..15), so the accumulator of a layer is still the B operand of the next
// one after its registers have been split (5.5 vector instructions per value), and the weights are staged in LDS as ready-made A fragments:
//     image[term 0..2][out tile][k-block][lane] = uint4 of eight bf16 = term(M[32 tile + lane % 32][feature(k-block, lane / 32, 0..7)])
// Per kernel (bit 0: the layers' products W x on split bf16, bit 1: the weight gradients' outer products): chosen from measurements at 2 M
// rows (profiles/r04_mlp_split_variants.txt) -- the split costs 176 vector instructions per 32 x 32 tile, which a kernel with few MFMAs per
// tile (the forward of the narrow heads) does not earn back.
#ifndef MLP_SP_F11
#define MLP_SP_F11 1          /* forward, one output tile (dx / do / feat heads); round 5: 1 -- with the narrow output layer on the vector pipe the hidden
                                 layer is what is left on the matrix pipe, and there the split wins (fine stage 13.61 -> 13.46 ms; 0 before: the split cost more
                                 than it saved while a third of the MFMAs were a padded output tile) */
#endif
#ifndef MLP_SP_F12
#define MLP_SP_F12 1          /* forward, two output tiles (the dshs head) */
#endif
#ifndef MLP_SP_B11
#define MLP_SP_B11 3          /* backward, one output tile */
#endif
#ifndef MLP_SP_B12
#define MLP_SP_B12 1          /* backward, two output tiles */
#endif
#ifndef MLP_SP_TF
#define MLP_SP_TF 1           /* trunk forward (xa block) */
#endif
#ifndef MLP_SP_TB
#define MLP_SP_TB 3           /* trunk backward (xa block) */
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Split { uint4 t[3][2]; };          // [term][k-block of the tile]
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 r = __builtin_convertvector((f32x2){a, b}, bf16x2);        // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);         // exact residuals
    p2 = pack_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
    p3 = pack_bf16(s0, s1);
}
__device__ __forceinline__ Split split_tile(const f32x16& x) {
    Split s;
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
        uint32_t p[3][4];
#pragma unroll
        for (int q = 0; q < 4; q++) split2(x[8 * kb + 2 * q], x[8 * kb + 2 * q + 1], p[0][q], p[1][q], p[2][q]);
#pragma unroll
        for (int m = 0; m < 3; m++) s.t[m][kb] = make_uint4(p[m][0], p[m][1], p[m][2], p[m][3]);
    }
    return s;
}
__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six products of one k-block, smallest first
__device__ __forceinline__ f32x16 mfma_split(f32x16 c, uint4 a1, uint4 a2, uint4 a3, uint4 b1, uint4 b2, uint4 b3) {
    c = mfma_bf16(a3, b1, c); c = mfma_bf16(a2, b2, c); c = mfma_bf16(a1, b3, c);
    c = mfma_bf16(a2, b1, c); c = mfma_bf16(a1, b2, c);
    return mfma_bf16(a1, b1, c);
}
// floats (4-byte units) of the image of a [32 NT][16 KB] matrix
constexpr int split_floats(int nt, int kb) { return 3 * nt * kb * 64 * 4; }
// layer_fwd on an image: acc[to] += M[32 to + .][k] in[k], k-blocks kb0 .. of the image (NB = k-blocks of an input tile that are non-zero)
template <int KT, int NT, int NB = 2>
__device__ __forceinline__ void layer_fwd_s(f32x16 (&acc)[NT], const f32x16 (&in)[KT], const uint4* __restrict__ img, int nt_img, int kb_img, int kb0, int lane) {
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        const Split x = split_tile(in[kt]);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int kb = kb0 + 2 * kt + b;
#pragma unroll
            for (int to = 0; to < NT; to++) {
                const uint4 w1 = img[((0 * nt_img + to) * kb_img + kb) * 64 + lane], w2 = img[((1 * nt_img + to) * kb_img + kb) * 64 + lane],
                            w3 = img[((2 * nt_img + to) * kb_img + kb) * 64 + lane];
                acc[to] = mfma_split(acc[to], w1, w2, w3, x.t[0][b], x.t[1][b], x.t[2][b]);
            }
        }
    }
}
// M[o][f] = src[o ld + col0 + f] (o < rows, f < cols), or transposed: M[o][f] = src[f ld + col0 + o] (f < rows, o < cols); zero elsewhere
__device__ __forceinline__ void stage_split(float* __restrict__ dst_f, int nt_img, int kb_img, const float* __restrict__ src, int ld, int col0, int rows, int cols,
                                            bool transposed) {
    uint4* dst = reinterpret_cast<uint4*>(dst_f);
    const int total = nt_img * kb_img * 64;
    for (int idx = threadIdx.x; idx < total; idx += MLP_THREADS) {
        const int ln = idx & 63, kb = (idx >> 6) % kb_img, to = (idx >> 6) / kb_img, hh = ln >> 5, o = 32 * to + (ln & 31);
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int v = 8 * (kb & 1) + t, f = 32 * (kb >> 1) + 8 * (v >> 2) + 4 * hh + (v & 3);
            const bool in = src && (transposed ? (f < rows && o < cols) : (o < rows && f < cols));
            w[t] = in ? (transposed ? src[(size_t)f * ld + col0 + o] : src[(size_t)o * ld + col0 + f]) : 0.f;
        }
        uint32_t p[3][4];
#pragma unroll
        for (int q = 0; q < 4; q++) split2(w[2 * q], w[2 * q + 1], p[0][q], p[1][q], p[2][q]);
#pragma unroll
        for (int m = 0; m < 3; m++) dst[((m * nt_img + to) * kb_img + kb) * 64 + ln] = make_uint4(p[m][0], p[m][1], p[m][2], p[m][3]);
    }
}
// dW += gf (x) af over the 32 rows of the fragments (registers 0..7 / 8..15 of both lane halves = the two k-blocks), split the same way
__device__ __forceinline__ f32x16 outer_acc_s(f32x16 acc, const Split& g, const Split& a) {
#pragma unroll
    for (int b = 0; b < 2; b++) acc = mfma_split(acc, g.t[0][b], g.t[1][b], g.t[2][b], a.t[0][b], a.t[1][b], a.t[2][b]);
    return acc;
}

// ---- transposed fragments for the weight gradients -------------------------------------------------------------------------------
// tile (lane = row, registers = features) -> fragment (lane r = feature, registers = the 16 rows 16 hh .. 16 hh + 15) through the wave's
// private LDS scratch.  LDS serves the DS instructions of one wave in order, so the reads see the writes; the fences only pin the
// compiler's order.
__device__ __forceinline__ f32x16 transpose_tile(f32x16 t, float* __restrict__ T, int r, int hh) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int v = 0; v < 16; v++) T[(8 * (v >> 2) + 4 * hh + (v & 3)) * TS + r] = t[v];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x16 f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float4 q = *(const float4*)(T + r * TS + 16 * hh + 4 * i);
        f[4 * i] = q.x; f[4 * i + 1] = q.y; f[4 * i + 2] = q.z; f[4 * i + 3] = q.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return f;
}
__device__ __forceinline__ float frag_sum(f32x16 f) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 16; v++) s += f[v];
    return s;
}
// dW[32 to + o][32 ti + i] += sum_rows gf[o][row] af[i][row]: both operands as fragments; the k-step s pairs row s (lanes 0-31)
// with row 16 + s (lanes 32-63) on both sides
__device__ __forceinline__ f32x16 outer_acc(f32x16 acc, f32x16 gf, f32x16 af) {
#pragma unroll
    for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[s], af[s], acc, 0, 0, 0);
    return acc;
}
// accumulator tile D[o][i] (lane r = i, register v = o = 8 (v / 4) + 4 hh + v % 4) -> dW[(o0 + o) * ld + i0 + i], bounded
__device__ __forceinline__ void flush_dw(float* __restrict__ dW, int ld, int o0, int i0, int rows, int cols, f32x16 acc, int r, int hh) {
    if (!dW || i0 + r >= cols) return;
#pragma unroll
    for (int v = 0; v < 16; v++) {
        const int o = o0 + 8 * (v >> 2) + 4 * hh + (v & 3);
        if (o < rows) unsafeAtomicAdd(dW + (size_t)o * ld + i0 + r, acc[v]);
    }
}
// Sum an accumulator tile / a per-lane partial over the workgroup's waves through LDS (`buf`: (MLP_WAVES - 1) x 17 x 64 floats, the
// transpose scratch after the tile loop); the total is valid in wave 0, which alone issues the atomics: a quarter of the traffic onto
// the few thousand addresses every workgroup of the grid adds to at the same moment.
__device__ __forceinline__ f32x16 wg_sum16(f32x16 acc, float* __restrict__ buf, int wave, int lane) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int v = 0; v < 16; v++) buf[((wave - 1) * 17 + v) * 64 + lane] = acc[v];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < MLP_WAVES - 1; w++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[v] += buf[(w * 17 + v) * 64 + lane];
    }
    return acc;
}
__device__ __forceinline__ float wg_sum1(float x, float* __restrict__ buf, int wave, int lane) {
    __syncthreads();
    if (wave > 0) buf[((wave - 1) * 17 + 16) * 64 + lane] = x;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < MLP_WAVES - 1; w++) x += buf[(w * 17 + 16) * 64 + lane];
    }
    return x;
}

// per-lane bias-gradient partials (lane r = feature of the tile, both halves hold half of the rows) -> db[o0 + r]
__device__ __forceinline__ void flush_db(float* __restrict__ db, int o0, int rows, float part, int r) {
    if (db && o0 + r < rows) unsafeAtomicAdd(db + o0 + r, part);
}

// stage a [rows, cols] block of a row-major matrix (row stride ld, first column col0) into LDS [rows_pad][stride], zero padded.
// 16-byte pieces, four independent loads in flight per thread: the one-dword-at-a-time loop this replaces spent 30-50 us per launch
// on dependent load -> write round trips, 15 % of a forward kernel.
__device__ __forceinline__ void stage_matrix(float* __restrict__ dst, int stride, int rows_pad, int cols_pad, const float* __restrict__ src, int ld, int col0,
                                             int rows, int cols) {
    const bool vec = src && !(cols & 3) && !(ld & 3) && !(col0 & 3) && !(cols_pad & 3) && !(stride & 3) && !((uintptr_t)src & 15);
    if (vec) {
        const int q = cols_pad >> 2, total = rows_pad * q;
#pragma unroll 4
        for (int idx = threadIdx.x; idx < total; idx += MLP_THREADS) {
            const int rr = idx / q, c = (idx - rr * q) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rr < rows && c < cols) v = *(const float4*)(src + (size_t)rr * ld + col0 + c);
            *(float4*)(dst + rr * stride + c) = v;
        }
        return;
    }
#pragma unroll 4
    for (int idx = threadIdx.x; idx < rows_pad * cols_pad; idx += MLP_THREADS) {
        const int rr = idx / cols_pad, c = idx - rr * cols_pad;
        dst[rr * stride + c] = (src && rr < rows && c < cols) ? src[(size_t)rr * ld + col0 + c] : 0.f;
    }
}
// the transposed image: dst[c][rr] = src[rr][col0 + c] for rr < rows, c < cols; zero elsewhere in [cols_pad][rows_pad]
__device__ __forceinline__ void stage_matrix_t(float* __restrict__ dst, int stride, int rows_pad, int cols_pad, const float* __restrict__ src, int ld, int col0,
                                               int rows, int cols) {
#pragma unroll 4
    for (int idx = threadIdx.x; idx < rows_pad * cols_pad; idx += MLP_THREADS) {
        const int rr = idx / cols_pad, c = idx - rr * cols_pad;          // consecutive threads read consecutive source columns
        dst[c * stride + rr] = (src && rr < rows && c < cols) ? src[(size_t)rr * ld + col0 + c] : 0.f;
    }
}
__device__ __forceinline__ void stage_vector(float* __restrict__ dst, int n_pad, const float* __restrict__ src, int n) {
    for (int idx = threadIdx.x; idx < n_pad; idx += MLP_THREADS) dst[idx] = (src && idx < n) ? src[idx] : 0.f;
}

// the xb block as ONE k-chunk of 8: register j of lane half hh holds xb[row][4 hh + j]
template <bool STRAIGHT>
__device__ __forceinline__ void load_xb(const float* __restrict__ xb, int kb, size_t row, bool ok, int hh, float (&v)[4]) {
    if (STRAIGHT && xb && kb > 0) {
        const float* p = xb + (ok ? row : 0) * (size_t)kb;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = 4 * hh + j;
            const float x = p[c < kb ? c : 0];                   // (unconditional load, see load_tile)
            v[j] = (ok && c < kb) ? x : 0.f;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = 4 * hh + j;
        v[j] = (ok && xb && c < kb) ? xb[row * (size_t)kb + c] : 0.f;
    }
}

// The heads that recompute h: the next tile's embedding values travel RAW across the loop iteration (clamped address, unconditional loads) and are
// masked where they are used -- masking at the load (a select, or a product) makes the compiler wait for the load right there, inside the tile loop.
__device__ __forceinline__ void load_xb_raw(const float* __restrict__ xb, int kb, size_t row, bool ok, int hh, float (&v)[4]) {
    const float* p = xb + (ok ? row : 0) * (size_t)kb;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = 4 * hh + j; v[j] = p[c < kb ? c : 0]; }
}
__device__ __forceinline__ void mask_xb(const float (&raw)[4], int kb, bool ok, int hh, float (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = raw[j] * ((ok && 4 * hh + j < kb) ? 1.f : 0.f);
}
// h = b + Wb xb for one 32-row tile: the xb part of the trunk's forward (trunk_forward_tile with no xa block, the same MFMAs in the same order: the
// same bits), from the [64][12] image of Wb and the bias behind it -- used by the trunk kernel and by the heads that recompute h (EmdMlpBranch.xb)
__device__ __forceinline__ void embed_h(const float* __restrict__ wb, const float* __restrict__ b, int r, int hh, const float (&xb)[4], f32x16 (&h)[2], bool init = true) {
    if (init) { h[0] = bias_tile(b, 0, hh); h[1] = bias_tile(b, 32, hh); }
#pragma unroll
    for (int to = 0; to < 2; to++) {
        const float4 w = *(const float4*)(wb + (32 * to + r) * 12 + 4 * hh);
        h[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, xb[0], h[to], 0, 0, 0);
        h[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, xb[1], h[to], 0, 0, 0);
        h[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, xb[2], h[to], 0, 0, 0);
        h[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, xb[3], h[to], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// branch: [relu] -> Linear(64, 64) -> relu [-> Linear(64, 64) -> relu] -> Linear(64, out_dim)
// ---------------------------------------------------------------------------------------------------------------------------------
// LDS (floats): W1 [64][WS] | W2 [64][WS] (DEPTH 2) | Wo [32 NTO][WS] | b1 [64] | b2 [64] | bo [64] | scratch [waves][32][TS] (backward)
#ifndef MLP_SP_F2
#define MLP_SP_F2 1           /* forward of the two-hidden-layer feature head: both hidden layers on split bf16 (round 5, late) */
#endif
constexpr int branch_mode(int depth, int nto, bool bwd) {
    // (the BACKWARD of the two-hidden-layer feature head keeps fp32 MFMA: its six images + transposes would need 167 KB of LDS; its forward has
    //  three images only)
    return depth != 1 ? (bwd ? 0 : MLP_SP_F2) : (bwd ? (nto == 1 ? MLP_SP_B11 : MLP_SP_B12) : (nto == 1 ? MLP_SP_F11 : MLP_SP_F12));
}
template <int DEPTH, int NTO, int MODE = 0>
struct BranchLds {
    static constexpr bool SP = (MODE & 1) != 0;            // weights as split-bf16 images
    static constexpr bool SPO = (MODE & 2) != 0;           // outer products on split bf16
    static constexpr int W64 = SP ? split_floats(2, 4) : 64 * WS;                    // a 64 x 64 matrix
    static constexpr int w1 = 0;
    static constexpr int w2 = w1 + W64;
    static constexpr int wo = w2 + (DEPTH == 2 ? W64 : 0);
    static constexpr int b1 = wo + (SP ? split_floats(NTO, 4) : 32 * NTO * WS);
    static constexpr int b2 = b1 + 64;
    static constexpr int bo = b2 + 64;
    static constexpr int fwd_floats = bo + 64;
    // backward only: the transposed images W1^T [64][WS], W2^T, Wo^T [64][WOT] and one transpose tile per wave
    static constexpr int WOT = 32 * NTO + 4;
    static constexpr int w1t = fwd_floats;
    static constexpr int w2t = w1t + W64;
    static constexpr int wot = w2t + (DEPTH == 2 ? W64 : 0);
    static constexpr int scratch = wot + (SP ? split_floats(2, 2 * NTO) : 64 * WOT);
    static constexpr int bwd_floats = scratch + MLP_WAVES * 32 * TS;
    // RC kernels (the head recomputes h from the embedding): Win [64][12] | b_in [64] behind the kernel's other regions
    static constexpr int rc_floats = 64 * 12 + 64;
};

// one call for both weight forms: the fp32 path reads rows of a padded LDS matrix, the split path ready-made bf16 fragments
template <bool SP, int KT, int NT, int NA = 4>
__device__ __forceinline__ void mm(f32x16 (&acc)[NT], const f32x16 (&in)[KT], const float* __restrict__ W, int stride, int k0, int nt_img, int kb_img, int r, int hh,
                                   int lane) {
    if constexpr (SP) layer_fwd_s<KT, NT, (NA + 1) / 2>(acc, in, reinterpret_cast<const uint4*>(W), nt_img, kb_img, k0 / 16, lane);
    else layer_fwd<KT, NT, NA>(acc, in, W, stride, k0, r, hh);
}
// dW[t][i] += gf (x) af[i], i = 0 .. NI - 1 (the fragments are split once per call on the bf16 path)
template <bool SP, int NI>
__device__ __forceinline__ void outer_all(f32x16 (&dW)[NI], const f32x16& gf, const f32x16 (&af)[NI], const Split (&afs)[NI]) {
    if constexpr (SP) {
        const Split gs = split_tile(gf);
#pragma unroll
        for (int i = 0; i < NI; i++) dW[i] = outer_acc_s(dW[i], gs, afs[i]);
    } else {
#pragma unroll
        for (int i = 0; i < NI; i++) dW[i] = outer_acc(dW[i], gf, af[i]);
    }
}
template <bool SP, int NI>
__device__ __forceinline__ void split_all(Split (&afs)[NI], const f32x16 (&af)[NI]) {
    if constexpr (SP) {
#pragma unroll
        for (int i = 0; i < NI; i++) afs[i] = split_tile(af[i]);
    }
}

template <int DEPTH, int NTO, bool BWD, bool RC = false>
__device__ __forceinline__ void branch_stage(float* lds, const EmdMlpBranch& a) {
    typedef BranchLds<DEPTH, NTO, branch_mode(DEPTH, NTO, BWD)> L;
    if constexpr (RC) {      // the trunk's embedding block and bias, in the trunk kernel's own layout (trunk_stage)
        float* rc = lds + (BWD ? L::bwd_floats : L::fwd_floats);
        stage_matrix(rc, 12, 64, 8, a.w_in, a.ld_w_in, a.col_in, 64, a.kb_in);
        stage_vector(rc + 64 * 12, 64, a.b_in, 64);
    }
    if constexpr (L::SP) {
        if (BWD) {
            stage_split(lds + L::w1t, 2, 4, a.w_hidden[0], 64, 0, 64, 64, true);
            stage_split(lds + L::wot, 2, 2 * NTO, a.w_out, 64, 0, a.out_dim, 64, true);
        }
        stage_split(lds + L::w1, 2, 4, a.w_hidden[0], 64, 0, 64, 64, false);
        if (DEPTH == 2) stage_split(lds + L::w2, 2, 4, a.w_hidden[1], 64, 0, 64, 64, false);        // (forward only: branch_mode)
        // (a narrow output layer runs on the vector pipe in the forward and reads plain fp32 rows: they fit the image's allocation)
        if (!BWD && MLP_NARROW_OUT && NTO == 1 && a.out_dim <= 4) stage_matrix(lds + L::wo, WS, 32, 64, a.w_out, 64, 0, a.out_dim, 64);
        else stage_split(lds + L::wo, NTO, 4, a.w_out, 64, 0, a.out_dim, 64, false);
    } else {
        if (BWD) {
            stage_matrix_t(lds + L::w1t, WS, 64, 64, a.w_hidden[0], 64, 0, 64, 64);
            if (DEPTH == 2) stage_matrix_t(lds + L::w2t, WS, 64, 64, a.w_hidden[1], 64, 0, 64, 64);
            stage_matrix_t(lds + L::wot, L::WOT, 32 * NTO, 64, a.w_out, 64, 0, a.out_dim, 64);
        }
        stage_matrix(lds + L::w1, WS, 64, 64, a.w_hidden[0], 64, 0, 64, 64);
        if (DEPTH == 2) stage_matrix(lds + L::w2, WS, 64, 64, a.w_hidden[1], 64, 0, 64, 64);
        stage_matrix(lds + L::wo, WS, 32 * NTO, 64, a.w_out, 64, 0, a.out_dim, 64);
    }
    stage_vector(lds + L::b1, 64, a.b_hidden[0], 64);
    stage_vector(lds + L::b2, 64, DEPTH == 2 ? a.b_hidden[1] : nullptr, 64);
    stage_vector(lds + L::bo, 64, a.b_out, a.out_dim);
    __syncthreads();
}

// L1: the head's regulariser mean |out| is formed beside the outputs (EmdMlpBranch.l1_sum); a separate instantiation keeps the registers of
// the usual kernels as they were
// RC: the head forms h = b_in + W_in xb itself (EmdMlpBranch.xb, a level without HexPlane features) instead of reading a [N,64] tensor
template <int DEPTH, int NTO, bool L1 = false, bool RC = false>
__global__ void __launch_bounds__(MLP_THREADS) MLP_OCC((DEPTH == 1 && NTO == 1 && (!L1 || MLP_L1_F11_TWO)) ? MLP_W_F11 : MLP_FWD_WAVES) k_mlp_branch_fwd(EmdMlpBranch a) {
    typedef BranchLds<DEPTH, NTO, branch_mode(DEPTH, NTO, false)> L;
    extern __shared__ float lds[];
    branch_stage<DEPTH, NTO, false, RC>(lds, a);
    const float* rc = lds + L::fwd_floats;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const size_t N = (size_t)a.num_points, tiles = (N + 31) / 32;
    const bool wide_out = (a.out_dim & 3) == 0;
    const bool narrow_out = a.out_dim <= 4;
    float l1_acc = 0.f;
    const size_t tile0 = (size_t)blockIdx.x * MLP_WAVES + wave, tstep = (size_t)gridDim.x * MLP_WAVES;
    // the next tile's rows are loaded while this tile computes: one wave per SIMD has nothing else to hide the HBM latency behind
    f32x16 nx[2];
    float nxb[4];
    // (h travels raw: a row past the end reads row 0 and its outputs are neither stored nor counted -- a product or select behind the load puts the
    //  wait for the prefetch AT the load)
    auto hrow = [&](size_t row_) -> size_t { return row_ < N ? row_ : 0; };
    if constexpr (RC) load_xb_raw(a.xb, a.kb_in, tile0 * 32 + r, tile0 * 32 + r < N, hh, nxb);
    else { nx[0] = load_tile_raw(a.h, 64, hrow(tile0 * 32 + r), 0, hh); nx[1] = load_tile_raw(a.h, 64, hrow(tile0 * 32 + r), 32, hh); }
    for (size_t tile = tile0; tile < tiles; tile += tstep) {
        const size_t row = tile * 32 + r;
        const bool ok = row < N;
        f32x16 x[2];
        const size_t nrow = (tile + tstep) * 32 + r;
        if constexpr (RC) {
            float xb[4];
            mask_xb(nxb, a.kb_in, ok, hh, xb);
            load_xb_raw(a.xb, a.kb_in, nrow, nrow < N, hh, nxb);
            embed_h(rc, rc + 64 * 12, r, hh, xb, x);
        } else {
            x[0] = nx[0]; x[1] = nx[1];
            nx[0] = load_tile_raw(a.h, 64, hrow(nrow), 0, hh); nx[1] = load_tile_raw(a.h, 64, hrow(nrow), 32, hh);
        }
        if (a.relu_input) { x[0] = relu16(x[0]); x[1] = relu16(x[1]); }
        f32x16 m[2] = {bias_tile(lds + L::b1, 0, hh), bias_tile(lds + L::b1, 32, hh)};
        mm<L::SP, 2, 2>(m, x, lds + L::w1, WS, 0, 2, 4, r, hh, lane);
        m[0] = relu16(m[0]); m[1] = relu16(m[1]);
        if (DEPTH == 2) {
            f32x16 m2[2] = {bias_tile(lds + L::b2, 0, hh), bias_tile(lds + L::b2, 32, hh)};
            mm<L::SP, 2, 2>(m2, m, lds + L::w2, WS, 0, 2, 4, r, hh, lane);
            m[0] = relu16(m2[0]); m[1] = relu16(m2[1]);
        }
        if constexpr (MLP_NARROW_OUT && NTO == 1) {
            // Round 5: an output layer of at most four features (dx: 3, do: 1) on the VECTOR pipe.  As an MFMA tile it is padded to 32 output rows --
            // 32 of the kernel's 96 MFMAs per 32-row tile for 3 or 1 useful rows -- while the contraction itself is 64 multiply-adds per output and
            // row: a lane holds 32 of its row's 64 activations (the two lane halves hold the two feature halves), so it forms 4 x 32 products
            // against weight rows read as broadcast ds_read_b128 and the halves meet in one cross-lane add.  (Summation order differs from the MFMA's
            // k-ordered chain: fp32 rounding only.)
            if (narrow_out) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int f0 = 32 * t + 8 * c + 4 * hh;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float4 w = *(const float4*)(lds + L::wo + q * WS + f0);           // (rows >= out_dim of the staged matrix are zero)
                            acc[q] = fmaf(w.x, m[t][4 * c], acc[q]); acc[q] = fmaf(w.y, m[t][4 * c + 1], acc[q]);
                            acc[q] = fmaf(w.z, m[t][4 * c + 2], acc[q]); acc[q] = fmaf(w.w, m[t][4 * c + 3], acc[q]);
                        }
                    }
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = acc[q] + __shfl_xor(acc[q], 32) + lds[L::bo + q];
                if (hh == 0 && ok) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (q < a.out_dim) {
                            a.out[row * (size_t)a.out_dim + q] = acc[q];
                            if (L1 && a.l1_sum) l1_acc += fabsf(acc[q]);
                        }
                }
                continue;
            }
        }
        f32x16 o[NTO];
#pragma unroll
        for (int t = 0; t < NTO; t++) o[t] = bias_tile(lds + L::bo, 32 * t, hh);
        mm<L::SP, 2, NTO>(o, m, lds + L::wo, WS, 0, NTO, 4, r, hh, lane);
        if (L1 && a.l1_sum && ok) {                   // the head's L1 regulariser, formed while the outputs are in registers
#pragma unroll
            for (int t = 0; t < NTO; t++)
#pragma unroll
                for (int v = 0; v < 16; v++)
                    if (32 * t + 8 * (v >> 2) + 4 * hh + (v & 3) < a.out_dim) l1_acc += fabsf(o[t][v]);
        }
#pragma unroll
        for (int t = 0; t < NTO; t++) {
            // (a wide tile may still hang over the row's end: out_dim = 48 -> the second tile holds features 32..47: whole 16-byte pieces, bounded by the width)
            if (wide_out) store_tile(a.out, a.out_dim, row, ok, 32 * t, hh, o[t], a.out_dim);
            else store_tile_narrow(a.out, a.out_dim, row, ok, 32 * t, hh, o[t]);
        }
    }
    if (L1 && a.l1_sum) {                             // mean |out| over the N x out_dim outputs: ONE atomic per workgroup (a float atomic per wave was
        __shared__ float s_l1[MLP_WAVES];             // 1 024 - 2 048 atomics on one address at the very end of the kernel, served one after the other: ~40 us)
        const float s = wave_reduce_to_lane63(l1_acc);
        if (lane == 63) s_l1[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < MLP_WAVES; w++) t += s_l1[w];
            atomicAdd(a.l1_sum, t / ((float)a.num_points * (float)a.out_dim));
        }
    }
}

// GO: how dL/dout reaches the kernel.  0: at most four outputs (dx / do heads, NTO = 1): four raw values per row; 1: out_dim a multiple of 4
// (dshs: 48): raw 16-byte pieces, a partial tile masked by whole pieces; 2: any width (guarded loads, masked at the load: the old path)
// CHAIN (GO 0 / 1): g_h_in is present.  A runtime branch around a group of loads would do: but then the number of loads in flight differs between the
// paths, the compiler's wait for the PREVIOUS iteration's prefetch has to assume the smaller one, and that wait lands on the loads just issued.
// GO 0 / 1 therefore take every such decision at compile time (and need g_out; without it the launcher picks GO 2).
template <int DEPTH, int NTO, bool L1 = false, bool RC = false, int GO = 2, bool CHAIN = false>
__global__ void __launch_bounds__(MLP_THREADS) MLP_OCC(MLP_BWD_WAVES) k_mlp_branch_bwd(EmdMlpBranch a, EmdMlpBranchGrads g) {
    typedef BranchLds<DEPTH, NTO, branch_mode(DEPTH, NTO, true)> L;
    extern __shared__ float lds[];
    branch_stage<DEPTH, NTO, true, RC>(lds, a);
    const float* rc = lds + L::bwd_floats;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    float* T = lds + L::scratch + wave * 32 * TS;
    const size_t N = (size_t)a.num_points, tiles = (N + 31) / 32;
    const bool wide_out = (a.out_dim & 3) == 0;
    f32x16 dW1[2][2], dW2[2][2], dWo[NTO][2];
    float db1[2] = {0.f, 0.f}, db2[2] = {0.f, 0.f}, dbo[NTO];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) { dW1[i][j] = zero16(); dW2[i][j] = zero16(); }
#pragma unroll
    for (int t = 0; t < NTO; t++) { dWo[t][0] = zero16(); dWo[t][1] = zero16(); dbo[t] = 0.f; }

    const size_t tile0 = (size_t)blockIdx.x * MLP_WAVES + wave, tstep = (size_t)gridDim.x * MLP_WAVES;
    // g.l1_grad: the gradient of the head's L1 regulariser (mean |out|, EmdMlpBranch.l1_sum) joins dL/dout while it is loaded:
    // + sign(out) l1_grad / (N out_dim); g_out itself may then be absent
    const float l1_s = (L1 && g.l1_grad) ? g.l1_grad[0] / ((float)a.num_points * (float)a.out_dim) : 0.f;
    auto load_go = [&](size_t row_, bool ok_, int t) -> f32x16 {
        const bool wide = wide_out && 32 * t + 32 <= a.out_dim;
        f32x16 v = (L1 && !g.g_out) ? zero16() : (wide ? load_tile(g.g_out, a.out_dim, row_, ok_, 32 * t, hh) : load_tile_narrow(g.g_out, a.out_dim, row_, ok_, 32 * t, hh));
        if (L1 && g.l1_grad) {
            const f32x16 o = wide ? load_tile(g.out, a.out_dim, row_, ok_, 32 * t, hh) : load_tile_narrow(g.out, a.out_dim, row_, ok_, 32 * t, hh);
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] += o[k] > 0.f ? l1_s : (o[k] < 0.f ? -l1_s : 0.f);      // (rows / features past the end read 0: no term)
        }
        return v;
    };
    // the next tile's h and g_out are loaded while this tile computes (one wave per SIMD: nothing else hides the HBM latency).  Every load of the
    // loop is unconditional and raw (see load_tile_raw); rows past the end read row 0 and their dL/dout is masked to zero where it is used, which
    // zeroes everything they could contribute (their h needs no mask: a finite activation times a zero gradient).
    auto rowc = [&](size_t row_) -> size_t { return row_ < N ? row_ : 0; };
    f32x16 nh[2];
    float nxb[4];
    f32x16 ngo[NTO];
    float ngn[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](size_t row_) {                      // h (or the embedding) and dL/dout of one tile
        const size_t rcl = rowc(row_);
        if constexpr (RC) load_xb_raw(a.xb, a.kb_in, row_, row_ < N, hh, nxb);
        else { nh[0] = load_tile_raw(a.h, 64, rcl, 0, hh); nh[1] = load_tile_raw(a.h, 64, rcl, 32, hh); }
        if constexpr (GO == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) ngn[j] = g.g_out[rcl * (size_t)a.out_dim + (j < a.out_dim ? j : 0)];
        } else if constexpr (GO == 1) {
#pragma unroll
            for (int t = 0; t < NTO; t++) ngo[t] = load_tile_raw(g.g_out, a.out_dim, rcl, 32 * t, hh, a.out_dim);
        } else {
#pragma unroll
            for (int t = 0; t < NTO; t++) ngo[t] = load_go(row_, row_ < N, t);
        }
    };
    fetch(tile0 * 32 + r);
    for (size_t tile = tile0; tile < tiles; tile += tstep) {
        const size_t row = tile * 32 + r;
        const bool ok = row < N;
        // ---- recompute the forward
        f32x16 hin[2];
        f32x16 go[NTO];
        float gn[4];
        if constexpr (GO == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) gn[j] = ngn[j];
        } else {
#pragma unroll
            for (int t = 0; t < NTO; t++) go[t] = ngo[t];
        }
        float xbm[4];
        if constexpr (RC) mask_xb(nxb, a.kb_in, ok, hh, xbm);
        else { hin[0] = nh[0]; hin[1] = nh[1]; }
        // this tile's forward outputs (the L1 term's sign) and the earlier heads' dL/dh (EmdMlpBranchGrads.g_h_in, added before the store) are
        // requested first -- older than the prefetches behind them, so waiting for them later does not wait for those
        f32x16 oraw[NTO];
        float on[4];
        if constexpr (L1 && GO == 0) {                    // (an L1 kernel is launched with l1_grad and the forward's outputs present)
#pragma unroll
            for (int j = 0; j < 4; j++) on[j] = g.out[rowc(row) * (size_t)a.out_dim + (j < a.out_dim ? j : 0)];
        } else if constexpr (L1 && GO == 1) {
#pragma unroll
            for (int t = 0; t < NTO; t++) oraw[t] = load_tile_raw(g.out, a.out_dim, rowc(row), 32 * t, hh, a.out_dim);
        }
        f32x16 gin[2] = {zero16(), zero16()};
        if constexpr (GO == 2) {
            if (DEPTH == 1 && g.g_h_in) { gin[0] = load_tile(g.g_h_in, 64, row, ok, 0, hh); gin[1] = load_tile(g.g_h_in, 64, row, ok, 32, hh); }
        } else if constexpr (CHAIN) { gin[0] = load_tile_raw(g.g_h_in, 64, rowc(row), 0, hh); gin[1] = load_tile_raw(g.g_h_in, 64, rowc(row), 32, hh); }
        fetch((tile + tstep) * 32 + r);
        if constexpr (RC) embed_h(rc, rc + 64 * 12, r, hh, xbm, hin);
        f32x16 x[2] = {hin[0], hin[1]};
        if (a.relu_input) { x[0] = relu16(x[0]); x[1] = relu16(x[1]); }
        f32x16 m1[2] = {bias_tile(lds + L::b1, 0, hh), bias_tile(lds + L::b1, 32, hh)};
        mm<L::SP, 2, 2>(m1, x, lds + L::w1, WS, 0, 2, 4, r, hh, lane);
        m1[0] = relu16(m1[0]); m1[1] = relu16(m1[1]);
        f32x16 m2[2];
        if (DEPTH == 2) {
            m2[0] = bias_tile(lds + L::b2, 0, hh); m2[1] = bias_tile(lds + L::b2, 32, hh);
            layer_fwd<2, 2>(m2, m1, lds + L::w2, WS, 0, r, hh);
            m2[0] = relu16(m2[0]); m2[1] = relu16(m2[1]);
        }
        f32x16 (&last)[2] = DEPTH == 2 ? m2 : m1;          // the activation that feeds the output layer
        // ---- dL/dout of this tile from its raw pieces: masked here, after the forward's MFMAs, where it is first needed
        if constexpr (GO == 0) {
            go[0] = zero16();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = gn[j];
                if constexpr (L1) v += on[j] > 0.f ? l1_s : (on[j] < 0.f ? -l1_s : 0.f);
                go[0][j] = v * ((ok && hh == 0 && j < a.out_dim) ? 1.f : 0.f);
            }
        } else if constexpr (GO == 1) {
#pragma unroll
            for (int t = 0; t < NTO; t++) {
                if constexpr (L1) {
#pragma unroll
                    for (int k = 0; k < 16; k++) go[t][k] += oraw[t][k] > 0.f ? l1_s : (oraw[t][k] < 0.f ? -l1_s : 0.f);
                }
                go[t] = mask_tile(go[t], ok, 32 * t, hh, a.out_dim);
            }
        }
        // ---- output layer
        f32x16 gl[2] = {zero16(), zero16()};
        if (NTO == 1 && a.out_dim <= 8) mm<L::SP, NTO, 2, 1>(gl, go, lds + L::wot, L::WOT, 0, 2, 2 * NTO, r, hh, lane);  // dx / do / feat: one 8-feature chunk carries everything
        else if (NTO == 2 && a.out_dim <= 48) {       // dshs (48 outputs): the second tile's upper two 8-feature chunks are padding
            const f32x16 g0[1] = {go[0]}, g1[1] = {go[NTO - 1]};
            mm<L::SP, 1, 2>(gl, g0, lds + L::wot, L::WOT, 0, 2, 2 * NTO, r, hh, lane);
            mm<L::SP, 1, 2, 2>(gl, g1, lds + L::wot, L::WOT, 32, 2, 2 * NTO, r, hh, lane);
        } else mm<L::SP, NTO, 2>(gl, go, lds + L::wot, L::WOT, 0, 2, 2 * NTO, r, hh, lane);
        gl[0] = mask16(gl[0], last[0]); gl[1] = mask16(gl[1], last[1]);
        {   // dWo += go (x) last, dbo += rowsum(go)
            const f32x16 af[2] = {transpose_tile(last[0], T, r, hh), transpose_tile(last[1], T, r, hh)};
            Split afs[2];
            split_all<L::SPO, 2>(afs, af);
#pragma unroll
            for (int t = 0; t < NTO; t++) {
                const f32x16 gf = transpose_tile(go[t], T, r, hh);
                dbo[t] += frag_sum(gf);
                outer_all<L::SPO, 2>(dWo[t], gf, af, afs);
            }
        }
        // ---- second hidden layer (feature head)
        f32x16 g1[2];
        if (DEPTH == 2) {
            g1[0] = zero16(); g1[1] = zero16();
            layer_fwd<2, 2>(g1, gl, lds + L::w2t, WS, 0, r, hh);
            g1[0] = mask16(g1[0], m1[0]); g1[1] = mask16(g1[1], m1[1]);
            f32x16 af[2] = {transpose_tile(m1[0], T, r, hh), transpose_tile(m1[1], T, r, hh)};
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f32x16 gf = transpose_tile(gl[t], T, r, hh);
                db2[t] += frag_sum(gf);
                dW2[t][0] = outer_acc(dW2[t][0], gf, af[0]);
                dW2[t][1] = outer_acc(dW2[t][1], gf, af[1]);
            }
        } else {
            g1[0] = gl[0]; g1[1] = gl[1];
        }
        // ---- first hidden layer
        f32x16 gx[2] = {zero16(), zero16()};
        mm<L::SP, 2, 2>(gx, g1, lds + L::w1t, WS, 0, 2, 4, r, hh, lane);
        if (a.relu_input) { gx[0] = mask16(gx[0], hin[0]); gx[1] = mask16(gx[1], hin[1]); }
        if ((GO == 2 && DEPTH == 1 && g.g_h_in) || (GO != 2 && CHAIN)) { gx[0] += gin[0]; gx[1] += gin[1]; }     // (one-hidden-layer heads only: the other kernel has no registers left)
        store_tile(g.g_h, 64, row, ok, 0, hh, gx[0]);
        store_tile(g.g_h, 64, row, ok, 32, hh, gx[1]);
        {
            const f32x16 af[2] = {transpose_tile(x[0], T, r, hh), transpose_tile(x[1], T, r, hh)};
            Split afs[2];
            split_all<L::SPO, 2>(afs, af);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f32x16 gf = transpose_tile(g1[t], T, r, hh);
                db1[t] += frag_sum(gf);
                outer_all<L::SPO, 2>(dW1[t], gf, af, afs);
            }
        }
    }
    // ---- workgroup sums, then one atomic add per accumulator element and workgroup
    float* red = lds + L::scratch;
#pragma unroll
    for (int to = 0; to < 2; to++) {
#pragma unroll
        for (int ti = 0; ti < 2; ti++) {
            const f32x16 t1 = wg_sum16(dW1[to][ti], red, wave, lane);
            if (wave == 0) flush_dw(g.d_w_hidden[0], 64, 32 * to, 32 * ti, 64, 64, t1, r, hh);
            if (DEPTH == 2) {
                const f32x16 t2 = wg_sum16(dW2[to][ti], red, wave, lane);
                if (wave == 0) flush_dw(g.d_w_hidden[1], 64, 32 * to, 32 * ti, 64, 64, t2, r, hh);
            }
        }
        const float b1s = wg_sum1(db1[to], red, wave, lane);
        if (wave == 0) flush_db(g.d_b_hidden[0], 32 * to, 64, b1s, r);
        if (DEPTH == 2) {
            const float b2s = wg_sum1(db2[to], red, wave, lane);
            if (wave == 0) flush_db(g.d_b_hidden[1], 32 * to, 64, b2s, r);
        }
    }
#pragma unroll
    for (int t = 0; t < NTO; t++) {
        const f32x16 o0 = wg_sum16(dWo[t][0], red, wave, lane), o1 = wg_sum16(dWo[t][1], red, wave, lane);
        const float bos = wg_sum1(dbo[t], red, wave, lane);
        if (wave == 0) {
            flush_dw(g.d_w_out, 64, 32 * t, 0, a.out_dim, 64, o0, r, hh);
            flush_dw(g.d_w_out, 64, 32 * t, 32, a.out_dim, 64, o1, r, hh);
            flush_db(g.d_b_out, 32 * t, a.out_dim, bos, r);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// trunk: h = b + W[:, col_a : col_a + ka] xa + W[:, col_b : col_b + kb] xb           (ka = 32 KTA, kb <= 8)
// ---------------------------------------------------------------------------------------------------------------------------------
// LDS: Wa [64][SA] | Wb [64][12] | b [64] | scratch (backward)
template <int KTA, int MODE = 0>
struct TrunkLds {
    static constexpr bool SP = (MODE & 1) != 0 && KTA > 0, SPO = (MODE & 2) != 0 && KTA > 0;
    static constexpr int SA = 32 * KTA + 4;                 // 4 x odd
    static constexpr int wa = 0;
    static constexpr int wb = wa + (KTA ? (SP ? split_floats(2, 2 * KTA) : 64 * SA) : 0);
    static constexpr int b = wb + 64 * 12;
    static constexpr int fwd_floats = b + 64;
    // backward: only the transposed images Wa^T [32 KTA][WS], Wb^T [32][WS] (rows >= kb zero) and one transpose tile per wave
    static constexpr int wat = 0;
    static constexpr int wbt = wat + (SP ? split_floats(KTA, 4) : 32 * KTA * WS);
    static constexpr int scratch = wbt + 32 * WS;
    static constexpr int bwd_floats = scratch + MLP_WAVES * 32 * TS;
};

template <int KTA>
__device__ __forceinline__ void trunk_stage(float* lds, const EmdMlpTrunk& a) {
    typedef TrunkLds<KTA, MLP_SP_TF> L;
    if (KTA) {
        if constexpr (L::SP) stage_split(lds + L::wa, 2, 2 * KTA, a.w, a.ld_w, a.col_a, 64, a.ka, false);
        else stage_matrix(lds + L::wa, L::SA, 64, 32 * KTA, a.w, a.ld_w, a.col_a, 64, a.ka);
    }
    stage_matrix(lds + L::wb, 12, 64, 8, a.kb > 0 ? a.w : nullptr, a.ld_w, a.col_b, 64, a.kb);
    stage_vector(lds + L::b, 64, a.b, 64);
    __syncthreads();
}

// this tile's xa / xb rows (already in registers) -> h
template <int KTA>
__device__ __forceinline__ void trunk_forward_tile(const float* lds, int r, int hh, const f32x16 (&xa)[KTA ? KTA : 1], const float (&xb)[4], f32x16 (&h)[2]) {
    typedef TrunkLds<KTA, MLP_SP_TF> L;
    h[0] = bias_tile(lds + L::b, 0, hh); h[1] = bias_tile(lds + L::b, 32, hh);
    if (KTA) mm<L::SP, (KTA ? KTA : 1), 2>(h, xa, lds + L::wa, L::SA, 0, 2, 2 * KTA, r, hh, r + 32 * hh);
    embed_h(lds + L::wb, lds + L::b, r, hh, xb, h, false);
}

// raw rows for the forward's prefetch: clamped row, a column past the width reads column 0 (its weights are zero: the staged images are zero padded);
// nothing is masked -- the rows past the end are not stored
template <int KTA>
__device__ __forceinline__ void trunk_load_x_raw(const EmdMlpTrunk& a, size_t row_c, int hh, f32x16 (&xa)[KTA ? KTA : 1], float (&xb)[4]) {
    if (KTA) {
#pragma unroll
        for (int t = 0; t < KTA; t++) xa[t] = load_tile_raw(a.xa, a.ka, row_c, 32 * t, hh, a.ka);
    }
    if (a.kb > 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) xb[j] = a.xb[row_c * (size_t)a.kb + (4 * hh + j < a.kb ? 4 * hh + j : 0)];
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) xb[j] = 0.f;
    }
}
template <int KTA, bool STRAIGHT>
__device__ __forceinline__ void trunk_load_x(const EmdMlpTrunk& a, size_t row, bool ok, int hh, f32x16 (&xa)[KTA ? KTA : 1], float (&xb)[4]) {
    if (KTA) {
#pragma unroll
        for (int t = 0; t < KTA; t++) xa[t] = load_tile<STRAIGHT>(a.xa, a.ka, row, ok, 32 * t, hh, a.ka);
    }
    load_xb<STRAIGHT>(a.xb, a.kb, row, ok, hh, xb);
}

template <int KTA>
__global__ void __launch_bounds__(MLP_THREADS) MLP_OCC(KTA == 0 ? MLP_W_T0 : MLP_FWD_WAVES) k_mlp_trunk_fwd(EmdMlpTrunk a) {
    extern __shared__ float lds[];
    trunk_stage<KTA>(lds, a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const size_t N = (size_t)a.num_points, tiles = (N + 31) / 32;
    const size_t tile0 = (size_t)blockIdx.x * MLP_WAVES + wave, tstep = (size_t)gridDim.x * MLP_WAVES;
    // the next tile's rows are loaded while this tile computes (one wave per SIMD: nothing else hides the HBM latency)
    f32x16 nxa[KTA ? KTA : 1];
    float nxb[4];
    trunk_load_x_raw<KTA>(a, tile0 * 32 + r < N ? tile0 * 32 + r : 0, hh, nxa, nxb);
    for (size_t tile = tile0; tile < tiles; tile += tstep) {
        const size_t row = tile * 32 + r;
        const bool ok = row < N;
        f32x16 xa[KTA ? KTA : 1], h[2];
        float xb[4];
#pragma unroll
        for (int t = 0; t < (KTA ? KTA : 1); t++) xa[t] = nxa[t];
#pragma unroll
        for (int j = 0; j < 4; j++) xb[j] = nxb[j];
        {
            const size_t nrow = (tile + tstep) * 32 + r;
            trunk_load_x_raw<KTA>(a, nrow < N ? nrow : 0, hh, nxa, nxb);
        }
        trunk_forward_tile<KTA>(lds, r, hh, xa, xb, h);
        store_tile(a.h, 64, row, ok, 0, hh, h[0]);
        store_tile(a.h, 64, row, ok, 32, hh, h[1]);
    }
}

// The loop's loads (round 5): dL/dh of the NEXT tile is the prefetch (32 registers; it is the first thing a tile needs), xa / xb of THIS tile are
// requested at the top of the iteration and first read after the dL/dxa product (96 MFMAs later) -- before, xa / xb of the next tile travelled (64
// registers) while dL/dh was loaded and waited for in place, a full HBM round trip per tile.  All of them raw and unconditional (load_tile_raw): rows
// past the end read row 0, and dL/dh is zeroed for them where it is used, which zeroes all they could add.  MULTI: more than one dL/dh tensor (a
// two-hidden-layer head keeps its own): the others are loaded inside the loop, guarded as before.
template <int KTA, bool MULTI = false>
__global__ void __launch_bounds__(MLP_THREADS) MLP_OCC(KTA == 0 ? MLP_W_TB0 : MLP_BWD_WAVES) k_mlp_trunk_bwd(EmdMlpTrunk a, EmdMlpTrunkGrads g) {
    typedef TrunkLds<KTA, MLP_SP_TB> L;
    extern __shared__ float lds[];
    if (KTA) {
        if constexpr (L::SP) stage_split(lds + L::wat, KTA, 4, a.w, a.ld_w, a.col_a, 64, a.ka, true);
        else stage_matrix_t(lds + L::wat, WS, 64, 32 * KTA, a.w, a.ld_w, a.col_a, 64, a.ka);
    }
    stage_matrix_t(lds + L::wbt, WS, 64, 32, a.kb > 0 ? a.w : nullptr, a.ld_w, a.col_b, 64, a.kb);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    float* T = lds + L::scratch + wave * 32 * TS;
    const size_t N = (size_t)a.num_points, tiles = (N + 31) / 32;
    f32x16 dWa[2][KTA ? KTA : 1], dWb[2];
    float db[2] = {0.f, 0.f};
#pragma unroll
    for (int to = 0; to < 2; to++) {
        dWb[to] = zero16();
#pragma unroll
        for (int t = 0; t < (KTA ? KTA : 1); t++) dWa[to][t] = zero16();
    }
    const size_t tile0 = (size_t)blockIdx.x * MLP_WAVES + wave, tstep = (size_t)gridDim.x * MLP_WAVES;
    auto rowc = [&](size_t row_) -> size_t { return row_ < N ? row_ : 0; };
    f32x16 ngh[2] = {load_tile_raw(g.g_h[0], 64, rowc(tile0 * 32 + r), 0, hh), load_tile_raw(g.g_h[0], 64, rowc(tile0 * 32 + r), 32, hh)};
    for (size_t tile = tile0; tile < tiles; tile += tstep) {
        const size_t row = tile * 32 + r;
        const bool ok = row < N;
        // dL/dh: the sum of the branches' contributions
        f32x16 gh[2] = {mask_tile(ngh[0], ok, 0, hh, 64), mask_tile(ngh[1], ok, 32, hh, 64)};
        f32x16 xa[KTA ? KTA : 1];
        float xb[4];
        if (KTA) {
#pragma unroll
            for (int t = 0; t < KTA; t++) xa[t] = load_tile_raw(a.xa, a.ka, rowc(row), 32 * t, hh, a.ka);
        }
        if (a.kb > 0) {                                  // (kernel-uniform, the same loads every iteration)
#pragma unroll
            for (int j = 0; j < 4; j++) xb[j] = a.xb[rowc(row) * (size_t)a.kb + (4 * hh + j < a.kb ? 4 * hh + j : 0)];
        }
        {
            const size_t nrow = rowc((tile + tstep) * 32 + r);
            ngh[0] = load_tile_raw(g.g_h[0], 64, nrow, 0, hh); ngh[1] = load_tile_raw(g.g_h[0], 64, nrow, 32, hh);
        }
        if constexpr (MULTI) {
            for (int k = 1; k < g.num_gh; k++) {
                const f32x16 p0 = load_tile(g.g_h[k], 64, row, ok, 0, hh), p1 = load_tile(g.g_h[k], 64, row, ok, 32, hh);
#pragma unroll
                for (int v = 0; v < 16; v++) { gh[0][v] += p0[v]; gh[1][v] += p1[v]; }
            }
        }
        const f32x16 gf[2] = {transpose_tile(gh[0], T, r, hh), transpose_tile(gh[1], T, r, hh)};
        db[0] += frag_sum(gf[0]); db[1] += frag_sum(gf[1]);
        if (KTA) {
            if (g.d_xa) {
                f32x16 gx[KTA ? KTA : 1];
#pragma unroll
                for (int t = 0; t < KTA; t++) gx[t] = zero16();
                mm<L::SP, 2, (KTA ? KTA : 1)>(gx, gh, lds + L::wat, WS, 0, KTA, 4, r, hh, lane);
#pragma unroll
                for (int t = 0; t < KTA; t++) store_tile(g.d_xa, a.ka, row, ok, 32 * t, hh, gx[t], a.ka);
            }
            if constexpr (L::SPO) {
                const Split gs[2] = {split_tile(gf[0]), split_tile(gf[1])};
#pragma unroll
                for (int t = 0; t < KTA; t++) {
                    const Split as = split_tile(transpose_tile(xa[t], T, r, hh));
                    dWa[0][t] = outer_acc_s(dWa[0][t], gs[0], as);
                    dWa[1][t] = outer_acc_s(dWa[1][t], gs[1], as);
                }
            } else {
#pragma unroll
                for (int t = 0; t < KTA; t++) {
                    const f32x16 af = transpose_tile(xa[t], T, r, hh);
                    dWa[0][t] = outer_acc(dWa[0][t], gf[0], af);
                    dWa[1][t] = outer_acc(dWa[1][t], gf[1], af);
                }
            }
        }
        if (a.kb > 0) {
            // xb as a tile: feature c = 8 (v / 4) + 4 hh + v % 4 < kb <= 8 lives in registers 0..3 (columns >= kb were read from column 0: zeroed here)
            f32x16 xt = zero16();
#pragma unroll
            for (int j = 0; j < 4; j++) xt[j] = xb[j] * ((4 * hh + j < a.kb) ? 1.f : 0.f);
            const f32x16 af = transpose_tile(xt, T, r, hh);
            dWb[0] = outer_acc(dWb[0], gf[0], af);
            dWb[1] = outer_acc(dWb[1], gf[1], af);
            if (g.d_xb) {
                // d xb[c] = sum_k Wb[k][c] gh[k]: one tile whose features c >= kb are zero (the rows of Wb^T beyond kb are zero)
                f32x16 gx1[1] = {zero16()};
                layer_fwd<2, 1>(gx1, gh, lds + L::wbt, WS, 0, r, hh);
                const f32x16 gx = gx1[0];
                if (ok) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int c = 4 * hh + j;
                        if (c < a.kb) g.d_xb[row * (size_t)a.kb + c] = gx[j];
                    }
                }
            }
        }
    }
    float* red = lds + L::scratch;
#pragma unroll
    for (int to = 0; to < 2; to++) {
        if (KTA) {
#pragma unroll
            for (int t = 0; t < KTA; t++) {
                const f32x16 sa = wg_sum16(dWa[to][t], red, wave, lane);
                if (wave == 0) flush_dw(g.d_w ? g.d_w + a.col_a : nullptr, a.ld_w, 32 * to, 32 * t, 64, a.ka, sa, r, hh);
            }
        }
        if (a.kb > 0) {
            const f32x16 sb = wg_sum16(dWb[to], red, wave, lane);
            if (wave == 0) flush_dw(g.d_w ? g.d_w + a.col_b : nullptr, a.ld_w, 32 * to, 0, 64, a.kb, sb, r, hh);
        }
        const float bs = wg_sum1(db[to], red, wave, lane);
        if (wave == 0) flush_db(g.d_b, 32 * to, 64, bs, r);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// trunk backward of a level WITHOUT HexPlane features (ka = 0, kb <= 4) on the vector pipe (round 5).  The MFMA form above pads the [64 x kb]
// weight gradient and the [kb]-wide dL/dxb to 32-column tiles: 64 fp32 MFMAs + three LDS transposes per 32 rows for 2 x 64 x kb x 32 useful
// multiply-adds, 244 us at 2 M rows while the kernel's only real traffic, the heads' dL/dh, is 512 MB.  Here a lane keeps the tile layout of its
// row (32 of the row's 64 features), forms its share of dL/dxb as kb x 32 multiply-adds against broadcast LDS rows (the halves meet in one
// cross-lane add) and keeps dWb[feature][c] += dL/dh[feature] xb[c] and db[feature] += dL/dh[feature] in 32 (kb + 1) per-lane accumulators that
// are summed over the lanes once, at the end: no MFMA, no transpose, bound by the read of dL/dh.
// ---------------------------------------------------------------------------------------------------------------------------------
#ifndef MLP_EMBED_BWD_VALU
#define MLP_EMBED_BWD_VALU 1
#endif
// One wave per SIMD: the 512-entry register file holds the lane's 32 x kb weights (no LDS round trip in the tile loop), the 32 (kb + 1)
// accumulators and the next two tiles' registers (8 KB per wave and tile; the rotation's register copy waits for the nearer one).  Every load of
// the loop is unconditional (rows past the end read row 0 and are zeroed by load_tile's product; xb of such a row meets a zero dL/dh): a guarded load
// makes the compiler wait for ALL outstanding loads -- the prefetches too -- at the join.  MULTI: more than one dL/dh tensor (a two-hidden-layer head
// keeps its own): the others are added inside the loop.
template <int KB, bool MULTI>          // KB = kb = 4
__global__ void __launch_bounds__(MLP_THREADS) MLP_OCC(1) k_mlp_embed_bwd(EmdMlpTrunk a, EmdMlpTrunkGrads g) {
    __shared__ float wt[KB][64];                                  // Wb^T: wt[c][feature]
    __shared__ float red[MLP_WAVES][2][32][KB + 1];
    for (int idx = threadIdx.x; idx < KB * 64; idx += MLP_THREADS) {
        const int c = idx >> 6, k = idx & 63;
        wt[c][k] = a.w[(size_t)k * a.ld_w + a.col_b + c];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const size_t N = (size_t)a.num_points, tiles = (N + 31) / 32;
    float4 wreg[KB][2][4];
#pragma unroll
    for (int c = 0; c < KB; c++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) wreg[c][t][q] = *(const float4*)&wt[c][32 * t + 8 * q + 4 * hh];
    float accw[2][16][KB], accb[2][16];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int v = 0; v < 16; v++) {
            accb[t][v] = 0.f;
#pragma unroll
            for (int c = 0; c < KB; c++) accw[t][v][c] = 0.f;
        }
    const size_t tile0 = (size_t)blockIdx.x * MLP_WAVES + wave, tstep = (size_t)gridDim.x * MLP_WAVES;
    auto load_x = [&](size_t row_) -> float4 { return *(const float4*)(a.xb + (row_ < N ? row_ : 0) * 4); };
    // (measured: three buffers in rotation with the loop unrolled by three -- no register copy of a travelling tile -- spill at 512 registers: 143 -> 358 us)
    f32x16 n1[2], n2[2];
    float4 x1, x2;
    {
        const size_t r1 = tile0 * 32 + r, r2 = (tile0 + tstep) * 32 + r;
        n1[0] = load_tile<true>(g.g_h[0], 64, r1, r1 < N, 0, hh); n1[1] = load_tile<true>(g.g_h[0], 64, r1, r1 < N, 32, hh); x1 = load_x(r1);
        n2[0] = load_tile<true>(g.g_h[0], 64, r2, r2 < N, 0, hh); n2[1] = load_tile<true>(g.g_h[0], 64, r2, r2 < N, 32, hh); x2 = load_x(r2);
    }
    for (size_t tile = tile0; tile < tiles; tile += tstep) {
        const size_t row = tile * 32 + r;
        const bool ok = row < N;
        f32x16 gh[2] = {n1[0], n1[1]};                            // dL/dh (rows past the end: zero)
        const float xb[KB] = {x1.x, x1.y, x1.z, x1.w};
        n1[0] = n2[0]; n1[1] = n2[1]; x1 = x2;
        {
            const size_t nrow = (tile + 2 * tstep) * 32 + r;
            n2[0] = load_tile<true>(g.g_h[0], 64, nrow, nrow < N, 0, hh); n2[1] = load_tile<true>(g.g_h[0], 64, nrow, nrow < N, 32, hh); x2 = load_x(nrow);
        }
        if constexpr (MULTI) {
            for (int k = 1; k < g.num_gh; k++) {
                const f32x16 p0 = load_tile<true>(g.g_h[k], 64, row, ok, 0, hh), p1 = load_tile<true>(g.g_h[k], 64, row, ok, 32, hh);
#pragma unroll
                for (int v = 0; v < 16; v++) { gh[0][v] += p0[v]; gh[1][v] += p1[v]; }
            }
        }
        if (g.d_xb) {
            float d[KB];
#pragma unroll
            for (int c = 0; c < KB; c++) {
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float4 w = wreg[c][t][q];
                        acc = fmaf(w.x, gh[t][4 * q], acc); acc = fmaf(w.y, gh[t][4 * q + 1], acc);
                        acc = fmaf(w.z, gh[t][4 * q + 2], acc); acc = fmaf(w.w, gh[t][4 * q + 3], acc);
                    }
                d[c] = acc + __shfl_xor(acc, 32);
            }
            if (hh == 0 && ok) *(float4*)(g.d_xb + row * 4) = make_float4(d[0], d[1], d[2], d[3]);       // whole 16-byte rows: 512 contiguous bytes per instruction
        }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int v = 0; v < 16; v++) {
                accb[t][v] += gh[t][v];
#pragma unroll
                for (int c = 0; c < KB; c++) accw[t][v][c] = fmaf(gh[t][v], xb[c], accw[t][v][c]);
            }
    }
    // ---- sums over the 32 rows of a lane half (xor shuffles stay inside the half), the waves, then one atomic per element and workgroup
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int v = 0; v < 16; v++) {
            float s[KB + 1];
#pragma unroll
            for (int c = 0; c < KB; c++) s[c] = accw[t][v][c];
            s[KB] = accb[t][v];
#pragma unroll
            for (int c = 0; c <= KB; c++) {
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) s[c] += __shfl_xor(s[c], m);
            }
            if (r == 0) {
#pragma unroll
                for (int c = 0; c <= KB; c++) red[wave][t][8 * (v >> 2) + 4 * hh + (v & 3)][c] = s[c];
            }
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * (KB + 1); idx += MLP_THREADS) {
        const int k = idx / (KB + 1), c = idx - k * (KB + 1);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < MLP_WAVES; w++) s += red[w][k >> 5][k & 31][c];
        if (c < KB) { if (g.d_w) unsafeAtomicAdd(g.d_w + (size_t)k * a.ld_w + a.col_b + c, s); }
        else if (g.d_b) unsafeAtomicAdd(g.d_b + k, s);
    }
}

unsigned mlp_grid(int num_points, int per_cu) {
    const size_t tiles = ((size_t)num_points + 31) / 32;
    const size_t wgs = (tiles + MLP_WAVES - 1) / MLP_WAVES, cap = (size_t)256 * per_cu;
    return (unsigned)(wgs < cap ? (wgs ? wgs : 1) : cap);          // persistent: `per_cu` workgroups per CU (their LDS holds the weights)
}

int check_branch(const EmdMlpBranch* a, const char* who) {
    if (!a) { emd_set_error("%s: null args", who); return EMD_ERR_INVALID; }
    if (a->num_points < 0 || (a->depth != 1 && a->depth != 2) || a->out_dim < 1 || a->out_dim > 64) {
        emd_set_error("%s: bad sizes (depth 1..2, out_dim 1..64)", who); return EMD_ERR_INVALID;
    }
    if (a->num_points == 0) return EMD_OK;
    if ((!a->h && !a->xb) || !a->w_hidden[0] || !a->b_hidden[0] || !a->w_out || !a->b_out || (a->depth == 2 && (!a->w_hidden[1] || !a->b_hidden[1]))) {
        emd_set_error("%s: null pointer", who); return EMD_ERR_INVALID;
    }
    if (a->xb) {          // the head recomputes h from the embedding (ABI 26)
        if (a->depth != 1) { emd_set_error("%s: xb (recomputed h) is served for depth 1 only", who); return EMD_ERR_INVALID; }
        if (!a->w_in || !a->b_in || a->kb_in < 1 || a->kb_in > 8 || a->col_in < 0 || a->ld_w_in < a->col_in + a->kb_in) {
            emd_set_error("%s: xb needs w_in, b_in, kb_in 1..8 and col_in + kb_in <= ld_w_in", who); return EMD_ERR_INVALID;
        }
    } else if (((uintptr_t)a->h & 15)) { emd_set_error("%s: h must be 16-byte aligned", who); return EMD_ERR_INVALID; }
    return EMD_OK;
}

int check_trunk(const EmdMlpTrunk* a, const char* who, bool need_h = true) {
    if (!a) { emd_set_error("%s: null args", who); return EMD_ERR_INVALID; }
    if (a->num_points < 0 || (a->ka < 0 || a->ka > 128 || (a->ka & 3)) || a->kb < 0 || a->kb > 8 || a->ka + a->kb == 0 || a->ld_w < a->ka + a->kb ||
        a->col_a < 0 || a->col_b < 0 || a->col_a + a->ka > a->ld_w || a->col_b + a->kb > a->ld_w) {
        emd_set_error("%s: bad sizes (ka a multiple of 4 up to 128, kb <= 8)", who); return EMD_ERR_INVALID;
    }
    if (a->num_points == 0) return EMD_OK;
    if (!a->w || !a->b || (need_h && !a->h) || (a->ka && !a->xa) || (a->kb && !a->xb)) { emd_set_error("%s: null pointer", who); return EMD_ERR_INVALID; }
    if ((need_h && ((uintptr_t)a->h & 15)) || (a->ka && ((uintptr_t)a->xa & 15))) { emd_set_error("%s: xa / h must be 16-byte aligned", who); return EMD_ERR_INVALID; }
    return EMD_OK;
}

template <auto kernel, int PER_CU, typename... Args>
int mlp_launch(int floats, int num_points, hipStream_t st, Args... args) {
    const size_t bytes = (size_t)floats * sizeof(float);
    if (bytes > 64 * 1024) {
        // once per kernel (this function template is instantiated per kernel: the kernel is its template argument): not a stream operation, so it is kept
        // out of the per-launch path and out of any stream capture after the first call
        // (the attribute is per DEVICE: one flag per device ordinal; an unsynchronised double set by two threads is harmless)
        static std::atomic<bool> raised[64];
        int dev = 0;
        EMD_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !raised[dev].load(std::memory_order_acquire)) {
            EMD_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            if (dev >= 0 && dev < 64) raised[dev].store(true, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(kernel, dim3(mlp_grid(num_points, PER_CU)), dim3(MLP_THREADS), bytes, st, args...);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

}  // namespace

extern "C" int emd_mlp_branch_forward(const EmdMlpBranch* a, void* hip_stream) {
    int rc = check_branch(a, "mlp_branch_forward");
    if (rc || a->num_points == 0) return rc;
    if (!a->out) { emd_set_error("mlp_branch_forward: null output"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    const int nto = a->out_dim > 32 ? 2 : 1;
    if (a->xb) {
        typedef BranchLds<1, 1, branch_mode(1, 1, false)> L1_; typedef BranchLds<1, 2, branch_mode(1, 2, false)> L2_;
        if (a->l1_sum) {
            if (nto == 1) return mlp_launch<k_mlp_branch_fwd<1, 1, true, true>, (MLP_L1_F11_TWO ? MLP_W_F11 : MLP_FWD_WAVES)>(L1_::fwd_floats + L1_::rc_floats, a->num_points, st, *a);
            return mlp_launch<k_mlp_branch_fwd<1, 2, true, true>, MLP_FWD_WAVES>(L2_::fwd_floats + L2_::rc_floats, a->num_points, st, *a);
        }
        if (nto == 1) return mlp_launch<k_mlp_branch_fwd<1, 1, false, true>, MLP_W_F11>(L1_::fwd_floats + L1_::rc_floats, a->num_points, st, *a);
        return mlp_launch<k_mlp_branch_fwd<1, 2, false, true>, MLP_FWD_WAVES>(L2_::fwd_floats + L2_::rc_floats, a->num_points, st, *a);
    }
    if (a->l1_sum) {
        if (a->depth == 1 && nto == 1) return mlp_launch<k_mlp_branch_fwd<1, 1, true>, (MLP_L1_F11_TWO ? MLP_W_F11 : MLP_FWD_WAVES)>(BranchLds<1, 1, branch_mode(1, 1, false)>::fwd_floats, a->num_points, st, *a);
        if (a->depth == 1) return mlp_launch<k_mlp_branch_fwd<1, 2, true>, MLP_FWD_WAVES>(BranchLds<1, 2, branch_mode(1, 2, false)>::fwd_floats, a->num_points, st, *a);
        if (nto == 1) return mlp_launch<k_mlp_branch_fwd<2, 1, true>, MLP_FWD_WAVES>(BranchLds<2, 1, branch_mode(2, 1, false)>::fwd_floats, a->num_points, st, *a);
        return mlp_launch<k_mlp_branch_fwd<2, 2, true>, MLP_FWD_WAVES>(BranchLds<2, 2, branch_mode(2, 2, false)>::fwd_floats, a->num_points, st, *a);
    }
    if (a->depth == 1 && nto == 1) return mlp_launch<k_mlp_branch_fwd<1, 1>, MLP_W_F11>(BranchLds<1, 1, branch_mode(1, 1, false)>::fwd_floats, a->num_points, st, *a);
    if (a->depth == 1) return mlp_launch<k_mlp_branch_fwd<1, 2>, MLP_FWD_WAVES>(BranchLds<1, 2, branch_mode(1, 2, false)>::fwd_floats, a->num_points, st, *a);
    if (nto == 1) return mlp_launch<k_mlp_branch_fwd<2, 1>, MLP_FWD_WAVES>(BranchLds<2, 1, branch_mode(2, 1, false)>::fwd_floats, a->num_points, st, *a);
    return mlp_launch<k_mlp_branch_fwd<2, 2>, MLP_FWD_WAVES>(BranchLds<2, 2, branch_mode(2, 2, false)>::fwd_floats, a->num_points, st, *a);
}

template <int DEPTH, int NTO, bool L1, bool RC, int GO, bool CHAIN = false>
int launch_branch_bwd(const EmdMlpBranch* a, const EmdMlpBranchGrads* g, hipStream_t st) {
    typedef BranchLds<DEPTH, NTO, branch_mode(DEPTH, NTO, true)> L;
    return mlp_launch<k_mlp_branch_bwd<DEPTH, NTO, L1, RC, GO, CHAIN>, MLP_BWD_WAVES>(L::bwd_floats + (RC ? L::rc_floats : 0), a->num_points, st, *a, *g);
}
// one-hidden-layer heads: by output tiles, regulariser, recomputed h, the form dL/dout is loaded in and the chained dL/dh (k_mlp_branch_bwd's GO / CHAIN)
template <int NTO, bool L1, bool RC>
int launch_branch_bwd1(const EmdMlpBranch* a, const EmdMlpBranchGrads* g, hipStream_t st) {
    const bool chain = g->g_h_in != nullptr;
    if (g->g_out) {
        if constexpr (NTO == 1) {
            if (a->out_dim <= 4) return chain ? launch_branch_bwd<1, 1, L1, RC, 0, true>(a, g, st) : launch_branch_bwd<1, 1, L1, RC, 0, false>(a, g, st);
        }
        if ((a->out_dim & 3) == 0) return chain ? launch_branch_bwd<1, NTO, L1, RC, 1, true>(a, g, st) : launch_branch_bwd<1, NTO, L1, RC, 1, false>(a, g, st);
    }
    return launch_branch_bwd<1, NTO, L1, RC, 2>(a, g, st);
}

extern "C" int emd_mlp_branch_backward(const EmdMlpBranch* a, const EmdMlpBranchGrads* g, void* hip_stream) {
    int rc = check_branch(a, "mlp_branch_backward");
    if (rc || a->num_points == 0) return rc;
    if (!g || (!g->g_out && !g->l1_grad) || !g->g_h) { emd_set_error("mlp_branch_backward: null gradient pointer"); return EMD_ERR_INVALID; }
    if (g->g_h_in && a->depth != 1) { emd_set_error("mlp_branch_backward: g_h_in is served for depth 1 only"); return EMD_ERR_INVALID; }
    if (g->g_h_in && ((uintptr_t)g->g_h_in & 15)) { emd_set_error("mlp_branch_backward: g_h_in must be 16-byte aligned"); return EMD_ERR_INVALID; }
    if (g->l1_grad && !g->out) { emd_set_error("mlp_branch_backward: l1_grad needs the forward's output tensor"); return EMD_ERR_INVALID; }
    if (((uintptr_t)g->g_h & 15)) { emd_set_error("mlp_branch_backward: g_h must be 16-byte aligned"); return EMD_ERR_INVALID; }
    if (!(a->out_dim & 3) && (((uintptr_t)g->g_out & 15) || ((uintptr_t)g->out & 15))) {
        emd_set_error("mlp_branch_backward: g_out / out must be 16-byte aligned when out_dim is a multiple of 4"); return EMD_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int nto = a->out_dim > 32 ? 2 : 1;
    const bool l1 = g->l1_grad != nullptr, rcm = a->xb != nullptr;
    if (a->depth == 1) {
        if (nto == 1) {
            if (l1) return rcm ? launch_branch_bwd1<1, true, true>(a, g, st) : launch_branch_bwd1<1, true, false>(a, g, st);
            return rcm ? launch_branch_bwd1<1, false, true>(a, g, st) : launch_branch_bwd1<1, false, false>(a, g, st);
        }
        if (l1) return rcm ? launch_branch_bwd1<2, true, true>(a, g, st) : launch_branch_bwd1<2, true, false>(a, g, st);
        return rcm ? launch_branch_bwd1<2, false, true>(a, g, st) : launch_branch_bwd1<2, false, false>(a, g, st);
    }
    // two hidden layers (the feature head): the raw-load forms where dL/dout allows them (no chained dL/dh: refused above)
    if (g->g_out && nto == 1 && a->out_dim <= 4)
        return l1 ? launch_branch_bwd<2, 1, true, false, 0>(a, g, st) : launch_branch_bwd<2, 1, false, false, 0>(a, g, st);
    if (g->g_out && (a->out_dim & 3) == 0) {
        if (l1) return nto == 1 ? launch_branch_bwd<2, 1, true, false, 1>(a, g, st) : launch_branch_bwd<2, 2, true, false, 1>(a, g, st);
        return nto == 1 ? launch_branch_bwd<2, 1, false, false, 1>(a, g, st) : launch_branch_bwd<2, 2, false, false, 1>(a, g, st);
    }
    if (l1) return nto == 1 ? launch_branch_bwd<2, 1, true, false, 2>(a, g, st) : launch_branch_bwd<2, 2, true, false, 2>(a, g, st);
    return nto == 1 ? launch_branch_bwd<2, 1, false, false, 2>(a, g, st) : launch_branch_bwd<2, 2, false, false, 2>(a, g, st);
}

extern "C" int emd_mlp_trunk_forward(const EmdMlpTrunk* a, void* hip_stream) {
    int rc = check_trunk(a, "mlp_trunk_forward");
    if (rc || a->num_points == 0) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    switch ((a->ka + 31) / 32) {            // input tiles of 32 columns (the last one zero-padded)
        case 0: return mlp_launch<k_mlp_trunk_fwd<0>, MLP_W_T0>(TrunkLds<0, MLP_SP_TF>::fwd_floats, a->num_points, st, *a);
        case 1: return mlp_launch<k_mlp_trunk_fwd<1>, MLP_FWD_WAVES>(TrunkLds<1, MLP_SP_TF>::fwd_floats, a->num_points, st, *a);
        case 2: return mlp_launch<k_mlp_trunk_fwd<2>, MLP_FWD_WAVES>(TrunkLds<2, MLP_SP_TF>::fwd_floats, a->num_points, st, *a);
        case 3: return mlp_launch<k_mlp_trunk_fwd<3>, MLP_FWD_WAVES>(TrunkLds<3, MLP_SP_TF>::fwd_floats, a->num_points, st, *a);
        default: return mlp_launch<k_mlp_trunk_fwd<4>, MLP_FWD_WAVES>(TrunkLds<4, MLP_SP_TF>::fwd_floats, a->num_points, st, *a);
    }
}

template <int KTA, int PER_CU>
int launch_trunk_bwd(const EmdMlpTrunk* a, const EmdMlpTrunkGrads* g, hipStream_t st) {
    if (g->num_gh == 1) return mlp_launch<k_mlp_trunk_bwd<KTA, false>, PER_CU>(TrunkLds<KTA, MLP_SP_TB>::bwd_floats, a->num_points, st, *a, *g);
    return mlp_launch<k_mlp_trunk_bwd<KTA, true>, PER_CU>(TrunkLds<KTA, MLP_SP_TB>::bwd_floats, a->num_points, st, *a, *g);
}

extern "C" int emd_mlp_trunk_backward(const EmdMlpTrunk* a, const EmdMlpTrunkGrads* g, void* hip_stream) {
    int rc = check_trunk(a, "mlp_trunk_backward", false);          // (the backward reads xa, xb and the heads' dL/dh, not h)
    if (rc || a->num_points == 0) return rc;
    if (!g || g->num_gh < 1 || g->num_gh > EMD_MLP_MAX_BRANCHES) { emd_set_error("mlp_trunk_backward: 1..%d gradient contributions", EMD_MLP_MAX_BRANCHES); return EMD_ERR_INVALID; }
    for (int k = 0; k < g->num_gh; k++)
        if (!g->g_h[k] || ((uintptr_t)g->g_h[k] & 15)) { emd_set_error("mlp_trunk_backward: g_h[%d] null or unaligned", k); return EMD_ERR_INVALID; }
    if (g->d_xa && ((uintptr_t)g->d_xa & 15)) { emd_set_error("mlp_trunk_backward: d_xa must be 16-byte aligned"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    switch ((a->ka + 31) / 32) {
        case 0:
            if (MLP_EMBED_BWD_VALU && a->kb == 4 && !((uintptr_t)a->xb & 15) && !((uintptr_t)g->d_xb & 15)) {     // (the reference's embedding width; static LDS)
                if (g->num_gh == 1) hipLaunchKernelGGL((k_mlp_embed_bwd<4, false>), dim3(mlp_grid(a->num_points, 1)), dim3(MLP_THREADS), 0, st, *a, *g);
                else hipLaunchKernelGGL((k_mlp_embed_bwd<4, true>), dim3(mlp_grid(a->num_points, 1)), dim3(MLP_THREADS), 0, st, *a, *g);
                EMD_LAUNCH_CHECK();
                return EMD_OK;
            }
            return launch_trunk_bwd<0, MLP_W_TB0>(a, g, st);
        case 1: return launch_trunk_bwd<1, MLP_BWD_WAVES>(a, g, st);
        case 2: return launch_trunk_bwd<2, MLP_BWD_WAVES>(a, g, st);
        case 3: return launch_trunk_bwd<3, MLP_BWD_WAVES>(a, g, st);
        default: return launch_trunk_bwd<4, MLP_BWD_WAVES>(a, g, st);
    }
}
