// microbench_k7_mfma.hip -- VERDICT r2 item 2: can the ten per-entry moment sums of K7 (k_render_backward_q) leave the vector pipe
// for the matrix pipe?
//
// Per pixel-pair iteration K7 (lane = list entry) forms u = G dL/dG and w = alpha T for two pixels and adds
//     m0 += u, gx += u tx, gy += u ty, m2xx += u dx dx, m2xy += u dx dy, m2yy += u dy dy       (six sums of u times a pixel-entry term)
//     dz += w dD, r += w dC0, g += w dC1, b += w dC2                                            (four sums of w times a PIXEL constant)
// as 16 packed fp32 instructions (render.hip).  The matrix formulation: out[entry][feature] += A[entry][k] B[k][feature] with
// k = pixel, v_mfma_f32_32x32x2_f32 (M = 32 entries, N = 32 features, K = 2 pixels): v_permlane32_swap(x_a, x_b) turns the two
// per-pixel registers of a quantity into the A operands of entries 0-31 and 32-63; B is one ds_read_b32 per lane of a per-pixel feature
// row.  Only sums whose pixel factor does not depend on the entry fit: the four w sums do; the six u sums need raw pixel-space
// moments (u, u px, u py, u px^2, u px py, u py^2) re-centred per entry afterwards (the cancellation DESIGN.md section 3 describes for
// gx / gy).  Counting generously -- all ten through the matrix pipe -- one iteration needs 2 swaps + 2 LDS reads + 4 MFMA (u and w
// products share accumulators through disjoint feature columns: two 16-register accumulators).
//
// This file times exactly those two forms of the accumulation block, alone and next to the rest of a K7 iteration (a filler of
// the same size and mix: 26 DPP + 21 packed + 43 plain VALU), at 4 waves per SIMD on every CU -- K7's occupancy with 114 VGPRs; the
// matrix form needs 32 more registers, i.e. 3 waves per SIMD in the real kernel (not modelled here: the bench is generous to it).
//
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench_k7_mfma.hip -o /tmp/mbk7 && /tmp/mbk7
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define ITERS 2048
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

enum { FORM_VALU = 0, FORM_MFMA = 1 };

// the rest of a K7 iteration, as issue load only: 26 DPP, 21 packed, 43 plain instructions on private registers
__device__ __forceinline__ void filler(float (&a)[8], v2f (&p)[8], float m, v2f m2) {
#pragma unroll
    for (int r = 0; r < 13; r++) {
        asm volatile("s_nop 0\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf"
                     : "+v"(a[r & 3]), "+v"(a[4 + (r & 3)]));
    }
#pragma unroll
    for (int r = 0; r < 21; r++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[r & 7]) : "v"(m2));
#pragma unroll
    for (int r = 0; r < 43; r++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[r & 7]) : "v"(m));
}

template <int FORM, bool WITH_FILLER>
__global__ void __launch_bounds__(1024) k_bench(unsigned long long* __restrict__ dt, float* __restrict__ sink) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1e-3f * (float)(i & 63);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    float a[8];
    v2f p[8];
    for (int k = 0; k < 8; k++) { a[k] = 1.0f + 1e-3f * (float)(lane + k); p[k] = (v2f){a[k], 0.5f * a[k]}; }
    const float m = 0.9999f;
    const v2f m2 = (v2f){m, m};
    // inputs of the accumulation block (kept live, perturbed every iteration so nothing folds)
    v2f u = (v2f){1e-3f * lane, 2e-3f * lane}, w = (v2f){3e-3f * lane, 1e-3f}, dx = (v2f){0.5f, -0.5f}, tx = (v2f){0.1f, 0.2f}, ty = (v2f){0.3f, 0.1f};
    float dy = 0.25f;
    const v2f c0 = (v2f){0.1f, 0.2f}, c1 = (v2f){0.3f, 0.4f}, c2 = (v2f){0.5f, 0.6f}, cd = (v2f){0.7f, 0.8f};
    v2f m0 = {0, 0}, gx = {0, 0}, gy = {0, 0}, mxx = {0, 0}, mxy = {0, 0}, myy = {0, 0}, dz = {0, 0}, rr = {0, 0}, gg = {0, 0}, bb = {0, 0};
    v16f acc1 = {0}, acc2 = {0};
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
        if (WITH_FILLER) filler(a, p, m, m2);
        if (FORM == FORM_VALU) {
            // the 16 packed instructions of render.hip's accumulation block
            const v2f ux = u * dx, uy = u * (v2f){dy, dy};
            m0 += u;
            gx = __builtin_elementwise_fma(u, tx, gx); gy = __builtin_elementwise_fma(u, ty, gy);
            mxx = __builtin_elementwise_fma(ux, dx, mxx); mxy = __builtin_elementwise_fma(ux, (v2f){dy, dy}, mxy);
            myy = __builtin_elementwise_fma(uy, (v2f){dy, dy}, myy);
            dz = __builtin_elementwise_fma(w, cd, dz); rr = __builtin_elementwise_fma(w, c0, rr);
            gg = __builtin_elementwise_fma(w, c1, gg); bb = __builtin_elementwise_fma(w, c2, bb);
            asm volatile("" : "+v"(m0), "+v"(gx), "+v"(gy), "+v"(mxx), "+v"(mxy), "+v"(myy));
            asm volatile("" : "+v"(dz), "+v"(rr), "+v"(gg), "+v"(bb));
        } else {
            // A operands by two swaps, B rows from LDS (pixel a features in lanes 0-31, pixel b in 32-63), four MFMA on two accumulators
            float ua = u.x, ub = u.y, wa = w.x, wb = w.y;
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ua), "+v"(ub));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(wa), "+v"(wb));
            const float bu = lds[(it & 31) * 64 + lane], bw = lds[2048 + (it & 31) * 64 + lane];
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ua, bu, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ub, bu, acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, bw, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb, bw, acc2, 0, 0, 0);
        }
        // next iteration's inputs (4 cheap ops, the same in both forms)
        u = u * m2; w = w * m2;
        asm volatile("" : "+v"(u), "+v"(w));
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + p[0].x + p[1].y + p[2].x + p[3].y + p[4].x + p[5].y + p[6].x + p[7].y;
    s += m0.x + gx.y + gy.x + mxx.y + mxy.x + myy.y + dz.x + rr.y + gg.x + bb.y;
    for (int k = 0; k < 16; k++) s += acc1[k] + acc2[k];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) { dt[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = r0; dt[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = r1; }
}

template <int FORM, bool WITH_FILLER>
double run(int waves_per_simd) {
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int threads = 256 * waves_per_simd;            // one workgroup per CU (96 KB of LDS): waves_per_simd waves on every SIMD
    const int waves = cus * threads / 64;
    unsigned long long* dt;
    float* sink;
    hipMalloc(&dt, sizeof(unsigned long long) * 2 * waves);
    hipMalloc(&sink, sizeof(float) * cus * threads);
    hipFuncSetAttribute((const void*)k_bench<FORM, WITH_FILLER>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 3; rep++) k_bench<FORM, WITH_FILLER><<<cus, threads, 96 * 1024>>>(dt, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), dt, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
    // per CU: (latest end - earliest start) / iterations / waves per SIMD = ns of SIMD time per wave-iteration (100 MHz realtime clock)
    std::vector<double> per_cu;
    const int wpc = threads / 64;
    for (int c = 0; c < cus; c++) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < wpc; w++) { lo = std::min(lo, h[2 * (c * wpc + w)]); hi = std::max(hi, h[2 * (c * wpc + w) + 1]); }
        per_cu.push_back((double)(hi - lo) * 10.0 / ITERS / waves_per_simd);
    }
    std::sort(per_cu.begin(), per_cu.end());
    hipFree(dt); hipFree(sink);
    return per_cu[per_cu.size() / 2];
}

int main() {
    printf("# ns of SIMD time per wave-iteration (median over CUs), one workgroup per CU, ITERS = %d\n", ITERS);
    printf("# block = the ten moment sums of one K7 pixel-pair iteration; filler = the other 90 vector instructions of that iteration\n");
    for (int w : {3, 4}) {
        const double v0 = run<FORM_VALU, false>(w), m0 = run<FORM_MFMA, false>(w), v1 = run<FORM_VALU, true>(w), m1 = run<FORM_MFMA, true>(w);
        printf("waves/SIMD %d:  block alone  valu %.1f ns  mfma %.1f ns   |  block + filler  valu %.1f ns  mfma %.1f ns\n", w, v0, m0, v1, m1);
    }
    return 0;
}
