#include <hip/hip_runtime.h>
#include <stdio.h>
// Sustained rate of v_mfma_f32_32x32x2_f32 (the fused MLP kernels' instruction) with nothing else going on: NCHAIN independent
// accumulator chains per wave, WAVES waves per SIMD, 256 CUs busy.  Nominal: 16 passes = 64 clocks per instruction and SIMD,
// 256 FLOP/clk/CU -> 157 TFLOP/s at 2.4 GHz.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NCHAIN, bool TOGGLE = false>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    f32x16 acc[NCHAIN];
    for (int c = 0; c < NCHAIN; c++) for (int v = 0; v < 16; v++) acc[c][v] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; it++) {
        if (TOGGLE) { a = __uint_as_float((__float_as_uint(a) * 1664525u + 1013904223u) & 0x3fffffffu | 0x30000000u); b = __uint_as_float((__float_as_uint(b) * 22695477u + 1u) & 0x3fffffffu | 0x30000000u); }
#pragma unroll
        for (int c = 0; c < NCHAIN; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < NCHAIN; c++) for (int v = 0; v < 16; v++) s += acc[c][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NCHAIN, bool TOGGLE = false>
void run(float* out, int blocks_per_cu, int iters = 20000, int reps = 3) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    for (int rep = 0; rep < reps; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NCHAIN, TOGGLE>), dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * iters * NCHAIN, flop = mfma * 4096.0;
    printf("chains %d  waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s   %6.1f clocks per MFMA and SIMD at 2.4 GHz (i.e. %.2f GHz if it takes 64)\n", NCHAIN, blocks_per_cu, ms,
           flop / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * NCHAIN * blocks_per_cu), 64.0 * iters * NCHAIN * blocks_per_cu / (ms * 1e-3) / 1e9);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<1>(out, 1); run<2>(out, 1); run<4>(out, 1); run<8>(out, 1);
    run<1>(out, 2); run<4>(out, 2); run<4>(out, 4);
    // sustained: the same instruction stream for ~0.1 s and ~1 s per launch (the clocks of a loaded chip settle below the boost clock)
    printf("sustained:\n");
    run<4>(out, 1, 1000000, 2); run<4>(out, 1, 10000000, 1); run<4>(out, 1, 20000, 3);
    printf("operands that change every iteration (random mantissas, exponents 2^-31 .. 2^0):\n");
    run<4, true>(out, 1, 20000, 3); run<4, true>(out, 1, 1000000, 2); run<4, true>(out, 1, 5000000, 1);
    return 0;
}
