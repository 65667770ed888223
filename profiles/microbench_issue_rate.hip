// microbench_issue_rate.hip -- measured issue ceilings of one gfx950 SIMD / CU for the instruction mix of the render kernels.
//
// Settles which ceiling K6 / K7 are priced against: cycles per wave64 instruction per SIMD for
// v_fma_f32, v_pk_fma_f32, v_add_f32_dpp, v_exp_f32, v_rcp_f32, v_cndmask, v_readlane, and LDS bytes per clock per CU
// for the broadcast ds_read_b128 the compositing loops use, each at 1 / 2 / 4 / 8 resident waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench_issue_rate.hip -o /tmp/mb && /tmp/mb
//
// Method: ONE workgroup per CU (forced by a 96 KB dynamic LDS allocation; two 64 KB workgroups per CU for w = 8), 256 w threads per
// workgroup = w waves on every SIMD, all resident for the whole kernel.  Every wave runs ITERS x 32 instructions of one kind
// (8 independent register chains) between two stamps of s_memrealtime (100 MHz wall clock) and s_memtime (shader clock).
// Reported per (instruction, w):  ns per wave-instruction per SIMD = (latest end - earliest start on the CU) / (ITERS * 32 * w);
// the shader clock actually held = d s_memtime / d s_memrealtime x 100 MHz;  cycles = ns x clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define ITERS 4096
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define BODY32(op) REP8(op) REP8(op) REP8(op) REP8(op)

enum { K_FMA = 0, K_PKFMA, K_ADD_DPP, K_MUL_DPP_CHAIN, K_EXP, K_RCP, K_CNDMASK, K_READLANE, K_LDS_B128_BCAST, K_LDS_B128_LANE,
       K_LDS_B32_BCAST, K_FMA_PLUS_LDS, K_PKMUL, K_MOV, K_COUNT };
const char* kNames[K_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32_dpp row_shr:1 (independent)", "v_mul_f32_dpp 6-step scan chain (dependent)",
                               "v_exp_f32", "v_rcp_f32", "v_cndmask_b32", "v_readlane_b32", "ds_read_b128 broadcast", "ds_read_b128 lane-consecutive",
                               "ds_read_b32 broadcast", "3 v_fma_f32 + 1 ds_read_b128 broadcast", "v_pk_mul_f32", "v_mov_b32"};

template <int KIND>
__global__ void __launch_bounds__(1024) k_bench(unsigned long long* __restrict__ dt, float* __restrict__ sink) {
    extern __shared__ float4 lds[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    float a[8];
    v2f p[8];
    for (int k = 0; k < 8; k++) { a[k] = 1.0f + 0.001f * (threadIdx.x + k); p[k] = (v2f){a[k], a[k] * 0.5f}; }
    const float m = 0.99999f, c = 1e-7f;
    const v2f m2 = (v2f){m, m}, c2 = (v2f){c, c};
    float4 acc4 = make_float4(0, 0, 0, 0);
    const unsigned baddr = 16u * (blockIdx.x & 63u);                // wave-uniform address (broadcast)
    const unsigned laddr = 16u * (threadIdx.x & 63u);               // lane-consecutive 16 B
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if (KIND == K_FMA) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            BODY32(OP)
#undef OP
        } else if (KIND == K_PKFMA) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
            BODY32(OP)
#undef OP
        } else if (KIND == K_PKMUL) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            BODY32(OP)
#undef OP
        } else if (KIND == K_ADD_DPP) {
#define OP(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
            BODY32(OP)
#undef OP
        } else if (KIND == K_MUL_DPP_CHAIN) {
            // the product scan of K7: six dependent DPP steps on ONE register (x 5 + 2 filler = 32 instructions per iteration)
#define STEP(ctrl) asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 " ctrl : "+v"(a[0]));
#define SCAN STEP("row_shr:1 row_mask:0xf bank_mask:0xf") STEP("row_shr:2 row_mask:0xf bank_mask:0xf") STEP("row_shr:4 row_mask:0xf bank_mask:0xf") \
             STEP("row_shr:8 row_mask:0xf bank_mask:0xf") STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
            SCAN SCAN SCAN SCAN SCAN
            asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %0, %1" : "+v"(a[0]) : "v"(m));
#undef SCAN
#undef STEP
        } else if (KIND == K_EXP) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            BODY32(OP)
#undef OP
        } else if (KIND == K_RCP) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            BODY32(OP)
#undef OP
        } else if (KIND == K_CNDMASK) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            BODY32(OP)
#undef OP
        } else if (KIND == K_MOV) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
            BODY32(OP)
#undef OP
        } else if (KIND == K_READLANE) {
            int s;
#define OP(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a[i])); asm volatile("" :: "s"(s));
            BODY32(OP)
#undef OP
        } else if (KIND == K_LDS_B128_BCAST || KIND == K_LDS_B128_LANE) {
            // one asm statement per 8 reads (distinct destinations, no compiler-inserted hazard nops inside), drained once per 32
            const unsigned ad = KIND == K_LDS_B128_BCAST ? baddr : laddr;
            float4 r0, r1, r2, r3, r4, r5, r6, r7;
#define LDS8(rd, tail) asm volatile(rd " %0, %8 offset:0\n\t" rd " %1, %8 offset:1024\n\t" rd " %2, %8 offset:2048\n\t" rd " %3, %8 offset:3072\n\t" \
                              rd " %4, %8 offset:4096\n\t" rd " %5, %8 offset:5120\n\t" rd " %6, %8 offset:6144\n\t" rd " %7, %8 offset:7168" tail \
                              : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(ad))
            LDS8("ds_read_b128", ""); LDS8("ds_read_b128", ""); LDS8("ds_read_b128", ""); LDS8("ds_read_b128", "\n\ts_waitcnt lgkmcnt(0)");
            acc4.x += r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
        } else if (KIND == K_LDS_B32_BCAST) {
            float r0, r1, r2, r3, r4, r5, r6, r7;
            const unsigned ad = baddr;
            LDS8("ds_read_b32", ""); LDS8("ds_read_b32", ""); LDS8("ds_read_b32", ""); LDS8("ds_read_b32", "\n\ts_waitcnt lgkmcnt(0)");
            acc4.x += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
        } else if (KIND == K_FMA_PLUS_LDS) {
            float4 r0, r1, r2, r3;
#define MIX(rr, i, off, tail) asm volatile("ds_read_b128 %0, %4 offset:" off "\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" tail \
                                     : "=&v"(rr), "+v"(a[i]) : "v"(m), "v"(c), "v"(baddr))
            MIX(r0, 0, "0", ""); MIX(r1, 1, "1024", ""); MIX(r2, 2, "2048", ""); MIX(r3, 3, "3072", "\n\ts_waitcnt lgkmcnt(0)");
            acc4.x += r0.x + r1.x + r2.x + r3.x;
            MIX(r0, 4, "4096", ""); MIX(r1, 5, "5120", ""); MIX(r2, 6, "6144", ""); MIX(r3, 7, "7168", "\n\ts_waitcnt lgkmcnt(0)");
            acc4.x += r0.x + r1.x + r2.x + r3.x;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = acc4.x;
    for (int k = 0; k < 8; k++) s += a[k] + p[k].x + p[k].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = dt + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 4;
        o[0] = r0; o[1] = r1; o[2] = t1 - t0; o[3] = 0;
    }
}

template <int KIND>
void run(int w, unsigned long long* d_dt, float* d_sink) {
    // w <= 4: one workgroup of 256 w threads per CU (96 KB LDS keeps a second one out); w = 8: two workgroups of 1024 (64 KB each)
    const int per_cu = w == 8 ? 2 : 1, threads = w == 8 ? 1024 : 256 * w, blocks = 256 * per_cu;
    const size_t lds = w == 8 ? 64 * 1024 : 96 * 1024;
    hipFuncSetAttribute((const void*)k_bench<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipMemset(d_dt, 0, 512 * 16 * 4 * 8);
    hipLaunchKernelGGL(k_bench<KIND>, dim3(blocks), dim3(threads), lds, 0, d_dt, d_sink);   // warm-up
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_bench<KIND>, dim3(blocks), dim3(threads), lds, 0, d_dt, d_sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)blocks * 16 * 4);
    hipMemcpy(h.data(), d_dt, h.size() * 8, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    // the chip as a whole: earliest start to latest end (all workgroups are resident together: 256 or 512 <= capacity)
    unsigned long long rmin = ~0ull, rmax = 0;
    std::vector<double> clk;
    for (int b = 0; b < blocks; b++)
        for (int v = 0; v < waves; v++) {
            const unsigned long long* o = &h[((size_t)b * 16 + v) * 4];
            rmin = std::min(rmin, o[0]); rmax = std::max(rmax, o[1]);
            clk.push_back((double)o[2] / (double)(o[1] - o[0]) * 0.1);       // GHz: shader ticks per 10 ns tick
        }
    std::sort(clk.begin(), clk.end());
    const double ghz = clk[clk.size() / 2];
    const double n_inst = (double)ITERS * 32.0;
    const double ns = (double)(rmax - rmin) * 10.0 / (n_inst * w);               // per wave-instruction per SIMD
    const double cyc = ns * ghz;
    printf("%-46s w=%d  ns/wave-instr/SIMD=%6.3f  shader_clock_GHz=%5.2f  cycles/wave-instr/SIMD=%6.3f", kNames[KIND], w, ns, ghz, cyc);
    if (KIND == K_LDS_B128_BCAST || KIND == K_LDS_B128_LANE) printf("  LDS B/clk/CU %.1f", 64.0 * 16.0 * 4.0 / cyc);
    if (KIND == K_LDS_B32_BCAST) printf("  LDS B/clk/CU %.1f", 64.0 * 4.0 * 4.0 / cyc);
    if (KIND == K_FMA_PLUS_LDS) printf("  LDS B/clk/CU %.1f (1 read per 4 instructions)", 64.0 * 16.0 * 4.0 / (cyc * 4.0));
    printf("\n");
}

template <int KIND>
void sweep(unsigned long long* d_dt, float* d_sink) {
    const int ws[4] = {1, 2, 4, 8};
    for (int i = 0; i < 4; i++) run<KIND>(ws[i], d_dt, d_sink);
}

int main() {
    unsigned long long* d_dt;
    float* d_sink;
    hipMalloc(&d_dt, 512 * 16 * 4 * 8);
    hipMalloc(&d_sink, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("# device %s, %d CUs, clock %d kHz; w = resident waves per SIMD (256-thread workgroups, w per CU)\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate);
    sweep<K_FMA>(d_dt, d_sink);
    sweep<K_PKFMA>(d_dt, d_sink);
    sweep<K_PKMUL>(d_dt, d_sink);
    sweep<K_MOV>(d_dt, d_sink);
    sweep<K_ADD_DPP>(d_dt, d_sink);
    sweep<K_MUL_DPP_CHAIN>(d_dt, d_sink);
    sweep<K_EXP>(d_dt, d_sink);
    sweep<K_RCP>(d_dt, d_sink);
    sweep<K_CNDMASK>(d_dt, d_sink);
    sweep<K_READLANE>(d_dt, d_sink);
    sweep<K_LDS_B128_BCAST>(d_dt, d_sink);
    sweep<K_LDS_B128_LANE>(d_dt, d_sink);
    sweep<K_LDS_B32_BCAST>(d_dt, d_sink);
    sweep<K_FMA_PLUS_LDS>(d_dt, d_sink);
    return 0;
}
