#include <hip/hip_runtime.h>
#include <stdio.h>
// Round 3: which LDS accumulation primitive should the HexPlane backward's windows use?  Same access pattern as
// microbench_lds_atomic.hip (conflict-free 32-lane rows, 1024 threads), more spellings:
//   0 CAS-loop float add (the r01 baseline)      1 ds_add_f64 (native double atomic)     2 ds_add_u64
//   3 ds_add_u32                                 4 ds_add_rtn_f32 (returning form)       5 CAS-loop, 8-byte (two floats per lane, ds_cmpst_b64)
//   6 plain read-add-write b32                   7 plain read-add-write b64 (two floats per lane)
//   8 CAS-loop float add, upper half of every wave idle                                  9 ds_add_f32 native
//  10 ds_pk_add_f16 (packed half, for the rate only)
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
    __shared__ double win64[480 * 32];
    float* win = (float*)win64;
    const int tid = threadIdx.x;
    for (int i = tid; i < 960 * 32; i += 1024) win[i] = 0.f;
    __syncthreads();
    const int c = tid & 31, group = tid >> 5;
    unsigned cell = group * 29 + 1;
    float v = 1.0f + tid;
    for (int it = 0; it < iters; it++) {
        cell = (cell * 37 + 11) % 480;
        if (MODE == 0) {
            unsigned* addr = (unsigned*)&win[cell * 32 + c];
            unsigned old = *addr, assumed;
            do { assumed = old; old = atomicCAS(addr, assumed, __float_as_uint(__uint_as_float(assumed) + v)); } while (old != assumed);
        } else if (MODE == 1) {
            atomicAdd(&win64[cell * 32 + c], (double)v);
        } else if (MODE == 2) {
            atomicAdd((unsigned long long*)&win64[cell * 32 + c], (unsigned long long)(long long)v);
        } else if (MODE == 3) {
            atomicAdd((int*)&win[cell * 32 + c], (int)v);
        } else if (MODE == 4) {
            v += 1e-30f * atomicAdd(&win[cell * 32 + c], v);
        } else if (MODE == 5) {
            unsigned long long* addr = (unsigned long long*)&win64[cell * 32 + c];
            unsigned long long old = *addr, assumed;
            do {
                assumed = old;
                const float lo = __uint_as_float((unsigned)assumed) + v, hi = __uint_as_float((unsigned)(assumed >> 32)) + v;
                old = atomicCAS(addr, assumed, ((unsigned long long)__float_as_uint(hi) << 32) | __float_as_uint(lo));
            } while (old != assumed);
        } else if (MODE == 6) {
            float x = win[cell * 32 + c]; win[cell * 32 + c] = x + v;
        } else if (MODE == 7) {
            float2 x = ((float2*)win64)[cell * 32 + c]; x.x += v; x.y += v; ((float2*)win64)[cell * 32 + c] = x;
        } else if (MODE == 8) {
            if ((tid & 32) == 0) {
                unsigned* addr = (unsigned*)&win[cell * 32 + c];
                unsigned old = *addr, assumed;
                do { assumed = old; old = atomicCAS(addr, assumed, __float_as_uint(__uint_as_float(assumed) + v)); } while (old != assumed);
            }
        } else if (MODE == 9) {
            atomicAdd(&win[cell * 32 + c], v);
        } else if (MODE == 10) {
            const unsigned addr = (unsigned)(size_t)&win[cell * 32 + c];
            asm volatile("ds_pk_add_f16 %0, %1" ::"v"(addr), "v"(__float_as_uint(v)) : "memory");
        }
    }
    __syncthreads();
    if (tid < 32) out[blockIdx.x * 32 + tid] = win[tid] + v;
}
template <int MODE>
void run(const char* name, float* out, int per_lane_floats) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 1024;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 16 * iters;       // wave instructions
    printf("%-44s %8.3f ms  %7.1f G wave-ops/s  %6.1f clk per wave op per CU (2.4 GHz)  %7.1f G floats/s\n", name, ms, instr / ms / 1e6,
           (ms * 1e-3) * 2.4e9 * 256 / instr, instr * 64 * per_lane_floats / ms / 1e6);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 32 * 4);
    run<0>("CAS-loop float add (b32)", out, 1);
    run<1>("ds_add_f64", out, 1);
    run<2>("ds_add_u64", out, 1);
    run<3>("ds_add_u32", out, 1);
    run<4>("ds_add_rtn_f32", out, 1);
    run<5>("CAS-loop two floats (b64)", out, 2);
    run<6>("read + add + write b32", out, 1);
    run<7>("read + add + write b64 (two floats)", out, 2);
    run<8>("CAS-loop float add, half the lanes", out, 1);
    run<9>("ds_add_f32", out, 1);
    run<10>("ds_pk_add_f16", out, 2);
    return 0;
}
