#include <hip/hip_runtime.h>
#include <stdio.h>
// raw throughput of LDS float atomics vs plain LDS read-modify-write, conflict-free rows
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
    __shared__ float win[960 * 32];
    const int tid = threadIdx.x;
    for (int i = tid; i < 960 * 32; i += 1024) win[i] = 0.f;
    __syncthreads();
    const int c = tid & 31, group = tid >> 5;
    unsigned cell = group * 29 + 1;
    float v = 1.0f + tid;
    for (int it = 0; it < iters; it++) {
        cell = (cell * 37 + 11) % 960;
        if (MODE == 0) atomicAdd(&win[cell * 32 + c], v);
        else if (MODE == 1) { float x = win[cell * 32 + c]; win[cell * 32 + c] = x + v; }
        else if (MODE == 2) win[cell * 32 + c] = v;
        else if (MODE == 3) atomicAdd((int*)&win[cell * 32 + c], (int)v);
        else if (MODE == 4) {
            unsigned* addr = (unsigned*)&win[cell * 32 + c];
            unsigned old = *addr, assumed;
            do { assumed = old; old = atomicCAS(addr, assumed, __float_as_uint(__uint_as_float(assumed) + v)); } while (old != assumed);
        }
    }
    __syncthreads();
    if (tid < 32) out[blockIdx.x * 32 + tid] = win[tid];
}
int main() {
    float* out; hipMalloc(&out, 4096 * 32 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 1024;
    const char* names[5] = {"ds_add_f32 atomic", "read+add+write", "plain write", "ds_add_u32 atomic", "CAS-loop float add"};
    for (int m = 0; m < 5; m++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, out, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, out, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, out, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 0, 0, out, iters);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double lane_ops = (double)blocks * 1024 * iters;
        printf("%-20s %.3f ms  %.1f G lane-ops/s  = %.2f lanes/clk/CU (256 CUs, 2.4 GHz)\n", names[m], ms, lane_ops / ms / 1e6, lane_ops / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
