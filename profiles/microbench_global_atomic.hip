#include <hip/hip_runtime.h>
#include <stdio.h>
// global atomics: rows of 32 floats at pseudo-random rows of a big buffer (like the HexPlane plane gradients)
template <int MODE>
__global__ void __launch_bounds__(256) k(float* buf, unsigned rows, int iters) {
    const int c = threadIdx.x & 31;
    unsigned r = (blockIdx.x * 8 + (threadIdx.x >> 5)) * 2654435761u;
    float v = 1.0f;
    for (int it = 0; it < iters; it++) {
        r = r * 1664525u + 1013904223u;
        float* addr = buf + (size_t)(r % rows) * 32 + c;
        if (MODE == 0) atomicAdd(addr, v);
        else if (MODE == 1) atomicAdd((unsigned*)addr, 1u);
        else if (MODE == 2) {
            unsigned* a = (unsigned*)addr;
            unsigned old = __builtin_nontemporal_load(a), assumed;
            do { assumed = old; old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + v)); } while (old != assumed);
        } else if (MODE == 3) unsafeAtomicAdd(addr, v);
        else if (MODE == 4) *addr = v;
    }
}
int main() {
    const unsigned rows = 1u << 20;   // 128 MB
    float* buf; (void)hipMalloc(&buf, (size_t)rows * 128); (void)hipMemset(buf, 0, (size_t)rows * 128);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 64, blocks = 65536;
    const char* names[5] = {"atomicAdd float", "atomicAdd uint", "CAS-loop float", "unsafeAtomicAdd float", "plain store"};
    for (int m = 0; m < 5; m++) {
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, rows, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, rows, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, rows, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, buf, rows, iters);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, buf, rows, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double lane_ops = (double)blocks * 256 * iters;
        printf("%-24s %.3f ms  %.1f G lane-ops/s  (%.1f G rows/s)\n", names[m], ms, lane_ops / ms / 1e6, lane_ops / 32 / ms / 1e6);
    }
    return 0;
}
