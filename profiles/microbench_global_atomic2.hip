#include <hip/hip_runtime.h>
#include <stdio.h>
// Round 3: what bounds the HexPlane backward's global float atomics (128-byte rows), and does the memory scope or locality change it?
//   0 agent scope (atomicAdd), random rows of 128 MB      1 workgroup scope, same rows        2 wavefront scope, same rows
//   3 agent scope, rows random inside a 1 MB region per block (L2-resident lines)             4 workgroup scope, same
//   5 agent scope, random rows of a 16 MB slice chosen by the XCC the block runs on (per-XCD private copies)
//   6 workgroup scope, same slices                         7 plain store
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xf; }   // HW_REG_XCC_ID[3:0]
template <int MODE>
__global__ void __launch_bounds__(256) k(float* buf, unsigned rows, int iters, unsigned* xcc_hist) {
    const int c = threadIdx.x & 31;
    unsigned r = (blockIdx.x * 8 + (threadIdx.x >> 5)) * 2654435761u;
    float v = 1.0f;
    const unsigned xcc = xcc_id();
    if (xcc_hist && threadIdx.x == 0) atomicAdd(&xcc_hist[(blockIdx.x & 7) * 16 + xcc], 1u);
    for (int it = 0; it < iters; it++) {
        r = r * 1664525u + 1013904223u;
        unsigned row = r % rows;
        if (MODE == 3 || MODE == 4) row = ((blockIdx.x * 8192u) + (r >> 8) % 8192u) % rows;        // 8192 rows = 1 MB per block
        if (MODE == 5 || MODE == 6) row = xcc * (rows / 8) + (r >> 4) % (rows / 8);
        float* addr = buf + (size_t)row * 32 + c;
        if (MODE == 0 || MODE == 3 || MODE == 5) atomicAdd(addr, v);
        else if (MODE == 1 || MODE == 4 || MODE == 6) __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else if (MODE == 7) *addr = v;
    }
}
template <int MODE>
void run(const char* name, float* buf, unsigned rows, unsigned* hist) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 64, blocks = 65536;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipMemset(buf, 0, (size_t)rows * 128);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, rows, iters, rep == 0 && MODE == 0 ? hist : nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double rows_done = (double)blocks * 8 * iters;
    // checksum: every add must have landed whatever the scope
    static float* host = (float*)malloc((size_t)rows * 128);
    (void)hipMemcpy(host, buf, (size_t)rows * 128, hipMemcpyDeviceToHost);
    double sum = 0; for (size_t i = 0; i < (size_t)rows * 32; i++) sum += host[i];
    printf("%-58s %7.3f ms  %6.1f G rows/s   sum %.0f (expected %.0f)\n", name, ms, rows_done / ms / 1e6, sum, MODE == 7 ? 0.0 : rows_done * 32);
}
int main() {
    const unsigned rows = 1u << 20;   // 128 MB
    float* buf; (void)hipMalloc(&buf, (size_t)rows * 128);
    unsigned* hist; (void)hipMalloc(&hist, 128 * 4); (void)hipMemset(hist, 0, 128 * 4);
    run<0>("agent scope, random rows of 128 MB", buf, rows, hist);
    run<1>("workgroup scope, random rows of 128 MB", buf, rows, hist);
    run<2>("wavefront scope, random rows of 128 MB", buf, rows, hist);
    run<3>("agent scope, rows inside 1 MB per block", buf, rows, hist);
    run<4>("workgroup scope, rows inside 1 MB per block", buf, rows, hist);
    run<5>("agent scope, 16 MB slice per XCC", buf, rows, hist);
    run<6>("workgroup scope, 16 MB slice per XCC", buf, rows, hist);
    run<7>("plain store", buf, rows, hist);
    unsigned h[128]; (void)hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost);
    printf("blocks by (blockIdx & 7) -> XCC_ID histogram:\n");
    for (int b = 0; b < 8; b++) { printf("  blockIdx%%8=%d:", b); for (int x = 0; x < 16; x++) if (h[b * 16 + x]) printf(" xcc%d:%u", x, h[b * 16 + x]); printf("\n"); }
    return 0;
}
