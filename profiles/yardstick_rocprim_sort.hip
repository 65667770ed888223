// yardstick_rocprim_sort.hip -- a YARDSTICK ONLY (never linked into the product, nothing under emd_amd/ includes rocPRIM):
// rocprim::radix_sort_pairs (the library's onesweep / merge-sort dispatch) on the two sorts of the binning stage at the bench's sizes,
// timed with HIP events, next to what the hand-written passes of csrc/binning.hip take for the same job (profiles/r05_sort_yardstick.txt).
//
//   (a) depth sort: u32 keys of 27 significant bits (depth bits above the near plane), u32 values, V = 1.06 M pairs
//       (the product additionally compacts the N = 2 M keys to the V visible ones in its first pass; the yardstick is given the V pairs)
//   (b) tile sort:  u32 keys of 13 significant bits (tile ids below 6 700), u32 values, D = 3.9 M pairs, stable
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/yardstick_rocprim_sort.hip -o /tmp/yardstick && /tmp/yardstick
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static int run(const char* name, size_t n, unsigned bits, bool tile_like, int reps) {
    std::vector<uint32_t> hk(n), hv(n);
    std::mt19937 g(12345);
    for (size_t i = 0; i < n; i++) {
        if (tile_like) hk[i] = (uint32_t)(g() % 6700u);                                    // tile ids
        else { float z = 0.2f + (float)(g() % 1000000u) * 1.2e-4f; uint32_t b, nb; float np = 0.2f; memcpy(&b, &z, 4); memcpy(&nb, &np, 4); hk[i] = b - nb; }   // depth bits above the near plane
        hv[i] = (uint32_t)i;
    }
    uint32_t *k0, *k1, *v0, *v1;
    CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
    CK(hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice));
    size_t tmp_bytes = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0));
    void* tmp;
    CK(hipMalloc(&tmp, tmp_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(e0, 0));
        CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[i], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    // correctness of the yardstick itself (stable order)
    std::vector<uint32_t> ok(n), ov(n);
    CK(hipMemcpy(ok.data(), k1, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ov.data(), v1, n * 4, hipMemcpyDeviceToHost));
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i++) sorted = ok[i - 1] < ok[i] || (ok[i - 1] == ok[i] && ov[i - 1] < ov[i]);
    printf("%-28s n = %8zu  bits = %2u  rocprim::radix_sort_pairs: median %7.1f us  min %7.1f us  (temporary storage %.1f MB, stable order %s)\n", name, n,
           bits, ms[reps / 2] * 1e3, ms[0] * 1e3, tmp_bytes / 1e6, sorted ? "ok" : "VIOLATED");
    (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(v1); (void)hipFree(tmp);
    return 0;
}

int main() {
    int rc = 0;
    rc |= run("(a) depth sort, V pairs", 1060000, 27, false, 50);
    rc |= run("(a') depth sort, N pairs", 2000000, 27, false, 50);
    rc |= run("(b) tile sort, D pairs", 3900000, 13, true, 50);
    rc |= run("(b') tile sort, 14 bits", 3900000, 14, true, 50);
    return rc;
}
