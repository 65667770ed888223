// microbench_k6_pix2.hip -- VERDICT r3 item 5: the compositing loop of the render forward (csrc/render.hip, k_render_forward_q) with ONE
// pixel per lane (as built) against TWO horizontally adjacent pixels per lane in explicit packed fp32 (v_pk_fma / v_pk_mul / v_pk_add on
// register pairs, as the render backward already does).  Both forms use the pinned arithmetic of the image contract (gauss_power,
// pinned_exp, the fma chain per channel) and are compared bit for bit here.
//
// What is measured: nanoseconds per (pixel, list entry) pair with the kernel's own residency (64-thread workgroups, four waves per SIMD),
// the four 16-lane rows of a wave walking their own byte lists of a 128-slot LDS ring (broadcast ds_read_b128 of the records), every
// pair evaluated in the branch-free predicated form of the kernel.  What it cannot contain is the price of the coarser culling: a row
// of the packed form owns an 8 x 4 pixel region instead of a 4 x 4 sub-block, and on the bench scene 20 % more (pixel, entry) pairs pass the
// exact footprint test at that granularity (tests/analysis/pair_stats.py: 0.690 against 0.575 of the quadrant pairs), and the survivor
// lists the backward walks get longer.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o mb profiles/microbench_k6_pix2.hip && ./mb
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float x) { return (v2f){x, x}; }
#define RING 128

__device__ __forceinline__ float gauss_power(float A, float B, float C, float dx, float dy) {
#pragma clang fp contract(off)
    const float q = __builtin_fmaf(A * dx, dx, (C * dy) * dy);
    return __builtin_fmaf(-0.5f, q, -((B * dx) * dy));
}
__device__ __forceinline__ v2f gauss_power2(float A, float B, float C, v2f dx, float dy) {
#pragma clang fp contract(off)
    const v2f q = __builtin_elementwise_fma(splat2(A) * dx, dx, splat2((C * dy) * dy));
    return __builtin_elementwise_fma(splat2(-0.5f), q, -((splat2(B) * dx) * splat2(dy)));
}
__device__ __forceinline__ float pinned_exp(float x) {
#pragma clang fp contract(off)
    const float t = x * 1.44269504088896341f;
    const float n = __builtin_rintf(t);
    const float f = t - n;
    float p = 1.54035304e-4f;
    p = __builtin_fmaf(p, f, 1.33335581e-3f); p = __builtin_fmaf(p, f, 9.61812911e-3f); p = __builtin_fmaf(p, f, 5.55041087e-2f);
    p = __builtin_fmaf(p, f, 2.40226507e-1f); p = __builtin_fmaf(p, f, 6.93147181e-1f); p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}
__device__ __forceinline__ v2f pinned_exp2(v2f x) {
#pragma clang fp contract(off)
    const v2f t = x * splat2(1.44269504088896341f);
    const v2f n = __builtin_elementwise_rint(t);
    const v2f f = t - n;
    v2f p = splat2(1.54035304e-4f);
    p = __builtin_elementwise_fma(p, f, splat2(1.33335581e-3f)); p = __builtin_elementwise_fma(p, f, splat2(9.61812911e-3f));
    p = __builtin_elementwise_fma(p, f, splat2(5.55041087e-2f)); p = __builtin_elementwise_fma(p, f, splat2(2.40226507e-1f));
    p = __builtin_elementwise_fma(p, f, splat2(6.93147181e-1f)); p = __builtin_elementwise_fma(p, f, splat2(1.0f));
    return (v2f){__builtin_ldexpf(p.x, (int)n.x), __builtin_ldexpf(p.y, (int)n.y)};
}

// records of the ring: s0 = (x, y, depth, opacity), s1 = (A, B, C, -), s2 = (r, g, b, -), s3 = (nx, ny, nz, -)
struct Out { float c0, c1, c2, dz, n0, n1, n2, T; };

// ONE pixel per lane: 64 pixels of an 8 x 8 quadrant, row = 4 x 4 sub-block (the kernel's form)
__global__ void __launch_bounds__(64) k_pix1(const float4* __restrict__ recs, const uint8_t* __restrict__ lists, int n_list, int rounds, Out* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ float4 s0[RING], s1[RING], s2[RING], s3[RING];
    __shared__ uint8_t s_list[4][RING];
    const uint32_t lane = threadIdx.x, row = lane >> 4, l = lane & 15;
    for (int i = lane; i < RING; i += 64) { s0[i] = recs[4 * i]; s1[i] = recs[4 * i + 1]; s2[i] = recs[4 * i + 2]; s3[i] = recs[4 * i + 3]; }
    for (int i = lane; i < 4 * RING; i += 64) s_list[i / RING][i % RING] = lists[i];
    __syncthreads();
    const float pfx = (float)((row & 1) * 4 + (l & 3)), pfy = (float)((row >> 1) * 4 + (l >> 2));
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dz = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    bool done = false;
    for (int r = 0; r < rounds; r++) {
        if (r) { T = T * 0.5f + 0.5f; done = false; }               // (keeps the transmittance alive across rounds)
        const uint8_t* list = s_list[row];
        uint32_t j = list[0];
        float4 g0 = s0[j], g1 = s1[j];
#pragma unroll 2
        for (int i = 0; i < n_list; i++) {
            const uint32_t jn = (i + 1 < n_list) ? list[i + 1] : j;
            const float4 g0n = s0[jn], g1n = s1[jn];
            const float dx = g0.x - pfx, dy = g0.y - pfy;
            const float power = gauss_power(g1.x, g1.y, g1.z, dx, dy);
            const float alpha = fminf(0.99f, g0.w * pinned_exp(power));
            const bool hit = !done && power <= 0.f && alpha >= (1.f / 255.f);
            const float test_T = T * (1.f - alpha);
            const bool stop = hit && test_T < 0.0001f;
            const bool take = hit && !stop;
            const float w = take ? alpha * T : 0.f;
            const float4 g2 = s2[j], g3 = s3[j];
            C0 = __builtin_fmaf(g2.x, w, C0); C1 = __builtin_fmaf(g2.y, w, C1); C2 = __builtin_fmaf(g2.z, w, C2);
            Dz = __builtin_fmaf(g0.z, w, Dz);
            N0 = __builtin_fmaf(g3.x, w, N0); N1 = __builtin_fmaf(g3.y, w, N1); N2 = __builtin_fmaf(g3.z, w, N2);
            T = take ? test_T : T;
            done = done || stop;
            j = jn; g0 = g0n; g1 = g1n;
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = Out{C0, C1, C2, Dz, N0, N1, N2, T};
}

// TWO pixels per lane: 128 pixels (16 x 8), row = an 8 x 4 region, lane = the pixel pair (2 (l & 3), l >> 2), (+1, +0) of its row's region
__global__ void __launch_bounds__(64) k_pix2(const float4* __restrict__ recs, const uint8_t* __restrict__ lists, int n_list, int rounds, Out* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ float4 s0[RING], s1[RING], s2[RING], s3[RING];
    __shared__ uint8_t s_list[4][RING];
    const uint32_t lane = threadIdx.x, row = lane >> 4, l = lane & 15;
    for (int i = lane; i < RING; i += 64) { s0[i] = recs[4 * i]; s1[i] = recs[4 * i + 1]; s2[i] = recs[4 * i + 2]; s3[i] = recs[4 * i + 3]; }
    for (int i = lane; i < 4 * RING; i += 64) s_list[i / RING][i % RING] = lists[i];
    __syncthreads();
    // the same 64 pixels as two lanes of k_pix1 would hold: lane pair (row, l) covers pixels (2 (l & 3) [+1], l >> 2) of an 8 x 4 region;
    // for the bit-for-bit comparison the regions are laid over the quadrant of k_pix1: rows 0, 1 = its upper / lower half
    const float px0 = (float)(2 * (l & 3)), pfy = (float)((row & 1) * 4 + (l >> 2));
    const v2f pfx = (v2f){px0, px0 + 1.f};
    v2f T = splat2(1.f), C0 = splat2(0.f), C1 = C0, C2 = C0, Dz = C0, N0 = C0, N1 = C0, N2 = C0;
    bool done_a = false, done_b = false;
    for (int r = 0; r < rounds; r++) {
        if (r) { T = T * splat2(0.5f) + splat2(0.5f); done_a = done_b = false; }
        const uint8_t* list = s_list[row];
        uint32_t j = list[0];
        float4 g0 = s0[j], g1 = s1[j];
#pragma unroll 2
        for (int i = 0; i < n_list; i++) {
            const uint32_t jn = (i + 1 < n_list) ? list[i + 1] : j;
            const float4 g0n = s0[jn], g1n = s1[jn];
            const v2f dx = splat2(g0.x) - pfx;
            const float dy = g0.y - pfy;
            const v2f power = gauss_power2(g1.x, g1.y, g1.z, dx, dy);
            const v2f aw = splat2(g0.w) * pinned_exp2(power);
            const v2f alpha = (v2f){fminf(0.99f, aw.x), fminf(0.99f, aw.y)};
            const bool hit_a = !done_a && power.x <= 0.f && alpha.x >= (1.f / 255.f), hit_b = !done_b && power.y <= 0.f && alpha.y >= (1.f / 255.f);
            const v2f test_T = T * (splat2(1.f) - alpha);
            const bool stop_a = hit_a && test_T.x < 0.0001f, stop_b = hit_b && test_T.y < 0.0001f;
            const bool take_a = hit_a && !stop_a, take_b = hit_b && !stop_b;
            const v2f aT = alpha * T;
            const v2f w = (v2f){take_a ? aT.x : 0.f, take_b ? aT.y : 0.f};
            const float4 g2 = s2[j], g3 = s3[j];
            C0 = __builtin_elementwise_fma(splat2(g2.x), w, C0); C1 = __builtin_elementwise_fma(splat2(g2.y), w, C1);
            C2 = __builtin_elementwise_fma(splat2(g2.z), w, C2); Dz = __builtin_elementwise_fma(splat2(g0.z), w, Dz);
            N0 = __builtin_elementwise_fma(splat2(g3.x), w, N0); N1 = __builtin_elementwise_fma(splat2(g3.y), w, N1);
            N2 = __builtin_elementwise_fma(splat2(g3.z), w, N2);
            T = (v2f){take_a ? test_T.x : T.x, take_b ? test_T.y : T.y};
            done_a = done_a || stop_a; done_b = done_b || stop_b;
            j = jn; g0 = g0n; g1 = g1n;
        }
    }
    Out* o = out + (size_t)blockIdx.x * 128 + 2 * lane;
    o[0] = Out{C0.x, C1.x, C2.x, Dz.x, N0.x, N1.x, N2.x, T.x};
    o[1] = Out{C0.y, C1.y, C2.y, Dz.y, N0.y, N1.y, N2.y, T.y};
}

template <typename F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    // a ring of 128 records around an 8 x 8 pixel window: elongated, partly faint footprints (about half of the pairs contribute)
    std::vector<float> recs(RING * 16);
    srand(11);
    auto rnd = []() { return (float)(rand() & 0xffffff) / 16777216.f; };
    for (int i = 0; i < RING; i++) {
        float* r = &recs[16 * i];
        r[0] = rnd() * 12.f - 2.f; r[1] = rnd() * 12.f - 2.f; r[2] = 5.f + rnd() * 30.f; r[3] = 0.02f + 0.3f * rnd();
        const float a = 0.03f + 0.5f * rnd(), c = 0.03f + 0.5f * rnd(), b = (rnd() - 0.5f) * 1.6f * sqrtf(a * c);
        r[4] = a; r[5] = b; r[6] = c; r[7] = 0.f;
        r[8] = rnd(); r[9] = rnd(); r[10] = rnd(); r[11] = 0.f; r[12] = rnd() - 0.5f; r[13] = rnd() - 0.5f; r[14] = rnd() - 0.5f; r[15] = 0.f;
    }
    const int n_list = 96;
    // per-row lists.  For the bit-for-bit comparison the packed kernel's rows 0 / 1 (8 x 4 regions = the quadrant's upper / lower half) walk the
    // SAME list as the two sub-block rows of k_pix1 they cover: all four lists are made equal here (timing is not affected by which slots they name)
    std::vector<uint8_t> lists(4 * RING);
    for (int i = 0; i < RING; i++) { const uint8_t s = (uint8_t)((i * 37 + 11) % RING); for (int r = 0; r < 4; r++) lists[r * RING + i] = s; }
    float4* d_recs; uint8_t* d_lists; Out *d_o1, *d_o2;
    const int blocks = 256 * 16 * 8;                 // 16 waves per CU resident (four per SIMD), eight generations
    CHECK(hipMalloc(&d_recs, recs.size() * 4)); CHECK(hipMalloc(&d_lists, lists.size()));
    CHECK(hipMalloc(&d_o1, sizeof(Out) * 64 * (size_t)blocks)); CHECK(hipMalloc(&d_o2, sizeof(Out) * 128 * (size_t)blocks));
    CHECK(hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_lists, lists.data(), lists.size(), hipMemcpyHostToDevice));
    // ---- bit-for-bit: the 64 pixels of k_pix1 against the first two rows' 64 pixels of k_pix2 (same pixel coordinates, same lists)
    hipLaunchKernelGGL(k_pix1, dim3(1), dim3(64), 0, 0, d_recs, d_lists, n_list, 1, d_o1);
    hipLaunchKernelGGL(k_pix2, dim3(1), dim3(64), 0, 0, d_recs, d_lists, n_list, 1, d_o2);
    CHECK(hipDeviceSynchronize());
    std::vector<Out> o1(64), o2(128);
    CHECK(hipMemcpy(o1.data(), d_o1, sizeof(Out) * 64, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(o2.data(), d_o2, sizeof(Out) * 128, hipMemcpyDeviceToHost));
    int diff = 0, hits = 0;
    for (int lane = 0; lane < 64; lane++) {                       // pixel (x, y) of k_pix1's lane
        const int row = lane >> 4, l = lane & 15, x = (row & 1) * 4 + (l & 3), y = (row >> 1) * 4 + (l >> 2);
        // k_pix2: row2 = y / 4 (rows 0, 1), l2 = (y % 4) * 4 + x / 2, component x & 1
        const int row2 = y / 4, l2 = (y % 4) * 4 + x / 2, idx = 2 * (row2 * 16 + l2) + (x & 1);
        if (memcmp(&o1[lane], &o2[idx], sizeof(Out)) != 0) diff++;
        if (o1[lane].T < 0.999f) hits++;
    }
    printf("# one pixel per lane vs two pixels per lane (packed fp32), pinned arithmetic of the image contract; MI355X\n");
    printf("bit-for-bit: %d of 64 pixels differ (%d pixels received a contribution)\n", diff, hits);
    const int rounds = 8, reps = 10;
    const float t1 = time_ms([&] { hipLaunchKernelGGL(k_pix1, dim3(blocks), dim3(64), 0, 0, d_recs, d_lists, n_list, rounds, d_o1); }, reps);
    const float t2 = time_ms([&] { hipLaunchKernelGGL(k_pix2, dim3(blocks), dim3(64), 0, 0, d_recs, d_lists, n_list, rounds, d_o2); }, reps);
    const double pairs1 = (double)blocks * 64 * n_list * rounds, pairs2 = (double)blocks * 128 * n_list * rounds;
    printf("one pixel per lane : %.3f ms for %.2e (pixel, entry) pairs = %.2f ps per pair\n", t1, pairs1, t1 * 1e9 / pairs1);
    printf("two pixels per lane: %.3f ms for %.2e (pixel, entry) pairs = %.2f ps per pair  (%.2f x)\n", t2, pairs2, t2 * 1e9 / pairs2, (t1 / pairs1) / (t2 / pairs2));
    printf("with the culling granularity of the bench scene (0.575 vs 0.690 of the quadrant pairs evaluated): relative cost of the pixel loop %.3f\n",
           (t2 / pairs2 * 0.690) / (t1 / pairs1 * 0.575));
    return 0;
}
