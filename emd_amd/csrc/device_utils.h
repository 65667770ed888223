// device_utils.h -- wave64 cross-lane primitives for gfx950 (DPP row shifts / broadcasts, ballot).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// DPP controls (gfx9): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f32(float v) {
    // lanes whose DPP source is invalid (or masked rows) receive 0 and keep v
    int src = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(src);
}

// Inclusive wave64 prefix sum; lane 63 ends up with the wave total.  6 v_add_f32_dpp.
__device__ __forceinline__ float wave_scan_add_f32(float v) {
    v = dpp_add_f32<DPP_ROW_SHR(1), 0xf>(v);
    v = dpp_add_f32<DPP_ROW_SHR(2), 0xf>(v);
    v = dpp_add_f32<DPP_ROW_SHR(4), 0xf>(v);
    v = dpp_add_f32<DPP_ROW_SHR(8), 0xf>(v);
    v = dpp_add_f32<DPP_ROW_BCAST15, 0xa>(v);
    v = dpp_add_f32<DPP_ROW_BCAST31, 0xc>(v);
    return v;
}

// Shift the whole wave up by one lane (lane l receives lane l-1; lane 0 receives `fill`): DPP wave_shr:1.
#define DPP_WAVE_SHR1 0x138
__device__ __forceinline__ float wave_shift_up1_f32(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), DPP_WAVE_SHR1, 0xf, 0xf, false));
}

// Hand-fused scans: one VALU op per step (v_mul_f32_dpp / v_add_f32_dpp with the destination doubling as `old`, so
// lanes without a valid source keep their value); hipcc emits v_mov_b32_dpp + v_mul_f32 (+ an identity v_mov) per
// step for the product scan.  The compiler does not pad the VALU-write -> DPP-read hazard inside asm statements.
// Two independent scans interleaved: each DPP step of one chain sits between two steps of the other, so one s_nop 0
// per pair covers the VALU-write -> DPP-read hazard (2 wait states) of both chains.
#define EMD_DPP_STEP2(op, a, b, ctrl)                                                                          \
    asm volatile("s_nop 0\n\t" op " %0, %0, %0 " ctrl "\n\t" op " %1, %1, %1 " ctrl : "+v"(a), "+v"(b))
__device__ __forceinline__ void wave_scan_mul2_f32_asm(float& a, float& b) {
    asm volatile("s_nop 0");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_shr:1 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_shr:2 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_shr:4 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_shr:8 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_bcast:31 row_mask:0xc bank_mask:0xf");
}
__device__ __forceinline__ void wave_scan_add2_f32_asm(float& a, float& b) {
    asm volatile("s_nop 0");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_shr:1 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_shr:2 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_shr:4 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_shr:8 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_bcast:31 row_mask:0xc bank_mask:0xf");
}

// The same scans inside each 32-lane HALF of the wave (the first five steps: lanes 31 and 63 end up with their half's totals)
__device__ __forceinline__ void half_scan_mul2_f32_asm(float& a, float& b) {
    asm volatile("s_nop 0");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_shr:1 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_shr:2 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_shr:4 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_shr:8 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_mul_f32_dpp", a, b, "row_bcast:15 row_mask:0xa bank_mask:0xf");
}
__device__ __forceinline__ void half_scan_add2_f32_asm(float& a, float& b) {
    asm volatile("s_nop 0");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_shr:1 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_shr:2 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_shr:4 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_shr:8 row_mask:0xf bank_mask:0xf");
    EMD_DPP_STEP2("v_add_f32_dpp", a, b, "row_bcast:15 row_mask:0xa bank_mask:0xf");
}

// Sum over the wave, valid in lane 63 only.
__device__ __forceinline__ float wave_reduce_to_lane63(float v) { return wave_scan_add_f32(v); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
    int src = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    return v + (uint32_t)src;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t v) {
    int src = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);          // (lanes without a source read 0: the identity of max on uint32)
    return max(v, (uint32_t)src);
}
// inclusive running maximum over the wave's lanes
__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t v) {
    v = dpp_max_u32<DPP_ROW_SHR(1), 0xf>(v);
    v = dpp_max_u32<DPP_ROW_SHR(2), 0xf>(v);
    v = dpp_max_u32<DPP_ROW_SHR(4), 0xf>(v);
    v = dpp_max_u32<DPP_ROW_SHR(8), 0xf>(v);
    v = dpp_max_u32<DPP_ROW_BCAST15, 0xa>(v);
    v = dpp_max_u32<DPP_ROW_BCAST31, 0xc>(v);
    return v;
}

__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
    v = dpp_add_u32<DPP_ROW_SHR(1), 0xf>(v);
    v = dpp_add_u32<DPP_ROW_SHR(2), 0xf>(v);
    v = dpp_add_u32<DPP_ROW_SHR(4), 0xf>(v);
    v = dpp_add_u32<DPP_ROW_SHR(8), 0xf>(v);
    v = dpp_add_u32<DPP_ROW_BCAST15, 0xa>(v);
    v = dpp_add_u32<DPP_ROW_BCAST31, 0xc>(v);
    return v;
}

// Float add into LDS shared between waves.  The native ds_add_f32 retires about one wave64 instruction per ~190 cycles on gfx950
// (measured: 0.33 lanes / clk / CU, 24x below ds_add_u32 and the plain read-modify-write); a compare-and-swap loop on the
// integer path runs ~10x faster (3.2 lanes / clk / CU) and is exact in the same way -- one rounding per add.
__device__ __forceinline__ void lds_add_f32(float* addr, float v) {
    unsigned* a = (unsigned*)addr;
    unsigned old = *a, assumed;
    do {
        assumed = old;
        old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + v));
    } while (old != assumed);
}

// fp64 add into LDS: ds_add_f64, native and at rate (8.9 clocks per wave64 instruction against 16.9 for the fp32 compare-and-swap
// loop and 193 for ds_add_f32: profiles/r03_lds_atomic_microbench.txt), no return value, so nothing waits on it.
__device__ __forceinline__ void lds_add_f64(double* addr, float v) {
    __hip_atomic_fetch_add(addr, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ float readlane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// Block-wide (256 threads = 4 waves) inclusive scan of one uint32 per thread.
// `smem` needs 4 uint32.  Returns the inclusive prefix; *total gets the block sum.
__device__ __forceinline__ uint32_t block_scan_add_u32(uint32_t v, uint32_t* smem, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_scan_add_u32(v);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    uint32_t s0 = smem[0], s1 = smem[1], s2 = smem[2], s3 = smem[3];
    uint32_t base = (wave > 0 ? s0 : 0) + (wave > 1 ? s1 : 0) + (wave > 2 ? s2 : 0);
    *total = s0 + s1 + s2 + s3;
    __syncthreads();
    return inc + base;
}
