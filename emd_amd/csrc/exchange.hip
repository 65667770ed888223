// exchange.hip -- visibility-compacted rows for the view-parallel gradient exchange (no reference counterpart: the
// reference trains one view per step on one GPU, S3Gaussian/train.py:203).
//
// A view's gradient is zero for every Gaussian the view does not see (47 % of them on the bench scene).  Instead of all-reducing dense
// [N, .] tensors, a rank can send only the rows of its V visible Gaussians -- (index, values) -- with an all-gather, and every rank adds the
// gathered rows into a dense buffer in RANK ORDER (a fixed order of float additions: replicas stay bit-identical).  At 2 and 4 ranks that moves
// fewer bytes over each xGMI link than the ring all-reduce; at 8 it does not, and emd_amd.dp chooses per world size.
//
//   emd_compact_rows  : rows[1 + j] = (index, v_0 .. v_{w-1}) for the j-th visible Gaussian (radii > 0), values gathered from up to four
//                       [N, w_k] sources; rows[0] = (count, overflow, -, ...) is the header that travels with the rows.  Workgroups claim
//                       their output ranges with one atomic each: the rows of a view are unordered (their indices are unique, so the order
//                       does not matter to the adds).
//   emd_scatter_rows  : one gathered view: for j < count: dst_k[index] (+)= scale * v.  No atomics: indices are unique within a view;
//                       the caller launches the views one after the other.
#include "common.h"
#include "device_utils.h"

namespace {

struct RowSrc { const float* p[EMD_ROW_SOURCES]; float* d[EMD_ROW_SOURCES]; int w[EMD_ROW_SOURCES]; int n; };

// A workgroup takes COMPACT_TILES x 256 consecutive Gaussians and claims its output range with ONE atomic (round 5, late): with a claim per 256
// Gaussians the kernel was 7 800 returning atomics on one address, served one after the other -- 94 us for 100 MB at 2 M Gaussians, whatever the loads
// and stores around them did (requesting a row's values together, assembling the rows in LDS: both measured, no change).  Ranks inside the range: wave
// ballots + a 32-entry table of (tile, wave) counts, so the rows of a workgroup stay in index order.
#define COMPACT_TILES 8
__global__ void __launch_bounds__(EMD_BLOCK) k_compact_rows(int N, const int32_t* __restrict__ radii, RowSrc s, int row_words, uint32_t cap,
                                                            uint32_t* __restrict__ rows, uint32_t* __restrict__ counter) {
    __shared__ uint32_t s_cnt[COMPACT_TILES][EMD_BLOCK / 64];
    __shared__ uint32_t s_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t chunk0 = (size_t)blockIdx.x * (COMPACT_TILES * EMD_BLOCK);
    int32_t rad[COMPACT_TILES];
#pragma unroll
    for (int t = 0; t < COMPACT_TILES; t++) {                       // (all of the thread's radii together)
        const size_t i = chunk0 + (size_t)t * EMD_BLOCK + threadIdx.x;
        rad[t] = radii[i < (size_t)N ? i : 0];
    }
    unsigned long long bal[COMPACT_TILES];
#pragma unroll
    for (int t = 0; t < COMPACT_TILES; t++) {
        const size_t i = chunk0 + (size_t)t * EMD_BLOCK + threadIdx.x;
        bal[t] = __ballot(i < (size_t)N && rad[t] > 0);
        if (lane == 0) s_cnt[t][wave] = (uint32_t)__popcll(bal[t]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int t = 0; t < COMPACT_TILES; t++)
            for (int w = 0; w < EMD_BLOCK / 64; w++) total += s_cnt[t][w];
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t run = s_base;                                          // rows in front of (tile t, this wave)
#pragma unroll
    for (int t = 0; t < COMPACT_TILES; t++) {
        uint32_t before = run;
#pragma unroll
        for (int w = 0; w < EMD_BLOCK / 64; w++) { if ((uint32_t)w < wave) before += s_cnt[t][w]; run += s_cnt[t][w]; }
        const bool vis = (bal[t] >> lane) & 1ull;
        if (!vis) continue;
        const uint32_t slot = before + (uint32_t)__popcll(bal[t] & ((1ull << lane) - 1ull));
        if (slot >= cap) continue;                                  // (the header reports it: the caller must not use this step's exchange)
        const size_t i = chunk0 + (size_t)t * EMD_BLOCK + threadIdx.x;
        uint32_t* r = rows + (size_t)(slot + 1u) * row_words;
        r[0] = (uint32_t)i;
        int o = 1;
#pragma unroll
        for (int k = 0; k < EMD_ROW_SOURCES; k++) {
            if (k >= s.n) break;
            for (int c = 0; c < s.w[k]; c++) r[o + c] = __float_as_uint(s.p[k][i * s.w[k] + c]);
            o += s.w[k];
        }
    }
}

// the header row, written once every workgroup's claim has landed (a launch of its own: kernel boundary = the only ordering used)
__global__ void k_rows_header(const uint32_t* __restrict__ counter, uint32_t cap, int row_words, uint32_t* __restrict__ rows) {
    const uint32_t c = *counter;
    if (threadIdx.x == 0) { rows[0] = c < cap ? c : cap; rows[1] = c > cap ? 1u : 0u; }
    for (int k = 2 + (int)threadIdx.x; k < row_words; k += (int)blockDim.x) rows[k] = 0u;
}

template <bool ADD>
__global__ void __launch_bounds__(EMD_BLOCK) k_scatter_rows(const uint32_t* __restrict__ rows, int row_words, uint32_t cap, uint32_t n_dst, RowSrc s,
                                                            float scale, uint32_t* __restrict__ overflow_out) {
    const uint32_t count = min(rows[0], cap);
    if (blockIdx.x == 0 && threadIdx.x == 0 && rows[1] && overflow_out) atomicOr(overflow_out, 1u);
    for (uint32_t j = blockIdx.x * EMD_BLOCK + threadIdx.x; j < count; j += gridDim.x * EMD_BLOCK) {
        const uint32_t* r = rows + (size_t)(j + 1u) * row_words;
        const uint32_t i = r[0];
        if (i >= n_dst) continue;                                   // (a corrupt row must not write out of bounds)
        int o = 1;
#pragma unroll
        for (int k = 0; k < EMD_ROW_SOURCES; k++) {
            if (k >= s.n) break;
            for (int c = 0; c < s.w[k]; c++) {
                float* dst = s.d[k] + (size_t)i * s.w[k] + c;
                const float v = scale * __uint_as_float(r[o + c]);
                *dst = ADD ? *dst + v : v;
            }
            o += s.w[k];
        }
    }
}

}  // namespace

extern "C" int emd_compact_rows(int32_t n, const int32_t* radii, int32_t num_sources, const float* const* sources, const int32_t* widths,
                                int64_t capacity, uint32_t* rows, uint32_t* counter, void* hip_stream) {
    if (n < 0 || num_sources < 1 || num_sources > EMD_ROW_SOURCES || !sources || !widths || !rows || !counter || capacity < 0 || (n > 0 && !radii)) {
        emd_set_error("compact_rows: bad argument");
        return EMD_ERR_INVALID;
    }
    RowSrc s;
    s.n = num_sources;
    int row_words = 1;
    for (int k = 0; k < EMD_ROW_SOURCES; k++) {
        s.p[k] = k < num_sources ? sources[k] : nullptr;
        s.d[k] = nullptr;
        s.w[k] = k < num_sources ? widths[k] : 0;
        if (k < num_sources && ((!sources[k] && n > 0) || widths[k] < 1 || widths[k] > 16)) { emd_set_error("compact_rows: source %d: null or width outside 1..16", k); return EMD_ERR_INVALID; }
        row_words += s.w[k];
    }
    if (row_words < 2) { emd_set_error("compact_rows: empty rows"); return EMD_ERR_INVALID; }
    hipStream_t st = (hipStream_t)hip_stream;
    int rc = emd_zero_async(counter, 4, st);
    if (rc) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(k_compact_rows, dim3((n + COMPACT_TILES * EMD_BLOCK - 1) / (COMPACT_TILES * EMD_BLOCK)), dim3(EMD_BLOCK), 0, st, n, radii, s, row_words,
                           (uint32_t)capacity, rows, counter);
        EMD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_rows_header, dim3(1), dim3(64), 0, st, counter, (uint32_t)capacity, row_words, rows);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

extern "C" int emd_scatter_rows(const uint32_t* rows, int64_t capacity, int32_t num_dst_rows, int32_t num_dests, float* const* dests, const int32_t* widths,
                                int32_t add, float scale, uint32_t* overflow_out, void* hip_stream) {
    if (!rows || capacity < 0 || num_dst_rows < 0 || num_dests < 1 || num_dests > EMD_ROW_SOURCES || !dests || !widths) {
        emd_set_error("scatter_rows: bad argument");
        return EMD_ERR_INVALID;
    }
    RowSrc s;
    s.n = num_dests;
    int row_words = 1;
    for (int k = 0; k < EMD_ROW_SOURCES; k++) {
        s.p[k] = nullptr;
        s.d[k] = k < num_dests ? dests[k] : nullptr;
        s.w[k] = k < num_dests ? widths[k] : 0;
        if (k < num_dests && ((!dests[k] && num_dst_rows > 0) || widths[k] < 1 || widths[k] > 16)) { emd_set_error("scatter_rows: destination %d: null or width outside 1..16", k); return EMD_ERR_INVALID; }
        row_words += s.w[k];
    }
    if (capacity == 0 || num_dst_rows == 0) return EMD_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    const unsigned nb = (unsigned)((capacity + EMD_BLOCK - 1) / EMD_BLOCK);
    if (add) hipLaunchKernelGGL((k_scatter_rows<true>), dim3(nb < 8192u ? nb : 8192u), dim3(EMD_BLOCK), 0, st, rows, row_words, (uint32_t)capacity, (uint32_t)num_dst_rows, s, scale, overflow_out);
    else hipLaunchKernelGGL((k_scatter_rows<false>), dim3(nb < 8192u ? nb : 8192u), dim3(EMD_BLOCK), 0, st, rows, row_words, (uint32_t)capacity, (uint32_t)num_dst_rows, s, scale, overflow_out);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
