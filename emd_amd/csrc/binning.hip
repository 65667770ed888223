// binning.hip -- K2..K5: depth order, tile duplication, stable partition by tile, tile ranges.
// [UPSTREAM K2-K5 in SURVEY.md section 2.4: CUB DeviceScan / duplicateWithKeys / DeviceRadixSort / identifyTileRanges]
//
// Upstream sorts D = sum(tiles touched) 64-bit keys (tile << 32 | depth bits) with their 32-bit values: 12 bytes per
// duplicate through every radix pass.  The depth bits of a key are a property of the GAUSSIAN, not of the duplicate, so
// the least-significant 32 bits of the LSD sort can be done BEFORE duplicating, on N (depth, id) pairs instead of
// D (key, id) triples (D ~ 3.8 V on the bench scene):
//     1. stable LSD radix sort of the VISIBLE Gaussians by depth: the keys are (depth bits - bits(near plane)), non-negative floats
//        order like their bit patterns, so 3 passes of 9 bits cover depths up to 65 536 x near (EMD_ERR_DEPTH_RANGE -> the caller
//        retries with EMD_FLAG_WIDE_DEPTH_SORT: 4 passes of 8 bits on the raw bits).  The first pass reads the N keys, drops the
//        culled ones (0xFFFFFFFF) and publishes V; the other two run over V (key, id) pairs,
//     2. gather the tile rectangle / tile count of each Gaussian in that order, scan, duplicate: the duplicates appear
//        ordered by (depth, Gaussian id) and carry only (tile id, Gaussian id); the duplicate kernel also builds the digit
//        histogram of the first tile pass,
//     3. stable LSD radix sort of the D pairs by the ceil(log2(tiles)) tile bits (2 passes for up to 64 K tiles).
// A stable sort by a low key followed by a stable sort by a high key IS the LSD sort of the concatenated key, and the
// initial order is the Gaussian index in both formulations, so the final order is bit-for-bit the order of the upstream
// sort of the 64-bit keys: (tile, depth bits, Gaussian id).  emd_raster_export_binning rebuilds the 64-bit keys from
// (tile id, depth bits of the Gaussian) for the parity tests.  Bytes through the radix passes on the bench scene
// (N = 2 M, V = 1.06 M, D = 4.6 M): 6 x 24 B x 4.6 M = 670 MB upstream, 4 N + 8 V + 2 x 16 V + 2 x 16 D = 200 MB here.
//
// Launches of the stage (19): per depth pass histogram -> digit scan -> scatter (9); the binning records gathered into depth order with
// the pairs every 256 of them emit (1); the duplicate kernel, which scans those block sums itself (every workgroup, 31 KB from L2) and
// publishes D / overflow / V -- the single-workgroup scan kernel of rounds 1-3 is gone (1); two tile passes, the first one's histogram
// built by the duplicate kernel (5); tile ranges; the dispatch order of the render kernels.
//
// Design for gfx950:
//   - the duplicate count D stays on the device (EmdStatus.num_rendered); every kernel here is launched on the
//     caller-provided capacity and bounds itself by D, so the forward pass needs no host read-back to proceed.
//   - duplication is balanced over output slots, not Gaussians: a 256-thread block scans the tile counts of its
//     256 Gaussians (DPP wave scan + LDS), then lane e writes slot e, finding its Gaussian by binary search in
//     LDS -- consecutive lanes write consecutive pairs (coalesced 4 B + 4 B stores) regardless of footprint size.
//   - ranking inside a radix block is wave-ballot based (8 ballots per key give the set of lanes with the same digit;
//     no LDS atomics in the ranking loop), which keeps every pass stable.
#include <string.h>

#include "common.h"
#include "device_utils.h"
#include "footprint.h"       // (EMD_ID_BITS: the quadrant mask rides above the 28-bit Gaussian id of a list word)

// EMD_BIN_CARRY (round 5, measured and NOT kept; the code stays as the record of the experiment): the 8-byte binning records travel through the
// depth passes as a second value instead of being gathered in depth order afterwards.  The gather kernel shrinks from 22 to 6 us, every one
// of the three depth scatters grows by 7 us (16 KB more LDS per block, 8 more bytes per element each way): 798 against 803 it/s on one box.
#ifndef EMD_BIN_CARRY
#define EMD_BIN_CARRY 0
#endif
#ifndef EMD_DUP_SCAN_OWNER
#define EMD_DUP_SCAN_OWNER 1
#endif

namespace {

// number of elements of a radix pass: a launch-time constant (first pass of the Gaussian depth sort), or a device-side count
// (visible Gaussians after that pass; the duplicate count of the tile passes) that reads as 0 while the overflow word is set
struct SortN { const uint32_t* count; const uint32_t* overflow; uint32_t fixed; };
__device__ __forceinline__ uint32_t sort_n(const SortN& c) {
    return c.count ? ((c.overflow && *c.overflow) ? 0u : *c.count) : c.fixed;
}

// ---------------------------------------------------------------------------------------------------
// scan geometry of the single-workgroup scans below: 1024 elements per trip
// ---------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 4
#define SCAN_TILE (EMD_BLOCK * SCAN_ITEMS)

#define DUP_SLOTS EMD_SORT_TILE

// ---------------------------------------------------------------------------------------------------
// K2': binning records of the Gaussians in depth order (one 8-byte gather each) + the pairs every block of 256 of them emits
// ---------------------------------------------------------------------------------------------------
// (Round 4, measured and dropped: this gather inside the last depth pass's scatter kernel, with the block sums as one float... integer
//  atomic per (wave, output block) -- 117 k single-lane atomic instructions at ~117 clocks each on the CU's memory path made that pass
//  78 us instead of 18; as its own launch with 4900 independent workgroups the gather takes 20.)
// CARRIED (round 5): the records arrive in depth order already -- the depth passes carry them as a second value (the first pass reads them in
// index order, coalesced) -- so the gather (150 MB of fabric traffic for 8.5 MB of records, 22 us) is gone and this kernel only adds up the
// pairs of every block of 256 and clears the tile ranges.
template <bool CARRIED>
__global__ void __launch_bounds__(EMD_BLOCK) k_sorted_counts(int N, const uint32_t* __restrict__ num_sorted,
                                                             const uint32_t* __restrict__ perm,
                                                             const uint2* __restrict__ binrec,
                                                             uint2* __restrict__ bin_s,
                                                             uint32_t* __restrict__ block_sums,
                                                             uint32_t* __restrict__ ranges, uint32_t n_ranges) {
    __shared__ uint32_t s_scan[4];
    const int i = blockIdx.x * EMD_BLOCK + threadIdx.x;
    uint2 br = make_uint2(0u, 0u);
    if (i < N && (uint32_t)i < *num_sorted) br = CARRIED ? bin_s[i] : binrec[perm[i]];         // the depth sort kept the V visible Gaussians only
    if (!CARRIED && i < N) bin_s[i] = br;
    // empty tiles keep the range (0, 0): cleared here instead of by a separate memset launch
    for (uint32_t r = (uint32_t)i; r < n_ranges; r += gridDim.x * EMD_BLOCK) ranges[r] = 0u;
    uint32_t total;
    block_scan_add_u32(((br.x >> 20) & 1023u) * (br.y & 1023u), s_scan, &total);   // pairs this Gaussian emits: width x height
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = total;
        if (blockIdx.x == gridDim.x - 1) { block_sums[gridDim.x] = 0u; block_sums[gridDim.x + 1] = 0u; block_sums[gridDim.x + 2] = 0u; }   // (read as uint4)
    }
}

// ---------------------------------------------------------------------------------------------------
// K3 duplicate: (tile id, Gaussian id) pairs in (depth, Gaussian id) order
// ---------------------------------------------------------------------------------------------------
// Balanced over OUTPUT slots: workgroup b writes slots [2048 b, 2048 (b + 1)) and walks the blocks of 256 depth-ordered
// Gaussians that own them (the nearest Gaussians cover thousands of tiles each and all sit at the front of the order, so
// a Gaussian-block-per-workgroup split would leave a long tail).  Inside a Gaussian block the owner of a slot is found
// by binary search over the block's exclusive offsets in LDS; consecutive lanes write consecutive pairs.
// An output block of 2048 slots IS a block of the first tile-sort pass, so the digit histogram of that pass is built here, from
// the tile ids while they are in registers (hist0: [bin][block], bin-major over the capacity block count): the pass's own
// histogram launch (a second read of all D tile ids) is gone.
//
// Round 4: the rectangle a Gaussian enumerates is no longer upstream's 3-sigma square of the LARGEST eigenvalue but that square cut down
// to the tiles its alpha >= 1/255 bounding box reaches (K1, preprocess.hip): on street scenes -- elongated, often faint footprints -- 25 %
// of upstream's (tile, Gaussian) pairs lie outside that box (31 % reach no pixel of their tile at all: tests/analysis/pair_stats.py), and
// everything from here on -- this kernel, both tile passes, the range kernel, the list scans of the render forward -- shrinks with them.
// Images, radii and gradients are unchanged bit for bit: a pair that is left out is one upstream's render loop skips at every pixel of
// the tile (its alpha < 1/255 `continue`).  EMD_FLAG_KEEP_ALL_PAIRS restores upstream's list entry for entry.
// The same box gives every emitted pair the mask of the tile's four 8x8 quadrants that hold a pixel centre inside it (K1 leaves four
// bits per Gaussian: is the outer quadrant column / row of the rectangle's first / last tile outside the box); it rides in the top four
// bits of the list word, and the quadrant waves of the render
// forward read 4 bytes per list entry and fetch the 64-byte records of their own entries only.  (Measured and dropped on the way: the
// exact ellipse-rectangle test of footprint.h per pair in this kernel -- its inputs are a 32-byte gather per Gaussian through 128-byte
// lines, 26 -> 67 us, whatever the arithmetic behind it costs.)
__global__ void __launch_bounds__(EMD_BLOCK) k_duplicate(int N, int gx, const uint2* __restrict__ bin_s,
                                                         const uint32_t* __restrict__ perm,
                                                         const uint32_t* __restrict__ block_sums,
                                                         uint64_t capacity, EmdStatus* __restrict__ status,
                                                         uint32_t* __restrict__ tkeys, uint32_t* __restrict__ vals,
                                                         uint32_t* __restrict__ hist0, uint32_t mask0, uint32_t nblocks_cap) {
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_first[2];            // block of Gaussians that holds this workgroup's first slot, pairs in front of that block
    __shared__ uint32_t s_excl[EMD_BLOCK];     // exclusive offsets inside the block
    __shared__ uint32_t s_rect[EMD_BLOCK];     // x0 | y0 << 10 | width << 20   (grid dims < 1024 tiles = 16K px)
    __shared__ uint32_t s_id[EMD_BLOCK];
    __shared__ uint32_t s_hs[EMD_BLOCK];       // height | skip-top << 10 | skip-bottom << 11
    __shared__ uint32_t s_h0[EMD_BLOCK];       // digit histogram of tile pass 0 (at most 8 bits per pass)
#if EMD_DUP_SCAN_OWNER
    __shared__ __attribute__((aligned(16))) uint16_t s_own[DUP_SLOTS];      // owner (index in the Gaussian block + 1) of every slot of the window: head marks, then their running maximum
    __shared__ uint32_t s_wmax[8];             // per-wave maxima of the owner scan [0..3], of the first-slot owner [4..7]
#endif
    s_h0[threadIdx.x] = 0;
    // ---- every workgroup scans the pair counts of the 256-Gaussian blocks itself (the last depth pass left them in block_sums; 7812
    //      words at 2 M Gaussians, from L2): D, and where its own 2048 slots start.  This replaces a single-workgroup scan kernel whose
    //      14 us sat on the critical path of every forward.
    const uint32_t V = status->reserved;                                      // Gaussians the depth sort kept
    // chunks of 1024 block sums: a thread takes four consecutive ones (one dwordx4), eight chunks in flight together
    const uint32_t nbv = (V + EMD_BLOCK - 1) / EMD_BLOCK;
    const uint64_t S0l = (uint64_t)blockIdx.x * DUP_SLOTS;
    const uint4* bs4 = reinterpret_cast<const uint4*>(block_sums);            // (256-byte aligned, padded with zeros to a multiple of 4)
    uint32_t D = 0;
    bool found = false;
    for (uint32_t c0 = 0; c0 < nbv; c0 += 8u * 4u * EMD_BLOCK) {
        uint4 v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint32_t e0 = c0 + (uint32_t)c * 4u * EMD_BLOCK + 4u * threadIdx.x;
            v[c] = bs4[(e0 < nbv ? e0 : 0u) / 4u];                               // (unconditional: a guarded load is waited for where it is issued --
        }                                                                       //  the eight chunks were eight trips to L2 one after the other)
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint32_t e0 = c0 + (uint32_t)c * 4u * EMD_BLOCK + 4u * threadIdx.x;
            if (!(e0 < nbv)) v[c] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint32_t e0 = c0 + (uint32_t)c * 4u * EMD_BLOCK + 4u * threadIdx.x;
            if (c0 + (uint32_t)c * 4u * EMD_BLOCK >= nbv) break;                // (uniform)
            const uint32_t sum = v[c].x + v[c].y + v[c].z + v[c].w;
            uint32_t total;
            const uint32_t incl = block_scan_add_u32(sum, s_scan, &total);
            uint32_t run = D + incl - sum;
            if (!found && S0l >= run && S0l < (uint64_t)run + sum) {           // exactly one thread over all chunks: its four sums hold slot S0
                const uint32_t w4[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (S0l < (uint64_t)run + w4[q]) { s_first[0] = e0 + (uint32_t)q; s_first[1] = run; break; }
                    run += w4[q];
                }
            }
            D += total;
            found = found || S0l < D;
        }
    }
    const bool bad = (uint64_t)D > capacity || (status->overflow & 2u) != 0u;   // (bit 1: depth range, set by the depth sort)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        status->num_rendered = D;
        status->overflow = (status->overflow & 2u) | (((uint64_t)D > capacity) ? 1u : 0u);
        status->num_visible = V;
    }
    if (bad || S0l >= D) {                     // (uniform per block) nothing to write: the pass still reads this block's histogram column
        if (hist0) hist0[(size_t)threadIdx.x * nblocks_cap + blockIdx.x] = 0u;
        return;
    }
    const uint32_t S0 = (uint32_t)S0l, S1 = (uint32_t)min((uint64_t)D, S0l + DUP_SLOTS);
    __syncthreads();
    uint32_t base = s_first[1];
    for (uint32_t gb = s_first[0]; gb < nbv && base < S1; gb++) {
        const uint32_t i = gb * EMD_BLOCK + threadIdx.x;
        const uint2 br_raw = bin_s[i < V ? i : 0u];
        const uint32_t pid_raw = perm[i < V ? i : 0u];                          // (requested with the record, not behind its count)
        const uint2 br = (i < V) ? br_raw : make_uint2(0u, 0u);
        const uint32_t cnt = ((br.x >> 20) & 1023u) * (br.y & 1023u);
#if EMD_DUP_SCAN_OWNER
        // Round 5: the owner of a slot comes from a running maximum over head marks instead of an 8-step binary search per slot (64 dependent
        // LDS reads per thread and Gaussian block: the kernel's longest chain).  Every Gaussian of the block with pairs marks the slot its run
        // starts at (distinct slots), the Gaussian that covers the window's first slot marks slot 0, and one block-wide max-scan over the window
        // (8 consecutive slots per thread + a DPP scan of the threads' maxima) leaves every slot's owner in LDS.
        reinterpret_cast<uint4*>(s_own)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);                 // (ordered before the marks by the scan's barriers)
#endif
        uint32_t total;
        const uint32_t inc = block_scan_add_u32(cnt, s_scan, &total);
        const uint32_t end = base + total;
        s_excl[threadIdx.x] = inc - cnt;
        s_rect[threadIdx.x] = br.x;
        s_id[threadIdx.x] = cnt ? pid_raw : 0u;
        s_hs[threadIdx.x] = br.y & 4095u;
        const uint32_t lo_slot = max(S0, base), hi_slot = min(S1, end);
#if EMD_DUP_SCAN_OWNER
        {
            const uint32_t st = base + inc - cnt;                                                    // first slot of this Gaussian's run
            if (cnt && st >= lo_slot && st < hi_slot) s_own[st - lo_slot] = (uint16_t)(threadIdx.x + 1u);
            // owner of the window's first slot: the last Gaussian with pairs that starts at or in front of it
            const unsigned long long bal = __ballot(cnt && st <= lo_slot);
            if ((threadIdx.x & 63) == 0) s_wmax[4 + (threadIdx.x >> 6)] = bal ? (threadIdx.x & ~63u) + (63u - (uint32_t)__builtin_clzll(bal)) + 1u : 0u;
        }
        __syncthreads();
        {
            const uint4 q = reinterpret_cast<const uint4*>(s_own)[threadIdx.x];
            uint32_t o[8] = {q.x & 0xffffu, q.x >> 16, q.y & 0xffffu, q.y >> 16, q.z & 0xffffu, q.z >> 16, q.w & 0xffffu, q.w >> 16};
            if (threadIdx.x == 0) o[0] = max(o[0], max(max(s_wmax[4], s_wmax[5]), max(s_wmax[6], s_wmax[7])));
#pragma unroll
            for (int k = 1; k < 8; k++) o[k] = max(o[k], o[k - 1]);
            const uint32_t incm = wave_scan_max_u32(o[7]);
            if ((threadIdx.x & 63) == 63) s_wmax[threadIdx.x >> 6] = incm;
            const uint32_t before_lane = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incm, DPP_WAVE_SHR1, 0xf, 0xf, false);   // the lanes in front of this one
            __syncthreads();
            uint32_t pre = before_lane;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) pre = max(pre, s_wmax[w]);
#pragma unroll
            for (int k = 0; k < 8; k++) o[k] = max(o[k], pre);
            reinterpret_cast<uint4*>(s_own)[threadIdx.x] = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
        }
#endif
        __syncthreads();
        for (uint32_t eg = lo_slot + threadIdx.x; eg < hi_slot; eg += EMD_BLOCK) {
            const uint32_t e = eg - base;
#if EMD_DUP_SCAN_OWNER
            const int lo = (int)s_own[eg - lo_slot] - 1;
#else
            // largest j with s_excl[j] <= e  (entries with cnt == 0 share offsets with their successor; the search
            // lands on the last of an equal run, which is the one that owns slot e)
            int lo = 0, hi = EMD_BLOCK - 1;
#pragma unroll
            for (int step = 0; step < 8; step++) {
                int mid = (lo + hi + 1) >> 1;
                if (s_excl[mid] <= e) lo = mid; else hi = mid - 1;
            }
#endif
            const uint32_t local_e = e - s_excl[lo];
            const uint32_t r = s_rect[lo];
            const uint32_t w = (r >> 20) & 1023u, x0 = r & 1023u, y0 = (r >> 10) & 1023u, hs = s_hs[lo];
            const uint32_t ly = local_e / w, lx = local_e - ly * w;
            const uint32_t ty = y0 + ly, tx = x0 + lx;
            const uint32_t tile = ty * (uint32_t)gx + tx;
            // quadrant mask of the pair: everything but the outer quadrant column / row of the rectangle's first / last tile where K1 says so
            const bool xl = !((r >> 30 & 1u) && lx == 0u), xr = !((r >> 31) && lx == w - 1u);
            const bool yt = !((hs >> 10 & 1u) && ly == 0u), yb = !((hs >> 11) && ly == (hs & 1023u) - 1u);
            const uint32_t qm = (xl && yt ? 1u : 0u) | (xr && yt ? 2u : 0u) | (xl && yb ? 4u : 0u) | (xr && yb ? 8u : 0u);
            tkeys[eg] = tile;
            vals[eg] = s_id[lo] | (qm << EMD_ID_BITS);
            atomicAdd(&s_h0[tile & mask0], 1u);
        }
        base = end;
        __syncthreads();   // LDS reused by the next block of Gaussians
    }
    if (hist0) hist0[(size_t)threadIdx.x * nblocks_cap + blockIdx.x] = s_h0[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------
// K4 radix pass on 32-bit keys with 32-bit values: (a) block histograms, (b) scan over [bin][block], (c) stable scatter.
// BITS = 8 (256 digits: the tile passes, the wide depth sort) or 9 (512 digits: the usual three-pass depth sort).
// `offset` is subtracted from every key before the digit is taken (depth bits relative to the near plane's).
// FIRST (first pass of the depth sort): the value of element idx is idx itself, culled Gaussians (key 0xFFFFFFFF) are skipped --
// they take part in neither the counts nor the scatter, so this stable pass also compacts the N Gaussians to the V visible
// ones in index order, and block 0 publishes V for the later passes; keys that do not fit `range_bits` raise bit 1 of the
// overflow word (the host then switches that camera to the wide sort, like a capacity overflow).
// ---------------------------------------------------------------------------------------------------
template <int BITS, bool FIRST>
__global__ void __launch_bounds__(EMD_BLOCK) k_radix_hist(const uint32_t* __restrict__ keys, SortN cnt, int shift, uint32_t mask, uint32_t offset,
                                                          uint32_t nblocks_cap, uint32_t* __restrict__ hist, int range_bits,
                                                          uint32_t* __restrict__ overflow_word) {
    constexpr int BINS = 1 << BITS, PER = BINS / EMD_BLOCK;
    __shared__ uint32_t s_h[BINS];
    const uint32_t D = sort_n(cnt);
    const uint32_t nblocks = (D + EMD_SORT_TILE - 1) / EMD_SORT_TILE;
#pragma unroll
    for (int k = 0; k < PER; k++) s_h[threadIdx.x + k * EMD_BLOCK] = 0;
    __syncthreads();
    if (blockIdx.x < nblocks) {
        const size_t base = (size_t)blockIdx.x * EMD_SORT_TILE;
        bool wide = false;
        // all of the thread's keys first, unconditionally (an index past the end reads the last key and is not counted): with the load inside the
        // guard the compiler waited for every key before asking for the next -- eight trips to memory one after the other per workgroup
        uint32_t kv[EMD_SORT_ITEMS];
#pragma unroll
        for (int k = 0; k < EMD_SORT_ITEMS; k++) {
            const size_t idx = base + (size_t)k * EMD_BLOCK + threadIdx.x;
            kv[k] = keys[idx < D ? idx : (size_t)D - 1];
        }
#pragma unroll
        for (int k = 0; k < EMD_SORT_ITEMS; k++) {
            size_t idx = base + (size_t)k * EMD_BLOCK + threadIdx.x;
            if (idx < D) {
                const uint32_t key = kv[k];
                if (FIRST && key == 0xFFFFFFFFu) continue;
                const uint32_t rel = key - offset;
                if (FIRST && range_bits < 32 && (rel >> range_bits)) wide = true;
                atomicAdd(&s_h[(rel >> shift) & mask], 1u);
            }
        }
        if (FIRST && wide) atomicOr(overflow_word, 2u);
    }
    __syncthreads();
    // bin-major layout over the *capacity* block count so the scan length is launch-time constant
#pragma unroll
    for (int k = 0; k < PER; k++) hist[(size_t)(threadIdx.x + k * EMD_BLOCK) * nblocks_cap + blockIdx.x] = s_h[threadIdx.x + k * EMD_BLOCK];
}

// One workgroup per digit: inclusive scan of that digit's per-block counts (row `bin` of the bin-major table) in place.
// Replaces three launch-bound generic scan launches per pass; the cross-digit offsets are formed in the scatter kernel.
__global__ void __launch_bounds__(EMD_BLOCK) k_radix_scan_bins(uint32_t* __restrict__ hist, uint32_t nblocks_cap) {
    __shared__ uint32_t s[4];
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks_cap;
    uint32_t carry = 0;
    // the counts of the NEXT tile travel while this one is scanned (two barriers and the stores): unconditional loads from clamped positions, masked
    // where they are used -- a row of a few thousand counts was load -> scan -> store, one round trip per 1024 counts in a 5 us kernel
    uint32_t nv[SCAN_ITEMS];
    const uint32_t last = nblocks_cap ? nblocks_cap - 1 : 0u;
    auto request = [&](uint32_t base) {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) nv[k] = row[min(base + threadIdx.x * SCAN_ITEMS + k, last)];
    };
    request(0);
    for (uint32_t base = 0; base < nblocks_cap; base += SCAN_TILE) {
        const uint32_t i0 = base + threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = (i0 + k < nblocks_cap) ? nv[k] : 0u; sum += v[k]; }
        request(base + SCAN_TILE);                    // (past the end: the row's last count again, unused)
        uint32_t total;
        const uint32_t inc = block_scan_add_u32(sum, s, &total);
        uint32_t run = carry + inc - sum;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { run += v[k]; if (i0 + k < nblocks_cap) row[i0 + k] = run; }
        carry += total;
    }
}

// CARRY: a second, 8-byte value travels with every pair (the binning record of the Gaussian; FIRST: read at the element's own index).
template <int BITS, bool FIRST, bool CARRY = false>
__global__ void __launch_bounds__(EMD_BLOCK) k_radix_scatter(const uint32_t* __restrict__ keys_in,
                                                             const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out,
                                                             uint32_t* __restrict__ vals_out, SortN cnt, int shift,
                                                             uint32_t mask, uint32_t offset, uint32_t nblocks_cap,
                                                             const uint32_t* __restrict__ hist_inc, uint32_t* __restrict__ count_out,
                                                             const uint2* __restrict__ carry_in, uint2* __restrict__ carry_out) {
    // wave w of the block owns the contiguous slice [w*512, (w+1)*512) of the block's 2048 keys and walks it in
    // 8 rounds of 64 consecutive keys: order inside the block = (wave, round, lane) = memory order => stable.
    constexpr int BINS = 1 << BITS, PER = BINS / EMD_BLOCK;
    __shared__ uint32_t s_cnt[4][BINS];   // running per-wave digit counts, then per-wave bases
    __shared__ uint32_t s_gbase[BINS];
    __shared__ uint32_t s_keys[EMD_SORT_TILE];
    __shared__ uint32_t s_vals[EMD_SORT_TILE];
    __shared__ uint2 s_carry[CARRY ? EMD_SORT_TILE : 1];
    __shared__ uint32_t s_scan[4];
    const uint32_t D = sort_n(cnt);
    const uint32_t nblocks = (D + EMD_SORT_TILE - 1) / EMD_SORT_TILE;
    if (blockIdx.x >= nblocks) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < PER; j++) s_cnt[k][threadIdx.x + j * EMD_BLOCK] = 0;
    __syncthreads();
    const size_t wbase = (size_t)blockIdx.x * EMD_SORT_TILE + (size_t)wave * (EMD_SORT_TILE / 4);
    uint32_t key[EMD_SORT_ITEMS];
    uint32_t val[EMD_SORT_ITEMS];
    uint32_t rank[EMD_SORT_ITEMS];
    uint2 car[CARRY ? EMD_SORT_ITEMS : 1];
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // all loads of the block first (keys, values, the digit rows of the scanned histogram further down): one round trip, not three
#pragma unroll
    for (int k = 0; k < EMD_SORT_ITEMS; k++) {
        const size_t idx = wbase + (size_t)k * 64 + lane;
        key[k] = idx < D ? keys_in[idx] : 0xFFFFFFFFu;
        val[k] = FIRST ? (uint32_t)idx : (idx < D ? vals_in[idx] : 0u);
        if (CARRY) car[k] = idx < D ? carry_in[idx] : make_uint2(0u, 0u);
    }
    uint32_t h_before[PER], h_tot[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint32_t* row = hist_inc + (size_t)(threadIdx.x * PER + j) * nblocks_cap;       // a thread owns PER consecutive digits
        h_before[j] = blockIdx.x ? row[blockIdx.x - 1] : 0u;
        h_tot[j] = row[nblocks_cap - 1];
    }
#pragma unroll
    for (int k = 0; k < EMD_SORT_ITEMS; k++) {
        const size_t idx = wbase + (size_t)k * 64 + lane;
        bool valid = idx < D;
        if (FIRST) valid = valid && key[k] != 0xFFFFFFFFu;          // culled Gaussian: dropped here
        const uint32_t digit = ((key[k] - offset) >> shift) & mask;
        // lanes with the same digit (invalid lanes form their own class and are ignored)
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const unsigned long long bal = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t before = (uint32_t)__popcll(same & lt_mask);
        const uint32_t prev = s_cnt[wave][digit];          // count from earlier rounds of this wave
        rank[k] = valid ? prev + before : 0xFFFFFFFFu;
        // the highest lane of each class publishes the new count (wave-private row: no atomics, no race)
        const bool leader = valid && ((same >> lane) >> 1) == 0ull;
        __builtin_amdgcn_wave_barrier();
        if (leader) s_cnt[wave][digit] = prev + (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // per-wave bases in the block's digit-sorted order + the block's global base for every digit
    {
        uint32_t c[PER][4], csum[PER], dtot[PER], before[PER];
        uint32_t csum_t = 0, dtot_t = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t d = threadIdx.x * PER + j;       // a thread owns PER consecutive digits
#pragma unroll
            for (int w = 0; w < 4; w++) c[j][w] = s_cnt[w][d];
            csum[j] = c[j][0] + c[j][1] + c[j][2] + c[j][3];
            // keys of digit d in earlier blocks (row-wise inclusive scan) + all keys of smaller digits (row totals)
            before[j] = h_before[j];
            dtot[j] = h_tot[j];
            csum_t += csum[j]; dtot_t += dtot[j];
        }
        uint32_t total;
        uint32_t g = block_scan_add_u32(dtot_t, s_scan, &total) - dtot_t;
        if (FIRST && count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = total;      // V: elements of the later passes
        uint32_t bpre = block_scan_add_u32(csum_t, s_scan, &total) - csum_t;
        __syncthreads();                                    // every thread has read its s_cnt columns
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t d = threadIdx.x * PER + j;
            s_gbase[d] = g + before[j] - bpre;              // global slot = s_gbase[digit] + position in block order
            s_cnt[0][d] = bpre;
            s_cnt[1][d] = bpre + c[j][0];
            s_cnt[2][d] = bpre + c[j][0] + c[j][1];
            s_cnt[3][d] = bpre + c[j][0] + c[j][1] + c[j][2];
            g += dtot[j]; bpre += csum[j];
        }
    }
    __syncthreads();
    // Reorder inside LDS first, then write: consecutive lanes hold consecutive output slots, so every digit run
    // leaves the block as one contiguous segment.  Scattering straight from registers wrote 4-byte fragments
    // of 256 different runs: 2.2x write amplification at the memory side (profiles/r01_pmc_hbm_traffic.csv).
    uint32_t nvalid_w = 0;
#pragma unroll
    for (int k = 0; k < EMD_SORT_ITEMS; k++) {
        if (rank[k] != 0xFFFFFFFFu) {
            const uint32_t digit = ((key[k] - offset) >> shift) & mask;
            const uint32_t pos = s_cnt[wave][digit] + rank[k];
            s_keys[pos] = key[k];
            s_vals[pos] = val[k];
            if (CARRY) s_carry[pos] = car[k];
            nvalid_w++;
        }
    }
    uint32_t nvalid;
    block_scan_add_u32(nvalid_w, s_scan, &nvalid);         // (ends with a barrier: the reordered tile is complete)
#pragma unroll
    for (int k = 0; k < EMD_SORT_ITEMS; k++) {
        const uint32_t pos = threadIdx.x + (uint32_t)k * EMD_BLOCK;
        if (pos < nvalid) {
            const uint32_t kk = s_keys[pos];
            const size_t dst = (size_t)s_gbase[((kk - offset) >> shift) & mask] + pos;
            keys_out[dst] = kk;
            vals_out[dst] = s_vals[pos];
            if (CARRY) carry_out[dst] = s_carry[pos];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K5 tile ranges
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(EMD_BLOCK) k_tile_ranges(const uint32_t* __restrict__ tkeys,
                                                           const EmdStatus* __restrict__ status,
                                                           uint32_t* __restrict__ ranges) {
    const uint32_t D = status->overflow ? 0u : status->num_rendered;
    for (size_t idx = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x; idx < D; idx += (size_t)gridDim.x * EMD_BLOCK) {
        const uint32_t t = tkeys[idx], tp = tkeys[idx ? idx - 1 : 0];          // (both requested before either is used)
        if (idx == 0) ranges[2 * t] = 0;
        else if (tp != t) { ranges[2 * tp + 1] = (uint32_t)idx; ranges[2 * t] = (uint32_t)idx; }
        if (idx == D - 1) ranges[2 * t + 1] = D;
    }
}

// Dispatch order of the render kernels: tiles by descending list length (1024 buckets of 16 entries), so the deep tiles
// start first and the kernel's tail is made of short ones (longest-processing-time-first).  Order inside a bucket is
// arbitrary: it only decides when a tile is rendered, never what is rendered.  One workgroup; above 64 K tiles the
// identity order is kept (plenty of tiles to fill the tail anyway).
// (Round 4, measured and dropped: formed by the last workgroup of the range kernel to finish -- a ticket behind a device-scope release
//  fence.  On gfx950 that fence writes the XCD's L2 back: 4096 workgroups x one fence made the range kernel 0.33 ms instead of 6 us.
//  The same cost sits under every decoupled look-back scheme, which is why the single-pass scans of round 1 lost as well.)
#define ORDER_BUCKETS 1024
__global__ void __launch_bounds__(ORDER_BUCKETS) k_tile_order(const uint32_t* __restrict__ ranges, uint32_t T, uint32_t* __restrict__ order) {
    __shared__ uint32_t s_h[ORDER_BUCKETS];
    __shared__ uint32_t s_w[ORDER_BUCKETS / 64];
    s_h[threadIdx.x] = 0;                                    // one thread per bucket
    __syncthreads();
    // a thread's tiles are requested together and kept for the second pass (as two plain loops the kernel made one trip to memory per tile and
    // pass, fourteen one after the other at 6 700 tiles: it is a single workgroup, nothing else hides them); tiles past 8 192 take the plain loops
    constexpr uint32_t KEEP = 8;
    uint32_t bk[KEEP];
    {
        uint2 rg[KEEP];
#pragma unroll
        for (uint32_t k = 0; k < KEEP; k++) {
            const uint32_t t = threadIdx.x + k * ORDER_BUCKETS;
            rg[k] = reinterpret_cast<const uint2*>(ranges)[t < T ? t : 0];
        }
#pragma unroll
        for (uint32_t k = 0; k < KEEP; k++) {
            const uint32_t t = threadIdx.x + k * ORDER_BUCKETS;
            bk[k] = ORDER_BUCKETS - 1 - min((rg[k].y - rg[k].x) >> 4, (uint32_t)ORDER_BUCKETS - 1);     // bucket 0 = longest
            if (t < T) atomicAdd(&s_h[bk[k]], 1u);
        }
    }
    for (uint32_t t = threadIdx.x + KEEP * ORDER_BUCKETS; t < T; t += ORDER_BUCKETS) {
        const uint32_t len = ranges[2 * t + 1] - ranges[2 * t];
        atomicAdd(&s_h[ORDER_BUCKETS - 1 - min(len >> 4, (uint32_t)ORDER_BUCKETS - 1)], 1u);
    }
    __syncthreads();
    {   // exclusive scan over the buckets: wave scans + the 16 wave totals
        const uint32_t v = s_h[threadIdx.x];
        const uint32_t inc = wave_scan_add_u32(v);
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += s_w[w];
        s_h[threadIdx.x] = base + inc - v;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < KEEP; k++) {
        const uint32_t t = threadIdx.x + k * ORDER_BUCKETS;
        if (t < T) order[atomicAdd(&s_h[bk[k]], 1u)] = t;
    }
    for (uint32_t t = threadIdx.x + KEEP * ORDER_BUCKETS; t < T; t += ORDER_BUCKETS) {
        const uint32_t len = ranges[2 * t + 1] - ranges[2 * t];
        order[atomicAdd(&s_h[ORDER_BUCKETS - 1 - min(len >> 4, (uint32_t)ORDER_BUCKETS - 1)], 1u)] = t;
    }
}
__global__ void __launch_bounds__(EMD_BLOCK) k_tile_order_identity(uint32_t T, uint32_t* __restrict__ order) {
    const uint32_t t = blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (t < T) order[t] = t;
}

// upstream's 64-bit keys, rebuilt for the parity tests: tile id << 32 | depth bits of the Gaussian
__global__ void __launch_bounds__(EMD_BLOCK) k_export_keys(size_t D, const uint32_t* __restrict__ tkeys,
                                                           const uint32_t* __restrict__ vals,
                                                           const uint32_t* __restrict__ depth_key,
                                                           uint64_t* __restrict__ keys, uint32_t* __restrict__ ids,
                                                           uint32_t* __restrict__ quad_masks) {
    const size_t i = (size_t)blockIdx.x * EMD_BLOCK + threadIdx.x;
    if (i >= D) return;
    const uint32_t w = vals[i], id = w & EMD_ID_MASK;          // (the top four bits of a list word are the pair's quadrant mask)
    if (keys) keys[i] = ((uint64_t)tkeys[i] << 32) | depth_key[id];
    if (ids) ids[i] = id;
    if (quad_masks) quad_masks[i] = w >> EMD_ID_BITS;
}

// one stable LSD pass over `n_cap` (launch bound) / sort_n(cnt) (actual) pairs; `skip_hist`: the histogram was built by the producer
template <int BITS, bool FIRST>
int radix_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, SortN cnt, size_t n_cap, int shift,
               int bits, uint32_t offset, uint32_t* hist, bool skip_hist, int range_bits, uint32_t* overflow_word, uint32_t* count_out,
               hipStream_t st, const uint2* carry_in = nullptr, uint2* carry_out = nullptr) {
    const uint32_t nsb = (uint32_t)((n_cap + EMD_SORT_TILE - 1) / EMD_SORT_TILE);
    const uint32_t mask = (1u << bits) - 1u;
    if (!skip_hist) {
        hipLaunchKernelGGL((k_radix_hist<BITS, FIRST>), dim3(nsb), dim3(EMD_BLOCK), 0, st, kin, cnt, shift, mask, offset, nsb, hist, range_bits,
                           overflow_word);
        EMD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_radix_scan_bins, dim3(1u << BITS), dim3(EMD_BLOCK), 0, st, hist, nsb);
    EMD_LAUNCH_CHECK();
    if (carry_in) hipLaunchKernelGGL((k_radix_scatter<BITS, FIRST, true>), dim3(nsb), dim3(EMD_BLOCK), 0, st, kin, vin, kout, vout, cnt, shift, mask, offset, nsb,
                                     hist, count_out, carry_in, carry_out);
    else hipLaunchKernelGGL((k_radix_scatter<BITS, FIRST, false>), dim3(nsb), dim3(EMD_BLOCK), 0, st, kin, vin, kout, vout, cnt, shift, mask, offset, nsb, hist,
                            count_out, carry_in, carry_out);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

}  // namespace

int emd_launch_binning(const EmdSettings& s, int flags, int N, const GeomWs& g, const BinWs& b, int64_t capacity, EmdStatus* status,
                       hipStream_t st) {
    const int gx = (s.image_width + EMD_TILE_X - 1) / EMD_TILE_X, gy = (s.image_height + EMD_TILE_Y - 1) / EMD_TILE_Y;
    const int T = gx * gy;
    if (gx >= 1024 || gy >= 1024) { emd_set_error("image too large: %d x %d tiles (max 1023)", gx, gy); return EMD_ERR_INVALID; }
    const int nb = (N + EMD_BLOCK - 1) / EMD_BLOCK;
    int rc;
    emd_prof_switch(PROF_PREPROCESS, PROF_SORT, st);
    if (N <= 0) {           // nothing to sort: empty ranges, identity order (the status block was not cleared by a projection kernel)
        { int zrc = emd_zero_async(status, sizeof(EmdStatus), st); if (zrc) return zrc; }
        { int zrc = emd_zero_async(b.ranges, (size_t)T * 8, st); if (zrc) return zrc; }
        hipLaunchKernelGGL(k_tile_order_identity, dim3((T + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, (uint32_t)T, b.tile_order);
        EMD_LAUNCH_CHECK();
        emd_prof_switch(PROF_SORT, PROF_RANGES, st);
        return EMD_OK;
    }
    // 1. visible Gaussians in depth order.  (The status word was cleared by K1: the depth passes may raise its overflow bit 1.)
    uint32_t* const sort_count = &status->reserved;          // V after the first (compacting) depth pass lives in the status block
    const bool wide = (flags & EMD_FLAG_WIDE_DEPTH_SORT) != 0;
    const int depth_passes = wide ? EMD_DEPTH_PASSES_WIDE : EMD_DEPTH_PASSES_NARROW;
    {
        const SortN c0 = {nullptr, nullptr, (uint32_t)N}, cv = {sort_count, nullptr, 0u};
        uint32_t near_bits = 0;
        if (!wide) { const float np = s.near_plane > 0.f ? s.near_plane : 0.f; memcpy(&near_bits, &np, 4); }
        uint32_t* ovf = &status->overflow;
        // the binning records travel with the pairs (EMD_BIN_CARRY): pass p writes them to cb[p & 1], chosen so that the last pass lands in bin_s
        uint2* const cb[2] = {(depth_passes & 1) ? g.bin_s : g.bin_t, (depth_passes & 1) ? g.bin_t : g.bin_s};
        const bool carry = EMD_BIN_CARRY != 0;
        if (wide) {
            rc = radix_pass<8, true>(g.depth_key, nullptr, g.gkeys[0], g.gvals[0], c0, (size_t)N, 0, 8, 0u, g.ghist, false, 32, ovf, sort_count, st,
                                     carry ? g.binrec : nullptr, cb[0]);
            for (int p = 1; p < depth_passes && !rc; p++)
                rc = radix_pass<8, false>(g.gkeys[(p - 1) & 1], g.gvals[(p - 1) & 1], g.gkeys[p & 1], g.gvals[p & 1], cv, (size_t)N, 8 * p, 8, 0u,
                                          g.ghist, false, 32, ovf, nullptr, st, carry ? cb[(p - 1) & 1] : nullptr, cb[p & 1]);
        } else {
            const int B = EMD_DEPTH_BITS_NARROW;
            rc = radix_pass<EMD_DEPTH_BITS_NARROW, true>(g.depth_key, nullptr, g.gkeys[0], g.gvals[0], c0, (size_t)N, 0, B, near_bits, g.ghist, false,
                                                         EMD_DEPTH_RANGE_NARROW, ovf, sort_count, st, carry ? g.binrec : nullptr, cb[0]);
            for (int p = 1; p < depth_passes && !rc; p++)
                rc = radix_pass<EMD_DEPTH_BITS_NARROW, false>(g.gkeys[(p - 1) & 1], g.gvals[(p - 1) & 1], g.gkeys[p & 1], g.gvals[p & 1], cv, (size_t)N,
                                                              B * p, B, near_bits, g.ghist, false, 32, ovf, nullptr, st,
                                                              carry ? cb[(p - 1) & 1] : nullptr, cb[p & 1]);
        }
        if (rc) return rc;
    }
    const uint32_t* perm = g.gvals[(depth_passes - 1) & 1];
    // 2. the binning records in depth order + block sums; duplicate: every workgroup scans the block sums, finds its 2048 output slots and
    //    writes their (tile, Gaussian | quadrant mask) pairs; workgroup 0 publishes D / overflow / V
    emd_prof_switch(PROF_SORT, PROF_DUPLICATE, st);
    const int passes = emd_tile_passes(T), bits = emd_tile_pass_bits(T);
    if (EMD_BIN_CARRY) hipLaunchKernelGGL((k_sorted_counts<true>), dim3(nb), dim3(EMD_BLOCK), 0, st, N, sort_count, perm, g.binrec, g.bin_s, g.block_sums, b.ranges,
                                          (uint32_t)(2 * T));
    else hipLaunchKernelGGL((k_sorted_counts<false>), dim3(nb), dim3(EMD_BLOCK), 0, st, N, sort_count, perm, g.binrec, g.bin_s, g.block_sums, b.ranges,
                            (uint32_t)(2 * T));
    EMD_LAUNCH_CHECK();
    if (capacity <= 0) {    // D and V are still reported (capacity 0 is how callers size the workspace)
        hipLaunchKernelGGL(k_duplicate, dim3(1), dim3(EMD_BLOCK), 0, st, N, gx, g.bin_s, perm, g.block_sums, (uint64_t)0, status, b.tkeys[0], b.vals[0],
                           (uint32_t*)nullptr, 0u, 1u);
        EMD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_tile_order_identity, dim3((T + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, (uint32_t)T, b.tile_order);
        EMD_LAUNCH_CHECK();
        emd_prof_switch(PROF_DUPLICATE, PROF_RANGES, st);
        return EMD_OK;
    }
    const uint32_t nslot = (uint32_t)(((size_t)capacity + DUP_SLOTS - 1) / DUP_SLOTS);
    // (the duplicate kernel also builds the digit histogram of the first tile pass: its 2048-slot output blocks are that pass's blocks)
    hipLaunchKernelGGL(k_duplicate, dim3(nslot), dim3(EMD_BLOCK), 0, st, N, gx, g.bin_s, perm, g.block_sums, (uint64_t)capacity, status,
                       b.tkeys[0], b.vals[0], passes > 0 ? b.hist : nullptr, passes > 0 ? (1u << bits) - 1u : 0u, nslot);
    EMD_LAUNCH_CHECK();
    // 3. stable partition by tile id
    emd_prof_switch(PROF_DUPLICATE, PROF_SORT, st);
    const SortN cd = {&status->num_rendered, &status->overflow, 0u};
    int cur = 0;
    for (int p = 0; p < passes; p++) {
        rc = radix_pass<8, false>(b.tkeys[cur], b.vals[cur], b.tkeys[cur ^ 1], b.vals[cur ^ 1], cd, (size_t)capacity, p * bits, bits, 0u, b.hist,
                                  p == 0, 32, nullptr, nullptr, st);
        if (rc) return rc;
        cur ^= 1;
    }
    emd_prof_switch(PROF_SORT, PROF_RANGES, st);
    const unsigned rb = (unsigned)(((size_t)capacity + EMD_BLOCK - 1) / EMD_BLOCK);
    hipLaunchKernelGGL(k_tile_ranges, dim3(rb < 4096u ? rb : 4096u), dim3(EMD_BLOCK), 0, st, b.tkeys[cur], status, b.ranges);
    EMD_LAUNCH_CHECK();
    if (T <= 65536) hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(ORDER_BUCKETS), 0, st, b.ranges, (uint32_t)T, b.tile_order);
    else hipLaunchKernelGGL(k_tile_order_identity, dim3((T + EMD_BLOCK - 1) / EMD_BLOCK), dim3(EMD_BLOCK), 0, st, (uint32_t)T, b.tile_order);
    EMD_LAUNCH_CHECK();
    return EMD_OK;   // PROF_RANGES is closed by the render-forward switch
}

int emd_launch_export_keys(int64_t D, const GeomWs& g, const BinWs& b, uint64_t* keys, uint32_t* ids, uint32_t* quad_masks, hipStream_t st) {
    if (D <= 0) return EMD_OK;
    hipLaunchKernelGGL(k_export_keys, dim3((unsigned)((D + EMD_BLOCK - 1) / EMD_BLOCK)), dim3(EMD_BLOCK), 0, st, (size_t)D,
                       b.tkeys[b.sorted_buf], b.vals[b.sorted_buf], g.depth_key, keys, ids, quad_masks);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}
