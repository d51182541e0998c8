// radix_sort.hip — stable LSD radix argsort of u32 key/value pairs (replaces crates/brush-sort).
//
// Reference: radix_argsort() crates/brush-sort/src/lib.rs:32-147 — FidelityFX-style, 4 bits per
// pass, 5 launches per pass (count / reduce / scan / scan_add / scatter), element count read
// from a GPU buffer, only the low `sorting_bits` sorted.  32-bit keys = 40 launches.
//
// gfx950 design: 8-bit digits (half the passes), 3 launches per pass, wave64 ranking.
//   k_upsweep   : 256-thread block per tile; per-wave LDS histograms -> counts[digit][tile]
//   k_scan      : one block per digit scans counts[digit][*] over tiles in place, writes totals[]
//   k_downsweep : re-reads the tile (coalesced); each wave ranks its contiguous chunk with ballot
//                 match-any masks (8 x v_cmp per key on 64-bit masks), the lowest peer lane bumps
//                 the wave's private LDS counter; 4-wave combine; the 256 digit bases come from a
//                 block scan of totals[]; scatter.
// The element count stays on the device (*d_n): grids are sized for max_n and surplus blocks exit
// on their first instruction.  The tile size is also chosen ON THE DEVICE from *d_n (1, 2, 4, 8 or
// 16 keys per lane, the smallest that keeps the tile count <= 256 — measured optimum of 64..2048): a 100 k-key sort then spreads
// over ~200 workgroups instead of 25, and an 8 M-key sort still uses 4096-key tiles.  All three
// kernels derive the same value, so the partition is consistent.
// Stability: tiles, wave chunks, rounds and lanes are all ranked in index order.
// Traffic per pass: 4 B/key (upsweep) + 16 B/pair (downsweep).  Roofline: HBM at large n, launch
// latency at n ~ 100 k.  The pass structure sorts exactly the reference's 4*ceil(bits/4) low bits.
#include "common.hpp"

namespace brush {
namespace {

constexpr uint32_t kSortThreads = 256;
constexpr uint32_t kSortWaves = kSortThreads / kWave;
constexpr uint32_t kSortMaxItems = 16;
constexpr uint32_t kSortTargetTiles = 256;
constexpr uint32_t kRadix = 256;

// Keys per lane for a sort of n keys: smallest power of two K in [1,16] with n/(256 K) <= 256.
__host__ __device__ __forceinline__ uint32_t sort_items(uint32_t n) {
    uint32_t k = 1;
    while (k < kSortMaxItems && (uint64_t)kSortThreads * k * kSortTargetTiles < n) k <<= 1;
    return k;
}
// Upper bound of the tile count over every n <= max_n (size of one row of the counts table).
inline uint32_t sort_max_tiles(uint32_t max_n) {
    uint32_t best = 1;
    for (uint32_t k = 1; k <= kSortMaxItems; k <<= 1) {
        // largest n that still selects k
        uint64_t hi = (k == kSortMaxItems) ? max_n : (uint64_t)kSortThreads * k * kSortTargetTiles;
        if (hi > max_n) hi = max_n;
        const uint32_t tiles = (uint32_t)((hi + (uint64_t)kSortThreads * k - 1) / ((uint64_t)kSortThreads * k));
        if (tiles > best) best = tiles;
    }
    return best;
}

__global__ __launch_bounds__(kSortThreads) void k_sort_upsweep(const uint32_t *__restrict__ keys,
                                                              const uint32_t *__restrict__ d_n,
                                                              uint32_t max_n, uint32_t shift, uint32_t mask,
                                                              uint32_t *__restrict__ counts,
                                                              uint32_t max_tiles) {
    const uint32_t n = min(*d_n, max_n);
    const uint32_t items = sort_items(n);
    const uint32_t tile_keys = kSortThreads * items;
    const uint32_t tile = blockIdx.x;
    if ((uint64_t)tile * tile_keys >= n) return;
    __shared__ uint32_t hist[kSortWaves][kRadix];
    for (uint32_t i = threadIdx.x; i < kSortWaves * kRadix; i += kSortThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t wid = threadIdx.x / kWave;
    const uint32_t base = tile * tile_keys;
    for (uint32_t i = 0; i < items; i++) {
        const uint32_t idx = base + i * kSortThreads + threadIdx.x;
        if (idx < n) atomicAdd(&hist[wid][(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    const uint32_t d = threadIdx.x;
    counts[(size_t)d * max_tiles + tile] = hist[0][d] + hist[1][d] + hist[2][d] + hist[3][d];
}

// Block d: exclusive scan over tiles of counts[d][*]; totals[d] = number of keys with digit d.
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t *__restrict__ counts,
                                                   const uint32_t *__restrict__ d_n, uint32_t max_n,
                                                   uint32_t max_tiles, uint32_t *__restrict__ totals) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry_s;
    const uint32_t n = min(*d_n, max_n);
    const uint32_t tile_keys = kSortThreads * sort_items(n);
    const uint32_t num_tiles = (n + tile_keys - 1) / tile_keys;
    uint32_t *row = counts + (size_t)blockIdx.x * max_tiles;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < num_tiles; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < num_tiles ? row[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane_id() == 63) wave_tot[threadIdx.x / kWave] = incl;
        __syncthreads();
        uint32_t off = carry_s;
        for (uint32_t w = 0; w < threadIdx.x / kWave; w++) off += wave_tot[w];
        if (i < num_tiles) row[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

template <uint32_t ITEMS>
__device__ __forceinline__ void downsweep_body(const uint32_t *__restrict__ keys_in,
                                               const uint32_t *__restrict__ vals_in,
                                               uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                               uint32_t n, uint32_t shift, uint32_t mask,
                                               const uint32_t *__restrict__ counts,
                                               const uint32_t *__restrict__ totals, uint32_t max_tiles,
                                               uint32_t (*wave_hist)[kRadix], uint32_t *digit_base,
                                               uint32_t *wave_tot) {
    const uint32_t tile = blockIdx.x;
    const uint32_t wid = threadIdx.x / kWave;
    const uint32_t lane = lane_id();
    for (uint32_t i = threadIdx.x; i < kSortWaves * kRadix; i += kSortThreads) (&wave_hist[0][0])[i] = 0;

    // Global base of digit d for this tile = (#keys with smaller digit) + (same digit, earlier tiles).
    {
        const uint32_t d = threadIdx.x;
        const uint32_t v = totals[d];
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t off = incl - v;
        for (uint32_t w = 0; w < wid; w++) off += wave_tot[w];
        digit_base[d] = off + counts[(size_t)d * max_tiles + tile];
    }

    // Each wave owns a contiguous chunk of 64*ITEMS keys; round i covers keys chunk + i*64 + lane.
    const uint32_t chunk = tile * (kSortThreads * ITEMS) + wid * (kWave * ITEMS);
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        const uint32_t idx = chunk + i * kWave + lane;
        key[i] = idx < n ? keys_in[idx] : 0u;
        val[i] = idx < n ? vals_in[idx] : 0u;
    }
    __syncthreads();  // wave_hist zeroed

    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        const uint32_t idx = chunk + i * kWave + lane;
        const bool valid = idx < n;
        const uint32_t digit = (key[i] >> shift) & mask;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t vote = __ballot(valid && bit);
            peers &= bit ? vote : ~vote;
        }
        const uint32_t below = __popcll(peers & lt);
        const uint32_t cnt = __popcll(peers);
        const int leader = valid ? (__ffsll((long long)peers) - 1) : 0;
        uint32_t old = 0;
        if (valid && (int)lane == leader) {
            old = wave_hist[wid][digit];
            wave_hist[wid][digit] = old + cnt;
        }
        old = __shfl(old, leader, 64);
        rank[i] = old + below;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        const uint32_t d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (uint32_t w = 0; w < kSortWaves; w++) {
            const uint32_t t = wave_hist[w][d];
            wave_hist[w][d] = run;
            run += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        const uint32_t idx = chunk + i * kWave + lane;
        if (idx < n) {
            const uint32_t digit = (key[i] >> shift) & mask;
            const uint32_t pos = digit_base[digit] + wave_hist[wid][digit] + rank[i];
            keys_out[pos] = key[i];
            vals_out[pos] = val[i];
        }
    }
}

__global__ __launch_bounds__(kSortThreads) void k_sort_downsweep(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ d_n,
    uint32_t max_n, uint32_t shift, uint32_t mask, const uint32_t *__restrict__ counts,
    const uint32_t *__restrict__ totals, uint32_t max_tiles) {
    const uint32_t n = min(*d_n, max_n);
    const uint32_t items = sort_items(n);
    if ((uint64_t)blockIdx.x * kSortThreads * items >= n) return;

    __shared__ uint32_t wave_hist[kSortWaves][kRadix];
    __shared__ uint32_t digit_base[kRadix];
    __shared__ uint32_t wave_tot[kSortWaves];
#define BRUSH_DOWN(K)                                                                                        \
    downsweep_body<K>(keys_in, vals_in, keys_out, vals_out, n, shift, mask, counts, totals, max_tiles, wave_hist, \
                      digit_base, wave_tot)
    switch (items) {  // block-uniform
        case 1: BRUSH_DOWN(1); break;
        case 2: BRUSH_DOWN(2); break;
        case 4: BRUSH_DOWN(4); break;
        case 8: BRUSH_DOWN(8); break;
        default: BRUSH_DOWN(16); break;
    }
#undef BRUSH_DOWN
}

__global__ void k_sort_copy(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                            uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                            const uint32_t *__restrict__ d_n, uint32_t max_n) {
    const uint32_t n = min(*d_n, max_n);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        keys_out[i] = keys_in[i];
        vals_out[i] = vals_in[i];
    }
}

struct SortWs {
    uint32_t *tmp_keys, *tmp_vals, *counts, *totals;
    uint32_t max_tiles;
    size_t bytes;
};

SortWs carve_sort(void *ws, uint32_t max_n) {
    SortWs w;
    Carver c(ws);
    w.max_tiles = sort_max_tiles(max_n ? max_n : 1);
    w.tmp_keys = c.take<uint32_t>(max_n ? max_n : 1);
    w.tmp_vals = c.take<uint32_t>(max_n ? max_n : 1);
    w.counts = c.take<uint32_t>((size_t)kRadix * w.max_tiles);
    w.totals = c.take<uint32_t>(kRadix);
    w.bytes = c.bytes();
    return w;
}

}  // namespace

size_t sort_workspace_bytes(uint32_t max_n) { return carve_sort(nullptr, max_n).bytes; }

hipError_t sort_launch(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out,
                       uint32_t *vals_out, const uint32_t *d_n, uint32_t max_n, uint32_t bits, void *ws,
                       hipStream_t s) {
    if (max_n == 0) return hipSuccess;
    const SortWs w = carve_sort(ws, max_n);
    const uint32_t total_bits = 4u * ((bits + 3u) / 4u);  // brush-sort/src/lib.rs:58
    const uint32_t passes = (total_bits + 7u) / 8u;
    if (passes == 0) {
        hipLaunchKernelGGL(k_sort_copy, dim3(min(w.max_tiles, 2048u)), dim3(256), 0, s, keys_in, vals_in,
                           keys_out, vals_out, d_n, max_n);
        return hipGetLastError();
    }
    const uint32_t *src_k = keys_in, *src_v = vals_in;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t shift = p * 8u;
        const uint32_t width = min(8u, total_bits - shift);
        const uint32_t mask = (1u << width) - 1u;
        const bool to_out = ((passes - 1 - p) % 2u) == 0;
        uint32_t *dst_k = to_out ? keys_out : w.tmp_keys;
        uint32_t *dst_v = to_out ? vals_out : w.tmp_vals;
        hipLaunchKernelGGL(k_sort_upsweep, dim3(w.max_tiles), dim3(kSortThreads), 0, s, src_k, d_n, max_n, shift,
                           mask, w.counts, w.max_tiles);
        hipLaunchKernelGGL(k_sort_scan, dim3(kRadix), dim3(256), 0, s, w.counts, d_n, max_n, w.max_tiles,
                           w.totals);
        hipLaunchKernelGGL(k_sort_downsweep, dim3(w.max_tiles), dim3(kSortThreads), 0, s, src_k, src_v, dst_k,
                           dst_v, d_n, max_n, shift, mask, w.counts, w.totals, w.max_tiles);
        src_k = dst_k;
        src_v = dst_v;
    }
    return hipGetLastError();
}

}  // namespace brush

using namespace brush;

extern "C" int brush_radix_argsort_workspace_size(uint32_t max_n, size_t *bytes) {
    if (!bytes) return BRUSH_ERR_INVALID_ARG;
    *bytes = sort_workspace_bytes(max_n);
    return BRUSH_OK;
}

extern "C" int brush_radix_argsort_u32(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out,
                                       uint32_t *vals_out, const uint32_t *d_n, uint32_t max_n,
                                       uint32_t sorting_bits, void *workspace, size_t workspace_bytes,
                                       brush_stream_t stream) {
    if (sorting_bits > 32) return BRUSH_ERR_INVALID_ARG;  // brush-sort/src/lib.rs:38-39
    if (max_n == 0) return BRUSH_OK;
    if (!keys_in || !vals_in || !keys_out || !vals_out || !d_n || !workspace) return BRUSH_ERR_INVALID_ARG;
    if (keys_in == keys_out || vals_in == vals_out) return BRUSH_ERR_INVALID_ARG;
    if (workspace_bytes < sort_workspace_bytes(max_n)) return BRUSH_ERR_WORKSPACE_SMALL;
    BRUSH_HIP_CHECK(sort_launch(keys_in, vals_in, keys_out, vals_out, d_n, max_n, sorting_bits, workspace,
                                static_cast<hipStream_t>(stream)));
    return BRUSH_OK;
}
