// radix_sort.hip — stable LSD radix argsort of u32 key/value pairs (replaces crates/brush-sort).
//
// Reference: radix_argsort() crates/brush-sort/src/lib.rs:32-147 — FidelityFX-style, 4 bits per
// pass, 5 launches per pass (count / reduce / scan / scan_add / scatter), element count read
// from a GPU buffer, only the low `sorting_bits` sorted.  32-bit keys = 40 launches.
//
// gfx950 design: 8-bit digits (half the passes), 1024-thread workgroups (16 waves), tiles of
// 1024 x {1,2,4,8,16} keys chosen ON THE DEVICE from *d_n: at most 128 tiles while that fits (a 103 k-key
// sort: 101 tiles of 1024 keys, a 419 k-key sort: 103 of 4096), 16384-key tiles when the sort is large.
//   k_upsweep   : per-wave LDS histograms -> counts of the tile's 256 digits
//   k_scan      : (large sorts only) one block per digit scans counts[digit][*] over the tiles
//   k_downsweep : every wave ranks its contiguous chunk with ballot match-any masks (8 x v_cmp per
//                 key on 64-bit masks, the lowest peer lane bumps the wave's private LDS counter),
//                 16-wave combine, then the pairs are REORDERED BY DIGIT IN LDS and written out so
//                 that consecutive lanes write consecutive addresses of a digit run (round 1 scattered
//                 one dword per lane into 256 runs: 2.1x write amplification at the HBM counters).
// Two launch shapes, chosen on the host from max_n:
//   * FUSED (max_n <= 512 tiles of 16384 keys, i.e. the reference's 8.39 M intersection cap and below):
//     2 launches per pass.  counts are laid out [tile][digit]; every downsweep block sums the rows of the
//     tiles before it and of all tiles itself (T coalesced 1-KiB row reads, T = the ACTUAL tile count of
//     *d_n, ~26-64 at the headline scene) instead of waiting for a scan launch: at 100 k - 400 k keys a
//     pass is bound by kernel boundaries (~4 us each), not by bytes.
//   * otherwise 3 launches per pass with counts laid out [digit][tile] and the scan kernel, fixed 8192-key tiles and
//     512-thread scatter workgroups whose 76 KiB of LDS let two of them share a CU (one's loads and stores overlap the
//     other's ranking): the 18 M-key tile sort of the 4K / 20 M-splat frame 326 -> 246 us (37 % of the HBM peak);
//     256-thread / 4096-key workgroups (four per CU) measured slower (278 us: shorter digit runs, twice the tiles).
// The element count stays on the device (*d_n): grids are sized for max_n and surplus blocks exit on
// their first instruction.  Stability: tiles, wave chunks, rounds and lanes are all ranked in index order.
// Traffic per pass: 4 B/key (upsweep) + 16 B/pair (downsweep).  Roofline: HBM at large n, kernel
// boundaries at n ~ 100 k.  The pass structure sorts exactly the reference's 4*ceil(bits/4) low bits.
#include "common.hpp"
#include "trace.hpp"

namespace brush {
namespace {

constexpr uint32_t kSortThreads = 1024;
constexpr uint32_t kSortWaves = kSortThreads / kWave;  // 16
constexpr uint32_t kSortMaxItems = 16;
// Measured at the headline scene (sweep of the keys per lane): the 103 k-key depth sort is fastest with 1 key per
// lane (101 tiles), the 419 k-key tile sort with 4 (103 tiles): ~100 workgroups either way.
constexpr uint32_t kSortTargetTiles = 128;
constexpr uint32_t kRadix = 256;
constexpr uint32_t kMaxTileKeys = kSortThreads * kSortMaxItems;  // 16384
constexpr uint32_t kFusedMaxTiles = 512;
// Sorts larger than that (3-launch shape): fixed 8192-key tiles, scattered by 512-thread workgroups of 16 keys per
// lane whose LDS image (74 KiB) lets TWO of them share a CU, so one's loads and stores overlap the other's ranking.
constexpr uint32_t kBigTileKeys = 8192;
constexpr uint32_t kBigThreads = 512;
// FUSED scatter: count-table rows every thread requests before the element count has arrived (4 quarters x 32 rows =
// the first kSortTargetTiles rows)
// Thread (group g8 = tid / 128, digit pair dp = tid % 128) owns rows g8 + 8 j of the 16-bit table: 16 dword loads
// cover digits 2 dp and 2 dp + 1 of the first 128 rows (32 registers of 32-bit counts spilled the 16-key variant).
constexpr uint32_t kSpecRows = 16;
constexpr uint32_t kTableGroups = 8;
static_assert(kTableGroups * kSpecRows == kSortTargetTiles, "the table always has the speculatively loaded rows");

// Keys per lane for a sort of n keys: smallest power of two K in [1,16] with n / (1024 K) <= 128.
__host__ __device__ __forceinline__ uint32_t sort_items(uint32_t n) {
    uint32_t k = 1;
    while (k < kSortMaxItems && (uint64_t)kSortThreads * k * kSortTargetTiles < n) k <<= 1;
    return k;
}
// Upper bound of the tile count over every n <= max_n.
inline uint32_t sort_max_tiles(uint32_t max_n) {
    const uint32_t big = (uint32_t)(((uint64_t)max_n + kMaxTileKeys - 1) / kMaxTileKeys);
    if (big > kFusedMaxTiles) return (uint32_t)(((uint64_t)max_n + kBigTileKeys - 1) / kBigTileKeys);
    return big > kSortTargetTiles ? big : kSortTargetTiles;  // n <= 128 * 1024 * K selects K: never more than 128 tiles
}
inline bool sort_is_fused(uint32_t max_n) {
    return (uint32_t)(((uint64_t)max_n + kMaxTileKeys - 1) / kMaxTileKeys) <= kFusedMaxTiles;
}

template <bool FUSED>
__global__ __launch_bounds__(kSortThreads) void k_sort_upsweep(const uint32_t *__restrict__ keys,
                                                              const uint32_t *__restrict__ d_n, uint32_t max_n,
                                                              uint32_t shift, uint32_t mask,
                                                              uint32_t *__restrict__ counts, uint32_t max_tiles) {
    BRUSH_KTRACE(kTrSortUp, shift | (((shift >> 3) + (max_n > (1u << 21) ? 4u : 0u)) << 24));
    const uint32_t n = min(*d_n, max_n);
    BRUSH_KTRACE_MARK(1, n);
    const uint32_t items = FUSED ? sort_items(n) : kBigTileKeys / kSortThreads;
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
        if (idx < n) {
            const uint32_t k = keys[idx];
            if (i == 0) BRUSH_KTRACE_MARK(2, k);
            atomicAdd(&hist[wid][(k >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    BRUSH_KTRACE_MARK(3, n);
    if (threadIdx.x < kRadix) {
        const uint32_t d = threadIdx.x;
        uint32_t c = 0;
#pragma unroll
        for (uint32_t w = 0; w < kSortWaves; w++) c += hist[w][d];
        // FUSED: 16-bit counts (a tile holds at most 16384 keys), row = 128 dwords of digit pairs
        if (FUSED) reinterpret_cast<uint16_t *>(counts)[(size_t)tile * kRadix + d] = (uint16_t)c;
        else counts[(size_t)d * max_tiles + tile] = c;
    }
}

// Large sorts: block d = exclusive scan over tiles of counts[d][*]; totals[d] = number of keys with digit d.
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t *__restrict__ counts,
                                                   const uint32_t *__restrict__ d_n, uint32_t max_n,
                                                   uint32_t max_tiles, uint32_t *__restrict__ totals) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry_s;
    const uint32_t n = min(*d_n, max_n);
    const uint32_t num_tiles = (n + kBigTileKeys - 1) / kBigTileKeys;
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

// LDS of the downsweep: the tile's pairs reordered by digit (128 KiB at 16 keys per lane) + the per-wave
// digit counters.  One 1024-thread workgroup per CU.
template <uint32_t TILE_KEYS, uint32_t WAVES, uint32_t PARTS>
struct DownLdsT {
    uint32_t keys[TILE_KEYS];
    uint32_t vals[TILE_KEYS];
    uint32_t wave_hist[WAVES][kRadix];
    uint32_t digit_base[kRadix];  // global position of the tile's first key of each digit
    uint32_t tile_start[kRadix];  // position inside the tile (after the reorder) of each digit's run
    uint32_t wave_tot2[4][2];
    // FUSED: partial column sums of the count table per row group (keys of earlier tiles / of all tiles).  They live
    // from before barrier A to barrier C, the reordered pairs from barrier C on: same storage.
    static_assert(2 * PARTS * kRadix <= TILE_KEYS, "part / part_all alias keys[]");
    __device__ __forceinline__ uint32_t (*part())[kRadix] { return reinterpret_cast<uint32_t(*)[kRadix]>(keys); }
    __device__ __forceinline__ uint32_t (*part_all())[kRadix] {
        return reinterpret_cast<uint32_t(*)[kRadix]>(keys + PARTS * kRadix);
    }
};
using DownLds = DownLdsT<kMaxTileKeys, kSortWaves, kTableGroups>;
using DownLdsBig = DownLdsT<kBigTileKeys, kBigThreads / kWave, 1>;
static_assert(2 * sizeof(DownLdsBig) <= 160 * 1024, "two large-sort workgroups per CU");

// Exclusive scans of two 256-entry arrays held by threads 0..255 (one value of each per thread) behind the same
// two barriers; all 1024 threads must call.
__device__ __forceinline__ void block_excl_scan256_pair(uint32_t &a, uint32_t &b, uint32_t (*wave_tot)[2]) {
    const uint32_t ia = wave_inclusive_scan(a), ib = wave_inclusive_scan(b);
    if (threadIdx.x < kRadix && lane_id() == 63) {
        wave_tot[threadIdx.x / kWave][0] = ia;
        wave_tot[threadIdx.x / kWave][1] = ib;
    }
    __syncthreads();
    uint32_t oa = ia - a, ob = ib - b;
    for (uint32_t w = 0; w < threadIdx.x / kWave && w < 4; w++) {
        oa += wave_tot[w][0];
        ob += wave_tot[w][1];
    }
    __syncthreads();
    a = oa, b = ob;
}

// Sums of the speculatively loaded count-table rows (thread = (group g8, digit pair dp), rows g8 + 8 j): keys of the
// pair's digits in the tiles before `tile` / in all `num_tiles` tiles, as packed 16-bit-pair sums of at most 16 rows
// each (16 * 16384 overflows 16 bits, so the halves are split before they are added up).
struct PairSums {
    uint32_t below[2], all[2];
};
// FUSED: the partial column sums of the [tile][digit] count table -> L.part() / L.part_all().  Rows 0 .. 127 were
// requested by the kernel before it knew n (`spec`, masked here); the rest of a large table (more than 128 tiles: sorts
// above 2 M keys) is read with kTableLoads independent row loads in flight per thread (latency-, not bandwidth-bound).
template <typename LDS>
__device__ __forceinline__ void table_sums_to_lds(const uint32_t (&spec)[kSpecRows], const uint32_t *__restrict__ counts,
                                                  uint32_t tile, uint32_t num_tiles, LDS &L, KTraceRef trace) {
    const uint32_t dp = threadIdx.x & (kRadix / 2 - 1), g8 = threadIdx.x / (kRadix / 2);
    PairSums ps{};
#pragma unroll
    for (uint32_t j = 0; j < kSpecRows; j++) {
        const uint32_t t = g8 + kTableGroups * j;
        const uint32_t a = t < num_tiles ? spec[j] : 0u;
        const uint32_t b = t < tile ? spec[j] : 0u;  // tile < num_tiles
        ps.all[0] += a & 0xFFFFu, ps.all[1] += a >> 16;
        ps.below[0] += b & 0xFFFFu, ps.below[1] += b >> 16;
    }
    constexpr uint32_t kTableLoads = 16;
    for (uint32_t t0 = kTableGroups * kSpecRows + g8; t0 < num_tiles; t0 += kTableGroups * kTableLoads) {
        uint32_t c[kTableLoads];
#pragma unroll
        for (uint32_t j = 0; j < kTableLoads; j++) {
            const uint32_t t = t0 + kTableGroups * j;
            c[j] = t < num_tiles ? counts[(size_t)t * (kRadix / 2) + dp] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kTableLoads; j++) {
            const uint32_t b = (t0 + kTableGroups * j) < tile ? c[j] : 0u;
            ps.all[0] += c[j] & 0xFFFFu, ps.all[1] += c[j] >> 16;
            ps.below[0] += b & 0xFFFFu, ps.below[1] += b >> 16;
        }
    }
    BRUSH_KTRACE_MARK_VIA(trace, 2, ps.below[0] + ps.all[1]);
    L.part()[g8][2 * dp] = ps.below[0], L.part()[g8][2 * dp + 1] = ps.below[1];
    L.part_all()[g8][2 * dp] = ps.all[0], L.part_all()[g8][2 * dp + 1] = ps.all[1];
}

template <bool FUSED, uint32_t ITEMS, uint32_t THREADS, typename LDS>
__device__ __forceinline__ void downsweep_body(const uint32_t *__restrict__ keys_in,
                                               const uint32_t *__restrict__ vals_in,
                                               uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                               uint32_t n, uint32_t shift, uint32_t mask,
                                               const uint32_t *__restrict__ counts,
                                               const uint32_t *__restrict__ totals, uint32_t max_tiles,
                                               uint32_t *__restrict__ edges, uint32_t edge_keys, LDS &L,
                                               const uint32_t (&spec)[kSpecRows], KTraceRef trace = KTraceRef{}) {
    static_assert(!FUSED || THREADS == kSortThreads, "the fused table sums assume 8 groups of 128 threads");
    constexpr uint32_t kTileKeys = THREADS * ITEMS;
    constexpr uint32_t kWaves = THREADS / kWave;
    // small sorts: the scatter is a few hundred KB, not worth two barriers — unless the pass also finds the run
    // edges (below), which needs every key's neighbours in output order
    const bool reorder = ITEMS > 2 || edges != nullptr;
    const uint32_t tile = blockIdx.x;
    const uint32_t wid = threadIdx.x / kWave;
    const uint32_t lane = lane_id();
    for (uint32_t i = threadIdx.x; i < kWaves * kRadix; i += THREADS) (&L.wave_hist[0][0])[i] = 0;

    // Each wave owns a contiguous chunk of 64*ITEMS keys; round i covers keys chunk + i*64 + lane.
    // The loads are issued first so that they are in flight during the table sums below.
    const uint32_t tile_base = tile * kTileKeys;
    const uint32_t chunk = tile_base + wid * (kWave * ITEMS);
    // Large tiles are register-bound (16 keys, 16 ranks and the ballot temporaries per lane): their values are loaded
    // late, just before they are placed, instead of being held across the ranking.
    constexpr bool kLateVals = ITEMS >= 8;
    uint32_t key[ITEMS], val[kLateVals ? 1 : ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        const uint32_t idx = chunk + i * kWave + lane;
        key[i] = idx < n ? keys_in[idx] : 0u;
        if constexpr (!kLateVals)
            val[i] = idx < n ? (vals_in ? vals_in[idx] : idx) : 0u;  // vals_in == nullptr: the positions themselves
    }
    // FUSED: counts[t][d] (16-bit); the small-tile variants sum their share of the table now, behind the key loads
    // issued above; for the large-tile variants (ITEMS >= 8, register-bound) the kernel did it before the switch.
    uint32_t glob_base = 0, glob_tot = 0;
    if (FUSED) {
        if constexpr (ITEMS < 8) table_sums_to_lds(spec, counts, tile, (n + kTileKeys - 1) / kTileKeys, L, trace);
    } else if (threadIdx.x < kRadix) {
        glob_tot = totals[threadIdx.x];
        glob_base = counts[(size_t)threadIdx.x * max_tiles + tile];
    }
    __syncthreads();  // A: wave_hist zeroed, partial column sums published
    BRUSH_KTRACE_MARK_VIA(trace, 3, key[0]);

    // rank of a key among the keys of its digit inside its wave's chunk: < 64 * ITEMS <= 1024.  The large-tile variants
    // are register-bound and keep two ranks per register.
    constexpr bool kPackRanks = ITEMS >= 8;
    uint32_t rank[kPackRanks ? ITEMS / 2 : ITEMS];
    auto rank_of = [&](uint32_t i) -> uint32_t {
        if constexpr (kPackRanks) return (rank[i / 2] >> ((i & 1u) * 16u)) & 0xFFFFu;
        else return rank[i];
    };
    const uint64_t lt = lanemask_lt();
    // digit bits this pass looks at (a 13-bit tile id is sorted as 7 + 6 bits: one and two ballots fewer per key).  Only
    // in the 1024-thread kernel: measured same box, tile sort 30.4 -> 29.6 us at S1 and 74.3 -> 70.9 on the dense scene,
    // but the 512-thread large-sort kernel lost 4 us per sort to the extra scalar branches (c3 386 -> 391).
    const uint32_t nbits = THREADS == kSortThreads ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__popc(mask)) : 8u;
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        const uint32_t idx = chunk + i * kWave + lane;
        const bool valid = idx < n;
        const uint32_t digit = (key[i] >> shift) & mask;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            if (b >= nbits) break;  // wave-uniform
            const bool bit = (digit >> b) & 1u;
            const uint64_t vote = __ballot(valid && bit);
            peers &= bit ? vote : ~vote;
        }
        const uint32_t below = __popcll(peers & lt);
        const uint32_t cnt = __popcll(peers);
        const int leader = valid ? (__ffsll((long long)peers) - 1) : 0;
        uint32_t old = 0;
        if (valid && (int)lane == leader) {
            old = L.wave_hist[wid][digit];
            L.wave_hist[wid][digit] = old + cnt;
        }
        old = __shfl(old, leader, 64);
        if constexpr (kPackRanks) rank[i / 2] = (i & 1u) ? (rank[i / 2] | ((old + below) << 16)) : (old + below);
        else rank[i] = old + below;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();  // B: every wave's digit counts are final
    // Threads 0..255 (digit d): 16-wave combine (wave_hist[w][d] <- keys of digit d in earlier waves of the tile),
    // the tile's count of d, and the global sums; then both exclusive scans over the digits behind one pair of
    // barriers: (#keys with a smaller digit anywhere) and (start of the digit's run inside the tile).
    uint32_t tile_cnt = 0;
    if (threadIdx.x < kRadix) {
        const uint32_t d = threadIdx.x;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; w++) {
            const uint32_t t = L.wave_hist[w][d];
            L.wave_hist[w][d] = tile_cnt;
            tile_cnt += t;
        }
        if (FUSED) {
#pragma unroll
            for (uint32_t g = 0; g < kTableGroups; g++) glob_tot += L.part_all()[g][d], glob_base += L.part()[g][d];
        }
    }
    uint32_t smaller = glob_tot, run_start = tile_cnt;
    block_excl_scan256_pair(smaller, run_start, L.wave_tot2);
    if (threadIdx.x < kRadix) {
        L.digit_base[threadIdx.x] = smaller + glob_base;
        L.tile_start[threadIdx.x] = run_start;
    }
    __syncthreads();  // C
    BRUSH_KTRACE_MARK_VIA(trace, 4, smaller);
    if (!reorder) {
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            const uint32_t idx = chunk + i * kWave + lane;
            if (idx < n) {
                const uint32_t digit = (key[i] >> shift) & mask;
                const uint32_t pos = L.digit_base[digit] + L.wave_hist[wid][digit] + rank_of(i);
                if (keys_out) keys_out[pos] = key[i];
                vals_out[pos] = kLateVals ? (vals_in ? vals_in[idx] : idx) : val[kLateVals ? 0 : i];
            }
        }
        return;
    }
    // reorder by digit in LDS (stable: wave order, then rank inside the wave) ...
    if constexpr (kLateVals) {
        // The late values are requested kValBatch at a time: one at a time (each load followed by its LDS write) the 16
        // loads of a lane were 16 dependent round trips — 12.7 of the 24 us a 8192-key tile's workgroup lives at 27 M
        // keys (in-kernel timeline, profiles/r04b_big_sort_timeline.json).  The ballot temporaries are dead by now.
        constexpr uint32_t kValBatch = THREADS == kSortThreads ? 4 : 16;
#pragma unroll
        for (uint32_t i0 = 0; i0 < ITEMS; i0 += kValBatch) {
            uint32_t v[kValBatch];
#pragma unroll
            for (uint32_t u = 0; u < kValBatch; u++) {
                const uint32_t idx = chunk + (i0 + u) * kWave + lane;
                v[u] = idx < n ? (vals_in ? vals_in[idx] : idx) : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < kValBatch; u++) {
                const uint32_t i = i0 + u, idx = chunk + i * kWave + lane;
                if (idx < n) {
                    const uint32_t digit = (key[i] >> shift) & mask;
                    const uint32_t lp = L.tile_start[digit] + L.wave_hist[wid][digit] + rank_of(i);
                    L.keys[lp] = key[i];
                    L.vals[lp] = v[u];
                }
            }
        }
    } else {
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            const uint32_t idx = chunk + i * kWave + lane;
            if (idx < n) {
                const uint32_t digit = (key[i] >> shift) & mask;
                const uint32_t lp = L.tile_start[digit] + L.wave_hist[wid][digit] + rank_of(i);
                L.keys[lp] = key[i];
                L.vals[lp] = val[i];
            }
        }
    }
    __syncthreads();
    BRUSH_KTRACE_MARK_VIA(trace, 2, L.keys[threadIdx.x]);  // (big tiles: mark 2 is free; reordered in LDS, values loaded)
    // ... and write out: position lp of the reordered tile belongs to the run of its digit, so consecutive lanes
    // write consecutive global addresses (round 1 scattered one dword per lane into 256 runs: 2.1x write
    // amplification at the HBM counters)
    const uint32_t tile_n = min(kTileKeys, n - tile_base);
    // 16 keys per lane at 1024 threads (128 registers): fully unrolled it spills
    constexpr uint32_t kOutUnroll = (ITEMS > 8 && THREADS == kSortThreads) ? 4 : ITEMS;
#pragma unroll kOutUnroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        const uint32_t lp = i * THREADS + threadIdx.x;
        if (lp < tile_n) {
            const uint32_t k = L.keys[lp];
            const uint32_t digit = (k >> shift) & mask;
            const uint32_t ts = L.tile_start[digit];
            const uint32_t pos = L.digit_base[digit] + (lp - ts);
            if (keys_out) keys_out[pos] = k;  // nullptr: nobody reads the sorted keys (the run edges are found right here)
            vals_out[pos] = L.vals[lp];
            // LAST pass only (edges != nullptr): the keys are in their final order, so the run of equal keys around
            // `pos` is a bin of the caller (GetTileBinEdges, get_tile_bin_edges.wgsl:15-42, without its launch).  A key
            // that differs from its predecessor / successor inside this workgroup's run of the digit proposes
            // [pos, pos + 1) as (start, end); so do the first and the last key of the run, whose outer neighbours belong
            // to other workgroups.  edges[2k] = max ~start, edges[2k+1] = max end over all proposals (zero-initialised
            // by the caller): proposals from the inside of a bin lose against the true edges.
            if (edges && k < edge_keys) {
                const bool first = lp == ts;
                const bool last = lp + 1 == tile_n || ((L.keys[lp + 1] >> shift) & mask) != digit;
                if (first || L.keys[lp - 1] != k) atomicMax(&edges[2 * k], ~pos);
                if (last || L.keys[lp + 1] != k) atomicMax(&edges[2 * k + 1], pos + 1u);
            }
        }
    }
}

template <bool FUSED>
__global__ __launch_bounds__(kSortThreads) void k_sort_downsweep(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ d_n,
    uint32_t max_n, uint32_t shift, uint32_t mask, const uint32_t *__restrict__ counts,
    const uint32_t *__restrict__ totals, uint32_t max_tiles, uint32_t *__restrict__ edges, uint32_t edge_keys) {
    BRUSH_KTRACE(kTrSortDown, shift | (((shift >> 3) + (max_n > (1u << 21) ? 4u : 0u)) << 24));
    // The first 128 rows of the [tile][digit] count table (every row a sort of up to 2 M keys has) are requested before
    // the element count is known: the table sum used to start only when *d_n had arrived and took two dependent batches
    // of loads (2.2 us of the kernel's 6.5 at 100 k keys, profiles/r04_small_kernel_timeline.json).  The table has at least
    // kSortTargetTiles = 128 rows (sort_max_tiles); rows of tiles that do not exist hold leftovers and are masked in the body.
    uint32_t spec[kSpecRows];
    {
        const uint32_t dp = threadIdx.x & (kRadix / 2 - 1), g8 = threadIdx.x / (kRadix / 2);
#pragma unroll
        for (uint32_t j = 0; j < kSpecRows; j++)
            spec[j] = FUSED ? counts[(size_t)(g8 + kTableGroups * j) * (kRadix / 2) + dp] : 0u;
    }
    const uint32_t n = min(*d_n, max_n);
    BRUSH_KTRACE_MARK(1, n);
    const uint32_t items = sort_items(n);
    if ((uint64_t)blockIdx.x * kSortThreads * items >= n) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    DownLds &L = *reinterpret_cast<DownLds *>(lds_raw);
#define BRUSH_DOWN(K, SPEC)                                                                                       \
    downsweep_body<FUSED, K, kSortThreads>(keys_in, vals_in, keys_out, vals_out, n, shift, mask, counts, totals, \
                                           max_tiles, edges, edge_keys, L, SPEC, BRUSH_KTRACE_REF)
    if (items >= 8) {  // block-uniform
        // large tiles are register-bound: the table sums are formed here, before their 8 / 16 keys per lane are live
        table_sums_to_lds(spec, counts, blockIdx.x, (n + kSortThreads * items - 1) / (kSortThreads * items), L,
                          BRUSH_KTRACE_REF);
        const uint32_t none[kSpecRows] = {};
        if (items == 8) BRUSH_DOWN(8, none);
        else BRUSH_DOWN(16, none);
        return;
    }
    switch (items) {
        case 1: BRUSH_DOWN(1, spec); break;
        case 2: BRUSH_DOWN(2, spec); break;
        default: BRUSH_DOWN(4, spec); break;
    }
#undef BRUSH_DOWN
}

// Large sorts: one 8192-key tile per 512-thread workgroup, two workgroups per CU.
__global__ __launch_bounds__(kBigThreads) void k_sort_downsweep_big(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ d_n,
    uint32_t max_n, uint32_t shift, uint32_t mask, const uint32_t *__restrict__ counts,
    const uint32_t *__restrict__ totals, uint32_t max_tiles, uint32_t *__restrict__ edges, uint32_t edge_keys) {
    BRUSH_KTRACE(kTrSortDownBig, shift | ((shift ? 1u : 0u) << 24));
    const uint32_t n = min(*d_n, max_n);
    BRUSH_KTRACE_MARK(1, n);
    if ((uint64_t)blockIdx.x * kBigTileKeys >= n) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    DownLdsBig &L = *reinterpret_cast<DownLdsBig *>(lds_raw);
    const uint32_t no_spec[kSpecRows] = {};
    downsweep_body<false, kBigTileKeys / kBigThreads, kBigThreads>(keys_in, vals_in, keys_out, vals_out, n, shift,
                                                                   mask, counts, totals, max_tiles, edges, edge_keys, L,
                                                                   no_spec, BRUSH_KTRACE_REF);
}

__global__ void k_sort_copy(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                            uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                            const uint32_t *__restrict__ d_n, uint32_t max_n) {
    const uint32_t n = min(*d_n, max_n);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        keys_out[i] = keys_in[i];
        vals_out[i] = vals_in ? vals_in[i] : i;
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

// The downsweep needs more LDS than the 64 KiB a kernel gets by default: opt in once per DEVICE (the attribute
// belongs to the function's instance on the current device; an embedder may drive several devices from one process).
hipError_t enable_big_lds() {
    static std::atomic<int> done[kMaxDevices];
    const int dev = current_device_slot();
    if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sort_downsweep<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DownLds));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sort_downsweep_big),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DownLdsBig));
    if (e == hipSuccess) done[dev].store(1, std::memory_order_release);
    return e;
}

}  // namespace

size_t sort_workspace_bytes(uint32_t max_n) { return carve_sort(nullptr, max_n).bytes; }

hipError_t sort_launch(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out,
                       uint32_t *vals_out, const uint32_t *d_n, uint32_t max_n, uint32_t bits, void *ws,
                       hipStream_t s, uint32_t *edges, uint32_t edge_keys) {
    if (max_n == 0) return hipSuccess;
    if (edges && bits == 0) return hipErrorInvalidValue;  // the edges come out of the last pass
    if (!keys_out && (bits == 0 || bits > 16)) return hipErrorInvalidValue;  // only the LAST of <= 2 passes may drop its keys
    const SortWs w = carve_sort(ws, max_n);
    const uint32_t total_bits = 4u * ((bits + 3u) / 4u);  // brush-sort/src/lib.rs:58
    const uint32_t passes = (total_bits + 7u) / 8u;
    if (passes == 0) {
        hipLaunchKernelGGL(k_sort_copy, dim3(min(w.max_tiles * 4u, 2048u)), dim3(256), 0, s, keys_in, vals_in,
                           keys_out, vals_out, d_n, max_n);
        return hipGetLastError();
    }
    const hipError_t lds = enable_big_lds();
    if (lds != hipSuccess) return lds;
    const bool fused = sort_is_fused(max_n);
    const uint32_t *src_k = keys_in, *src_v = vals_in;
    // Internal callers that bound the key values (edge_keys: every key < edge_keys, the tile sort) get the significant
    // bits split evenly over the passes instead of 8 + the rest: a 13-bit tile id is sorted as 7 + 6 bits, so the first
    // pass scatters into 128 runs per tile instead of 256 (runs twice as long: fewer partial lines written).  Same
    // result: the bits above the bound are zero in every key.
    uint32_t even_width = 0;
    if (edge_keys > 1u) {
        uint32_t sig = 0;
        while (sig < 32u && ((edge_keys - 1u) >> sig) != 0u) sig++;
        if (sig <= total_bits) even_width = (sig + passes - 1u) / passes;
    }
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t shift = even_width ? p * even_width : p * 8u;
        const uint32_t width = even_width ? even_width : min(8u, total_bits - shift);
        const uint32_t mask = (1u << width) - 1u;
        const bool to_out = ((passes - 1 - p) % 2u) == 0;
        uint32_t *dst_k = to_out ? keys_out : w.tmp_keys;  // keys_out may be nullptr: the last pass then writes no keys
        uint32_t *dst_v = to_out ? vals_out : w.tmp_vals;
        uint32_t *pass_edges = p + 1 == passes ? edges : nullptr;
        if (fused) {
            hipLaunchKernelGGL(k_sort_upsweep<true>, dim3(w.max_tiles), dim3(kSortThreads), 0, s, src_k, d_n, max_n,
                               shift, mask, w.counts, w.max_tiles);
            hipLaunchKernelGGL(k_sort_downsweep<true>, dim3(w.max_tiles), dim3(kSortThreads), sizeof(DownLds), s, src_k,
                               src_v, dst_k, dst_v, d_n, max_n, shift, mask, w.counts, w.totals, w.max_tiles, pass_edges,
                               edge_keys);
        } else {
            hipLaunchKernelGGL(k_sort_upsweep<false>, dim3(w.max_tiles), dim3(kSortThreads), 0, s, src_k, d_n, max_n,
                               shift, mask, w.counts, w.max_tiles);
            hipLaunchKernelGGL(k_sort_scan, dim3(kRadix), dim3(256), 0, s, w.counts, d_n, max_n, w.max_tiles, w.totals);
            hipLaunchKernelGGL(k_sort_downsweep_big, dim3(w.max_tiles), dim3(kBigThreads), sizeof(DownLdsBig), s,
                               src_k, src_v, dst_k, dst_v, d_n, max_n, shift, mask, w.counts, w.totals, w.max_tiles,
                               pass_edges, edge_keys);
        }
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
