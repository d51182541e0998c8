// scan.hip — inclusive u32 prefix sum (replaces crates/brush-prefix-sum).
//
// Reference: prefix_sum() crates/brush-prefix-sum/src/lib.rs:17-102 — 512-wide Hillis-Steele
// per workgroup, recursive block sums, 5 launches at N = 1 M.
//
// gfx950 design: reduce-then-scan.  Up to 2048 tiles (2 M elements) every scan block sums the tile
// sums before it itself (two launches); beyond that a single-block spine scans them (three launches).
// (Letting the producers accumulate the tile sums with atomics saves the reduce launch but was
// slower: near splats share a tile, and ~2000 same-address atomics drain for ~20 us after the last
// wave of the producing kernel has retired.)
//   k_reduce : one 256-thread block per 1024-element tile, one coalesced dwordx4 per lane,
//              wave64 shuffle reduction -> tile_sums[t]
//   k_spine  : one 1024-thread block scans tile_sums in place (exclusive), 1024 at a time with
//              a running carry; optionally publishes the grand total clamped to a capacity
//   k_down   : re-reads the tile (L2/MALL-warm), in-lane scan of 4 + wave64 scan + 4-wave
//              combine in LDS, adds the tile prefix, one coalesced dwordx4 store per lane
// HBM traffic: 8 B/element + a second read of the input that hits the Infinity Cache for the
// sizes on this path (N*4 B <= 80 MB).  Roofline: HBM.
#include "common.hpp"
#include "trace.hpp"

namespace brush {
namespace {

constexpr uint32_t kScanThreads = 256;
constexpr uint32_t kScanTile = kScanThreads * 4;

__device__ __forceinline__ uint4 load_tile4(const uint32_t *__restrict__ in, uint32_t idx, uint32_t n,
                                            uint32_t valid_n, bool aligned) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (idx >= valid_n) return v;
    if (aligned && idx + 4 <= valid_n && idx + 4 <= n) {
        v = *reinterpret_cast<const uint4 *>(in + idx);
    } else {
        const uint32_t lim = valid_n < n ? valid_n : n;
        if (idx + 0 < lim) v.x = in[idx + 0];
        if (idx + 1 < lim) v.y = in[idx + 1];
        if (idx + 2 < lim) v.z = in[idx + 2];
        if (idx + 3 < lim) v.w = in[idx + 3];
    }
    return v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_reduce(const uint32_t *__restrict__ in, uint32_t n,
                                                             const uint32_t *__restrict__ d_valid_n,
                                                             uint32_t *__restrict__ tile_sums) {
    __shared__ uint32_t wave_tot[kScanThreads / kWave];
    BRUSH_KTRACE(kTrScanReduce, 0);
    const uint32_t valid_n = d_valid_n ? min(*d_valid_n, n) : n;
    BRUSH_KTRACE_MARK(1, valid_n);
    const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
    const uint32_t idx = blockIdx.x * kScanTile + threadIdx.x * 4;
    const uint4 v = load_tile4(in, idx, n, valid_n, aligned);
    uint32_t s = v.x + v.y + v.z + v.w;
    BRUSH_KTRACE_MARK(2, s);
    s = wave_sum(s);
    if (lane_id() == 0) wave_tot[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// Single block: exclusive scan of tile_sums[0..num_tiles) in place.
__global__ __launch_bounds__(1024) void k_scan_spine(uint32_t *__restrict__ tile_sums, uint32_t num_tiles,
                                                     uint32_t *__restrict__ d_total, uint32_t cap,
                                                     uint32_t *__restrict__ d_overflow) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < num_tiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < num_tiles ? tile_sums[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane_id() == 63) wave_tot[threadIdx.x / kWave] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (uint32_t w = 0; w < threadIdx.x / kWave; w++) wave_off += wave_tot[w];
        const uint32_t carry = carry_s;
        if (i < num_tiles) tile_sums[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && d_total) {
        const uint32_t total = carry_s;
        if (total > cap) {
            *d_total = cap;
            if (d_overflow) *d_overflow = 1u;
        } else {
            *d_total = total;
        }
    }
}

// SELF: `tile_prefix` holds the raw tile sums; the block adds up those before it, and the last block
// publishes the grand total (clamped to cap, flagging the overflow).
template <bool SELF>
__global__ __launch_bounds__(kScanThreads) void k_scan_down(const uint32_t *__restrict__ in,
                                                           uint32_t *__restrict__ out, uint32_t n,
                                                           const uint32_t *__restrict__ d_valid_n,
                                                           const uint32_t *__restrict__ tile_prefix,
                                                           uint32_t *__restrict__ d_total, uint32_t cap,
                                                           uint32_t *__restrict__ d_overflow) {
    __shared__ uint32_t wave_tot[kScanThreads / kWave];
    __shared__ uint32_t pre_s[kScanThreads / kWave];
    BRUSH_KTRACE(kTrScanDown, 0);
    if (SELF) {
        uint32_t before = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kScanThreads) before += tile_prefix[i];
        BRUSH_KTRACE_MARK(1, before);
        before = wave_sum(before);
        if (lane_id() == 0) pre_s[threadIdx.x / kWave] = before;
    }
    const uint32_t valid_n = d_valid_n ? min(*d_valid_n, n) : n;
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    const uint32_t idx = blockIdx.x * kScanTile + threadIdx.x * 4;
    BRUSH_KTRACE_MARK(2, valid_n);
    uint4 v = load_tile4(in, idx, n, valid_n, aligned);
    BRUSH_KTRACE_MARK(3, v.x + v.w);
    v.y += v.x;
    v.z += v.y;
    v.w += v.z;
    const uint32_t incl = wave_inclusive_scan(v.w);
    const uint32_t wid = threadIdx.x / kWave;
    if (lane_id() == 63) wave_tot[wid] = incl;
    __syncthreads();
    const uint32_t prefix = SELF ? pre_s[0] + pre_s[1] + pre_s[2] + pre_s[3] : tile_prefix[blockIdx.x];
    if (SELF && d_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const uint32_t total = prefix + wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        if (total > cap) {
            *d_total = cap;
            if (d_overflow) *d_overflow = 1u;
        } else {
            *d_total = total;
        }
    }
    uint32_t off = prefix + (incl - v.w);
    for (uint32_t w = 0; w < wid; w++) off += wave_tot[w];
    v.x += off;
    v.y += off;
    v.z += off;
    v.w += off;
    BRUSH_KTRACE_MARK(4, v.w);
    if (idx >= n) return;
    if (aligned && idx + 4 <= n) {
        *reinterpret_cast<uint4 *>(out + idx) = v;
    } else {
        if (idx + 0 < n) out[idx + 0] = v.x;
        if (idx + 1 < n) out[idx + 1] = v.y;
        if (idx + 2 < n) out[idx + 2] = v.z;
        if (idx + 3 < n) out[idx + 3] = v.w;
    }
}

}  // namespace

size_t scan_workspace_bytes(uint32_t n) {
    return align_up((size_t)ceil_div(n ? n : 1, kScanTile) * sizeof(uint32_t), 256);
}

constexpr uint32_t kSelfScanTiles = 2048;

hipError_t scan_launch(const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *d_valid_n,
                       uint32_t *d_total, uint32_t cap, uint32_t *d_overflow, void *ws, hipStream_t s) {
    uint32_t *tile_sums = static_cast<uint32_t *>(ws);
    const uint32_t num_tiles = ceil_div(n, kScanTile);
    if (n == 0) {  // nothing to scan: the spine still writes d_total = 0
        hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, s, tile_sums, 0u, d_total, cap, d_overflow);
        return hipGetLastError();
    }
    if (num_tiles <= kSelfScanTiles) {
        hipLaunchKernelGGL(k_scan_reduce, dim3(num_tiles), dim3(kScanThreads), 0, s, in, n, d_valid_n, tile_sums);
        hipLaunchKernelGGL(k_scan_down<true>, dim3(num_tiles), dim3(kScanThreads), 0, s, in, out, n, d_valid_n,
                           tile_sums, d_total, cap, d_overflow);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_scan_reduce, dim3(num_tiles), dim3(kScanThreads), 0, s, in, n, d_valid_n, tile_sums);
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, s, tile_sums, num_tiles, d_total, cap, d_overflow);
    hipLaunchKernelGGL(k_scan_down<false>, dim3(num_tiles), dim3(kScanThreads), 0, s, in, out, n, d_valid_n,
                       tile_sums, d_total, cap, d_overflow);
    return hipGetLastError();
}

}  // namespace brush

using namespace brush;

extern "C" int brush_inclusive_scan_workspace_size(uint32_t n, size_t *bytes) {
    if (!bytes) return BRUSH_ERR_INVALID_ARG;
    *bytes = scan_workspace_bytes(n);
    return BRUSH_OK;
}

extern "C" int brush_inclusive_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, void *workspace,
                                        size_t workspace_bytes, brush_stream_t stream) {
    if (n == 0) return BRUSH_OK;
    if (!in || !out || !workspace) return BRUSH_ERR_INVALID_ARG;
    if (workspace_bytes < scan_workspace_bytes(n)) return BRUSH_ERR_WORKSPACE_SMALL;
    BRUSH_HIP_CHECK(scan_launch(in, out, n, nullptr, nullptr, 0xFFFFFFFFu, nullptr, workspace,
                                static_cast<hipStream_t>(stream)));
    return BRUSH_OK;
}
