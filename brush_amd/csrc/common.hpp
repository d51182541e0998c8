// common.hpp — shared host/device helpers for libbrush_hip (gfx950 only, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/brush_hip.h"

namespace brush {

constexpr uint32_t kTileWidth = BRUSH_TILE_WIDTH;      // helpers.wgsl:1
constexpr uint32_t kTileSize = kTileWidth * kTileWidth; // helpers.wgsl:3
constexpr uint32_t kWave = 64;                          // CDNA wavefront
constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr uint32_t kUniformWords = 28;
constexpr uint32_t kNumVisibleWord = 25;                // render.rs:145-149
// Compact-order gradient rows of the backward: [v_xy(2) v_conic(3) v_rgb(3) v_opac(1) pad(7)].  One 64-byte cache
// line per row: the L2 executes the compositing backward's float atomics line by line, and a 48-byte row straddles two
// lines half of the time (measured: 154 -> 133 us for the kernel at the headline scene).
constexpr uint32_t kCompactStride = 16;
constexpr uint32_t kCompactVec = kCompactStride / 4;  // float4 words per row

// Deterministic mode (BrushAux::flags & BRUSH_AUX_DETERMINISTIC, chosen per call): the compositing backward writes
// one gradient row per intersection instead of float atomics and the rows are summed per splat in a fixed order, so
// gradients are bitwise reproducible run to run; the forward then also records where every sorted intersection sat
// before the tile sort (BrushAux::isect_unsorted_pos).

// Records the failing hipError_t for brush_last_hip_error().
void set_last_hip_error(int e);

#define BRUSH_HIP_CHECK(expr)                                   \
    do {                                                        \
        hipError_t _e = (expr);                                 \
        if (_e != hipSuccess) {                                 \
            ::brush::set_last_hip_error((int)_e);               \
            return BRUSH_ERR_HIP;                               \
        }                                                       \
    } while (0)

// Per-device caches (function attributes, SIMD counts) are keyed on the CURRENT device of the calling thread: slot
// of that device in a small table (devices beyond it share the last slot, whose entries are then re-derived per call).
constexpr int kMaxDevices = 64;
static inline int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < kMaxDevices ? dev : kMaxDevices - 1;
}

__host__ __device__ static inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace (256-B aligned carve-outs).
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *p) : base(static_cast<char *>(p)) {}
    template <typename T>
    T *take(size_t count) {
        off = align_up(off, 256);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
    size_t bytes() const { return align_up(off, 256); }
};

// ---- device-side wave helpers -------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Wave64 inclusive prefix sum on the VALU: six DPP adds (row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast:15 / row_bcast:31 carry the row totals up).  __shfl_up is a ds_bpermute on gfx950 (LDS crossbar, ~60 cycles of
// latency each, six of them in a dependent chain): the in-kernel timeline of the small kernels (profiles/
// r04_small_kernel_timeline.json) showed those chains as a large share of every scan / reservation / ranking phase.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
    // lanes whose source lane does not exist, and rows outside ROW_MASK, add `old` = 0
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v = dpp_add_u32<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add_u32<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add_u32<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add_u32<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add_u32<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v = dpp_add_u32<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}
// Value of lane `src` (compile-time or wave-uniform) in every lane, through the scalar unit (v_readlane_b32).
__device__ __forceinline__ uint32_t wave_bcast(uint32_t v, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src);
}
// Sum over the wave, in every lane.
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return wave_bcast(wave_inclusive_scan(v), 63u); }

// ---- internal launch API (one per translation unit) ----------------------------------
// scan.hip
size_t scan_workspace_bytes(uint32_t n);
// Inclusive scan of in[0..n) -> out. If d_valid_n != nullptr, elements >= *d_valid_n read as 0.
// If d_total != nullptr, the grand total (out[n-1]) is also written there, clamped to `cap`
// with *d_overflow set when it exceeded cap (pass cap = 0xFFFFFFFF / nullptr to disable).
hipError_t scan_launch(const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *d_valid_n,
                       uint32_t *d_total, uint32_t cap, uint32_t *d_overflow, void *ws, hipStream_t s);

// radix_sort.hip
size_t sort_workspace_bytes(uint32_t max_n);
// vals_in == nullptr sorts the positions 0..n-1 themselves (argsort proper).
// edges != nullptr (zero-initialised [edge_keys][2], keys < edge_keys): the last pass also records the run of every
// key value k in the sorted output as edges[2k] = ~start, edges[2k+1] = end (both 0 for a key that does not occur).
// keys_out == nullptr (internal callers only, bits > 0): the sorted keys themselves are not written (4 B per pair less
// in the last pass); with an even number of passes the ping-pong puts the LAST pass's keys there, which is the only
// output dropped.
hipError_t sort_launch(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out,
                       uint32_t *vals_out, const uint32_t *d_n, uint32_t max_n, uint32_t bits,
                       void *ws, hipStream_t s, uint32_t *edges = nullptr, uint32_t edge_keys = 0);

}  // namespace brush
