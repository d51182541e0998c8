// trace.hpp — development-only in-kernel timeline (make -C brush_amd/csrc trace -> libbrush_hip_trace.so).
//
// In the product build (BRUSH_TRACE undefined) every macro below expands to nothing: no stamp executes, no kernel
// argument, register or instruction is added.  In the trace build wave 0 of every workgroup keeps up to six s_memrealtime
// stamps (100 MHz, one counter for the whole chip) — kernel entry, up to four marks each tied to the arrival of a value
// (the device-side count the kernel starts from, the first data load, ...), last instruction before exit — and appends one record
// to a ring the host reads back (brush_debug_trace_begin / brush_debug_trace_read, render.hip).  The record leaves
// through memory no kernel reads.  tools/debug/fwd_timeline.py turns the ring into the per-kernel table of
// profiles/r04_small_kernel_timeline.json.  Quote the SHARES of that build, not its run time.
#pragma once
#include "common.hpp"

#ifdef BRUSH_TRACE
namespace brush {

struct TraceRec {
    uint64_t t[6];  // entry, marks 1..4 (0 = not taken), exit (s_memrealtime ticks, 10 ns)
    uint32_t kid, aux, block, grid;
};
static_assert(sizeof(TraceRec) == 64, "one cache line per record");
// Slot of a record: (kernel id, launch slot, workgroup) — no cursor: a returning atomic per workgroup on one word
// serialises at ~88 per us and showed up as a 12 us tail behind every short 1024-workgroup kernel.
constexpr uint32_t kTraceKids = 20, kTraceLaunchSlots = 8, kTraceBlocks = 8192;
constexpr uint32_t kTraceCap = kTraceKids * kTraceLaunchSlots * kTraceBlocks;

// One ring per process; every translation unit keeps its own device-side pointer to it (no relocatable device code).
void trace_register(void (*attach)(TraceRec *, uint32_t *));

namespace {
__device__ TraceRec *g_trace_rec = nullptr;
__device__ uint32_t *g_trace_cur = nullptr;
void trace_attach_tu(TraceRec *r, uint32_t *c) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_rec), &r, sizeof(r));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_cur), &c, sizeof(c));
}
struct TraceRegistrar {
    TraceRegistrar() { trace_register(&trace_attach_tu); }
} g_trace_registrar;

struct KTrace {
    uint64_t t[5];
    uint32_t kid, aux;
    __device__ __forceinline__ KTrace(uint32_t k, uint32_t a) : kid(k), aux(a) {
        t[1] = t[2] = t[3] = t[4] = 0;
        t[0] = __builtin_amdgcn_s_memrealtime();
    }
    // `dep` must have arrived in its register before the stamp is taken (the compiler places the wait for it in front
    // of the empty asm statement that names it).
    template <typename T>
    __device__ __forceinline__ void mark(int i, T dep) {
        asm volatile("" ::"v"(dep) : "memory");
        t[i] = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ ~KTrace() {
        if (threadIdx.x == 0 && threadIdx.y == 0 && g_trace_rec) {
            const uint64_t t5 = __builtin_amdgcn_s_memrealtime();
            const uint32_t b = blockIdx.x + blockIdx.y * gridDim.x;
            const uint32_t i = (kid * kTraceLaunchSlots + (aux >> 24)) * kTraceBlocks + b;
            if (kid < kTraceKids && (aux >> 24) < kTraceLaunchSlots && b < kTraceBlocks) {
                TraceRec r;
                r.t[0] = t[0], r.t[1] = t[1], r.t[2] = t[2], r.t[3] = t[3], r.t[4] = t[4], r.t[5] = t5;
                r.kid = kid, r.aux = aux & 0xFFFFFFu, r.block = b, r.grid = gridDim.x * gridDim.y;
                g_trace_rec[i] = r;
            }
        }
    }
};
}  // namespace
}  // namespace brush
// aux: bits 24..26 = launch slot (distinguishes the launches of one kernel inside a step), low 24 bits free
#define BRUSH_KTRACE(kid, aux) ::brush::KTrace brush_ktrace_((kid), (aux))
#define BRUSH_KTRACE_MARK(i, dep) brush_ktrace_.mark((i), (dep))
// for device functions called by a traced kernel: pass BRUSH_KTRACE_REF as a KTraceRef argument
namespace brush { using KTraceRef = KTrace *; }
#define BRUSH_KTRACE_REF (&brush_ktrace_)
#define BRUSH_KTRACE_MARK_VIA(ref, i, dep) do { if (ref) (ref)->mark((i), (dep)); } while (0)
#else
#define BRUSH_KTRACE(kid, aux) ((void)0)
#define BRUSH_KTRACE_MARK(i, dep) ((void)0)
namespace brush { struct KTraceRef {}; }  // empty: costs nothing as an argument
#define BRUSH_KTRACE_REF (::brush::KTraceRef{})
#define BRUSH_KTRACE_MARK_VIA(ref, i, dep) ((void)0)
#endif

// Kernel ids of the timeline (tools/debug/fwd_timeline.py names them).
enum : uint32_t {
    kTrCull = 1, kTrCompact, kTrSortUp, kTrSortDown, kTrProjectVisible, kTrWalkCount, kTrScanReduce, kTrScanDown,
    kTrMap, kTrRasterize, kTrZeroGrads, kTrRasterizeBwd, kTrProjectBwd, kTrSortScan, kTrSortDownBig, kTrSortOne,
};
