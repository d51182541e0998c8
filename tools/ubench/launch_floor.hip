// launch_floor.hip — what a DEPENDENT small kernel costs on MI355X as a function of what it does before it exits.
// A chain of 200 launches of one kernel on one stream is captured into a hipGraph and replayed; time per launch =
// (replay time) / 200.  Variants add, one at a time, the pieces the forward's small kernels are made of:
//   empty          : nothing
//   count          : *d_n (one scalar load every kernel of the path starts with)
//   count_load     : *d_n -> in[idx] -> out[idx]                 (two dependent loads + a store)
//   count_load_lds : + LDS zeroing, two barriers, LDS atomics     (the shape of k_sort_upsweep)
//   spec_load      : in[idx] issued BEFORE *d_n is known (bounded by the capacity), masked afterwards
//   chain3         : *d_n -> idx[i] -> in[idx[i]] -> out           (three dependent loads: a gather by sorted index)
// for the launch shapes of the path (128 x 1024, 512 x 256, 1024 x 256, 4096 x 256) at 100 k live elements.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ void k_var(const unsigned *__restrict__ d_n, const unsigned *__restrict__ in, const unsigned *__restrict__ idx,
                      unsigned *__restrict__ out, unsigned cap) {
    __shared__ unsigned hist[16 * 256];
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (MODE == 0) return;
    if (MODE == 5) {  // speculative: data load first, count second
        const unsigned v = i < cap ? in[i] : 0u;
        const unsigned n = *d_n;
        if (i < n) out[i] = v + 1u;
        return;
    }
    const unsigned n = *d_n;
    if (MODE == 1) {
        if (n == 0xFFFFFFFFu) out[0] = 1u;
        return;
    }
    if (blockIdx.x * blockDim.x >= n) return;
    if (MODE == 2) {
        if (i < n) out[i] = in[i] + 1u;
        return;
    }
    if (MODE == 3) {
        for (unsigned k = threadIdx.x; k < 16 * 256; k += blockDim.x) hist[k] = 0;
        __syncthreads();
        if (i < n) atomicAdd(&hist[(threadIdx.x / 64 % 16) * 256 + (in[i] & 255u)], 1u);
        __syncthreads();
        if (threadIdx.x < 256) {
            unsigned c = 0;
            for (int w = 0; w < 16; w++) c += hist[w * 256 + threadIdx.x];
            out[blockIdx.x * 256 + threadIdx.x] = c;
        }
        return;
    }
    if (MODE == 4) {
        if (i < n) out[i] = in[idx[i]] + 1u;
        return;
    }
}

template <int MODE>
float run(unsigned grid, unsigned block, const unsigned *d_n, const unsigned *in, const unsigned *idx, unsigned *out,
          unsigned cap) {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    const int chain = 200;
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_var<MODE>, dim3(grid), dim3(block), 0, s, d_n, in, idx, out, cap);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; r++) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    const int reps = 10;
    for (int r = 0; r < reps; r++) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    CHECK(hipStreamDestroy(s));
    return ms * 1000.0f / (reps * chain);
}

int main() {
    const unsigned cap = 1u << 20, live = 102865;
    unsigned *d_n, *in, *idx, *out;
    CHECK(hipMalloc(&d_n, 256));
    CHECK(hipMalloc(&in, cap * 4));
    CHECK(hipMalloc(&idx, cap * 4));
    CHECK(hipMalloc(&out, cap * 4));
    unsigned *h = (unsigned *)malloc(cap * 4);
    for (unsigned i = 0; i < cap; i++) h[i] = (i * 2654435761u) % live;
    CHECK(hipMemcpy(idx, h, cap * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(in, h, cap * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_n, &live, 4, hipMemcpyHostToDevice));
    const unsigned shapes[][2] = {{128, 1024}, {101, 1024}, {512, 256}, {1024, 256}, {4096, 256}, {2040, 256}};
    const char *names[] = {"empty", "count", "count_load", "count_load_lds", "chain3", "spec_load"};
    printf("{\"what\": \"us per dependent launch, 200-launch chain in a replayed hipGraph, %u live of %u elements\", \"rows\": [\n", live, cap);
    bool first = true;
    for (auto &sh : shapes) {
        float t[6];
        t[0] = run<0>(sh[0], sh[1], d_n, in, idx, out, cap);
        t[1] = run<1>(sh[0], sh[1], d_n, in, idx, out, cap);
        t[2] = run<2>(sh[0], sh[1], d_n, in, idx, out, cap);
        t[3] = run<3>(sh[0], sh[1], d_n, in, idx, out, cap);
        t[4] = run<4>(sh[0], sh[1], d_n, in, idx, out, cap);
        t[5] = run<5>(sh[0], sh[1], d_n, in, idx, out, cap);
        printf("%s {\"grid\": %u, \"block\": %u", first ? "" : ",\n", sh[0], sh[1]);
        for (int k = 0; k < 6; k++) printf(", \"%s\": %.2f", names[k], t[k]);
        printf("}");
        first = false;
    }
    printf("\n]}\n");
    return 0;
}
