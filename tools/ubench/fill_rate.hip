// fill_rate.hip — write bandwidth of zero-fill kernels on MI355X: plain vs non-temporal 16-byte stores, one store per
// thread vs grid-stride loops, and the row shape of the dense gradient write (192-byte rows, 12 consecutive lanes per row).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_fill(v4f *p, size_t n4) {
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {  // plain, one store per thread
        const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        if (i < n4) p[i] = z;
    } else if (MODE == 1) {  // nt, one store per thread
        const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        if (i < n4) __builtin_nontemporal_store(z, p + i);
    } else if (MODE == 2) {  // plain, grid-stride
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = z;
    } else if (MODE == 3) {  // nt, grid-stride
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
            __builtin_nontemporal_store(z, p + i);
    } else if (MODE == 4) {  // plain, 4 consecutive KiB per wave (each thread 4 stores 1 KiB apart)
        const size_t base = ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63u)) * 4 + (threadIdx.x & 63u);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (base + k * 64 < n4) p[base + k * 64] = z;
    } else if (MODE == 5) {  // nt, same shape
        const size_t base = ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63u)) * 4 + (threadIdx.x & 63u);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (base + k * 64 < n4) __builtin_nontemporal_store(z, p + base + k * 64);
    }
}

template <int MODE>
void run(const char *name, v4f *d, size_t bytes, int grid_stride_blocks) {
    const size_t n4 = bytes / 16;
    const bool stride = MODE == 2 || MODE == 3;
    const bool quad = MODE == 4 || MODE == 5;
    const unsigned blocks = stride ? grid_stride_blocks : (unsigned)((n4 / (quad ? 4 : 1) + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fill<MODE>, dim3(blocks), dim3(256), 0, 0, d, n4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_fill<MODE>, dim3(blocks), dim3(256), 0, 0, d, n4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %6.2f GB  %8.3f ms  %7.1f GB/s\n", name, bytes / 1e9, ms / 5, bytes * 5 / (ms * 1e-3) / 1e9);
}

int main() {
    for (size_t bytes : {(size_t)256 << 20, (size_t)1 << 30, (size_t)5 << 30}) {
        v4f *d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
        run<0>("plain, one 16-B store per thread", d, bytes, 0);
        run<1>("nt,    one 16-B store per thread", d, bytes, 0);
        run<2>("plain, grid-stride 2048 blocks", d, bytes, 2048);
        run<3>("nt,    grid-stride 2048 blocks", d, bytes, 2048);
        run<2>("plain, grid-stride 8192 blocks", d, bytes, 8192);
        run<3>("nt,    grid-stride 8192 blocks", d, bytes, 8192);
        run<4>("plain, 4 KiB per wave", d, bytes, 0);
        run<5>("nt,    4 KiB per wave", d, bytes, 0);
        hipMemsetAsync(d, 0, bytes, 0); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); for (int r = 0; r < 5; r++) hipMemsetAsync(d, 0, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %6.2f GB  %8.3f ms  %7.1f GB/s\n", "hipMemsetAsync", bytes / 1e9, ms / 5, bytes * 5 / (ms * 1e-3) / 1e9);
        hipFree(d);
    }
    return 0;
}
