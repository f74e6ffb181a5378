// Streaming-read ceiling of this GPU for the access pattern the scoring kernels use: every byte of a 25.5 GB buffer
// (4096 x 1080p BGR) read once with 16-byte loads, xor-reduced, nothing written.  SURVEY.md 8(d) asks for the
// roofline fraction to be quoted next to a measured ceiling as well as the 8 TB/s nominal peak.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_read.hip -o tools/ubench/stream_read && tools/ubench/stream_read
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ void read_kernel(const u32x4* __restrict__ src, size_t n16, uint32_t* sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (size_t)(UNROLL - 1) * blockDim.x < n16; i += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = NT ? __builtin_nontemporal_load(src + i + (size_t)k * blockDim.x) : src[i + (size_t)k * blockDim.x];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) acc ^= v[k];
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) sink[0] = r;   // never true for random data; keeps the loads alive
}

template <int UNROLL, bool NT>
static void run(const u32x4* d, size_t n16, uint32_t* sink, int wg, int blocks_per_cu, const char* tag)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * blocks_per_cu;
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((read_kernel<UNROLL, NT>), dim3(grid), dim3(wg), 0, 0, d, n16, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-28s wg=%4d blocks/CU=%2d unroll=%d  %.3f ms  %.0f GB/s\n", tag, wg, blocks_per_cu, UNROLL, best, n16 * 16.0 / best / 1e6);
}

int main()
{
    const size_t bytes = (size_t)4096 * 1080 * 1920 * 3;
    u32x4* d; uint32_t* sink;
    if (hipMalloc((void**)&d, bytes) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0x5a, bytes);
    hipDeviceSynchronize();
    const size_t n16 = bytes / 16;
    for (int bpc : {4, 8, 16}) {
        run<4, true>(d, n16, sink, 256, bpc, "16-B loads, nontemporal");
        run<8, true>(d, n16, sink, 256, bpc, "16-B loads, nontemporal");
        run<4, false>(d, n16, sink, 256, bpc, "16-B loads, default policy");
        run<4, true>(d, n16, sink, 1024, bpc / 4 ? bpc / 4 : 1, "16-B loads, nontemporal");
    }
    return 0;
}
