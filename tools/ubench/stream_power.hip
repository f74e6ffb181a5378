// Energy of the frame stream by access path (round 2): the scoring pass runs at the package power limit, so what a byte
// costs on its way into the registers matters as much as the bandwidth.  Each mode streams the same 25.5 GB buffer
// over and over for a few seconds (tools/power_ab.sh samples rocm-smi meanwhile) and prints the bandwidth it sustained:
//   x4   16-byte non-temporal loads straight into registers, lanes 16 bytes apart (the streaming ceiling, stream_read.hip)
//   x3   12-byte non-temporal loads straight into registers, lanes 12 bytes apart (four BGR pixels per lane, no staging)
//   dma  global_load_lds_dwordx4 into wave-private LDS slots + three ds_read_b128 per lane (the scoring kernels' path)
//   usage: stream_power <x4|x3|dma> [seconds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 __attribute__((aligned(4))) u32x3_u;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int UNROLL = 4;

__global__ __launch_bounds__(256) void k_x4(const u32x4* __restrict__ src, size_t n16, u32* sink)
{
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (size_t)(UNROLL - 1) * 256 < n16; i += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = __builtin_nontemporal_load(src + i + (size_t)k * 256);
#pragma unroll
        for (int k = 0; k < UNROLL; k++) acc ^= v[k];
    }
    const u32 r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) sink[0] = r;
}

__global__ __launch_bounds__(256) void k_x3(const uint8_t* __restrict__ src, size_t n12, u32* sink)
{
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    u32x3 acc = {0, 0, 0};
    for (; i + (size_t)(UNROLL - 1) * 256 < n12; i += stride) {
        u32x3 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x3_u*>(src + (i + (size_t)k * 256) * 12));
#pragma unroll
        for (int k = 0; k < UNROLL; k++) acc ^= v[k];
    }
    const u32 r = acc.x ^ acc.y ^ acc.z;
    if (r == 0x12345678u) sink[0] = r;
}

// 4 waves; each wave streams 3 KiB pieces through its own LDS slot, one piece ahead.  AUX = cache policy bits of the
// LDS-DMA load (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX>
__global__ __launch_bounds__(256) void k_dma(const uint8_t* __restrict__ src, size_t n_pieces, u32* sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[2][4][3072];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wave_id = (size_t)blockIdx.x * 4 + wave, n_waves = (size_t)gridDim.x * 4;
    u32x4 acc = {0, 0, 0, 0};
    auto issue = [&](size_t piece, int buf) {
        const uint8_t* p = src + piece * 3072;
#pragma unroll
        for (int j = 0; j < 3; j++)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p + j * 1024 + lane * 16), (lds_ptr_t)(&stage[buf][wave][j * 1024]), 16, 0, AUX);
    };
    size_t piece = wave_id;
    int buf = 0;
    if (piece < n_pieces) issue(piece, 0);
    for (; piece < n_pieces; piece += n_waves, buf ^= 1) {
        if (piece + n_waves < n_pieces) {
            issue(piece + n_waves, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const u32x4* s = reinterpret_cast<const u32x4*>(&stage[buf][wave][lane * 48]);
        const u32x4 a = s[0], b = s[1], c = s[2];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc ^= a ^ b ^ c;
    }
    const u32 r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) sink[0] = r;
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "x4";
    const double seconds = argc > 2 ? atof(argv[2]) : 6.0;
    const int blocks_per_cu = argc > 3 ? atoi(argv[3]) : 8;
    const size_t bytes = (size_t)4096 * 1080 * 1920 * 3;
    uint8_t* d; u32* sink;
    if (hipMalloc((void**)&d, bytes) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0x5a, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * blocks_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    double total_ms = 0;
    int launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(a);
        for (int r = 0; r < 8; r++) {
            if (!strcmp(mode, "x4")) hipLaunchKernelGGL(k_x4, dim3(grid), dim3(256), 0, 0, (const u32x4*)d, bytes / 16, sink);
            else if (!strcmp(mode, "x3")) hipLaunchKernelGGL(k_x3, dim3(grid), dim3(256), 0, 0, d, bytes / 12, sink);
            else if (!strcmp(mode, "dma")) hipLaunchKernelGGL(k_dma<2>, dim3(grid), dim3(256), 0, 0, d, bytes / 3072, sink);
            else if (!strcmp(mode, "dma0")) hipLaunchKernelGGL(k_dma<0>, dim3(grid), dim3(256), 0, 0, d, bytes / 3072, sink);
            else if (!strcmp(mode, "dma1")) hipLaunchKernelGGL(k_dma<1>, dim3(grid), dim3(256), 0, 0, d, bytes / 3072, sink);
            else if (!strcmp(mode, "dma3")) hipLaunchKernelGGL(k_dma<3>, dim3(grid), dim3(256), 0, 0, d, bytes / 3072, sink);
            else if (!strcmp(mode, "dma16")) hipLaunchKernelGGL(k_dma<16>, dim3(grid), dim3(256), 0, 0, d, bytes / 3072, sink);
            else if (!strcmp(mode, "dma17")) hipLaunchKernelGGL(k_dma<17>, dim3(grid), dim3(256), 0, 0, d, bytes / 3072, sink);
            else if (!strcmp(mode, "dma18")) hipLaunchKernelGGL(k_dma<18>, dim3(grid), dim3(256), 0, 0, d, bytes / 3072, sink);
            else if (!strcmp(mode, "dma19")) hipLaunchKernelGGL(k_dma<19>, dim3(grid), dim3(256), 0, 0, d, bytes / 3072, sink);
            else { printf("unknown mode %s\n", mode); return 1; }
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        total_ms += ms; launches += 8;
    }
    printf("%s blocks/CU=%d: %d launches, %.3f ms each, %.0f GB/s\n", mode, blocks_per_cu, launches, total_ms / launches, bytes / (total_ms / launches) / 1e6);
    return 0;
}
