// Energy per operation: plain vs packed fp32 (round 2).  The HSV pass is power-bound, so an instruction that does two
// pixels' worth of work is only a gain if it costs less energy than two single ones.  Each mode issues a long stream of
// one instruction kind from six waves per SIMD for a few seconds (rocm-smi samples power meanwhile, tools/pk_power.sh) and
// prints the operation rate it sustained.
//   modes: fma (v_fma_f32), pkfma (v_pk_fma_f32), add (v_add_f32), pkadd (v_pk_add_f32), perm (v_perm_b32), mix (3 fma : 1 perm)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float b = 0.999f, c = 1e-4f;
    const f32x2 pb = {b, b}, pc = {c, c};
    unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                                        "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            if (MODE == 2) asm volatile("v_add_f32 %0, %0, %8\nv_add_f32 %1, %1, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\n"
                                        "v_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_add_f32 %6, %6, %8\nv_add_f32 %7, %7, %8\n"
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n"
                                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));
            if (MODE == 4) asm volatile("v_perm_b32 %0, %0, %1, %4\nv_perm_b32 %1, %1, %2, %4\nv_perm_b32 %2, %2, %3, %4\nv_perm_b32 %3, %3, %0, %4\n"
                                        "v_perm_b32 %0, %0, %1, %4\nv_perm_b32 %1, %1, %2, %4\nv_perm_b32 %2, %2, %3, %4\nv_perm_b32 %3, %3, %0, %4\n"
                                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x07020500u));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 + u1 + u2 + u3);
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "fma";
    const double seconds = argc > 2 ? atof(argv[2]) : 6.0;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * 6, iters = 20000;
    float* d;
    (void)hipMalloc(&d, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double ms_total = 0; int launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        (void)hipEventRecord(e0);
        if (!strcmp(mode, "fma")) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters);
        else if (!strcmp(mode, "pkfma")) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters);
        else if (!strcmp(mode, "add")) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, iters);
        else if (!strcmp(mode, "pkadd")) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, iters);
        else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        ms_total += ms; launches++;
    }
    // float operations (lanes x instructions x width) per second: 8 scalar or 4 packed (= 8 floats) or 8 perms per inner line
    const double ops = (double)grid * 256 * iters * 8 * 8;
    printf("%s: %d launches, %.3f ms each, %.2f T lane-ops/s\n", mode, launches, ms_total / launches, ops / (ms_total / launches) / 1e9);
    return 0;
}
