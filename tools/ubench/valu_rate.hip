// Micro-benchmark: issue rate (cycles per wave64 instruction per SIMD) of the integer VALU ops the
// scoring kernel is made of, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFK(NAME, ASM)                                                                             \
    __global__ void k_##NAME(uint32_t* out, int iters)                                               \
    {                                                                                                \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,  \
                 a6 = a0 + 6, a7 = a0 + 7;                                                           \
        uint32_t b = threadIdx.x * 3 + 1, c = 0x01020304u + threadIdx.x;                             \
        for (int i = 0; i < iters; i++) {                                                            \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c) : "vcc");                                                  \
        }                                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    }

#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_MAX(i) "v_max_u32 %" #i ", %" #i ", %8\n"
#define A_MAX3(i) "v_max3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_MADI24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 7, %8\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 8, %8\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %8, %9, %" #i "\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMP(i) "v_cmp_eq_u32 vcc, %" #i ", %8\n"
#define A_CMPS(i) "v_cmp_eq_u32 s[20:21], %" #i ", %8\n"
#define A_PKMAX(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n"
#define A_PKSUB(i) "v_pk_sub_u16 %" #i ", %" #i ", %8\n"
#define A_PKMAD(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define A_PKMUL(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define A_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define A_SDWADST(i) "v_add_u32_sdwa %" #i ", %8, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PKFMA(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define A_CVTU32(i) "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define A_MADU16(i) "v_mad_u32_u16 %" #i ", %8, %9, %" #i "\n"
#define A_SALU(i) "s_add_u32 s20, s20, 1\n"
#define A_MINU16(i) "v_min_u16 %" #i ", %" #i ", %8\n"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i "\n"
#define A_MSAD(i) "v_msad_u8 %" #i ", %8, %9, %" #i "\n"
#define A_LERP(i) "v_lerp_u8 %" #i ", %" #i ", %8, %9\n"

DEFK(add, A_ADD) DEFK(sub, A_SUB) DEFK(max, A_MAX) DEFK(max3, A_MAX3) DEFK(min3, A_MIN3) DEFK(madu24, A_MADU24)
DEFK(madi24, A_MADI24) DEFK(lshladd, A_LSHLADD) DEFK(lshlor, A_LSHLOR) DEFK(and_, A_AND) DEFK(bfe, A_BFE)
DEFK(lshr, A_LSHR) DEFK(ashr, A_ASHR) DEFK(perm, A_PERM) DEFK(sad, A_SAD) DEFK(cndmask, A_CNDMASK) DEFK(cmp, A_CMP)
DEFK(cmps, A_CMPS) DEFK(pkmax, A_PKMAX) DEFK(pksub, A_PKSUB) DEFK(pkmad, A_PKMAD) DEFK(pkmul, A_PKMUL)
DEFK(addsdwa, A_ADDSDWA) DEFK(sdwadst, A_SDWADST) DEFK(fma, A_FMA) DEFK(mulf, A_PKFMA) DEFK(mullo, A_MULLO)
DEFK(cvtub, A_CVTUB) DEFK(cvtu32, A_CVTU32) DEFK(mov, A_MOV) DEFK(xor_, A_XOR) DEFK(or3, A_OR3) DEFK(add3, A_ADD3)
DEFK(bfi, A_BFI) DEFK(madu16, A_MADU16) DEFK(salu, A_SALU) DEFK(minu16, A_MINU16) DEFK(dot4, A_DOT4) DEFK(msad, A_MSAD)
DEFK(lerp, A_LERP)

// LDS read rate: replicated-by-lane table, b32 reads (conflict-free) and random-conflict reads
__global__ void k_ldsread(uint32_t* out, int iters, int conflict)
{
    __shared__ uint32_t tab[256 * 32];
    for (int i = threadIdx.x; i < 256 * 32; i += blockDim.x) tab[i] = i * 2654435761u;
    __syncthreads();
    uint32_t l32 = threadIdx.x & 31, acc = 0, idx = threadIdx.x * 7;
    for (int i = 0; i < iters * 4; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t v = (idx >> (j * 3)) & 255;
            acc += conflict ? tab[(v * 37 + l32 * 5) & 8191] : tab[v * 32 + l32];
        }
        idx = idx * 1664525u + acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_ldsatomic(uint32_t* out, int iters, int conflict)
{
    __shared__ uint32_t tab[256 * 32];
    for (int i = threadIdx.x; i < 256 * 32; i += blockDim.x) tab[i] = 0;
    __syncthreads();
    uint32_t l32 = threadIdx.x & 31, idx = threadIdx.x * 7 + 1;
    for (int i = 0; i < iters * 4; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t v = conflict == 2 ? 5 : ((idx >> (j * 3)) & 255);
            __hip_atomic_fetch_add(conflict ? &tab[v] : &tab[v * 32 + l32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        idx = idx * 1664525u + 1013904223u;
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = tab[threadIdx.x];
}

int main()
{
    uint32_t* d;
    hipMalloc(&d, 4096 * 1024 * 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double ghz = clk / 1e6;
    printf("device %s CUs=%d clock=%.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct K { const char* name; void (*fn)(uint32_t*, int); };
#define E(n) {#n, k_##n}
    std::vector<K> ks = {E(add), E(sub), E(max), E(max3), E(min3), E(madu24), E(madi24), E(lshladd), E(lshlor), E(and_), E(bfe),
                         E(lshr), E(ashr), E(perm), E(sad), E(cndmask), E(cmp), E(cmps), E(pkmax), E(pksub), E(pkmad), E(pkmul),
                         E(addsdwa), E(sdwadst), E(fma), E(mulf), E(mullo), E(cvtub), E(cvtu32), E(mov), E(xor_), E(or3), E(add3),
                         E(bfi), E(madu16), E(salu), E(minu16), E(dot4), E(msad), E(lerp)};
    for (int wps : {1, 4}) {  // waves per SIMD
        printf("--- %d wave(s) per SIMD: cycles per wave64 instruction per SIMD (assuming %.2f GHz)\n", wps, ghz);
        for (auto& k : ks) {
            dim3 grid(p.multiProcessorCount), block(256 * wps);
            hipLaunchKernelGGL(k.fn, grid, block, 0, 0, d, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, grid, block, 0, 0, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double instr_per_simd = (double)iters * 32 * wps;
            printf("%-10s %7.3f ms  %.2f cyc/instr\n", k.name, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
        }
    }
    for (int c : {0, 1}) {
        hipLaunchKernelGGL(k_ldsread, dim3(p.multiProcessorCount), dim3(1024), 0, 0, d, 10, c);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ldsread, dim3(p.multiProcessorCount), dim3(1024), 0, 0, d, 1000, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("ldsread conflict=%d: %.3f ms, %.2f cyc per wave-read per CU\n", c, ms, ms * 1e-3 * ghz * 1e9 / (1000.0 * 4 * 8 * 16));
    }
    for (int c : {0, 1, 2}) {
        hipLaunchKernelGGL(k_ldsatomic, dim3(p.multiProcessorCount), dim3(1024), 0, 0, d, 10, c);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ldsatomic, dim3(p.multiProcessorCount), dim3(1024), 0, 0, d, 1000, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("ldsatomic mode=%d: %.3f ms, %.2f cyc per wave-atomic per CU\n", c, ms, ms * 1e-3 * ghz * 1e9 / (1000.0 * 4 * 8 * 16));
    }
    return 0;
}
