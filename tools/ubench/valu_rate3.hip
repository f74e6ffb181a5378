// Round 3 (prepared at the end of round 1, not yet run): the instruction classes the fp32-pipe formulation of the HSV
// arithmetic would lean on -- byte-to-float unpack, fp32 min/max/sub, float<->int conversions,
// byte packing from float, 16-bit shifts, dot / 16-bit multiply-add forms -- at 4, 5 and 6 waves per SIMD (the HSV
// pass now runs six).  Same method as valu_rate2.hip: 32 independent instructions per loop slot per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define DEFK(NAME, NI, ASM)                                                                          \
    __global__ void k_##NAME(uint32_t* out, int iters)                                               \
    {                                                                                                \
        __shared__ uint32_t lds[2048];                                                              \
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i * 2654435761u;              \
        __syncthreads();                                                                             \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,  \
                 a6 = a0 + 6, a7 = a0 + 7;                                                           \
        uint32_t b = threadIdx.x * 3 + 1, c = 0x01020304u + threadIdx.x;                             \
        uint32_t addr = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * 12 + (threadIdx.x >> 6) * 1024; \
        for (int i = 0; i < iters; i++) {                                                            \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         "s_waitcnt lgkmcnt(0)\n"                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c), "v"(addr) : "vcc", "s20", "s21", "memory");               \
        }                                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    }

#define A_ADD32(i) "v_add_u32_e32 %" #i ", %" #i ", %8\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define A_CVTUB0(i) "v_cvt_f32_ubyte0_e32 %" #i ", %" #i "\n"
#define A_CVTUB1(i) "v_cvt_f32_ubyte1_e32 %" #i ", %" #i "\n"
#define A_CVTUB3(i) "v_cvt_f32_ubyte3_e32 %" #i ", %" #i "\n"
#define A_MAXF(i) "v_max_f32_e32 %" #i ", %" #i ", %8\n"
#define A_MINF(i) "v_min_f32_e32 %" #i ", %" #i ", %8\n"
#define A_SUBF(i) "v_sub_f32_e32 %" #i ", %" #i ", %8\n"
#define A_MULF(i) "v_mul_f32_e32 %" #i ", %" #i ", %8\n"
#define A_FMAC(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX3F(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_CVTU32F(i) "v_cvt_u32_f32_e32 %" #i ", %" #i "\n"
#define A_CVTI32F(i) "v_cvt_i32_f32_e32 %" #i ", %" #i "\n"
#define A_CVTFU32(i) "v_cvt_f32_u32_e32 %" #i ", %" #i "\n"
#define A_FLOORF(i) "v_floor_f32_e32 %" #i ", %" #i "\n"
#define A_CVTPKU8(i) "v_cvt_pk_u8_f32 %" #i ", %8, 1, %" #i "\n"
#define A_CMPEQF(i) "v_cmp_eq_f32_e32 vcc, %" #i ", %8\nv_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n"
#define A_LSHR16(i) "v_lshrrev_b16_e32 %" #i ", 8, %" #i "\n"
#define A_LSHL16(i) "v_lshlrev_b16_e32 %" #i ", 4, %" #i "\n"
#define A_AND(i) "v_and_b32_e32 %" #i ", %" #i ", %8\n"
#define A_LSHR32(i) "v_lshrrev_b32_e32 %" #i ", 24, %" #i "\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 4, %8\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 4, %8\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX3U(i) "v_max3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_MADI24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n"
#define A_MADU32U16(i) "v_mad_u32_u16 %" #i ", %" #i ", %8, %9\n"
#define A_MADI32I16(i) "v_mad_i32_i16 %" #i ", %" #i ", %8, %9\n"
#define A_DOT2U(i) "v_dot2_u32_u16 %" #i ", %" #i ", %8, %9\n"
#define A_DOT4U(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define A_SADU8(i) "v_sad_u8 %" #i ", %8, %9, %" #i "\n"
#define A_MINSDWA(i) "v_min_u16_sdwa %" #i ", %8, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define A_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define A_DSB32(i) "ds_read_b32 %" #i ", %10 offset:" #i "*4\n"
#define A_DSB128(i) "ds_read_b128 %" #i ", %10 offset:16\n"

DEFK(add32, 1, A_ADD32) DEFK(bfe, 1, A_BFE) DEFK(cvtub0, 1, A_CVTUB0) DEFK(cvtub1, 1, A_CVTUB1) DEFK(cvtub3, 1, A_CVTUB3)
DEFK(maxf, 1, A_MAXF) DEFK(minf, 1, A_MINF) DEFK(subf, 1, A_SUBF) DEFK(mulf, 1, A_MULF) DEFK(fmac, 1, A_FMAC) DEFK(fma, 1, A_FMA)
DEFK(max3f, 1, A_MAX3F) DEFK(cvtu32f, 1, A_CVTU32F) DEFK(cvti32f, 1, A_CVTI32F) DEFK(cvtfu32, 1, A_CVTFU32) DEFK(floorf, 1, A_FLOORF)
DEFK(cvtpku8, 1, A_CVTPKU8) DEFK(cmpeqf, 2, A_CMPEQF) DEFK(lshr16, 1, A_LSHR16) DEFK(lshl16, 1, A_LSHL16) DEFK(and32, 1, A_AND)
DEFK(lshr32, 1, A_LSHR32) DEFK(lshladd, 1, A_LSHLADD) DEFK(lshlor, 1, A_LSHLOR) DEFK(andor, 1, A_ANDOR) DEFK(perm, 1, A_PERM)
DEFK(max3u, 1, A_MAX3U) DEFK(madu24, 1, A_MADU24) DEFK(madi24, 1, A_MADI24) DEFK(madu32u16, 1, A_MADU32U16)
DEFK(madi32i16, 1, A_MADI32I16) DEFK(dot2u, 1, A_DOT2U) DEFK(dot4u, 1, A_DOT4U) DEFK(sadu8, 1, A_SADU8) DEFK(minsdwa, 1, A_MINSDWA)
DEFK(addsdwa, 1, A_ADDSDWA) DEFK(dsb32, 1, A_DSB32)

int main()
{
    uint32_t* d;
    (void)hipMalloc(&d, 4096 * 2048 * 4);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const double ghz = 2.4;
    const int iters = 3000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct K { const char* name; int ni; void (*fn)(uint32_t*, int); };
#define E(n, ni) {#n, ni, k_##n}
    std::vector<K> ks = {E(add32, 1), E(bfe, 1), E(cvtub0, 1), E(cvtub1, 1), E(cvtub3, 1), E(maxf, 1), E(minf, 1), E(subf, 1),
                         E(mulf, 1), E(fmac, 1), E(fma, 1), E(max3f, 1), E(cvtu32f, 1), E(cvti32f, 1), E(cvtfu32, 1), E(floorf, 1),
                         E(cvtpku8, 1), E(cmpeqf, 2), E(lshr16, 1), E(lshl16, 1), E(and32, 1), E(lshr32, 1), E(lshladd, 1),
                         E(lshlor, 1), E(andor, 1), E(perm, 1), E(max3u, 1), E(madu24, 1), E(madi24, 1), E(madu32u16, 1),
                         E(madi32i16, 1), E(dot2u, 1), E(dot4u, 1), E(sadu8, 1), E(minsdwa, 1), E(addsdwa, 1), E(dsb32, 1)};
    for (int wps : {4, 5, 6}) {
        printf("--- %d wave(s) per SIMD: cycles per wave64 instruction per SIMD (at %.2f GHz)\n", wps, ghz);
        for (auto& k : ks) {
            dim3 grid(p.multiProcessorCount), block(256), grid2(p.multiProcessorCount * wps);
            hipLaunchKernelGGL(k.fn, grid2, block, 0, 0, d, 10);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, grid2, block, 0, 0, d, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            double instr_per_simd = (double)iters * 32 * wps * k.ni;
            printf("%-10s %7.3f ms  %.2f cyc/instr (%d instr per slot)\n", k.name, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd, k.ni);
        }
    }
    return 0;
}
