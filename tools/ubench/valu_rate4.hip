// Round 2: the same instruction-class measurement as valu_rate3.hip, but every wave also reads the shader clock
// (s_memtime) and the constant 100 MHz clock (s_memrealtime) around its loop, so the table gives TRUE shader cycles
// per wave64 instruction and the clock the chip actually sustained under that instruction stream -- the earlier
// tables divided wall time by the nominal 2.4 GHz.  6 waves per SIMD (the HSV pass's occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define DEFK(NAME, NI, ASM)                                                                          \
    __global__ void k_##NAME(uint32_t* out, unsigned long long* clk, int iters)                      \
    {                                                                                                \
        __shared__ uint32_t lds[2048];                                                              \
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i * 2654435761u;              \
        __syncthreads();                                                                             \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,  \
                 a6 = a0 + 6, a7 = a0 + 7;                                                           \
        uint32_t b = threadIdx.x * 3 + 1, c = 0x01020304u + threadIdx.x;                             \
        uint32_t addr = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * 12 + (threadIdx.x >> 6) * 1024; \
        unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();                   \
        for (int i = 0; i < iters; i++) {                                                            \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                     \
                         "s_waitcnt lgkmcnt(0)\n"                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c), "v"(addr) : "vcc", "s20", "s21", "memory");               \
        }                                                                                            \
        unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();                   \
        if ((threadIdx.x & 63) == 0) {                                                               \
            const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                              \
            clk[2 * w] = t1 - t0; clk[2 * w + 1] = r1 - r0;                                          \
        }                                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;          \
    }

#define A_ADD32(i) "v_add_u32_e32 %" #i ", %" #i ", %8\n"
#define A_SUBF(i) "v_sub_f32_e32 %" #i ", %" #i ", %8\n"
#define A_SUBFCL(i) "v_sub_f32_e64 %" #i ", %" #i ", %8 clamp\n"
#define A_MULFCL(i) "v_mul_f32_e64 %" #i ", %" #i ", -1.0 clamp\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_FMAK(i) "v_fma_f32 %" #i ", %" #i ", 2.0, %9\n"
#define A_FMAC(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
#define A_MAX3F(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MAXF(i) "v_max_f32_e32 %" #i ", %" #i ", %8\n"
#define A_MAXU16(i) "v_max_u16_e32 %" #i ", %" #i ", %8\n"
#define A_MINU16(i) "v_min_u16_e32 %" #i ", %" #i ", %8\n"
#define A_LSHR16(i) "v_lshrrev_b16_e32 %" #i ", 8, %" #i "\n"
#define A_LSHL16(i) "v_lshlrev_b16_e32 %" #i ", 4, %" #i "\n"
#define A_LSHL32(i) "v_lshlrev_b32_e32 %" #i ", 4, %" #i "\n"
#define A_LSHR32(i) "v_lshrrev_b32_e32 %" #i ", 24, %" #i "\n"
#define A_ASHR32(i) "v_ashrrev_i32_e32 %" #i ", 31, %" #i "\n"
#define A_AND(i) "v_and_b32_e32 %" #i ", %" #i ", %8\n"
#define A_ANDK(i) "v_and_b32_e32 %" #i ", 0xff, %" #i "\n"
#define A_OR(i) "v_or_b32_e32 %" #i ", %" #i ", %8\n"
#define A_XOR(i) "v_xor_b32_e32 %" #i ", %" #i ", %8\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX3U(i) "v_max3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 4, %8\n"
#define A_SADU8(i) "v_sad_u8 %" #i ", %8, %9, %" #i "\n"
#define A_MINSDWA(i) "v_min_u16_sdwa %" #i ", %8, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define A_CMPSEL(i) "v_cmp_eq_u32_e32 vcc, %" #i ", %8\nv_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n"
#define A_PKMAX(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n"
#define A_PKSUB(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define A_PKMAD(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define A_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define A_CVTUB1(i) "v_cvt_f32_ubyte1_e32 %" #i ", %" #i "\n"
#define A_MOV(i) "v_mov_b32_e32 %" #i ", %8\n"
#define A_DSB32(i) "ds_read_b32 %" #i ", %10 offset:" #i "*4\n"
#define A_NOP(i) "s_nop 0\n"
// a stream shaped like the HSV pass: 3 fast : 7 slow
#define A_MIXHSV(i) "v_sub_u32_e32 %" #i ", %" #i ", %8\nv_max3_u32 %" #i ", %" #i ", %8, %9\nv_lshl_add_u32 %" #i ", %" #i ", 1, %8\nv_perm_b32 %" #i ", %" #i ", %8, %9\n"
// fp-heavy stream: 3 fast fp : 1 slow
#define A_MIXFP(i) "v_sub_f32_e32 %" #i ", %" #i ", %8\nv_fma_f32 %" #i ", %" #i ", %8, %9\nv_fma_f32 %" #i ", %" #i ", 2.0, %9\nv_perm_b32 %" #i ", %" #i ", %8, %9\n"

DEFK(add32, 1, A_ADD32) DEFK(subf, 1, A_SUBF) DEFK(subfcl, 1, A_SUBFCL) DEFK(mulfcl, 1, A_MULFCL) DEFK(fma, 1, A_FMA)
DEFK(fmak, 1, A_FMAK) DEFK(fmac, 1, A_FMAC) DEFK(max3f, 1, A_MAX3F) DEFK(maxf, 1, A_MAXF) DEFK(maxu16, 1, A_MAXU16)
DEFK(minu16, 1, A_MINU16) DEFK(lshr16, 1, A_LSHR16) DEFK(lshl16, 1, A_LSHL16) DEFK(lshl32, 1, A_LSHL32) DEFK(lshr32, 1, A_LSHR32)
DEFK(ashr32, 1, A_ASHR32) DEFK(and32, 1, A_AND) DEFK(andk, 1, A_ANDK) DEFK(or32, 1, A_OR) DEFK(xor32, 1, A_XOR) DEFK(bfe, 1, A_BFE)
DEFK(perm, 1, A_PERM) DEFK(max3u, 1, A_MAX3U) DEFK(madu24, 1, A_MADU24) DEFK(lshladd, 1, A_LSHLADD) DEFK(sadu8, 1, A_SADU8)
DEFK(minsdwa, 1, A_MINSDWA) DEFK(cmpsel, 2, A_CMPSEL) DEFK(pkmax, 1, A_PKMAX) DEFK(pksub, 1, A_PKSUB) DEFK(pkmad, 1, A_PKMAD)
DEFK(bfi, 1, A_BFI) DEFK(cvtub1, 1, A_CVTUB1) DEFK(mov, 1, A_MOV) DEFK(dsb32, 1, A_DSB32) DEFK(nop, 1, A_NOP)
DEFK(mixhsv, 4, A_MIXHSV) DEFK(mixfp, 4, A_MIXFP)

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int wps = 6, iters = 3000;
    const int blocks = p.multiProcessorCount * wps, waves = blocks * 4;
    uint32_t* d;
    unsigned long long* dclk;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&dclk, (size_t)waves * 16);
    std::vector<unsigned long long> h(waves * 2);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct K { const char* name; int ni; void (*fn)(uint32_t*, unsigned long long*, int); };
#define E(n, ni) {#n, ni, k_##n}
    std::vector<K> ks = {E(add32, 1), E(subf, 1), E(subfcl, 1), E(mulfcl, 1), E(fma, 1), E(fmak, 1), E(fmac, 1), E(max3f, 1), E(maxf, 1),
                         E(maxu16, 1), E(minu16, 1), E(lshr16, 1), E(lshl16, 1), E(lshl32, 1), E(lshr32, 1), E(ashr32, 1), E(and32, 1),
                         E(andk, 1), E(or32, 1), E(xor32, 1), E(bfe, 1), E(perm, 1), E(max3u, 1), E(madu24, 1), E(lshladd, 1), E(sadu8, 1),
                         E(minsdwa, 1), E(cmpsel, 2), E(pkmax, 1), E(pksub, 1), E(pkmad, 1), E(bfi, 1), E(cvtub1, 1), E(mov, 1),
                         E(dsb32, 1), E(nop, 1), E(mixhsv, 4), E(mixfp, 4)};
    printf("%d waves per SIMD, %d CUs; true cycles = s_memtime ticks / instructions issued by the SIMD's %d waves\n", wps, p.multiProcessorCount, wps);
    printf("%-9s %8s %10s %12s %10s\n", "op", "ms", "cyc/instr", "clock GHz", "nominal");
    for (auto& k : ks) {
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, dclk, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, dclk, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), dclk, (size_t)waves * 16, hipMemcpyDeviceToHost);
        double cyc = 0, real = 0;
        for (int w = 0; w < waves; w++) { cyc += (double)h[2 * w]; real += (double)h[2 * w + 1]; }
        cyc /= waves; real /= waves;
        const double instr_per_simd = (double)iters * 32 * wps * k.ni;
        printf("%-9s %8.3f %10.2f %12.3f %10.2f\n", k.name, ms, cyc / instr_per_simd, cyc / real * 0.1, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    return 0;
}
