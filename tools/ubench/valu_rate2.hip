// Round 2: encoding / select / LDS-feed costs on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define DEFK(NAME, NI, ASM)                                                                          \
    __global__ void k_##NAME(uint32_t* out, int iters)                                               \
    {                                                                                                \
        __shared__ uint32_t lds[16384];                                                              \
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;              \
        __syncthreads();                                                                             \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,  \
                 a6 = a0 + 6, a7 = a0 + 7;                                                           \
        uint32_t b = threadIdx.x * 3 + 1, c = 0x01020304u + threadIdx.x;                             \
        uint32_t addr = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * 12 + (threadIdx.x >> 6) * 3072; \
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
#define A_ADD64(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define A_CMPSEL32(i) "v_cmp_gt_u32_e32 vcc, %" #i ", %8\nv_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n"
#define A_CMPSEL64(i) "v_cmp_gt_u32_e64 s[20:21], %" #i ", %8\nv_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n"
#define A_SEL64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n"
#define A_SEL32(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n"
#define A_FMAC(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
#define A_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MAXU16(i) "v_max_u16_e32 %" #i ", %" #i ", %8\n"
#define A_SUBU16(i) "v_sub_u16_e32 %" #i ", %" #i ", %8\n"
#define A_MADU16(i) "v_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define A_PKADDSEL(i) "v_pk_add_u16 %" #i ", %" #i ", %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
#define A_PKMIN(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define A_PKLSHL(i) "v_pk_lshlrev_b16 %" #i ", 2, %" #i "\n"
#define A_MULU24(i) "v_mul_u32_u24_e32 %" #i ", %" #i ", %8\n"
#define A_MULI24(i) "v_mul_i32_i24_e32 %" #i ", %" #i ", %8\n"
#define A_MULHI24(i) "v_mul_hi_u32_u24_e32 %" #i ", %" #i ", %8\n"
#define A_MAXI32(i) "v_max_i32_e32 %" #i ", %" #i ", %8\n"
#define A_MINU32(i) "v_min_u32_e32 %" #i ", %" #i ", %8\n"
#define A_LSHL32(i) "v_lshlrev_b32_e32 %" #i ", 7, %" #i "\n"
#define A_OR32(i) "v_or_b32_e32 %" #i ", %" #i ", %8\n"
#define A_SUBREV(i) "v_subrev_u32_e32 %" #i ", %" #i ", %8\n"
#define A_ADDC(i) "v_addc_co_u32_e32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define A_DSU8(i) "ds_read_u8 %" #i ", %10 offset:" #i "\n"
#define A_DSU8D16(i) "ds_read_u8_d16 %" #i ", %10 offset:" #i "\nds_read_u8_d16_hi %" #i ", %10 offset:768\n"
#define A_DSB32(i) "ds_read_b32 %" #i ", %10 offset:" #i "*4\n"
#define A_DSB128(i) "ds_read_b64 %" #i ", %10 offset:8\n"
#define A_MIX(i) "ds_read_u8 %" #i ", %10 offset:" #i "\nv_add_u32_e32 %8, %8, %9\nv_add_u32_e32 %9, %9, %8\nv_add_u32_e32 %8, %8, %9\n"
#define A_MADU24K(i) "v_mad_u32_u24 %" #i ", %" #i ", 9, %8\n"
#define A_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define A_CVTPK(i) "v_cvt_pk_u8_f32 %" #i ", %8, 1, %" #i "\n"
#define A_SADU16(i) "v_sad_u16 %" #i ", %8, %9, %" #i "\n"
#define A_SADU32(i) "v_sad_u32 %" #i ", %8, %9, %" #i "\n"
#define A_MED3(i) "v_med3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MED3I16(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"

DEFK(add32, 1, A_ADD32) DEFK(add64, 1, A_ADD64) DEFK(cmpsel32, 2, A_CMPSEL32) DEFK(cmpsel64, 2, A_CMPSEL64)
DEFK(sel64, 1, A_SEL64) DEFK(sel32, 1, A_SEL32) DEFK(fmac, 1, A_FMAC) DEFK(maxu16, 1, A_MAXU16)
DEFK(subu16, 1, A_SUBU16) DEFK(madu16, 1, A_MADU16) DEFK(pkaddsel, 1, A_PKADDSEL) DEFK(pkmin, 1, A_PKMIN)
DEFK(pklshl, 1, A_PKLSHL) DEFK(mulu24, 1, A_MULU24) DEFK(muli24, 1, A_MULI24) DEFK(mulhi24, 1, A_MULHI24)
DEFK(maxi32, 1, A_MAXI32) DEFK(minu32, 1, A_MINU32) DEFK(lshl32, 1, A_LSHL32) DEFK(or32, 1, A_OR32)
DEFK(subrev, 1, A_SUBREV) DEFK(addc, 1, A_ADDC) DEFK(dsu8, 1, A_DSU8) DEFK(dsu8d16, 2, A_DSU8D16) DEFK(dsb32, 1, A_DSB32)
DEFK(mix, 4, A_MIX) DEFK(madu24k, 1, A_MADU24K) DEFK(alignbyte, 1, A_ALIGNBYTE) DEFK(sadu16, 1, A_SADU16)
DEFK(sadu32, 1, A_SADU32) DEFK(med3, 1, A_MED3) DEFK(pkmaxi16, 1, A_MED3I16)

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
    std::vector<K> ks = {E(add32, 1), E(add64, 1), E(cmpsel32, 2), E(cmpsel64, 2), E(sel64, 1), E(sel32, 1), E(fmac, 1),
                         E(maxu16, 1), E(subu16, 1), E(madu16, 1), E(pkaddsel, 1), E(pkmin, 1), E(pklshl, 1), E(mulu24, 1),
                         E(muli24, 1), E(mulhi24, 1), E(maxi32, 1), E(minu32, 1), E(lshl32, 1), E(or32, 1), E(subrev, 1), E(addc, 1),
                         E(dsu8, 1), E(dsu8d16, 2), E(dsb32, 1), E(mix, 4), E(madu24k, 1), E(alignbyte, 1), E(sadu16, 1), E(sadu32, 1),
                         E(med3, 1), E(pkmaxi16, 1)};
    for (int wps : {2, 4}) {
        printf("--- %d wave(s) per SIMD: cycles per wave64 instruction per SIMD (at %.2f GHz)\n", wps, ghz);
        for (auto& k : ks) {
            dim3 grid(p.multiProcessorCount), block(256 * wps);
            hipLaunchKernelGGL(k.fn, grid, block, 0, 0, d, 10);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, grid, block, 0, 0, d, iters);
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
