// psd_score_kernels.hip -- fused per-frame scoring kernels for gfx950 (MI355X, CDNA4).
//
// One pass over packed BGR frames produces, per frame, the exact integer reductions that
// PySceneDetect's detectors derive from cv2/numpy (reference paths relative to its tree):
//   sad_h/s/v : sum |HSV_t - HSV_{t-1}|     content_detector.py:29-36,155,166-169
//   hist[256] : histogram of BT.601 luma Y   histogram_detector.py:156-159
//   byte_sum  : sum of all B,G,R bytes       threshold_detector.py:127
//
// Design:
//  * HSV needs the previous frame, so a workgroup owns a fixed spatial tile of the frame and WALKS THE TIME AXIS over
//    a chunk of T frames, carrying the previous frame's H,S,V for its pixels in registers (packed 4 px per dword).
//    Every pixel of the batch is read from HBM once (3 B/px algorithmic) and converted to HSV once; only the one
//    halo frame in front of each chunk is read twice.  The luma histogram / byte sum alone need no carry: that pass
//    (luma_hist_kernel) takes one tile of one frame per workgroup.
//  * A lane handles groups of 16 pixels = 48 contiguous bytes.  On the fast path the bytes travel HBM -> LDS with
//    global_load_lds_dwordx4 into wave-private slots (score_frames_dma_kernel); score_frames_kernel is the
//    register-load / byte-load form for strided, unaligned or ragged input.
//  * The two fixed-point division tables of the 8-bit HSV conversion live in LDS, replicated so the lanes of a wave
//    spread over the banks; the luma histogram is replicated the same way, so the LDS atomics never collide -- flat
//    frames (all pixels in one bin) cost the same as noise.
//  * Per-frame results leave the workgroup as one global atomic per non-empty bin / sum.
//    Everything is integer, so the result is independent of scheduling order.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <stdint.h>
#include <stdlib.h>

#include "psd_internal.h"

// PSD_ABLATE (build-time, experiments only -- results are WRONG when non-zero):
//   1 = no LDS table reads, 2 = no per-frame barrier/flush, 4 = no global loads inside the time loop,
//   8 = no HSV arithmetic (loads + byte SAD only), 16 = no histogram increments in the fused fp32 quad, 32 = no V-plane store (V mode).
//   tools/ablate.sh builds and times the variants.
#ifndef PSD_HIST_PAIR_ATOMICS
#define PSD_HIST_PAIR_ATOMICS 1
#endif
#ifndef PSD_ABLATE
#define PSD_ABLATE 0
#endif
#ifndef PSD_PACK3
#define PSD_PACK3 1
#endif

// Histogram increment in LDS that the compiler does not see as an LDS store.  hipcc orders every LDS store / atomic
// behind all outstanding LDS-DMA (global_load_lds) with an s_waitcnt vmcnt(0) -- a write-after-write guard it applies without
// looking at the addresses -- so a histogram update in the arithmetic of frame t waits for the staging slots of frame t + 1
// to be filled: the prefetch is serialised with the arithmetic inside every wave (the fused pass: 7.8 ms; frame loads alone
// 4.1 ms, arithmetic alone 5.8 ms).  The increments never touch the staging slots.  LDS operations of a wave complete in
// order, so the compiler's own lgkmcnt bookkeeping stays conservative; what it can no longer know is that increments are
// outstanding, hence lds_hidden_fence() in front of every barrier behind which another thread reads the accumulators.
__device__ __forceinline__ void lds_add_hidden(const uint32_t* p, uint32_t inc)
{
    asm volatile("ds_add_u32 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(inc) : "memory");
}
__device__ __forceinline__ void lds_add_hidden_at(uint32_t lds_byte_address, uint32_t inc)
{
    asm volatile("ds_add_u32 %0, %1" ::"v"(lds_byte_address), "v"(inc) : "memory");
}
__device__ __forceinline__ void lds_hidden_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#ifndef PSD_FUSED_PHASED
#define PSD_FUSED_PHASED 1   // the fused HSV + luma quad in phases (quad_fused); 0 = pixel by pixel as in round 1
#endif
// (Instruction-selection experiments that did NOT pay, measured on the HSV variant, N=1024 1080p:
//  0xff/180 from SGPRs instead of literals -1.7 %, shift+and instead of v_bfe -1.5 %, 2d/4d by
//  additions instead of v_lshl_add -4 %; the 16-bit v_min_u16 for the hue wrap +1.3 % is kept.)

namespace psd {

typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int COPIES = 32;           // replication factor of LUTs / accumulators (= b32 LDS banks)
constexpr int NACC = 260;            // 256 luma bins + sad_h, sad_s, sad_v, byte_sum
constexpr int ACC_SAD_H = 256, ACC_SAD_S = 257, ACC_SAD_V = 258, ACC_BYTES = 259;

// BT.601 luma in OpenCV's 14-bit fixed point (color_yuv.simd.hpp RGB2YCrCb_i).
constexpr u32 kB2Y = 1868, kG2Y = 9617, kR2Y = 4899;

struct Group {
    u32 w[12];  // 16 packed BGR pixels
};
struct Hsv16 {
    u32 h[4], s[4], v[4];  // 4 pixels per dword
};

// 24-bit multiply-adds, pinned to the full-rate VALU forms (left to itself hipcc widens the hue
// product to a quarter-rate v_mad_u64_u32).  All operands are provably within 24 bits:
// |hraw| <= 1275, hdiv180 <= 122880, diff <= 255, sdiv <= 1044480, BGR coefficients < 2^14.
__device__ __forceinline__ int mad_i24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ u32 mad_u24(u32 a, u32 b, u32 c)
{
    u32 d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ u32 mad_u24_vsv(u32 a, u32 b, u32 c)
{
    u32 d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}
// B2Y held in a VGPR the compiler cannot rematerialise as a literal move per pixel (the asm is pure,
// so it is hoisted out of the frame loop); VOP3 on gfx9 takes only one scalar/literal operand.
__device__ __forceinline__ u32 vgpr_b2y()
{
    u32 x;
    asm("v_mov_b32 %0, 0x74c" : "=v"(x));
    static_assert(kB2Y == 0x74c, "");
    return x;
}
__device__ __forceinline__ u32 mad_u24_vvs(u32 a, u32 b, u32 c)
{
    u32 d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

// (a == b) ? x : y as compare + v_cndmask.  Spelled in asm because hipcc otherwise turns the hue
// sector selection into real branches (s_and_saveexec / s_xor / s_or per pixel), which costs more
// scalar instructions than the three candidate subtractions it tries to skip.
__device__ __forceinline__ int sel_eq(u32 a, u32 b, int x, int y)
{
    int d;
    asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %4, %3, vcc" : "=v"(d) : "v"(a), "v"(b), "v"(x), "v"(y) : "vcc");
    return d;
}

// 16-bit unsigned min (VOP2, fast class; the upper halves are ignored and the result is zero-extended).
__device__ __forceinline__ u32 min_u16(u32 a, u32 b) { u32 d; asm("v_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// Packing without separate shift/or instructions:
//  * pack_byte2<K>: byte 2 of `src` -> byte K of the packed dword (v_perm_b32; K = 0 zero-fills the rest);
//    used for S (the table holds sdiv << 4, so the product's bits 16..23 are S) and for V.
//  * min_hi16_to_byte<K>: the hue-wrap min taken on the upper halves (the hdiv table is pre-shifted by 4
//    as well, so the hue sits in bits 16..31 of the product) and written straight into byte K (SDWA).
template <int K>
__device__ __forceinline__ u32 pack_byte2(u32 src, u32 packed)
{
    constexpr u32 sel = K == 0 ? 0x0c0c0c06u : K == 1 ? 0x03020600u : K == 2 ? 0x03060100u : 0x06020100u;
    u32 d;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(src), "v"(K == 0 ? src : packed), "s"(sel));
    return d;
}
template <int K>
__device__ __forceinline__ u32 pack_byte0(u32 src, u32 packed)
{
    constexpr u32 sel = K == 0 ? 0x0c0c0c04u : K == 1 ? 0x03020400u : K == 2 ? 0x03040100u : 0x04020100u;
    u32 d;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(src), "v"(K == 0 ? src : packed), "s"(sel));
    return d;
}
// min of the UPPER 16-bit halves of a and b (as unsigned), written into byte K of `packed`.
template <int K>
__device__ __forceinline__ u32 min_hi16_to_byte(u32 a, u32 b, u32 packed)
{
    if (K == 0) {
        u32 d;
        asm("v_min_u16_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(d) : "v"(a), "v"(b));
        return d;
    }
    if (K == 1) asm("v_min_u16_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(packed) : "v"(a), "v"(b));
    if (K == 2) asm("v_min_u16_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(packed) : "v"(a), "v"(b));
    if (K == 3) asm("v_min_u16_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(packed) : "v"(a), "v"(b));
    return packed;
}

// ---- loads ---------------------------------------------------------------------------------

template <bool FAST>
__device__ __forceinline__ void load_group(const ScoreParams& p, const uint8_t* __restrict__ frame,
                                           int group, Group& g)
{
    if (FAST) {
        const u32x4* src = reinterpret_cast<const u32x4*>(frame + (size_t)group * 48);
        u32x4 a = __builtin_nontemporal_load(src + 0);
        u32x4 b = __builtin_nontemporal_load(src + 1);
        u32x4 c = __builtin_nontemporal_load(src + 2);
        g.w[0] = a.x; g.w[1] = a.y; g.w[2] = a.z; g.w[3] = a.w;
        g.w[4] = b.x; g.w[5] = b.y; g.w[6] = b.z; g.w[7] = b.w;
        g.w[8] = c.x; g.w[9] = c.y; g.w[10] = c.z; g.w[11] = c.w;
        return;
    }
    // Generic path: any alignment, padded rows, ragged tail.  Pixels past the end read as 0.
    const int first_px = group * 16;
#pragma unroll
    for (int i = 0; i < 12; i++) g.w[i] = 0;
#pragma unroll 1
    for (int j = 0; j < 16; j++) {
        const int px = first_px + j;
        if (px < p.npix) {
            const int row = px / p.width, col = px - row * p.width;
            const uint8_t* s = frame + (size_t)row * p.row_stride + (size_t)col * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int bi = 3 * j + c;
                g.w[bi >> 2] |= (u32)s[c] << ((bi & 3) * 8);
            }
        }
    }
}

// ---- per-group arithmetic ------------------------------------------------------------------

// One pixel (K-th of a quad held in d[0..2]): exact 8-bit HSV as OpenCV's RGB2HSV_b, packed into
// byte K of hq/sq/vq, plus the luma histogram.  Both division tables are stored pre-shifted by 4 so
// that S and H sit 16-bit aligned in the products (255 * (sdiv << 4) + (2048 << 4) < 2^32).
template <bool HSV, bool LUMA, int LC, int AC, int K>
__device__ __forceinline__ void pixel(const u32 (&d)[3], u32& hq, u32& sq, u32& vq, const u32* __restrict__ lut_s,
                                      const u32* __restrict__ lut_h, u32* __restrict__ hist, u32 inc)
{
    constexpr int ib = 3 * K, ig = 3 * K + 1, ir = 3 * K + 2;
    const u32 b = (d[ib >> 2] >> ((ib & 3) * 8)) & 0xffu;
    const u32 gg = (d[ig >> 2] >> ((ig & 3) * 8)) & 0xffu;
    const u32 r = (d[ir >> 2] >> ((ir & 3) * 8)) & 0xffu;
    if (LUMA) {
        // (B2Y*b + G2Y*g + R2Y*r + 8192) >> 14; the rounding term rides on the first product
        const u32 y = mad_u24_vsv(r, kR2Y, mad_u24_vsv(gg, kG2Y, mad_u24_vvs(b, vgpr_b2y(), 8192u))) >> 14;
        // inc is 1, or 0 for a lane that owns no group here (cheaper than predicating every atomic)
        lds_add_hidden(&hist[y * AC], inc);
    }
#if PSD_ABLATE & 8
    if (HSV) { hq += b + (gg << 8); sq += r; vq ^= b; }
#else
    if (HSV) {
        const u32 v = max(max(b, gg), r);
        const u32 vmin = min(min(b, gg), r);
        const u32 diff = v - vmin;
#if PSD_ABLATE & 1
        const u32 sdiv16 = (v + 4097u) << 4;
        const int hdiv16 = ((int)diff + 480) << 4;
#else
        const u32 sdiv16 = lut_s[v * LC];
        const int hdiv16 = (int)lut_h[diff * LC];
#endif
        const u32 s16 = mad_u24(diff, sdiv16, 2048u << 4);   // S in bits 16..23
        const int dd = (int)diff;
        const int c_r = (int)gg - (int)b;
        const int c_g = (int)b - (int)r + 2 * dd;
        const int c_b = (int)r - (int)gg + 4 * dd;
        const int hraw = sel_eq(v, r, c_r, sel_eq(v, gg, c_g, c_b));
        // hh16 = (hraw*hdiv + 2048) << 4, so its upper half is OpenCV's (hraw*hdiv + 2048) >> 12 (arithmetic);
        // |hraw*hdiv| <= 5*122880 + 3*255, far inside 32 bits after the shift.
        const int hh16 = mad_i24(hraw, hdiv16, 2048 << 4);
        // hh < 0 ? hh + 180 : hh: as 16-bit unsigned values a negative hh is huge, so the min picks right
        hq = min_hi16_to_byte<K>((u32)hh16, (u32)hh16 + (180u << 16), hq);
#if PSD_PACK3
        // S and V leave the pixel unpacked; convert_group joins the four of a quad with three v_perm_b32 per
        // channel (pairwise, then pair of pairs) instead of one insertion per pixel
        sq = s16;
        vq = v;
#else
        sq = pack_byte2<K>(s16, sq);
        vq = pack_byte0<K>(v, vq);
#endif
    }
#endif
}

__device__ __forceinline__ u32 perm_b32(u32 s0, u32 s1, u32 sel) { return __builtin_amdgcn_perm(s0, s1, sel); }


// ---- fp32-pipe formulation of the same integers (PSD_HSV_FP32) ---------------------------------
//
// On gfx950 the plain fp32 add / sub / fma forms (and 32-bit add, logic, right shifts, 16-bit shifts) issue at twice
// the rate of everything else the integer formulation leans on (v_bfe, v_max3, 24-bit multiplies, v_lshl_add, compares
// and selects, v_perm, SDWA): tools/ubench/valu_rate3.hip / valu_rate4.hip.  Every quantity of OpenCV's 8-bit HSV is an
// integer far below 2^24 (or an integer / 4096 below 2^8), so float32 holds it exactly:
//   * a channel byte becomes the float 2^23 + byte with ONE v_perm_b32 (the byte under the three upper bytes of
//     0x4B000000); the bias cancels in every difference and leaves the integer in the low bits for the LDS address;
//   * v == r / v == g become the 0/1 floats clamp(v - r), clamp(v - g) (the VOP3 clamp modifier is free) and the two
//     selects become fused multiply-adds;
//   * S = RNE(diff * sdiv'[v] + 2^23) with sdiv' = nextafter(sdiv / 4096): the bump breaks the exact .5 ties upwards
//     like OpenCV's (x + 2048) >> 12 and moves nothing else (every other value is >= 2^-12 away from a tie);
//   * H = RNE((hraw * hdiv[diff] / 4096 + 2^-13) + 1.5 * 2^23): the first fma is exact (a multiple of 2^-12 below 2^8
//     plus 2^-13), the second rounds to floor(x + 0.5) for either sign and leaves h as a 16-bit two's complement in the
//     low half, where the wrap min(h, h + 180) of the integer formulation applies unchanged.
// tools/hsv_fp32_check.py replays this arithmetic in numpy float32 over all 2^24 triples; the -m gpu exhaustive test
// runs it through the kernel.
#ifndef PSD_HSV_FP32
#define PSD_HSV_FP32 1
#endif
#ifndef PSD_FUSED_FP32
#define PSD_FUSED_FP32 1    // the fused HSV + luma pass in the fp32 formulation too (luma from the hue differences, pixel_fp_luma_bits)
#endif
#ifndef PSD_HSV_SINGLE_S
#define PSD_HSV_SINGLE_S 1   // staged HSV-only pass: one copy of the sdiv table, its address is a single 16-bit shift (pixel_fp_front)
#endif

typedef const __attribute__((address_space(3))) float* lds_cf32_t;
typedef const __attribute__((address_space(3))) char* lds_cc_t;
typedef __attribute__((address_space(3))) char* lds_c_t;

// The LDS word at the absolute LDS byte address `abs`, which lies inside the object `obj` points into.  Written as an
// offset from `obj` (the arithmetic folds away: the address register is `abs` itself) so that the access keeps its
// provenance: an LDS address made from an integer may alias the staging slots as far as the compiler can tell, and it then
// puts an s_waitcnt vmcnt(0) -- "wait for the LDS-DMA of the NEXT frame" -- in front of the first table read of every step,
// which serialises the prefetch with the arithmetic inside a wave.
__device__ __forceinline__ float lds_f32_in(const u32* obj, u32 abs)
{
    const lds_cc_t b = (lds_cc_t)obj;
    return *(lds_cf32_t)(b + (abs - (u32)(uintptr_t)b));
}

struct FpLane {          // per-lane constants of the fp32 formulation
    u32 bias;            // 0x4B000000: float 2^23 with a zero low byte
    u32 off_s, off_h;    // LDS byte address of this lane's replica of the two tables
    const u32* obj_s;    // ... and the pointers they came from (provenance for lds_f32_in)
    const u32* obj_h;
    float bias_h;        // LS == 4 only: 2^21 + off_h / 16 (see pixel_fp_front)
    float bias_rel;      // REL only: 2^(23 - log2 LC) + replica / LC (see pixel_fp_front)
};

// (compiler builtins rather than inline asm wherever one exists: hipcc pads every use of an asm-defined register with
//  an s_nop because it cannot rule out the gfx950 trans-op hazard for an instruction it does not see)
template <int K>
__device__ __forceinline__ float unpack_biased(u32 w, u32 bias)
{
    // byte K of w under the upper three bytes of `bias`
    return __uint_as_float(__builtin_amdgcn_perm(w, bias, 0x03020104u + (u32)K));
}
// positive floats order like their bit patterns: integer max3 / min3 on the biased channels
__device__ __forceinline__ float max3_f32(float a, float b, float c)
{
    return __uint_as_float(max(max(__float_as_uint(a), __float_as_uint(b)), __float_as_uint(c)));
}
__device__ __forceinline__ float min3_f32(float a, float b, float c)
{
    return __uint_as_float(min(min(__float_as_uint(a), __float_as_uint(b)), __float_as_uint(c)));
}
// clamp(a - b) to [0, 1]: med3 with the constants 0 and 1 folds into the subtraction's clamp modifier
__device__ __forceinline__ float sub_clamp(float a, float b) { return __builtin_amdgcn_fmed3f(a - b, 0.0f, 1.0f); }
// (low 16 bits of x) << S, upper half zero
// (asm: written in C++ this becomes a 32-bit shift, which issues at half rate, plus an and)
template <int S>
__device__ __forceinline__ u32 lshl16(u32 x) { u32 d; asm("v_lshlrev_b16_e32 %0, %1, %2" : "=v"(d) : "n"(S), "v"(x)); return d; }
// ((low 16 bits of x) << S) + y
template <int S>
__device__ __forceinline__ u32 lshl16_add(u32 x, u32 y) { u32 d; asm("v_lshlrev_b16_e32 %0, %1, %2\n\tv_add_u32_e32 %0, %0, %3" : "=&v"(d) : "n"(S), "v"(x), "v"(y)); return d; }
template <int K>
__device__ __forceinline__ u32 min_lo16_to_byte(u32 a, u32 b, u32 packed)
{
    if (K == 0) {
        u32 d;
        asm("v_min_u16_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(d) : "v"(a), "v"(b));
        return d;
    }
    if (K == 1) asm("v_min_u16_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(packed) : "v"(a), "v"(b));
    if (K == 2) asm("v_min_u16_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(packed) : "v"(a), "v"(b));
    if (K == 3) asm("v_min_u16_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(packed) : "v"(a), "v"(b));
    return packed;
}

// The quad is written in three phases so that the eight table reads of its four pixels are in flight while the
// table-free hue arithmetic runs (the scheduler keeps this source order).  LS = log2(bytes per table entry across
// the replicas).
struct PxFp {
    float B, G, R, V, diff, sdiv, hdiv;
};

template <int K>
__device__ __forceinline__ float channel_fp(const u32 (&d)[3], int c, u32 bias)
{
    const int i = 3 * K + c;
    const u32 w = d[i >> 2];
    return (i & 3) == 0 ? unpack_biased<0>(w, bias) : (i & 3) == 1 ? unpack_biased<1>(w, bias)
         : (i & 3) == 2 ? unpack_biased<2>(w, bias) : unpack_biased<3>(w, bias);
}

// phase 1: unpack, V, diff, table addresses, table reads issued
template <int LS, int K, bool S1, bool REL = false>
__device__ __forceinline__ void pixel_fp_front(const u32 (&d)[3], const FpLane& fl, PxFp& x)
{
    x.B = channel_fp<K>(d, 0, fl.bias);
    x.G = channel_fp<K>(d, 1, fl.bias);
    x.R = channel_fp<K>(d, 2, fl.bias);
    x.V = max3_f32(x.B, x.G, x.R);                     // 2^23 + v
    x.diff = x.V - min3_f32(x.B, x.G, x.R);            // exact, 0..255
    // S1: ONE copy of the sdiv table at a 1 KiB-aligned LDS address whose number (address >> 10) sits in byte 1 of the
    // bias, i.e. in bits 8..15 of every channel float: a 16-bit shift of V's bits by 2 IS the table address, no addition.
    // (The copies only spread bank conflicts; the LDS pipe has headroom, the VALU and the power budget do not.)
    if constexpr (REL) {
        // The fused passes (any LDS layout, 2^LS / 4 replicas of the hdiv table): both addresses RELATIVE to their table, whose
        // base -- a link-time constant -- rides in the offset field of the ds_read.  sdiv: ONE copy, 4 v = a 16-bit shift of
        // V's bits.  hdiv: floats in [2^(23-k), 2^(24-k)) step by 2^-k, so the low bits of diff + (2^(23-k) + replica / 2^k)
        // are 2^k diff + replica and a 16-bit shift by 2 makes them the byte offset of this lane's replica of entry diff.
        // One instruction for the first, two for the second, where the absolute forms took two and three.
        const u32 r_s = lshl16<2>(__float_as_uint(x.V));
        const u32 r_h = lshl16<2>(__float_as_uint(x.diff + fl.bias_rel));
        x.sdiv = *(lds_cf32_t)((lds_cc_t)fl.obj_s + r_s);
        x.hdiv = *(lds_cf32_t)((lds_cc_t)fl.obj_h + r_h);
        return;
    }
    const u32 a_s = S1 ? lshl16<2>(__float_as_uint(x.V)) : lshl16_add<LS>(__float_as_uint(x.V), fl.off_s);
    u32 a_h;
    if constexpr (LS == 4) {
        // floats in [2^21, 2^22) step by 1/4: the low bits of diff + (2^21 + off_h / 16) are 4 diff + off_h / 4, and a
        // 16-bit shift by 2 turns them into 16 diff + off_h (< 2^16: the HSV-only pass has 24 KiB of LDS)
        a_h = lshl16<2>(__float_as_uint(x.diff + fl.bias_h));
    } else {
        a_h = lshl16_add<LS>(__float_as_uint(x.diff + 8388608.0f), fl.off_h);
    }
    x.sdiv = lds_f32_in(fl.obj_s, a_s);
    x.hdiv = lds_f32_in(fl.obj_h, a_h);
}

// phase 2: hraw = v == r ? g - b : v == g ? b - r + 2 diff : r - g + 4 diff (the biases cancel).
// With p = g - b, q = b - r and the 0/1 floats nr = (v != r), ng = (v != g):
//   hraw = p + nr * ((q - p + 2 diff) + ng * (2 diff - p - 2 q))
// (sector r: p; sector g: p + q - p + 2 diff = b - r + 2 diff; sector b: that + 2 diff - p - 2 q = r - g + 4 diff).
// Nine full-rate instructions: 2 differences, w = 2 diff - p, two more sums, two clamps, two fused multiply-adds.
__device__ __forceinline__ float pixel_fp_hraw(const PxFp& x)
{
    const float p = x.G - x.B, q = x.B - x.R;
    const float w = __builtin_fmaf(x.diff, 2.0f, -p);                     // 2 diff - p
    const float a = w + q;                                                 // c_g - c_r
    const float b = __builtin_fmaf(q, -2.0f, w);                           // c_b - c_g
    const float nr = sub_clamp(x.V, x.R), ng = sub_clamp(x.V, x.G);       // 0 where the channel IS the maximum
    return __builtin_fmaf(nr, __builtin_fmaf(ng, b, a), p);
}

// phase 3: the two table products; hue into byte K of hq, S in the low byte of the return value
template <int K>
__device__ __forceinline__ u32 pixel_fp_back(const PxFp& x, float hraw, u32& hq)
{
    const float t = __builtin_fmaf(hraw, x.hdiv, 0.0001220703125f);       // exact: a multiple of 2^-13 below 2^8
    const u32 hb = __float_as_uint(t + 12582912.0f);                       // low half = floor(. + .5), two's complement
    hq = min_lo16_to_byte<K>(hb, hb + 180u, hq);
    return __float_as_uint(__builtin_fmaf(x.diff, x.sdiv, 8388608.0f));    // low byte = S
}

#ifndef PSD_HPAIR
#define PSD_HPAIR 0     // the hue wrap on pixel PAIRS (pixel_fp_back_quad): measured, see DESIGN.md 4.1
#endif
// The hue wrap min(h, h + 180) on two pixels per instruction: the float add that rounds the second pixel's hue writes its low half
// (the 16-bit two's complement h) into the UPPER half of the first pixel's register (SDWA dst_sel:WORD_1), one v_pk_add_u16 and one
// v_pk_min_u16 wrap both, and one v_perm_b32 per quad takes the four low bytes: 13 instructions per quad where four integer adds
// and four SDWA minima made it 16.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 hue_pair_wrapped(float t0, float t1, float c)
{
    u32 hb = __float_as_uint(t0 + c);
    asm("v_add_f32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(hb) : "v"(t1), "v"(c));
    const u16x2 h = __builtin_bit_cast(u16x2, hb), k = {180, 180};
    return __builtin_bit_cast(u32, __builtin_elementwise_min(h, (u16x2)(h + k)));
}
__device__ __forceinline__ void pixel_fp_back_quad(const PxFp& x0, const PxFp& x1, const PxFp& x2, const PxFp& x3, float r0, float r1, float r2,
                                                   float r3, u32& hq, u32& s0, u32& s1, u32& s2, u32& s3)
{
    const float t0 = __builtin_fmaf(r0, x0.hdiv, 0.0001220703125f), t1 = __builtin_fmaf(r1, x1.hdiv, 0.0001220703125f);
    const float t2 = __builtin_fmaf(r2, x2.hdiv, 0.0001220703125f), t3 = __builtin_fmaf(r3, x3.hdiv, 0.0001220703125f);
    float c = 12582912.0f;
    asm("" : "+v"(c));                // (a VGPR for the additions: the SDWA form takes no literal; not volatile -- hoisted out of the walk)
    const u32 w01 = hue_pair_wrapped(t0, t1, c), w23 = hue_pair_wrapped(t2, t3, c);
    hq = __builtin_amdgcn_perm(w23, w01, 0x06040200u);
    s0 = __float_as_uint(__builtin_fmaf(x0.diff, x0.sdiv, 8388608.0f));
    s1 = __float_as_uint(__builtin_fmaf(x1.diff, x1.sdiv, 8388608.0f));
    s2 = __float_as_uint(__builtin_fmaf(x2.diff, x2.sdiv, 8388608.0f));
    s3 = __float_as_uint(__builtin_fmaf(x3.diff, x3.sdiv, 8388608.0f));
}

// luma of the fused pass from the hue differences: the coefficients of OpenCV's fixed-point BGR -> Y sum to 2^14, so
//   (1868 b + 9617 g + 4899 r + 8192) >> 14 = g + floor((-6767 p - 4899 q) / 16384 + 1/2),  p = g - b, q = b - r
// and in float32 (every product and sum below is exact: multiples of 2^-15 under 2^8; the 2^-15 pushes the exact ties up
// like the integer + 8192, nothing else is nearer than 2^-14 to a tie; g + u > 0 because it IS the weighted sum / 2^14)
//   u = p * (-6767 / 16384) + (q * (-4899 / 16384) + 2^-15),  2^23 + y = RNE((2^23 + g) + u)
// three full-rate instructions per pixel on values the hue already has, against three 24-bit multiplies and a shift.
__device__ __forceinline__ u32 pixel_fp_luma_bits(const PxFp& x)
{
    const float p = x.G - x.B, q = x.B - x.R;
    const float u = __builtin_fmaf(p, -6767.0f / 16384.0f, __builtin_fmaf(q, -4899.0f / 16384.0f, 0.000030517578125f));
    return __float_as_uint(x.G + u);   // low byte = y
}

typedef __attribute__((address_space(3))) u32* lds_u32_t;

// ---- the fused HSV + luma quad (integer formulation), written in phases -------------------------------------------------
// LDS operations of a wave complete in order, and the histogram increment is an (unreturned) LDS atomic: issued per pixel
// in front of that pixel's two table reads -- what pixel<true, true> does -- every pixel waits for its own atomic, slowed by
// bank conflicts, before its table values arrive, 16 times per group with four waves per SIMD to hide it (round 1: 47 % of
// the wave cycles parked in s_waitcnt, VALU 70 % busy at the full 2.4 GHz).  Here a quad first issues its eight table reads,
// computes the four luma values and the hue candidates while they are in flight, consumes them, and only then issues its
// four histogram atomics, which drain under the next quad's arithmetic.
template <int K>
__device__ __forceinline__ void extract_bgr(const u32 (&d)[3], u32& b, u32& g, u32& r)
{
    constexpr int ib = 3 * K, ig = 3 * K + 1, ir = 3 * K + 2;
    b = (d[ib >> 2] >> ((ib & 3) * 8)) & 0xffu;
    g = (d[ig >> 2] >> ((ig & 3) * 8)) & 0xffu;
    r = (d[ir >> 2] >> ((ir & 3) * 8)) & 0xffu;
}

template <int LC, int AC>
__device__ __forceinline__ void quad_fused(const u32 (&d)[3], u32& hq_out, u32& sq_out, u32& vq_out, const u32* __restrict__ lut_s,
                                           const u32* __restrict__ lut_h, u32* __restrict__ hist, u32 inc)
{
    u32 b[4], g[4], r[4], v[4], diff[4], sdiv16[4], y[4];
    int hdiv16[4], hraw[4];
    extract_bgr<0>(d, b[0], g[0], r[0]);
    extract_bgr<1>(d, b[1], g[1], r[1]);
    extract_bgr<2>(d, b[2], g[2], r[2]);
    extract_bgr<3>(d, b[3], g[3], r[3]);
    // phase A: the eight table reads
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = max(max(b[k], g[k]), r[k]);
        diff[k] = v[k] - min(min(b[k], g[k]), r[k]);
        sdiv16[k] = lut_s[v[k] * LC];
        hdiv16[k] = (int)lut_h[diff[k] * LC];
    }
    // (scheduling fences: left alone, hipcc sinks the reads below phase B and waits for them right after issuing them)
    __builtin_amdgcn_sched_barrier(0);
    // phase B: everything that needs no table value
#pragma unroll
    for (int k = 0; k < 4; k++) {
        y[k] = mad_u24_vsv(r[k], kR2Y, mad_u24_vsv(g[k], kG2Y, mad_u24_vvs(b[k], vgpr_b2y(), 8192u))) >> 14;
        const int dd = (int)diff[k];
        const int c_r = (int)g[k] - (int)b[k];
        const int c_g = (int)b[k] - (int)r[k] + 2 * dd;
        const int c_b = (int)r[k] - (int)g[k] + 4 * dd;
        hraw[k] = sel_eq(v[k], r[k], c_r, sel_eq(v[k], g[k], c_g, c_b));
    }
    __builtin_amdgcn_sched_barrier(0);
    // phase C: the two products per pixel, packing
    u32 hq = 0, s16[4];
#pragma unroll
    for (int k = 0; k < 4; k++) s16[k] = mad_u24(diff[k], sdiv16[k], 2048u << 4);
    {
        const int h0 = mad_i24(hraw[0], hdiv16[0], 2048 << 4), h1 = mad_i24(hraw[1], hdiv16[1], 2048 << 4);
        const int h2 = mad_i24(hraw[2], hdiv16[2], 2048 << 4), h3 = mad_i24(hraw[3], hdiv16[3], 2048 << 4);
        hq = min_hi16_to_byte<0>((u32)h0, (u32)h0 + (180u << 16), hq);
        hq = min_hi16_to_byte<1>((u32)h1, (u32)h1 + (180u << 16), hq);
        hq = min_hi16_to_byte<2>((u32)h2, (u32)h2 + (180u << 16), hq);
        hq = min_hi16_to_byte<3>((u32)h3, (u32)h3 + (180u << 16), hq);
    }
    const u32 s01 = perm_b32(s16[1], s16[0], 0x0c0c0602u), v01 = perm_b32(v[1], v[0], 0x0c0c0400u);
    const u32 s23 = perm_b32(s16[3], s16[2], 0x0c0c0602u), v23 = perm_b32(v[3], v[2], 0x0c0c0400u);
    hq_out = hq;
    sq_out = perm_b32(s23, s01, 0x05040100u);
    vq_out = perm_b32(v23, v01, 0x05040100u);
    // phase D: the histogram increments, behind the reads in the LDS queue
#pragma unroll
    for (int k = 0; k < 4; k++)
        lds_add_hidden(&hist[y[k] * AC], inc);
}

// 8-bit BGR -> HSV exactly as OpenCV's RGB2HSV_b (hsv_shift = 12, hue range 180), plus luma
// histogram and byte sum.  lut_s/lut_h/hist are already offset by the lane's replica index.
template <bool HSV, bool LUMA, int LC = COPIES, int AC = COPIES, bool FP = false, bool S1 = false, bool VM = false, bool REL = false>
__device__ __forceinline__ void convert_group(Group& g, Hsv16& o, const u32* __restrict__ lut_s,
                                              const u32* __restrict__ lut_h, u32* __restrict__ hist,
                                              u32& byte_sum, u32 inc = 1u)
{
    if constexpr (FP) {
        static_assert(HSV && !(LUMA && S1), "");
        // VM (with LUMA): the histogram slots count V instead of the luma
        constexpr int LA = AC == 4 ? 4 : AC == 8 ? 5 : AC == 16 ? 6 : AC == 32 ? 7 : AC == 2 ? 3 : AC == 1 ? 2 : -1;   // log2(bytes per bin)
        static_assert(!LUMA || LA > 0, "");
        const u32 off_a = LUMA ? (u32)(uintptr_t)hist : 0u;
        if constexpr (LUMA && !VM) {
            u32 local = 0;
#pragma unroll
            for (int i = 0; i < 12; i++) local = __builtin_amdgcn_sad_u8(g.w[i], 0u, local);
            byte_sum += local * inc;
        }
        constexpr int LS = LC == 4 ? 4 : LC == 8 ? 5 : LC == 16 ? 6 : LC == 32 ? 7 : LC == 2 ? 3 : -1;
        static_assert(LS > 0, "");
        FpLane fl;
        fl.off_s = (u32)(uintptr_t)lut_s;   // low half of a flat LDS address = the LDS byte address
        fl.bias = S1 ? (0x4B000000u | ((fl.off_s >> 10) << 8)) : 0x4B000000u;
        fl.off_h = (u32)(uintptr_t)lut_h;
        fl.obj_s = lut_s;
        fl.obj_h = lut_h;
        fl.bias_h = 2097152.0f + (float)fl.off_h * 0.0625f;   // exact: off_h is a multiple of 4 below 2^16
        // REL: lut_s / lut_h are the TABLES (no replica offset); this lane reads replica threadIdx.x % LC of the hdiv table
        static_assert(!REL || !S1, "");
        fl.bias_rel = (float)(8388608 / LC) + (float)(threadIdx.x & (LC - 1)) * (1.0f / LC);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (q > 0)
                asm volatile("" : "+v"(g.w[3 * q]), "+v"(g.w[3 * q + 1]), "+v"(g.w[3 * q + 2]) : "v"(o.h[q - 1]), "v"(o.s[q - 1]), "v"(o.v[q - 1]));
            const u32 d[3] = {g.w[3 * q], g.w[3 * q + 1], g.w[3 * q + 2]};
#if PSD_ABLATE & 8
            o.h[q] = d[0]; o.s[q] = d[1]; o.v[q] = d[2];
            continue;
#endif
            PxFp x0, x1, x2, x3;
            pixel_fp_front<LS, 0, S1, REL>(d, fl, x0);
            pixel_fp_front<LS, 1, S1, REL>(d, fl, x1);
            pixel_fp_front<LS, 2, S1, REL>(d, fl, x2);
            pixel_fp_front<LS, 3, S1, REL>(d, fl, x3);
            if constexpr (LUMA) __builtin_amdgcn_sched_barrier(0);   // the eight table reads stay in front (see quad_fused)
            const float r0 = pixel_fp_hraw(x0), r1 = pixel_fp_hraw(x1), r2 = pixel_fp_hraw(x2), r3 = pixel_fp_hraw(x3);
            u32 ya[4];
            if constexpr (LUMA) {
                // V mode (the edge term's front end, psd_edge_kernels.hip): the histogram counts V = max(B, G, R), whose bits'
                // low byte is v, instead of the luma
                if constexpr (REL) {
                    // the bits are 0x4B000000 + y exactly (a float in [2^23, 2^23 + 256)), so (bits << LA) + (base - (0x4B000000 <<
                    // LA)) is the address modulo 2^32: ONE v_lshl_add_u32 where the 16-bit shift + addition took two
                    const u32 off_b = off_a - (0x4B000000u << LA);
                    ya[0] = ((VM ? __float_as_uint(x0.V) : pixel_fp_luma_bits(x0)) << LA) + off_b;
                    ya[1] = ((VM ? __float_as_uint(x1.V) : pixel_fp_luma_bits(x1)) << LA) + off_b;
                    ya[2] = ((VM ? __float_as_uint(x2.V) : pixel_fp_luma_bits(x2)) << LA) + off_b;
                    ya[3] = ((VM ? __float_as_uint(x3.V) : pixel_fp_luma_bits(x3)) << LA) + off_b;
                } else {
                ya[0] = lshl16_add<LA>(VM ? __float_as_uint(x0.V) : pixel_fp_luma_bits(x0), off_a);
                ya[1] = lshl16_add<LA>(VM ? __float_as_uint(x1.V) : pixel_fp_luma_bits(x1), off_a);
                ya[2] = lshl16_add<LA>(VM ? __float_as_uint(x2.V) : pixel_fp_luma_bits(x2), off_a);
                ya[3] = lshl16_add<LA>(VM ? __float_as_uint(x3.V) : pixel_fp_luma_bits(x3), off_a);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            u32 hq = 0;
#if PSD_HPAIR
            u32 s0, s1, s2, s3;
            pixel_fp_back_quad(x0, x1, x2, x3, r0, r1, r2, r3, hq, s0, s1, s2, s3);
#else
            const u32 s0 = pixel_fp_back<0>(x0, r0, hq), s1 = pixel_fp_back<1>(x1, r1, hq);
            const u32 s2 = pixel_fp_back<2>(x2, r2, hq), s3 = pixel_fp_back<3>(x3, r3, hq);
#endif
            const u32 s01 = perm_b32(s1, s0, 0x0c0c0400u), v01 = perm_b32(__float_as_uint(x1.V), __float_as_uint(x0.V), 0x0c0c0400u);
            const u32 s23 = perm_b32(s3, s2, 0x0c0c0400u), v23 = perm_b32(__float_as_uint(x3.V), __float_as_uint(x2.V), 0x0c0c0400u);
            o.h[q] = hq;
            o.s[q] = perm_b32(s23, s01, 0x05040100u);
            o.v[q] = perm_b32(v23, v01, 0x05040100u);
            if constexpr (LUMA) {
                // the histogram increments go last: LDS operations complete in order (see quad_fused)
#if PSD_ABLATE & 16
                asm volatile("" :: "v"(ya[0]), "v"(ya[1]), "v"(ya[2]), "v"(ya[3]));
                continue;
#endif
#pragma unroll
                for (int k = 0; k < 4; k++)
                    lds_add_hidden_at(ya[k], inc);
            }
        }
        return;
    }
    if (LUMA) {
        u32 local = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) local = __builtin_amdgcn_sad_u8(g.w[i], 0u, local);
        byte_sum += local * inc;   // 16 px * 3 * 255 < 2^24: a 24-bit multiply
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u32 hq, sq, vq;
        if (!HSV) { hq = 0; sq = 0; vq = 0; }
        // The 4 pixels of this quad live in 3 dwords.  Tie them to the previous quad's results so
        // the 16 pixels are processed quad by quad: fully interleaved (what the scheduler does when
        // left alone) the live temporaries exceed the 128-VGPR budget of a 16-wave workgroup and
        // spill.  Latency is hidden by the other 3 waves of the SIMD, not by ILP across quads.
        // (in place on g: the group's input registers are dead after this call, a copy would cost 3 v_mov)
        if (HSV && q > 0)
            asm volatile("" : "+v"(g.w[3 * q]), "+v"(g.w[3 * q + 1]), "+v"(g.w[3 * q + 2]) : "v"(o.h[q - 1]), "v"(o.s[q - 1]), "v"(o.v[q - 1]));
        const u32 d[3] = {g.w[3 * q], g.w[3 * q + 1], g.w[3 * q + 2]};
#if PSD_FUSED_PHASED && PSD_PACK3 && !(PSD_ABLATE & 8)
        if constexpr (HSV && LUMA) {
            quad_fused<LC, AC>(d, o.h[q], o.s[q], o.v[q], lut_s, lut_h, hist, inc);
            continue;
        }
#endif
#if PSD_PACK3 && !(PSD_ABLATE & 8)
        if (HSV) {
            u32 s0, s1, s2, s3, v0, v1, v2, v3;
            pixel<HSV, LUMA, LC, AC, 0>(d, hq, s0, v0, lut_s, lut_h, hist, inc);
            pixel<HSV, LUMA, LC, AC, 1>(d, hq, s1, v1, lut_s, lut_h, hist, inc);
            const u32 s01 = perm_b32(s1, s0, 0x0c0c0602u), v01 = perm_b32(v1, v0, 0x0c0c0400u);
            pixel<HSV, LUMA, LC, AC, 2>(d, hq, s2, v2, lut_s, lut_h, hist, inc);
            pixel<HSV, LUMA, LC, AC, 3>(d, hq, s3, v3, lut_s, lut_h, hist, inc);
            const u32 s23 = perm_b32(s3, s2, 0x0c0c0602u), v23 = perm_b32(v3, v2, 0x0c0c0400u);
            o.h[q] = hq;
            o.s[q] = perm_b32(s23, s01, 0x05040100u);
            o.v[q] = perm_b32(v23, v01, 0x05040100u);
            continue;
        }
#endif
        pixel<HSV, LUMA, LC, AC, 0>(d, hq, sq, vq, lut_s, lut_h, hist, inc);
        pixel<HSV, LUMA, LC, AC, 1>(d, hq, sq, vq, lut_s, lut_h, hist, inc);
        pixel<HSV, LUMA, LC, AC, 2>(d, hq, sq, vq, lut_s, lut_h, hist, inc);
        pixel<HSV, LUMA, LC, AC, 3>(d, hq, sq, vq, lut_s, lut_h, hist, inc);
        if (HSV) { o.h[q] = hq; o.s[q] = sq; o.v[q] = vq; }
    }
}

__device__ __forceinline__ void sad_group(const Hsv16& a, const Hsv16& b, u32& sh, u32& ss, u32& sv)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        sh = __builtin_amdgcn_sad_u8(a.h[q], b.h[q], sh);
        ss = __builtin_amdgcn_sad_u8(a.s[q], b.s[q], ss);
        sv = __builtin_amdgcn_sad_u8(a.v[q], b.v[q], sv);
    }
}

// ---- workgroup flush of one frame's accumulators -------------------------------------------

// acc: [NACC][COPIES] u32 in LDS.  Sums the replicas, zeroes them, and adds the totals to the
// frame's record in global memory.  Called by all WG threads after a __syncthreads().
template <bool LUMA, int WG>
__device__ __forceinline__ void flush_frame(u32* acc, psd_frame_scores* rec, int tid)
{
    static_assert(WG % 4 == 0, "");
    const int part = tid & 3;
    const u32x4 zero = {0, 0, 0, 0};
    if (LUMA) {
        for (int a = tid >> 2; a < 256; a += WG / 4) {
            u32x4* p = reinterpret_cast<u32x4*>(acc + a * COPIES + part * 8);
            u32x4 v0 = p[0], v1 = p[1];
            p[0] = zero; p[1] = zero;
            u32 s = v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w;
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (part == 0 && s) atomicAdd(&rec->hist[a], s);
        }
    }
    if (tid < 16) {
        const int a = 256 + (tid >> 2);
        u32x4* p = reinterpret_cast<u32x4*>(acc + a * COPIES + part * 8);
        u32x4 v0 = p[0], v1 = p[1];
        p[0] = zero; p[1] = zero;
        u32 s = v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w;
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (part == 0 && s) {
            unsigned long long* dst =
                a == ACC_SAD_H ? (unsigned long long*)&rec->sad_h
              : a == ACC_SAD_S ? (unsigned long long*)&rec->sad_s
              : a == ACC_SAD_V ? (unsigned long long*)&rec->sad_v
                               : (unsigned long long*)&rec->byte_sum;
            atomicAdd(dst, (unsigned long long)s);
        }
    }
}

// ---- the kernel ----------------------------------------------------------------------------

// grid.x = n_tiles * n_chunks.  Block (tile, chunk) scores frames [chunk*T, chunk*T+T) of its
// spatial tile.  G = 16-pixel groups per lane per frame.
template <bool HSV, bool LUMA, int G, bool FAST, int WG>
__global__ __launch_bounds__(WG) void score_frames_kernel(const ScoreParams p)
{
    __shared__ __attribute__((aligned(16))) u32 lut_s[HSV ? 256 * COPIES : 4];
    __shared__ __attribute__((aligned(16))) u32 lut_h[HSV ? 256 * COPIES : 4];
    __shared__ __attribute__((aligned(16))) u32 acc[2][NACC * COPIES];

    const int tid = threadIdx.x;
    const int l32 = tid & 31;
    const int tile = blockIdx.x % p.n_tiles;
    const int chunk = blockIdx.x / p.n_tiles;
    constexpr bool FPK = HSV && !LUMA && PSD_HSV_FP32;   // HSV-only: the fp32 formulation and its float tables

    // Fill the replicated LUTs and clear the accumulators.
    if (HSV) {
        const uint32_t* tab = FPK ? p.lutf : p.lut;
        for (int i = tid; i < 256 * COPIES; i += WG) {
            lut_s[i] = tab[i / COPIES];
            lut_h[i] = tab[256 + i / COPIES];
        }
    }
    for (int i = tid; i < 2 * NACC * COPIES; i += WG) (&acc[0][0])[i] = 0;
    __syncthreads();

    const int t0 = chunk * p.frames_per_chunk;
    const int t1 = min(p.n, t0 + p.frames_per_chunk);
    const int g0 = p.group_begin + tile * p.groups_per_tile;
    const int g1 = min(p.group_end, g0 + p.groups_per_tile);

    int grp[G];
    bool live[G];
#pragma unroll
    for (int k = 0; k < G; k++) {
        grp[k] = g0 + k * WG + tid;
        live[k] = grp[k] < g1;
    }

    const u32* my_lut_s = lut_s + l32;
    const u32* my_lut_h = lut_h + l32;

    Hsv16 prev[G];
    bool have_prev = false;
    if (HSV) {
        const uint8_t* halo = (t0 > 0) ? p.frames + (size_t)(t0 - 1) * p.frame_stride : p.prev;
        if (halo != nullptr) {
            have_prev = true;
#pragma unroll
            for (int k = 0; k < G; k++) {
                if (live[k]) {
                    Group g;
                    load_group<FAST>(p, halo, grp[k], g);
                    u32 dummy = 0;
                    convert_group<true, false, COPIES, COPIES, FPK>(g, prev[k], my_lut_s, my_lut_h, nullptr, dummy);
                }
            }
        }
    }

    Group cur[G];
    if (t0 < t1) {
        const uint8_t* f = p.frames + (size_t)t0 * p.frame_stride;
#pragma unroll
        for (int k = 0; k < G; k++)
            if (live[k]) load_group<FAST>(p, f, grp[k], cur[k]);
    }

    // Pixels past the end of the frame in the last group read as black: Y = 0.
    const int last_group = (int)((p.npix + 15) / 16) - 1;
    const int pad_px = FAST ? 0 : (last_group + 1) * 16 - (int)p.npix;

    for (int t = t0; t < t1; t++) {
        const int buf = (t - t0) & 1;
        u32* my_hist = &acc[buf][0] + l32;
        // Prefetch the next frame of the chunk while this one is being scored.
        Group nxt[G];
#if PSD_ABLATE & 4
#pragma unroll
        for (int k = 0; k < G; k++) {
            nxt[k] = cur[k];
#pragma unroll
            for (int i = 0; i < 12; i++) asm volatile("" : "+v"(nxt[k].w[i]));
        }
#else
        if (t + 1 < t1) {
            const uint8_t* f = p.frames + (size_t)(t + 1) * p.frame_stride;
#pragma unroll
            for (int k = 0; k < G; k++)
                if (live[k]) load_group<FAST>(p, f, grp[k], nxt[k]);
        }
#endif
        u32 sh = 0, ss = 0, sv = 0, bs = 0;
        // the first frame of a clip packed into the batch has no predecessor (p.seg, psd_score_segments_device)
        const bool chain = have_prev && !(p.seg != nullptr && p.seg[t] != 0);
#pragma unroll
        for (int k = 0; k < G; k++) {
            if (live[k]) {
                Hsv16 c;
                convert_group<HSV, LUMA, COPIES, COPIES, FPK>(cur[k], c, my_lut_s, my_lut_h, my_hist, bs);
                if (HSV) {
                    if (chain) sad_group(c, prev[k], sh, ss, sv);
                    prev[k] = c;
                }
                if (!FAST && LUMA && pad_px && grp[k] == last_group)
                    __hip_atomic_fetch_add(&my_hist[0], (u32)(0 - pad_px), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (HSV) {
            __hip_atomic_fetch_add(&my_hist[ACC_SAD_H * COPIES], sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&my_hist[ACC_SAD_S * COPIES], ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&my_hist[ACC_SAD_V * COPIES], sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (LUMA)
            __hip_atomic_fetch_add(&my_hist[ACC_BYTES * COPIES], bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        have_prev = true;
#if !(PSD_ABLATE & 2)
        if (LUMA) lds_hidden_fence();
        __syncthreads();
        // Frame t is complete in acc[buf]; frame t+1 accumulates into the other buffer, so no
        // second barrier is needed (the barrier of frame t+1 orders this flush before reuse).
        flush_frame<LUMA, WG>(&acc[buf][0], p.out + t, tid);
#endif
        if (t + 1 < t1) {
#pragma unroll
            for (int k = 0; k < G; k++) cur[k] = nxt[k];
        }
    }
}

// ---- fast path: LDS-DMA staged loads ----------------------------------------------------------
//
// Same tiling and time walk as score_frames_kernel, but the frame bytes travel
// HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip, issued one frame ahead) and are
// then picked up by each lane as its 48 contiguous bytes with three ds_read_b128.
//   * global side: lane i of a wave moves bytes [16 i, 16 i + 16) of each 1 KiB piece, so every
//     128-byte line is touched by exactly one instruction (the direct kernel's 48-byte lane stride
//     makes three instructions touch every line and tops out near 4.2 TB/s);
//   * LDS side: the wave-private 3 KiB slot is a linear image of its 64 groups; lane l reads
//     dwords [12 l, 12 l + 12) -- a 12-dword stride is conflict-free for ds_read_b128.
// A wave only ever reads the slot it filled itself, so the hand-off needs no barrier: the wave's
// own s_waitcnt vmcnt(0) covers the DMA, lgkmcnt(0) covers the reads before the slot is refilled.
// Accumulators: HSV-only keeps 2 x F frames of {sad_h,sad_s,sad_v} so the workgroup barrier and
// flush happen once per F frames; with the luma histogram it is one frame per barrier (F = 1).

constexpr int LCD_MAX = 16;  // LUT replicas in the staged kernel (2-way conflicts at worst, 32 KiB for both tables)
constexpr int ACD_MAX = 16;  // accumulator replicas
// The fused HSV+luma variant has to fit two tables, the histogram slots and G=2 staging in 160 KiB:
#ifndef PSD_FUSED_LC
#define PSD_FUSED_LC 16
#endif
#ifndef PSD_FUSED_AC
// histogram replicas of the two fused 16-wave passes: the lanes of a wave that share a replica serialise when they count the
// same bin, which uniform noise never shows and every real frame does (equal neighbours).  Round 4, A/B on one box
// (profiles/r04_w_*): 4 / 8 / 16 replicas -- edge term's front end + pipeline on frames with objects 0.377 / 0.395 / 0.404 of
// its roofline, constant frames 0.36 / 0.47 / 0.50; all-detectors pass on constant frames 0.31 / 0.46 / 0.51, shot-like 0.432 /
// 0.446 / 0.451, noise unchanged.  32 do not fit beside the 96 KiB of staging slots (2 x 260 x 32 x 4 bytes).
#define PSD_FUSED_AC 16
#endif
#ifndef PSD_FUSED_F
#define PSD_FUSED_F 1
#endif
// HSV-only variant: small workgroups (kHsvWG threads) with fewer table replicas, so that four of them share a CU
// and their barriers / flushes interleave (A/B on 2048 x 1080p: 3.04 ms vs 3.18 ms for one 1024-thread workgroup
// with 16 replicas; 512 threads with 8 replicas 3.13 ms; 256 threads with 2 / 8 replicas 3.08 / 3.09 ms; 128 threads 3.4+ ms)
#ifndef PSD_HSV_LC
#define PSD_HSV_LC 4
#endif
#ifndef PSD_FUSED_WG
#define PSD_FUSED_WG 1024   // threads per workgroup of the fused HSV + luma pass (PSD_FUSED_SMALL overrides with kHsvWG)
#endif
constexpr int kFusedWG = PSD_FUSED_WG;
#ifndef PSD_HSV_WAVES
#define PSD_HSV_WAVES 6      // waves per SIMD hipcc has to leave room for in the HSV pass (77 instead of 83 VGPRs, no spills: six workgroups
                            // per CU again after the swapped register sets; A/B +0.3 ... +1.1 %)
#endif
#ifndef PSD_FILL_LAST_ROUND
#define PSD_FILL_LAST_ROUND 1   // launch_range: pick the number of time chunks so that the last round of workgroups is full
#endif
#ifndef PSD_FUSED_REL
#define PSD_FUSED_REL 1    // fused passes: one sdiv table + table-relative addresses with the base in the ds_read offset field
#endif
#ifndef PSD_FUSED_SMALL
#define PSD_FUSED_SMALL 0   // 1: the fused HSV+luma variant also runs on kHsvWG-thread workgroups (measured: 4.32-4.43 ms
                            // vs 4.36 ms on 2048 x 1080p with 4 table / 8 histogram replicas -- no gain, off)
#endif
// luma-only variant (histogram + byte sum): replicas of the accumulators and frames per barrier
#ifndef PSD_LUMA_AC
#define PSD_LUMA_AC 16
#endif
#ifndef PSD_LUMA_F
#define PSD_LUMA_F 1
#endif

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int G, int NW, int AUX>
__device__ __forceinline__ void dma_issue(const ScoreParams& p, const uint8_t* frame, int wave_group0, int tile_end, int wave,
                                          int lane, uint8_t* stage)
{
    // wave_group0[k] = first group of this wave's k-th slot; 64 groups = 3072 contiguous bytes.
#pragma unroll
    for (int k = 0; k < G; k++) {
        const long gfirst = (long)wave_group0 + (long)k * NW * 64;
        // Never load past the tile: lanes of the last wave that own no group then keep whatever bytes
        // their staging area holds -- the SAME bytes every frame, so their HSV never changes, their SADs
        // are 0 and the HSV arithmetic needs no per-lane predicate (which would turn `prev = cur` into
        // 12 real moves per group).  Their luma side effects are predicated instead.
        const long limit = (long)tile_end * 48;
        uint8_t* slot = stage + (size_t)(k * NW + wave) * 3072;
        // One per-lane 32-bit offset (a frame is far below 4 GiB) on top of the uniform frame pointer, the three 1 KiB pieces
        // told apart by the instruction's immediate offset, which moves the global and the LDS address alike: the frame
        // pointer advances in SGPRs and no address arithmetic is left on the VALU (it was a 64-bit add per piece and frame).
        const u32 voff = (u32)(gfirst * 48) + (u32)lane * 16u;
        typedef const __attribute__((address_space(1))) uint8_t* gbl_u8_t;
        gbl_u8_t fb = (gbl_u8_t)frame;
        asm volatile("" : "+s"(fb));   // (pins the frame pointer to an SGPR pair: global_load_lds v_off, s[base] offset:imm)
        if ((long)voff + 16 <= limit) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(fb + voff), (lds_ptr_t)slot, 16, 0, AUX);
        if ((long)voff + 1024 + 16 <= limit) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(fb + voff), (lds_ptr_t)slot, 16, 1024, AUX);
        if ((long)voff + 2048 + 16 <= limit) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(fb + voff), (lds_ptr_t)slot, 16, 2048, AUX);
    }
}

template <int G, int NW>
__device__ __forceinline__ void stage_read(const uint8_t* stage, int wave, int lane, Group (&out)[G])
{
#pragma unroll
    for (int k = 0; k < G; k++) {
        const u32x4* src = reinterpret_cast<const u32x4*>(stage + (size_t)(k * NW + wave) * 3072 + lane * 48);
        const u32x4 a = src[0], b = src[1], c = src[2];
        out[k].w[0] = a.x; out[k].w[1] = a.y; out[k].w[2] = a.z; out[k].w[3] = a.w;
        out[k].w[4] = b.x; out[k].w[5] = b.y; out[k].w[6] = b.z; out[k].w[7] = b.w;
        out[k].w[8] = c.x; out[k].w[9] = c.y; out[k].w[10] = c.z; out[k].w[11] = c.w;
    }
}

// Sum the AC replicas of one accumulator of one slot, zero them, add to the record.
template <bool LUMA, int AC>
__device__ __forceinline__ void flush_slot(u32* slot_acc, psd_frame_scores* rec, int idx, u32* hist_dst = nullptr)
{
    // idx in [0, 2*NA): accumulator a = idx >> 1, half = idx & 1 (AC/2 replicas each)
    const int a = idx >> 1, half = idx & 1;
    u32 s;
    if (AC == 16) {
        u32x4* q = reinterpret_cast<u32x4*>(slot_acc + a * AC + half * 8);
        const u32x4 v0 = q[0], v1 = q[1];
        const u32x4 zero = {0, 0, 0, 0};
        q[0] = zero; q[1] = zero;
        s = v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w;
    } else if (AC == 8) {
        u32x4* q = reinterpret_cast<u32x4*>(slot_acc + a * AC + half * 4);
        const u32x4 v0 = q[0];
        const u32x4 zero = {0, 0, 0, 0};
        q[0] = zero;
        s = v0.x + v0.y + v0.z + v0.w;
    } else {
        static_assert(AC == 16 || AC == 8 || AC == 4, "");
        u32* q = slot_acc + a * AC + half * 2;
        s = q[0] + q[1];
        q[0] = 0; q[1] = 0;
    }
    s += __shfl_xor(s, 1);
    if (half == 0 && s) {
        if (LUMA && a < 256) {
#if !(PSD_ABLATE & 64)
            atomicAdd(hist_dst ? &hist_dst[a] : &rec->hist[a], s);   // hist_dst: V mode, the edge term's per-frame V histogram
#else
            asm volatile("" :: "v"(s));
#endif
        } else {
            const int e = LUMA ? a - 256 : a;
            unsigned long long* dst = e == 0 ? (unsigned long long*)&rec->sad_h
                                    : e == 1 ? (unsigned long long*)&rec->sad_s
                                    : e == 2 ? (unsigned long long*)&rec->sad_v
                                             : (unsigned long long*)&rec->byte_sum;
#if !(PSD_ABLATE & 128)
            atomicAdd(dst, (unsigned long long)s);
#else
            asm volatile("" :: "v"(s), "v"(dst));
#endif
        }
    }
}

// PSD_PHASE_TIMING (experiments): per-wave shader-clock time of each phase of a step, summed into g_phase and read back
// with psd_debug_phases().  s_memtime shares lgkmcnt with the LDS queue, so the probes sit where the kernel drains it anyway.
#ifndef PSD_PHASE_TIMING
#define PSD_PHASE_TIMING 0
#endif
#if PSD_PHASE_TIMING
__device__ unsigned long long g_phase[8];
#define PT_INIT unsigned long long pt_acc[6] = {0, 0, 0, 0, 0, 0}; const unsigned long long pt_rt0 = wall_clock64(); unsigned long long pt_last = __builtin_readcyclecounter();
#define PT(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_readcyclecounter(); pt_acc[i] += n_ - pt_last; pt_last = n_; }
#define PT_DONE if (lane == 0) { for (int i = 0; i < 6; i++) atomicAdd(&g_phase[i], pt_acc[i]); atomicAdd(&g_phase[6], wall_clock64() - pt_rt0); atomicAdd(&g_phase[7], 1ull); }
#else
#define PT_INIT
#define PT(i)
#define PT_DONE
#endif

// PSD_WG_TIMELINE (experiments): every workgroup of the staged kernel leaves its start / end on the constant 100 MHz clock
// and the hardware id of its first wave in g_timeline[3 * blockIdx.x ..] (buffer set with psd_debug_timeline(); tools/wg_timeline.py
// turns it into a residency curve of the launch: ramp, steady state, tail).
#ifndef PSD_WG_TIMELINE
#define PSD_WG_TIMELINE 0
#endif
#if PSD_WG_TIMELINE
__device__ unsigned long long* g_timeline;
#define TL_BEGIN const unsigned long long tl_t0 = wall_clock64(); const unsigned tl_hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
#define TL_END { const unsigned long long tl_t1 = wall_clock64(); unsigned long long* tl = g_timeline; if (tl && threadIdx.x == 0) { tl[3 * (size_t)blockIdx.x] = tl_t0; tl[3 * (size_t)blockIdx.x + 1] = tl_t1; tl[3 * (size_t)blockIdx.x + 2] = tl_hw; } }
#else
#define TL_BEGIN
#define TL_END
#endif

// VM ("V mode", only with HSV && LUMA): the front end of the edge term rides on the HSV pass -- the histogram slots count
// V = max(B, G, R) instead of the luma and go to p.vhist[t], the V plane (which the HSV conversion has packed already) is
// stored to p.vout[t], no byte sum.  The frames are then read once for ContentDetector with weights.delta_edges > 0
// instead of once by this pass and once by value_plane_hist_dma_kernel.
// SEG: the instance for batches of packed clips (p.seg != nullptr); the common instance carries no clip-start code at all.
template <bool HSV, bool LUMA, int G, int WG, bool VM = false, bool SEG = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu((HSV && LUMA) ? 4 : (HSV && G == 1 && WG == kHsvWG && !VM) ? PSD_HSV_WAVES : 1)))
void score_frames_dma_kernel(const ScoreParams p)
{
    TL_BEGIN
    static_assert(!VM || HSV, "");
    constexpr int NW = WG / 64;
    constexpr int ACD = (HSV && LUMA) ? PSD_FUSED_AC : (LUMA ? PSD_LUMA_AC : ACD_MAX);
    constexpr int LCD = (HSV && LUMA) ? PSD_FUSED_LC : (HSV && WG == kHsvWG ? PSD_HSV_LC : LCD_MAX);
    static_assert(!(HSV && LUMA) || WG == kFusedWG || PSD_FUSED_SMALL, "");
    constexpr int NA = LUMA ? NACC : 4;       // accumulators per frame slot
    constexpr int F = LUMA ? (HSV ? PSD_FUSED_F : PSD_LUMA_F) : 8;   // frames per barrier
    constexpr int SLOTS = 2 * F;
    // nt policy on the frame stream: +2 % for the HSV pass, +5 % for the luma pass, but -3 % for the fused one (A/B)
    constexpr int DMA_AUX = (HSV && LUMA) ? 0 : PSD_DMA_AUX;
    constexpr bool FPK = HSV && (LUMA ? PSD_FUSED_FP32 : PSD_HSV_FP32);   // the fp32 formulation and its float tables
    constexpr bool S1 = FPK && !LUMA && PSD_HSV_SINGLE_S && LCD == 4;   // one copy of the sdiv table, addressed without an addition
    constexpr bool REL = FPK && LUMA && PSD_FUSED_REL;                   // fused passes: table-relative addresses (pixel_fp_front)
    __shared__ __attribute__((aligned(1024))) u32 lut_s[HSV ? ((S1 || REL) ? 256 : 256 * LCD) : 4];
    // (512-byte alignment: hipcc lays LDS objects out by descending alignment, so the two tables come first and their bases
    //  fit the 16-bit offset field of ds_read_b32 -- REL; behind the 96 KiB of staging slots they would not)
    __shared__ __attribute__((aligned(512))) u32 lut_h[HSV ? 256 * LCD : 4];
    __shared__ __attribute__((aligned(16))) u32 acc[SLOTS][NA * ACD];
    __shared__ __attribute__((aligned(16))) uint8_t stage[G * NW * 3072];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: slot addresses and M0 stay on the scalar unit)
    const int l16 = tid & (LCD - 1);
    const int lacc = tid & (ACD - 1);
    const int tile = blockIdx.x % p.n_tiles;
    const int chunk = blockIdx.x / p.n_tiles;

    if (HSV) {
        const uint32_t* tab = FPK ? p.lutf : p.lut;
        for (int i = tid; i < 256 * LCD; i += WG) {
            if (!S1 && !REL) lut_s[i] = tab[i / LCD];
            else if (i < 256) lut_s[i] = tab[i];
            lut_h[i] = tab[256 + i / LCD];
        }
    }
    for (int i = tid; i < SLOTS * NA * ACD; i += WG) (&acc[0][0])[i] = 0;
    __syncthreads();

    const int t0 = chunk * p.frames_per_chunk;
    const int t1 = min(p.n, t0 + p.frames_per_chunk);
    const int g0 = p.group_begin + tile * p.groups_per_tile;
    const int g1 = min(p.group_end, g0 + p.groups_per_tile);
    const int wave_group0 = g0 + wave * 64;

    bool live[G];
#pragma unroll
    for (int k = 0; k < G; k++) live[k] = (wave_group0 + k * NW * 64 + lane) < g1;

    const u32* my_lut_s = (S1 || REL) ? lut_s : lut_s + l16;
    const u32* my_lut_h = REL ? lut_h : lut_h + l16;

    Hsv16 prev[G], other[G];   // the previous frame's H, S, V planes of this lane's pixels / the frame being converted (they swap)
    bool have_prev = false;
    u32 seg_flag = (SEG && t0 < t1) ? p.seg[t0] : 0u;   // clip-start flag of the next frame to be stepped (packed clips)
    Group cur[G];
    if (HSV) {
        const uint8_t* halo = (t0 > 0) ? p.frames + (size_t)(t0 - 1) * p.frame_stride : p.prev;
        if (halo != nullptr) {
            have_prev = true;
            dma_issue<G, NW, DMA_AUX>(p, halo, wave_group0, g1, wave, lane, stage);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage_read<G, NW>(stage, wave, lane, cur);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < G; k++) {
                u32 dummy = 0;
                convert_group<true, false, LCD, ACD, FPK, S1, false, REL>(cur[k], prev[k], my_lut_s, my_lut_h, nullptr, dummy);
            }
        }
    }
    if (t0 < t1) dma_issue<G, NW, DMA_AUX>(p, p.frames + (size_t)t0 * p.frame_stride, wave_group0, g1, wave, lane, stage);

    PT_INIT
    // One step of the walk: frame t against the HSV planes in `prv`, its own left in `nxt`.  The loop below calls it twice per
    // iteration with the two register sets swapped, so "prev = current" is a renaming instead of twelve v_mov_b32 per group
    // and frame (0.75 of the 28 VALU instructions per pixel of the HSV pass).
    auto step = [&](const int t, Hsv16 (&prv)[G], Hsv16 (&nxt)[G]) __attribute__((always_inline)) {
        const int slot = (t - t0) % SLOTS;
        u32* my_acc = &acc[slot][0] + lacc;
        // Frame t has been in flight since the previous step; take it out of the staging slot and
        // immediately refill the slot with frame t+1.
#if !(PSD_ABLATE & 4)
        PT(5)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PT(0)
        // The first frame of a clip packed into the batch has no predecessor (p.seg, psd_score_segments_device).  Its flag was
        // requested one step earlier, behind that step's DMA issue, so the wait above covered it and its latency lay under a
        // whole step (read at the top of its own step it cost every wave a memory round trip per frame: 2 % on packed clips).
        // It goes into an SGPR HERE, in front of the next frame's DMA issue: hipcc waits for a loaded register where it is first
        // read and knows nothing of the wait above, so read behind the issue (round 4) its own s_waitcnt vmcnt(0) drained the
        // prefetch the wave had just issued -- in the 4-wave pass in every step of every launch, packed clips or not (the
        // register was read whether or not it had been loaded), so the prefetch overlapped nothing inside a wave; the 16-wave
        // passes turned the loaded flag into an SGPR at once, a bare memory round trip per step on packed clips.  Round 5: the
        // flag's code only exists in the SEG instances (HSV pass +1.3 ... +2.1 % on plain batches, +2.6 ... +4 % on packed
        // clips; the fused pass on packed clips +5 %, profiles/r05_h_*, r05_i_*).
        u32 flag_now = 0;
        if constexpr (SEG) {
            flag_now = __builtin_amdgcn_readfirstlane(seg_flag);
            asm volatile("" ::"s"(flag_now) : "memory");
        }
        stage_read<G, NW>(stage, wave, lane, cur);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PT(1)
        if (t + 1 < t1) dma_issue<G, NW, DMA_AUX>(p, p.frames + (size_t)(t + 1) * p.frame_stride, wave_group0, g1, wave, lane, stage);
        const bool clip_start = SEG && flag_now != 0;
        if constexpr (SEG) {
            if (t + 1 < t1) seg_flag = p.seg[t + 1];
        }
#else
        const bool clip_start = SEG && p.seg[t] != 0;
#pragma unroll
        for (int k = 0; k < G; k++)
#pragma unroll
            for (int i = 0; i < 12; i++) asm volatile("" : "+v"(cur[k].w[i]));
#endif
        u32 sh = 0, ss = 0, sv = 0, bs = 0;
        const bool chain = have_prev && !clip_start;
#pragma unroll
        for (int k = 0; k < G; k++) {
            if (HSV || live[k]) {
                Hsv16& c = nxt[k];
                convert_group<HSV, LUMA, LCD, ACD, FPK, S1, VM, REL>(cur[k], c, my_lut_s, my_lut_h, my_acc, bs, live[k] ? 1u : 0u);
                if (HSV) {
                    if (chain) sad_group(c, prv[k], sh, ss, sv);
                }
                if constexpr (VM) {
                    // 16 V bytes of the group, in pixel order (a group is 16 consecutive pixels: byte offset 16 * group)
#if PSD_ABLATE & 32
                    asm volatile("" :: "v"(c.v[0]), "v"(c.v[1]), "v"(c.v[2]), "v"(c.v[3]));
#else
                    if (live[k]) {
                        u32x4 pk;
                        pk.x = c.v[0]; pk.y = c.v[1]; pk.z = c.v[2]; pk.w = c.v[3];
                        u32x4* dst = reinterpret_cast<u32x4*>(p.vout + (size_t)t * p.npix + (size_t)(wave_group0 + k * NW * 64 + lane) * 16);
                        // (Rounds 4-5 chased a "content dependence" of this store -- 0.17 ms per 1024 shot-like frames, 0.04 ms on
                        //  noise -- through non-temporal / write-through forms and a dozen XOR keys on the stored bytes.  It is not
                        //  the store: keyed planes that the Sobel kernel un-keys changed nothing, and the variants that "helped" were
                        //  the ones that fed the kernels BEHIND this one garbage -- their long, low-power runs bank package power
                        //  that this pass then spends as clock.  profiles/r05_klm_*, r05_no_*.)
                        *dst = pk;
                    }
#endif
                }
            }
        }
        constexpr int E0 = LUMA ? 256 : 0;  // index of sad_h among the slot's accumulators
        // (the frame's sums go into the accumulators with LDS adds the compiler does not see, like the histogram increments:
        //  in front of a visible LDS atomic hipcc waits for the staging of frame t + 1 -- s_waitcnt vmcnt(0) -- at the end of
        //  every step; lds_hidden_fence() below, in front of the barrier behind which the flush reads them)
        if (HSV && chain) {   // (uniform; without a predecessor the sums are zero: nothing to add, no registers to clear)
            lds_add_hidden(&my_acc[(E0 + 0) * ACD], sh);
            lds_add_hidden(&my_acc[(E0 + 1) * ACD], ss);
            lds_add_hidden(&my_acc[(E0 + 2) * ACD], sv);
        }
        if (LUMA) lds_add_hidden(&my_acc[(E0 + 3) * ACD], bs);
        have_prev = true;
        PT(2)
#if !(PSD_ABLATE & 2)
        // Every F frames (and at the end of the chunk) the block's sums for the last <= F frames
        // are complete in one half of the slot ring: flush that half while the other half fills.
        const int done = t - t0 + 1;
        if (done % F == 0 || t + 1 == t1) {
            lds_barrier();      // (lgkmcnt(0) + s_barrier: the waves share LDS only; __syncthreads() would also drain the staging of frame t + 1)
            PT(3)
            const int nf = (done % F == 0) ? F : done % F;       // frames in this half
            const int first = done - nf;                          // chunk-relative index of the first
            for (int i = tid; i < nf * 2 * NA; i += WG) {
                const int fi = i / (2 * NA), idx = i - fi * 2 * NA;
                flush_slot<LUMA, ACD>(&acc[(first + fi) % SLOTS][0], p.out + t0 + first + fi, idx,
                                      VM ? p.vhist + (size_t)(t0 + first + fi) * 256 : nullptr);
            }
            PT(4)
        }
#endif
    };
    // (A/B on 4096 x 1080p: HSV pass 5.30 -> 5.18 ms, the V-mode pass of the edge term +0.8 %; the fused HSV + luma pass lost 3 %
    //  with two copies of its 1350-instruction step, and gains 1 % since the step has shrunk to 1250: profiles/r03_ad_*)
    {
        int t = t0;
        for (; t + 1 < t1; t += 2) {
            step(t, prev, other);
            step(t + 1, other, prev);
        }
        if (t < t1) step(t, prev, other);
    }
    PT_DONE
    TL_END
}

// ---- luma histogram + byte sum without the time walk -------------------------------------------
//
// HistogramDetector / ThresholdDetector need nothing from the previous frame, so the luma-only pass does not
// have to walk the time axis: one small workgroup (4 waves) takes one spatial tile of ONE frame, streams it
// HBM -> LDS exactly like the kernel above (wave-private 3 KiB slots, one step ahead), counts into its 16x
// replicated LDS histogram and flushes once at the end -- no per-frame workgroup barrier, three workgroups per
// CU.  grid = (tiles per frame, frames).
constexpr int kLumaWG = 256, kLumaG = 2, kLumaStepsPerTile = 8;

template <int G>
__global__ __launch_bounds__(kLumaWG) void luma_hist_kernel(const ScoreParams p)
{
    constexpr int WG = kLumaWG, NW = WG / 64, AC = 16, STEP = G * NW * 64;
    __shared__ __attribute__((aligned(16))) u32 acc[NACC * AC];
    __shared__ __attribute__((aligned(16))) uint8_t stage[G * NW * 3072];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, t = blockIdx.y;
    for (int i = tid; i < NACC * AC; i += WG) acc[i] = 0;
    __syncthreads();
    const int g0 = p.group_begin + tile * p.groups_per_tile;
    const int g1 = min(p.group_end, g0 + p.groups_per_tile);
    const uint8_t* frame = p.frames + (size_t)t * p.frame_stride;
    const long limit = (long)g1 * 48;
    auto issue = [&](int sbase) {
#pragma unroll
        for (int k = 0; k < G; k++) {
            // (uniform base pointer pinned to SGPRs + the lane's 16 bytes + immediate piece offsets: see dma_issue)
            const long first = ((long)sbase + (long)(k * NW + wave) * 64) * 48;
            uint8_t* slot = stage + (size_t)(k * NW + wave) * 3072;
            typedef const __attribute__((address_space(1))) uint8_t* gbl_u8_t;
            gbl_u8_t fb = (gbl_u8_t)(frame + first);
            asm volatile("" : "+s"(fb));
            const u32 voff = (u32)lane * 16u;
            const long room = limit - first - (long)voff;      // bytes of the tile from this lane's first byte on
            if (room >= 16) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(fb + voff), (lds_ptr_t)slot, 16, 0, PSD_DMA_AUX);
            if (room >= 1024 + 16) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(fb + voff), (lds_ptr_t)slot, 16, 1024, PSD_DMA_AUX);
            if (room >= 2048 + 16) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(fb + voff), (lds_ptr_t)slot, 16, 2048, PSD_DMA_AUX);
        }
    };
    u32* my_acc = acc + (tid & (AC - 1));
    u32 bs = 0;
    if (g0 < g1) issue(g0);
    for (int sbase = g0; sbase < g1; sbase += STEP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Group cur[G];
        stage_read<G, NW>(stage, wave, lane, cur);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (sbase + STEP < g1) issue(sbase + STEP);
#pragma unroll
        for (int k = 0; k < G; k++) {
            if (sbase + (k * NW + wave) * 64 + lane < g1) {
                Hsv16 unused;
                convert_group<false, true, COPIES, AC>(cur[k], unused, nullptr, nullptr, my_acc, bs, 1u);
            }
        }
    }
    __hip_atomic_fetch_add(&my_acc[ACC_BYTES * AC], bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_hidden_fence();
    __syncthreads();
#if PSD_HIST_PAIR_ATOMICS
    // The tile's flush (third session of round 6).  A device-scope atomic leaves the XCD as a memory transaction of its own whatever its
    // width, and this pass -- memory-bound -- pays for their number: without the histogram's atomics it is 2.8 % faster, without any 5 %
    // (profiles/r06_ag_*).  So thread b sums the copies of bin b, and bins 2 i and 2 i + 1 leave in ONE 64-bit atomic (a bin counts pixels of
    // one frame: nothing ever carries from the low word into the high one; rec->hist is 8-byte aligned): half the transactions of the
    // one-per-bin form, which also took three passes over 2 x 260 half-accumulators.  4K Histogram + Threshold +2 %.
    static_assert(WG == 256 && AC == 16, "one thread per bin");
    psd_frame_scores* rec = p.out + t;
    {
        const u32x4* q = reinterpret_cast<const u32x4*>(acc + tid * AC);
        const u32x4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        const u32 s = (v0.x + v0.y + v0.z + v0.w) + (v1.x + v1.y + v1.z + v1.w) + (v2.x + v2.y + v2.z + v2.w) + (v3.x + v3.y + v3.z + v3.w);
        const u32 up = __shfl_down(s, 1);
        if (!(tid & 1) && (s | up))
            atomicAdd(reinterpret_cast<unsigned long long*>(&rec->hist[tid]), (unsigned long long)s | ((unsigned long long)up << 32));
    }
    if (tid < AC) {        // the byte sum's sixteen copies: lanes 0 .. 15 of wave 0
        u32 v = acc[ACC_BYTES * AC + tid];
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        if (tid == 0 && v) atomicAdd((unsigned long long*)&rec->byte_sum, (unsigned long long)v);
    }
#else
    for (int i = tid; i < 2 * NACC; i += WG) flush_slot<true, AC>(acc, p.out + t, i);
#endif
}

static hipError_t launch_luma_hist(ScoreParams p, int group_begin, int group_end, hipStream_t stream, int* launches)
{
    constexpr int STEP = kLumaG * (kLumaWG / 64) * 64;
    const int n_groups = group_end - group_begin;
    if (n_groups <= 0) return hipSuccess;
    // long batches: 8 steps (4096 groups) per workgroup; short ones (the per-frame API): one step, for parallelism
    const int cap = STEP * (p.n >= 32 ? kLumaStepsPerTile : 1);
    p.group_begin = group_begin;
    p.group_end = group_end;
    p.n_tiles = (n_groups + cap - 1) / cap;
    p.groups_per_tile = (n_groups + p.n_tiles - 1) / p.n_tiles;
    p.frames_per_chunk = 1;
    const uint8_t* frames = p.frames;
    psd_frame_scores* out = p.out;
    const int n = p.n;
    for (int t0 = 0; t0 < n; t0 += 32768) {   // grid.y limit
        p.frames = frames + (size_t)t0 * p.frame_stride;
        p.out = out + t0;
        p.n = n - t0 < 32768 ? n - t0 : 32768;
        hipLaunchKernelGGL((luma_hist_kernel<kLumaG>), dim3(p.n_tiles, p.n), dim3(kLumaWG), 0, stream, p);
        *launches += 1;
    }
    return hipGetLastError();
}

// PSD_LUMA_WALK=1 keeps the luma-only pass on the time-walking kernel (experiments / A-B runs).
static bool luma_time_walk()
{
    static const bool d = [] { const char* e = getenv("PSD_LUMA_WALK"); return e && atoi(e) != 0; }();
    return d;
}

// ---- host-side launcher ---------------------------------------------------------------------

// PSD_SCORE_DIRECT=1 selects the register-staged loads on the fast path too (experiments).
static bool direct_loads()
{
    static const bool d = [] { const char* e = getenv("PSD_SCORE_DIRECT"); return e && atoi(e) != 0; }();
    return d;
}

template <bool HSV, bool LUMA, int G, bool FAST>
static hipError_t launch_one(const ScoreParams& p, int grid, hipStream_t stream)
{
    constexpr int WG = kScoreWG;
    if constexpr (FAST && HSV && LUMA) {
        if (p.vout != nullptr) {   // V mode (launch_score_frames checked the preconditions)
            if (p.seg != nullptr) hipLaunchKernelGGL((score_frames_dma_kernel<true, true, G, kFusedWG, true, true>), dim3(grid), dim3(kFusedWG), 0, stream, p);
            else hipLaunchKernelGGL((score_frames_dma_kernel<true, true, G, kFusedWG, true, false>), dim3(grid), dim3(kFusedWG), 0, stream, p);
            return hipGetLastError();
        }
    }
    if constexpr (FAST) {
        if (!direct_loads()) {
            constexpr int SWG = (HSV && (!LUMA || PSD_FUSED_SMALL)) ? kHsvWG : (HSV && LUMA) ? kFusedWG : kScoreWG;
            // (the clip-start flags only matter to the HSV carry: the luma-only instance has no SEG form)
            if (HSV && p.seg != nullptr) hipLaunchKernelGGL((score_frames_dma_kernel<HSV, LUMA, G, SWG, false, HSV>), dim3(grid), dim3(SWG), 0, stream, p);
            else hipLaunchKernelGGL((score_frames_dma_kernel<HSV, LUMA, G, SWG, false, false>), dim3(grid), dim3(SWG), 0, stream, p);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((score_frames_kernel<HSV, LUMA, G, FAST, WG>), dim3(grid), dim3(WG), 0, stream, p);
    return hipGetLastError();
}

template <bool FAST, int G>
static hipError_t launch_flags(const ScoreParams& p, int grid, bool hsv, bool luma, hipStream_t s)
{
    if (hsv && luma) return launch_one<true, true, G, FAST>(p, grid, s);
    if (hsv) return launch_one<true, false, G, FAST>(p, grid, s);
    return launch_one<false, true, G, FAST>(p, grid, s);
}

// Groups of 16 pixels per lane per frame.  Tunable for experiments with PSD_SCORE_G=1|2.
static int groups_per_lane()
{
    static const int g = [] {
        const char* e = getenv("PSD_SCORE_G");
        const int v = e ? atoi(e) : kScoreG;
        return v == 2 ? 2 : 1;
    }();
    return g;
}

static hipError_t launch_range(ScoreParams p, int group_begin, int group_end, bool hsv, bool luma,
                               bool fast, int target_blocks, hipStream_t stream)
{
    // Tile geometry: spread the 16-px groups evenly over the fewest tiles that fit.
    const bool staged = fast && !direct_loads();
    const int wg = (staged && hsv && (!luma || PSD_FUSED_SMALL)) ? kHsvWG : (staged && hsv && luma) ? kFusedWG : kScoreWG;
    const bool mid = wg != kScoreWG && wg != kHsvWG;   // a fused pass on PSD_FUSED_WG-thread workgroups keeps G and the walk rules
    // The small-workgroup HSV pass takes one group per lane unless PSD_SCORE_G says otherwise: 80 VGPRs and 24 KiB of
    // LDS, six workgroups (24 waves) per CU.  Sweep on 4096 x 1080p: G=1 5.71 ms, G=2 5.81 ms (1024-thread kernel 5.9-6.0).
    static const bool g_forced = getenv("PSD_SCORE_G") != nullptr;
    const int gpl = !fast ? 1 : (wg != kScoreWG && !mid && !g_forced) ? 1 : groups_per_lane();
    const int cap = wg * gpl;
    const int n_groups = group_end - group_begin;
    if (n_groups <= 0) return hipSuccess;
    const int target_blocks_in = target_blocks;
    p.group_begin = group_begin;
    p.group_end = group_end;
    p.n_tiles = (n_groups + cap - 1) / cap;
    p.groups_per_tile = (n_groups + p.n_tiles - 1) / p.n_tiles;
    // Time chunks: enough workgroups to keep every CU busy for several rounds, but chunks long
    // enough that the re-read halo frame stays a small fraction.
    if (staged && hsv && luma) {
        // (experiments: workgroups per launch of the 16-wave fused passes; default 8 per CU)
        static const int env_fused = [] { const char* e = getenv("PSD_FUSED_BLOCKS"); return e ? atoi(e) : 0; }();
        if (env_fused > 0) target_blocks = env_fused;
    }
    if (mid) target_blocks *= kScoreWG / wg;
    else if (wg != kScoreWG) {
        // small workgroups: several share a CU, so ask for proportionally more of them (PSD_HSV_BLOCKS overrides)
        static const int env_blocks = [] { const char* e = getenv("PSD_HSV_BLOCKS"); return e ? atoi(e) : 0; }();
        // sweep on 4096 x 1080p (G=1): 2048 blocks 6.13 ms, 8192 5.86, 16384 5.79, 32768 5.71, 65536 5.76
        target_blocks = env_blocks > 0 ? env_blocks : target_blocks * (kScoreWG / wg) * 4;
    }
    // `target_blocks` bounds the parallelism from above.  With the HSV carry every chunk also converts one halo
    // frame, so small frames (few tiles) should not be cut into very short walks: prefer >= 32 (8) frames per chunk
    // as long as that still leaves about one workgroup per resident slot (short batches keep maximum parallelism).
    // 4096 x 256x144 (the reference's default downscaled size): 0.121 ms vs 0.160 ms without this rule.
    const int chunks_hi = (target_blocks + p.n_tiles - 1) / p.n_tiles;
    int chunks = chunks_hi;
    if (hsv) {
        const int slots = (target_blocks_in / 8) * (mid ? kScoreWG / wg : wg != kScoreWG ? 6 : 1);   // target_blocks_in = 8 per CU
        const int chunks_lo = (slots + p.n_tiles - 1) / p.n_tiles;
        const int walk = (wg != kScoreWG && !mid) ? 32 : 8;   // 16-wave workgroups (one per CU) need the parallelism more
        chunks = (p.n + walk - 1) / walk;
        if (chunks < chunks_lo) chunks = chunks_lo;
        if (chunks > chunks_hi) chunks = chunks_hi;
#if PSD_FILL_LAST_ROUND
        // Workgroups of one launch all take about the same time, so the launch runs in rounds of `slots` resident workgroups
        // and a last round that is 10 % full costs as much as a full one (tools/wg_timeline.py: 21.1 rounds of 230 us at 4096 x
        // 1080p = 0.2 ms of a 5 ms launch with the chip nearly empty).  Among the chunk counts within 20 % of the target
        // take the one whose last round is fullest.
        if (fast && chunks > 1 && slots > 0) {
            double best = 2.0;
            int best_c = chunks;
            for (int c = std::max(1, chunks - chunks / 5); c <= chunks + chunks / 5 && c <= p.n; c++) {
                const int fpc = (p.n + c - 1) / c, cc = (p.n + fpc - 1) / fpc;
                const double rounds = (double)p.n_tiles * cc / slots;
                const double waste = (std::ceil(rounds) - rounds) / std::ceil(rounds);
                if (waste < best - 1e-9 || (waste < best + 1e-9 && std::abs(cc - chunks) < std::abs(best_c - chunks))) { best = waste; best_c = cc; }
            }
            chunks = best_c;
        }
#endif
    }
    if (chunks > p.n) chunks = p.n;
    if (chunks < 1) chunks = 1;
    p.frames_per_chunk = (p.n + chunks - 1) / chunks;
    chunks = (p.n + p.frames_per_chunk - 1) / p.frames_per_chunk;
    const int grid = p.n_tiles * chunks;
    if (hsv && fast) note_walk_geometry(p.frames_per_chunk, p.n_tiles);
    if (!fast) return launch_flags<false, 1>(p, grid, hsv, luma, stream);
    return gpl == 2 ? launch_flags<true, 2>(p, grid, hsv, luma, stream)
                    : launch_flags<true, 1>(p, grid, hsv, luma, stream);
}

bool score_v_mode_available(long npix) { return !direct_loads() && npix % 16 == 0; }

hipError_t launch_score_frames(ScoreParams p, bool hsv, bool luma, bool fast, int target_blocks,
                               hipStream_t stream, int* launches)
{
    const int total_groups = (int)((p.npix + 15) / 16);
    const int full_groups = (int)(p.npix / 16);
    if (p.vout != nullptr) {
        // V mode: the HSV term and the edge term's V plane + V histogram from one pass (the staged 16-wave fused kernel, its
        // histogram slots counting V; whole groups only)
        if (!hsv || luma || !fast || direct_loads() || total_groups != full_groups || p.vhist == nullptr) return hipErrorInvalidValue;
        *launches += 1;
        return launch_range(p, 0, full_groups, true, true, true, target_blocks, stream);
    }
    if (!fast) {
        *launches += 1;
        return launch_range(p, 0, total_groups, hsv, luma, false, target_blocks, stream);
    }
    // Fast kernel over the full 48-byte groups; the ragged tail (< 16 px per frame), if any, goes
    // through the generic kernel.  Both add into the same records.
    hipError_t err = hipSuccess;
    if (full_groups > 0) {
        if (luma && !hsv && !direct_loads() && !luma_time_walk()) {
            err = launch_luma_hist(p, 0, full_groups, stream, launches);
        } else {
            *launches += 1;
            err = launch_range(p, 0, full_groups, hsv, luma, true, target_blocks, stream);
        }
    }
    if (err == hipSuccess && total_groups > full_groups) {
        *launches += 1;
        err = launch_range(p, full_groups, total_groups, hsv, luma, false, 1, stream);
    }
    return err;
}

}  // namespace psd

#if PSD_WG_TIMELINE
extern "C" int psd_debug_timeline(unsigned long long* d_buffer)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(psd::g_timeline), &d_buffer, sizeof d_buffer) == hipSuccess ? 0 : -1;
}
#endif

#if PSD_PHASE_TIMING
// phases: 0 wait for the frame's DMA, 1 staging slot -> registers, 2 arithmetic + LDS drain, 3 workgroup barrier, 4 flush,
// 5 rest (loop overhead, DMA issue of the next frame is inside 2); [6] = the same span on the constant 100 MHz clock
// (s_memrealtime), so sum(0..5) / [6] x 100 MHz is the shader clock the waves saw; [7] = waves counted.  reset != 0 clears the counters.
extern "C" int psd_debug_phases(unsigned long long* out, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(psd::g_phase), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(psd::g_phase), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
