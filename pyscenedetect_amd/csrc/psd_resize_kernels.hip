// psd_resize_kernels.hip -- the reference's default pipeline for gfx950: cv2.resize(INTER_LINEAR) in front of the detectors.
//
// SceneManager scores frames of about 256 pixels width unless told otherwise (reference scenedetect/scene_manager.py:110,
// 123-140, 666-678): every frame goes through cv2.resize(frame, (round(w / f), round(h / f)), INTER_LINEAR) first.
// For a decimation by f only TWO of every f source rows carry taps (and two of every f pixels inside them), so the
// algorithmic traffic of "downscale, then ContentDetector" is 2 * dst_h source rows per frame -- 1,658,880 B of the
// 6,220,800 B of a 1080p frame at the default 256 x 144 -- and nothing of the small frame has to exist in HBM.
//
//   resize_walk_kernel<STORE, HSV>
//     A workgroup owns R destination rows and walks a chunk of consecutive frames (like the scoring kernels).  Per frame
//     it streams the 2 R source rows it needs HBM -> LDS with global_load_lds_dwordx4 one frame ahead (double buffered,
//     one barrier per frame), every thread gathers the four taps of its destination pixels from LDS, interpolates in
//     OpenCV's 11-bit fixed point (resize.cpp HResizeLinear / VResizeLinear: ((b0 * (h0 >> 4)) >> 16) + ... + 2) >> 2),
//     and then either stores the BGR pixel (STORE: the plain cv2.resize) and / or converts it to HSV and adds
//     |HSV_t - HSV_{t-1}| to the frame's record (HSV: content_detector.py:155,166-169 on the resized frame), carrying the
//     previous frame's HSV of its pixels in registers.  Everything integer, so records and pixels equal the oracle's.
//   resize_linear_generic_kernel
//     any alignment / odd row strides: one destination pixel per thread straight from global memory.
//
// Coefficient tables (xofs / ialpha / yofs / ibeta in OpenCV's float arithmetic) are built once per (src, dst) shape and
// stay cached in the engine; no allocation or synchronisation on the call path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "psd_internal.h"

extern "C" void psd_set_error(const char* fmt, ...);

namespace psd {

void** engine_resize_cache(psd_engine* e);
int engine_hist_scratch(psd_engine* e, size_t bytes, hipStream_t stream, void** out);   // grows on demand (psd_engine.cpp)
int engine_num_cus(psd_engine* e);
const uint32_t* engine_lut(psd_engine* e);

typedef uint32_t u32;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            psd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)hipGetLastError(); /* (the failure is reported here: do not leave it for the next launch check) */ \
            return PSD_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)

struct XTap { int o0, o1; int a; };   // byte offsets 3*sx, 3*min(sx+1, sw-1); a0 | a1 << 16
struct YTap { int s0, s1; int b; };   // clamped source rows; b0 | b1 << 16

struct ResizeTabs {
    XTap* x = nullptr;   // device, [dst_w]
    YTap* y = nullptr;   // device, [dst_h]
    bool area2 = false;  // exact 2x2 decimation: OpenCV routes INTER_LINEAR to INTER_AREA (rounded box mean)
    unsigned long long used = 0;
};

typedef std::tuple<int, int, int, int, int> ShapeKey;   // src h, w, dst h, w, kind
struct OtherTab { void* d; int mode; float inv_area; unsigned long long used; };
struct ResizeCache {
    std::map<ShapeKey, ResizeTabs> linear;   // INTER_LINEAR taps (kind 0; 1 = the INTER_AREA enlargement coefficients)
    std::map<ShapeKey, OtherTab> other;      // TableKind tables: INTER_NEAREST offsets, INTER_AREA runs, hash thumbnail runs
    unsigned long long tick = 0;             // use counter: the least recently used table goes when a map is full
};
// A long-running service that sees ever new resolutions or crop sizes must not accumulate tables for the life of the engine
// (a few KiB each, but unbounded): each map keeps the kMaxTables most recently used.
constexpr size_t kMaxTables = 48;

template <typename Map, typename Free>
static void evict_lru(Map& m, Free free_entry)
{
    while (m.size() >= kMaxTables) {
        auto victim = m.begin();
        for (auto it = m.begin(); it != m.end(); ++it)
            if (it->second.used < victim->second.used) victim = it;
        // (kernels queued earlier may still read it: a full-device wait on this cold path, once per 49th new shape)
        (void)hipDeviceSynchronize();
        free_entry(victim->second);
        m.erase(victim);
    }
}

static ResizeCache& cache_of(psd_engine* e)
{
    void** slot = engine_resize_cache(e);
    if (!*slot) *slot = new ResizeCache();
    return *static_cast<ResizeCache*>(*slot);
}

bool table_find(psd_engine* e, int kind, int sh, int sw, int dh, int dw, DevTable* out)
{
    ResizeCache& c = cache_of(e);
    auto it = c.other.find(std::make_tuple(sh, sw, dh, dw, kind));
    if (it == c.other.end()) return false;
    it->second.used = ++c.tick;
    out->ptr = it->second.d; out->mode = it->second.mode; out->inv_area = it->second.inv_area;
    return true;
}

int table_store(psd_engine* e, int kind, int sh, int sw, int dh, int dw, const void* host, size_t bytes, int mode, float inv_area,
                DevTable* out)
{
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    // once per shape: a synchronous copy (the host table dies with the caller)
    hipError_t err = hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        (void)hipFree(d);
        psd_set_error("coefficient table upload failed: %s", hipGetErrorString(err));
        return PSD_ERR_HIP;
    }
    ResizeCache& c = cache_of(e);
    evict_lru(c.other, [](OtherTab& t) { if (t.d) (void)hipFree(t.d); });
    c.other[std::make_tuple(sh, sw, dh, dw, kind)] = OtherTab{d, mode, inv_area, ++c.tick};
    out->ptr = d; out->mode = mode; out->inv_area = inv_area;
    return PSD_OK;
}

static short sat_s16_round(float v)
{
    long r = lrintf(v);
    return (short)(r > 32767 ? 32767 : (r < -32768 ? -32768 : r));
}

// OpenCV's coefficient tables (resize.cpp, INTER_LINEAR, 11-bit fixed point), in the same float arithmetic.
// area_mode: cv2.resize(INTER_AREA) that does not shrink along both axes -- OpenCV emulates it with the bilinear passes and
// other coefficients (resize.cpp: sx = cvFloor(dx * scale_x), fx = (dx + 1) - (sx + 1) * inv_scale_x, 0 if <= 0 else its
// fractional part); the kernels are the INTER_LINEAR ones.
static void linear_tabs_host(int sh, int sw, int dh, int dw, bool area_mode, std::vector<XTap>& xt, std::vector<YTap>& yt, bool* area2)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    *area2 = !area_mode && fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16 &&
             iscale_x == 2 && iscale_y == 2;
    xt.resize(dw);
    yt.resize(dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx;
        int sx;
        if (!area_mode) {
            fx = (float)((dx + 0.5) * scale_x - 0.5);
            sx = (int)floorf(fx);
            fx -= sx;
        } else {
            sx = (int)floor(dx * scale_x);
            fx = (float)((dx + 1) - (sx + 1) * inv_scale_x);
            fx = fx <= 0 ? 0.f : fx - floorf(fx);
        }
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
        const short a0 = sat_s16_round((1.f - fx) * 2048), a1 = sat_s16_round(fx * 2048);
        xt[dx].o0 = 3 * sx; xt[dx].o1 = 3 * sx1;
        xt[dx].a = (int)((u32)(uint16_t)a0 | ((u32)(uint16_t)a1 << 16));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy;
        int sy;
        if (!area_mode) {
            fy = (float)((dy + 0.5) * scale_y - 0.5);
            sy = (int)floorf(fy);
            fy -= sy;
        } else {
            sy = (int)floor(dy * scale_y);
            fy = (float)((dy + 1) - (sy + 1) * inv_scale_y);
            fy = fy <= 0 ? 0.f : fy - floorf(fy);
        }
        const short b0 = sat_s16_round((1.f - fy) * 2048), b1 = sat_s16_round(fy * 2048);
        yt[dy].s0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        yt[dy].s1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        yt[dy].b = (int)((u32)(uint16_t)b0 | ((u32)(uint16_t)b1 << 16));
    }
}

static int get_tabs(psd_engine* e, int sh, int sw, int dh, int dw, hipStream_t stream, ResizeTabs* out, bool area_mode = false)
{
    ResizeCache& rc_ = cache_of(e);
    std::map<ShapeKey, ResizeTabs>& cache = rc_.linear;
    const auto key = std::make_tuple(sh, sw, dh, dw, area_mode ? 1 : 0);
    auto it = cache.find(key);
    if (it != cache.end()) { it->second.used = ++rc_.tick; *out = it->second; return PSD_OK; }
    evict_lru(cache, [](ResizeTabs& t) { if (t.x) (void)hipFree(t.x); });
    ResizeTabs t;
    std::vector<XTap> xt;
    std::vector<YTap> yt;
    linear_tabs_host(sh, sw, dh, dw, area_mode, xt, yt, &t.area2);
    uint8_t* d = nullptr;
    const size_t xb = (sizeof(XTap) * dw + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc((void**)&d, xb + sizeof(YTap) * dh));
    t.x = reinterpret_cast<XTap*>(d);
    t.y = reinterpret_cast<YTap*>(d + xb);
    // once per shape: a synchronous copy (the host vectors die with this call)
    HIP_TRY(hipMemcpy(t.x, xt.data(), sizeof(XTap) * dw, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.y, yt.data(), sizeof(YTap) * dh, hipMemcpyHostToDevice));
    (void)stream;
    t.used = ++rc_.tick;
    cache[key] = t;
    *out = t;
    return PSD_OK;
}

void resize_release(psd_engine* e)
{
    void** slot = engine_resize_cache(e);
    if (!*slot) return;
    ResizeCache* cache = static_cast<ResizeCache*>(*slot);
    for (auto& kv : cache->linear)
        if (kv.second.x) (void)hipFree(kv.second.x);
    for (auto& kv : cache->other)
        if (kv.second.d) (void)hipFree(kv.second.d);
    delete cache;
    *slot = nullptr;
}

struct RsParams {
    const uint8_t* src;      // frame t at src + t * sstride, rows at srow
    const uint8_t* prev;     // source-size frame preceding frame 0, or null
    const uint8_t* seg;      // n flags: frame t starts a clip (no predecessor), or null
    size_t sstride, srow;
    int sh, sw, row_bytes;   // row_bytes = 3 * sw
    uint8_t* dst;            // STORE: frame t at dst + t * dstride, packed rows
    size_t dstride;
    int dh, dw;
    psd_frame_scores* out;   // HSV: n records, zero-initialised
    const uint32_t* lut;     // [0..255] = sdiv << 4, [256..511] = hdiv180 << 4
    const XTap* xt;
    const YTap* yt;
    int n, rows_per_tile, n_tiles, frames_per_chunk, row_pad;
    int area2;
    int store_vec;           // STORE: a tile's pixels leave through LDS as 16-byte stores (whole, 16-byte aligned tiles), else byte stores
    int depth;               // staging buffers: 2 = one frame ahead, 3 = two frames ahead
    uint8_t* vout;           // VOUT: the resized frame's V plane (max(B, G, R), one byte per pixel), frame t at vout + t * dh * dw
    u32* hpart;              // LUMA / VOUT: per (frame, tile) partial luma (V) histograms, `hstride` words each (rs_hist_flush)
    int hstride;
};

// 24-bit multiplies (v_mul_lo_u32 issues at a quarter of their rate; hipcc does not pick them for `>> 4`-ed or table operands)
__device__ __forceinline__ int mul_i24(int a, int b) { int d; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ u32 mul_u24(u32 a, u32 b) { u32 d; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

__device__ __forceinline__ u32 interp(u32 p00, u32 p01, u32 p10, u32 p11, int a0, int a1, int b0, int b1)
{
    // every factor is far below 2^23 (pixels < 2^8, coefficients <= 2^11, h >> 4 < 2^15): 24-bit multiplies, which issue at
    // four times the rate of v_mul_lo_u32
    const int h0 = mul_i24((int)p00, a0) + mul_i24((int)p01, a1);
    const int h1 = mul_i24((int)p10, a0) + mul_i24((int)p11, a1);
    return (u32)(((mul_i24(b0, h0 >> 4) >> 16) + (mul_i24(b1, h1 >> 4) >> 16) + 2) >> 2) & 0xffu;
}

#ifndef PSD_RS_WG
#define PSD_RS_WG 256
#endif
#ifndef PSD_RS_SWAP
#define PSD_RS_SWAP 0
#endif
#ifndef PSD_RS_HUE_SELECT
#define PSD_RS_HUE_SELECT 0
#endif
constexpr int RS_WG = PSD_RS_WG, RS_NW = RS_WG / 64, RS_SLOTS = 16, RS_REP = 16;
// copies of a tile's luma histogram (lane l counts into copy l % RS_HREP).  Neighbouring pixels of a natural frame share their
// luma, and the lanes of ONE ds_add_u32 that hit the same word are served one after the other -- which cost the score kernels up
// to 40 % on constant frames until they got 16 copies (psd_score_kernels.hip, round 4).  Not here: 1 / 4 / 8 copies measure the
// same on uniform, shot-like and object frames (1.18-1.29 / 1.20-1.22 / 1.20-1.32 ms per 4096 x 1080p, all four detectors;
// profiles/r06_j_ab_resize_hist_replicas.txt) -- a lane adds 2-8 pixels per frame between barriers, the atomics are not what
// the step waits for -- and one copy has the cheapest flush.
#ifndef PSD_RS_HREP
#define PSD_RS_HREP 1
#endif
constexpr int RS_HREP = PSD_RS_HREP;
// A tile's partial histogram in memory (third session of round 6).  A tile has at most 2048 pixels, and its 256 counts used to leave as 128
// words of two 16-bit counts: 512 bytes per (frame, tile), 165 MB per 4096 x 1080p -- scattered writes between streaming reads, which cost
// up to 0.16 ms of the pass's 1.34 ("two states").  Now the LOW BYTES of the counts leave as 64 words (bins 4 i .. 4 i + 3 in
// lane i of wave 0), and what a byte cannot hold -- a count >= 256: at most tile pixels / 256 lanes can own one -- as an escape: words 64 /
// 65 = the ballot of the lanes that own such a bin, words 66 .. = those lanes' four carries (count >> 8, four bits each), in lane order.
// 272 bytes per (frame, tile) at 1080p -> 256 x 144 instead of 512.
#ifndef PSD_RS_HPACK8
#define PSD_RS_HPACK8 1
#endif
constexpr bool RS_HPACK8 = PSD_RS_HPACK8 != 0;
static int rs_hist_stride_words(int tile_pixels) { return RS_HPACK8 ? 66 + std::min(64, (tile_pixels + 255) / 256) : 128; }

// LDS increment the compiler does not see as an LDS store: hipcc orders every LDS store / atomic behind ALL outstanding LDS-DMA
// (s_waitcnt vmcnt(0)), i.e. behind the staging of the NEXT frame, which would serialise the prefetch with this frame's
// arithmetic (psd_score_kernels.hip, lds_add_hidden).  The histogram / sum words never overlap the staging buffers; the barrier of
// the next frame is preceded by an explicit lgkmcnt(0).
__device__ __forceinline__ void rs_lds_add(const u32* p, u32 inc)
{
    asm volatile("ds_add_u32 %0, %1" ::"v"((u32)(uintptr_t)p), "v"(inc) : "memory");
}

typedef unsigned short rs_u16x2 __attribute__((ext_vector_type(2)));

// The four taps of one destination pixel, three channels (round 6).  Both taps of a row are ADJACENT source pixels (the table's
// second tap is only ever clamped onto the first where its coefficient is 0), i.e. six consecutive bytes b0 g0 r0 b1 g1 r1 at
// byte offset 3 * sx of the staged row: three dwords instead of six byte reads (rs_taps_load).  v_perm_b32 pairs a channel's two bytes as
// 16-bit halves and v_dot2_u32_u16 against (a0 << 4 | a1 << 20) IS OpenCV's horizontal pass times 16, in one instruction:
//   hx = (p0 a0 + p1 a1) << 4;   OpenCV: ((b0 (h0 >> 4)) >> 16) + ((b1 (h1 >> 4)) >> 16) + 2) >> 2
// and (hx & 0xffff00) = (h >> 4) << 8, so the vertical products are the upper halves of 24-bit multiplies by b << 8
// (v_mul_hi_u32_u24: (((h >> 4) << 8) (b << 8)) >> 32 = ((h >> 4) b) >> 16, exact: both factors are below 2^23).
// 36 VALU + 4 LDS instructions per pixel where the byte-wise form took ~65 + 12.
struct RsPx { u32 off, a4, b80, b81; };      // off: byte offset of the first tap inside a staging buffer (row 2 lr, 3 sx)

// How the six tap bytes of a row come out of LDS: three dwords from the dword-aligned address below the tap (ds_read2_b32 +
// ds_read_b32: dword alignment is all they ask for), then two v_alignbyte_b32 shift the tap to byte 0.  (The obvious form -- ONE
// unaligned 8-byte read at the tap, which hipcc emits as ds_read_b64 and gfx950 executes correctly -- runs on a slow path of the
// LDS: with 40 % fewer VALU instructions than the byte-wise kernel it was 8 - 27 % SLOWER, the aligned form 10 - 36 % faster:
// profiles/r06_d_ab_resize_taps.txt.)  The loads are inline assembly -- hipcc would fuse them into one "unaligned" ds_read_b96
// -- so the wait for them is the kernel's own, inside the same asm statement (rs_taps_load).
typedef u32 rs_u32x2 __attribute__((ext_vector_type(2)));
struct RsTaps { rs_u32x2 a01, b01; u32 a2, b2; };     // rows a / b: dwords 0..1 and dword 2 from the aligned address

// The loads of C pixels AND the wait for them in ONE asm statement: the compiler does not know that an asm load's register is
// only filled later, so between a load statement and a separate s_waitcnt statement it may copy the register (it did, in the
// G = 1 / 2 instances: stale taps on 1080p and 4K sources while the G = 4 instance happened to be scheduled safely).  Outputs are
// early-clobber: the later loads of the statement still read their address registers.
#define RS_LD4(i)                                        \
    "ds_read2_b32 %[a" #i "], %[p" #i "] offset1:1\n\t" \
    "ds_read_b32 %[c" #i "], %[p" #i "] offset:8\n\t"   \
    "ds_read2_b32 %[b" #i "], %[q" #i "] offset1:1\n\t" \
    "ds_read_b32 %[d" #i "], %[q" #i "] offset:8\n\t"
#define RS_OUT(i, t) [a##i] "=&v"(t.a01), [c##i] "=&v"(t.a2), [b##i] "=&v"(t.b01), [d##i] "=&v"(t.b2)
#define RS_IN(i, pa, rp) [p##i] "v"(pa), [q##i] "v"(pa + rp)

template <int C>
__device__ __forceinline__ void rs_taps_load(const u32 (&addr)[C], u32 row_pad, RsTaps (&t)[C])
{
    static_assert(C == 1 || C == 2 || C == 4, "");
    if constexpr (C == 1) {
        asm volatile(RS_LD4(0) "s_waitcnt lgkmcnt(0)" : RS_OUT(0, t[0]) : RS_IN(0, addr[0], row_pad) : "memory");
    } else if constexpr (C == 2) {
        asm volatile(RS_LD4(0) RS_LD4(1) "s_waitcnt lgkmcnt(0)" : RS_OUT(0, t[0]), RS_OUT(1, t[1])
                     : RS_IN(0, addr[0], row_pad), RS_IN(1, addr[1], row_pad) : "memory");
    } else {
        asm volatile(RS_LD4(0) RS_LD4(1) RS_LD4(2) RS_LD4(3) "s_waitcnt lgkmcnt(0)"
                     : RS_OUT(0, t[0]), RS_OUT(1, t[1]), RS_OUT(2, t[2]), RS_OUT(3, t[3])
                     : RS_IN(0, addr[0], row_pad), RS_IN(1, addr[1], row_pad), RS_IN(2, addr[2], row_pad), RS_IN(3, addr[3], row_pad) : "memory");
    }
}

template <bool AREA2>
__device__ __forceinline__ void rs_interp(const RsTaps& w, u32 shift, const RsPx& t, u32 (&c)[3])
{
    const u32 ax = __builtin_amdgcn_alignbyte(w.a01.y, w.a01.x, shift), ay = __builtin_amdgcn_alignbyte(w.a2, w.a01.y, shift);
    const u32 bx = __builtin_amdgcn_alignbyte(w.b01.y, w.b01.x, shift), by = __builtin_amdgcn_alignbyte(w.b2, w.b01.y, shift);
    const rs_u16x2 a4 = __builtin_bit_cast(rs_u16x2, t.a4);
    constexpr u32 SEL[3] = {0x0C030C00u, 0x0C040C01u, 0x0C050C02u};     // (byte k, byte 3 + k) of the eight as two 16-bit halves
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const u32 ha = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, __builtin_amdgcn_perm(ay, ax, SEL[k])), a4, 0u, false);
        const u32 hb = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, __builtin_amdgcn_perm(by, bx, SEL[k])), a4, 0u, false);
        if (AREA2) {
            c[k] = (ha + hb + 2u) >> 2;                // a4 = (1, 1): the rounded box mean of OpenCV's exact 2 x 2 case
        } else {
            const u32 xa = ha & 0x00ffff00u, xb = hb & 0x00ffff00u;
            __builtin_assume(t.b80 <= (2048u << 8) && t.b81 <= (2048u << 8));      // (24-bit factors: v_mul_hi_u32_u24)
            const u32 ta = (u32)(((unsigned long long)xa * (unsigned long long)t.b80) >> 32);
            const u32 tb = (u32)(((unsigned long long)xb * (unsigned long long)t.b81) >> 32);
            c[k] = (ta + tb + 2u) >> 2;
        }
        __builtin_assume(c[k] <= 255u);
    }
}

// OpenCV's 8-bit BGR -> HSV (RGB2HSV_b, hue range 180) for one pixel, tables pre-shifted by 4; every product fits 24 bits
// (sdiv << 4 <= 255 * 4096 * 16 / 255 ..., hdiv << 4 < 2^21, |hraw| <= 1275), the rounding additions ride in the multiply-adds
__device__ __forceinline__ void rs_hsv(u32 b, u32 g, u32 r, const u32* lut_s, const u32* lut_h, u32& h, u32& s, u32& v)
{
    v = max(max(b, g), r);
    const u32 vmin = min(min(b, g), r);
    const u32 diff = v - vmin;
    const u32 ls = lut_s[v], lh = lut_h[diff];
    __builtin_assume(ls < (1u << 24) && lh < (1u << 21) && diff <= 255u);  // (24-bit factors: v_mad_u32_u24 / v_mad_i32_i24)
    s = (diff * ls + (2048u << 4)) >> 16;
    const int d = (int)diff;
#if PSD_RS_HUE_SELECT
    // without branches: hipcc turns the nested ?: into two divergent branches per pixel (all three arms run in any wave that
    // holds all three cases, plus eight scalar exec-mask instructions)
    const int c0 = (int)g - (int)b, c1 = (int)b - (int)r + 2 * d, c2 = (int)r - (int)g + 4 * d;
    int c12, hraw;
    asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(c12) : "v"(v), "v"(g), "v"(c2), "v"(c1) : "vcc");
    asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(hraw) : "v"(v), "v"(r), "v"(c12), "v"(c0) : "vcc");
#else
    const int hraw = v == r ? (int)g - (int)b : v == g ? (int)b - (int)r + 2 * d : (int)r - (int)g + 4 * d;
#endif
    __builtin_assume(hraw >= -2048 && hraw < 2048);                        // (|hraw| <= 1275)
    const int hh = (hraw * (int)lh + (2048 << 4)) >> 16;
    h = min((u32)hh, (u32)(hh + 180));                                     // hh in [-90, 179]: the negative ones wrap by + 180
}

// A tile's finished histogram (RS_HREP copies in LDS) out to memory in the layout above; CLEAR: the slot is zeroed for the frame after next,
// with stores the compiler does not see, for the reason given at rs_lds_add: a visible LDS store here would make the wave wait for the NEXT
// frame's staging, issued a few lines up (the reads are of a different object than the staging buffers and are not held back).
template <bool CLEAR>
__device__ __forceinline__ void rs_hist_flush(u32* hs, u32* dst, int tid)
{
    if constexpr (RS_HPACK8) {
        if (tid >= 64) return;                       // wave 0, whole: the ballot below is over all 64 lanes
        u32 c[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < RS_HREP; r++) {
            const uint4 w = *reinterpret_cast<const uint4*>(&hs[r * 256 + 4 * tid]);
            c[0] += w.x; c[1] += w.y; c[2] += w.z; c[3] += w.w;
            if (CLEAR) {
                asm volatile("ds_write_b64 %0, %1" ::"v"((u32)(uintptr_t)&hs[r * 256 + 4 * tid]), "v"(0ull) : "memory");
                asm volatile("ds_write_b64 %0, %1" ::"v"((u32)(uintptr_t)&hs[r * 256 + 4 * tid + 2]), "v"(0ull) : "memory");
            }
        }
        const u32 bytes = (c[0] & 255u) | ((c[1] & 255u) << 8) | ((c[2] & 255u) << 16) | (c[3] << 24);
        const u32 carry = (c[0] >> 8) | ((c[1] >> 8) << 4) | ((c[2] >> 8) << 8) | ((c[3] >> 8) << 12);      // (a tile has <= 2048 pixels: <= 8 each)
        const unsigned long long over = __ballot(carry != 0u);
        dst[tid] = bytes;
        if (tid < 2) dst[64 + tid] = tid ? (u32)(over >> 32) : (u32)over;
        if (carry) dst[66 + __popcll(over & ((1ull << tid) - 1ull))] = carry;
    } else {
        if (tid >= 128) return;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int r = 0; r < RS_HREP; r++) {
            lo += hs[r * 256 + 2 * tid]; hi += hs[r * 256 + 2 * tid + 1];
            if (CLEAR) asm volatile("ds_write_b64 %0, %1" ::"v"((u32)(uintptr_t)&hs[r * 256 + 2 * tid]), "v"(0ull) : "memory");
        }
        dst[tid] = lo | (hi << 16);
    }
}

// grid.x = n_tiles * n_chunks.  Dynamic LDS: depth buffers x (2 R rows x row_pad bytes) + 16 bytes of slack (the 8-byte tap reads).
// LUMA: the luma histogram and the byte sum of the RESIZED frame as well (HistogramDetector / ThresholdDetector behind the
// reference's default downscale, histogram_detector.py:156-159 and threshold_detector.py:127 on what scene_manager.py:666-678
// hands them).  A workgroup counts its tile's pixels of frame t into one of two 256-bin LDS histograms and, one barrier
// later, writes it to hpart[t][tile] with plain coalesced stores (rs_hist_flush: low bytes + escapes); hist_reduce_kernel adds the tiles of a
// frame up.  (Global atomics instead -- up to 256 per tile and frame, 75 M per 4096-frame launch on one address per bin and
// frame from every XCD -- would cost more than the pixels.)  The byte sum rides with the three SADs.
// SEG: the instance for batches of packed clips (p.seg != nullptr; psd_score_segments_downscaled_device): a frame that starts
// a clip has no predecessor.  The plain instances carry none of the flag's code.
// VOUT (with HSV, without LUMA / STORE): the front end of the edge term behind the default downscale -- ContentDetector with
// weights.delta_edges or a StatsManager, content_detector.py:155-174 on what scene_manager.py:666-678 hands it.  The HSV term as
// always, and of the resized frame only what cv2.Canny needs leaves the CU: its V plane (a third of the frame's bytes, through
// LDS as 16-byte stores) and its V histogram (the luma instances' machinery counting V: numpy.median for the thresholds).
#ifndef PSD_RS_HPART_TILE_MAJOR
#define PSD_RS_HPART_TILE_MAJOR 1   // partial histograms laid out [tile][frame]: a workgroup's walk writes ONE contiguous run (0: [frame][tile], rounds 4-6.4)
#endif
#ifndef PSD_RS_STORE_WAIT
#define PSD_RS_STORE_WAIT 1
#endif
#ifndef PSD_RS_VOUT_HIST
#define PSD_RS_VOUT_HIST 0     // 1: the VOUT instances count their V histogram themselves (rounds 6.2-6.4); 0: vplane_hist_kernel below
#endif
template <bool STORE, bool HSV, int G, bool LUMA = false, bool SEG = false, bool VOUT = false>
__global__ __launch_bounds__(RS_WG) void resize_walk_kernel(const RsParams p)
{
    static_assert(!VOUT || (HSV && !LUMA && !STORE), "VOUT rides on the HSV conversion");
    constexpr bool HIST = LUMA || (VOUT && PSD_RS_VOUT_HIST);
    extern __shared__ __attribute__((aligned(16))) uint8_t rs_stage[];
    __shared__ u32 lut_s[HSV ? 256 : 1], lut_h[HSV ? 256 : 1];
    // per-frame sums (sad_h, sad_s, sad_v, byte_sum), RS_REP copies each: a lane adds into copy lane % RS_REP
    __shared__ __attribute__((aligned(16))) u32 sums[RS_SLOTS][4][RS_REP];
    __shared__ __attribute__((aligned(16))) u32 lhist[HIST ? 2 : 1][HIST ? RS_HREP * 256 : 1];     // [slot][copy][bin]
    __shared__ int srows[64];           // clamped source rows of this tile: slot 2 * lr + k
    // STORE: the tile's resized pixels of frame t (whole destination rows: ONE contiguous run of the frame) collect here and leave
    // one barrier later as 16-byte stores, two slots taking turns: a third of the store instructions of three byte stores per pixel,
    // 1 % of the kernel's time.  What makes the storing instance 40 % slower than the one that converts to HSV and scores (1.40
    // against 1.02 ms per 4096 x 1080p -> 256 x 144; 0.96 ms with the stores taken out) is the WRITE TRAFFIC ITSELF: 453 MB in 295 k
    // runs of 1.5 KB between 6.8 GB of streaming reads cost 0.44 ms whatever the instruction (bytes, dwordx4) and cache policy (nt,
    // sc0, sc1, sc0 sc1: 1.34-1.41 ms) -- about 1 ms per GB written, where the 2 MB-per-frame V plane of the full-resolution edge
    // front end costs 0.08 (profiles/r06_u_resize_store_cost.txt).
    __shared__ __attribute__((aligned(16))) uint8_t obuf[(STORE || VOUT) ? 2 : 1][STORE ? RS_WG * G * 3 : VOUT ? RS_WG * G : 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x % p.n_tiles, chunk = blockIdx.x / p.n_tiles;
    const int r0 = tile * p.rows_per_tile;
    const int nrows = min(p.rows_per_tile, p.dh - r0);
    const int t0 = chunk * p.frames_per_chunk, t1 = min(p.n, t0 + p.frames_per_chunk);
    if (HSV) {
        for (int i = tid; i < 256; i += RS_WG) { lut_s[i] = p.lut[i]; lut_h[i] = p.lut[256 + i]; }
    }
    if (HSV || LUMA) {
        for (int i = tid; i < RS_SLOTS * 4 * RS_REP; i += RS_WG) (&sums[0][0][0])[i] = 0;
    }
    if (HIST) {
        for (int i = tid; i < 2 * RS_HREP * 256; i += RS_WG) (&lhist[0][0])[i] = 0;
    }
    if (tid < 2 * nrows) {
        const YTap y = p.yt[r0 + (tid >> 1)];
        srows[tid] = p.area2 ? 2 * (r0 + (tid >> 1)) + (tid & 1) : ((tid & 1) ? y.s1 : y.s0);
    }
    // this thread's destination pixels: p = g * 256 + tid over the tile's nrows * dw pixels.  A slot beyond the tile's pixels
    // (`dead`) reads the tile's first taps with ALL coefficients zero: its pixel is (0, 0, 0) in every frame, so it adds nothing
    // to the SADs and the byte sum, and its histogram increment is 0 -- no branch around it.
    RsPx px[G];
    bool live[G];
    u32 dst_off[G];                     // STORE: byte offset of the pixel inside a destination frame
    const int npx = nrows * p.dw;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int q = g * RS_WG + tid;
        live[g] = q < npx;
        const int lr = live[g] ? q / p.dw : 0, col = live[g] ? q - lr * p.dw : 0;
        dst_off[g] = (u32)(((r0 + lr) * p.dw + col) * 3);
        if (p.area2) {
            px[g].off = (u32)(2 * lr * p.row_pad + 6 * col);
            px[g].a4 = live[g] ? 0x00010001u : 0u;
            px[g].b80 = px[g].b81 = 0;
        } else {
            const XTap x = p.xt[col];
            const u32 b = (u32)p.yt[r0 + lr].b;
            px[g].off = (u32)(2 * lr * p.row_pad + x.o0);
            // (coefficients are in [0, 2048]: << 4 fits the 16-bit halves the dot product reads)
            px[g].a4 = live[g] ? (((u32)x.a & 0xffffu) << 4) | (((u32)x.a >> 16) << 20) : 0u;
            px[g].b80 = live[g] ? (b & 0xffffu) << 8 : 0u;
            px[g].b81 = live[g] ? (b >> 16) << 8 : 0u;
        }
    }
    __syncthreads();
    const int pieces = (p.row_bytes + 1023) >> 10;                // 1 KiB wave-instructions per source row
    const size_t buf_bytes = (size_t)2 * p.rows_per_tile * p.row_pad;
    // This wave's LDS-DMA instructions are the same every frame: piece q = wave + 4 i of the tile's 2 nrows source rows.
    // Where they read and land is worked out ONCE, on the scalar unit (the wave index is uniform): per frame an instruction
    // costs one 64-bit add for the lane's 16 bytes.  (The loop this replaces divided q by `pieces` and read srows[] for every
    // instruction of every frame -- more VALU work than the interpolation and the HSV conversion of the tile's pixels.)
    constexpr int NI = 8;                                          // instructions per wave kept in registers; more take the loop
    const int total = pieces * 2 * nrows;
    size_t goff[NI];
    u32 loff[NI];
    int rem[NI];                                                   // bytes of the row behind the piece's start (>= 1024: all lanes)
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int q = wave + i * RS_NW;
        const bool in = q < total;
        const int slot = in ? q / pieces : 0, piece = in ? q - slot * pieces : 0;
        const int row = __builtin_amdgcn_readfirstlane(srows[slot]);
        goff[i] = (size_t)row * p.srow + (size_t)piece * 1024;
        loff[i] = (u32)slot * (u32)p.row_pad + (u32)piece * 1024u;
        rem[i] = in ? p.row_bytes - piece * 1024 : 0;
    }
    auto issue = [&](const uint8_t* frame, int buf) {
        uint8_t* base = rs_stage + (size_t)buf * buf_bytes;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            if (rem[i] >= 1024) {
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(frame + goff[i] + lane * 16), (lds_ptr_t)(base + loff[i]), 16, 0, PSD_DMA_AUX);
            } else if (rem[i] > 0) {
                if (lane * 16 + 16 <= rem[i])
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(frame + goff[i] + lane * 16), (lds_ptr_t)(base + loff[i]), 16, 0, PSD_DMA_AUX);
            }
        }
        for (int q = wave + NI * RS_NW; q < total; q += RS_NW) {
            const int slot = q / pieces, piece = q - slot * pieces;
            const int off = piece * 1024 + lane * 16;
            if (off + 16 <= p.row_bytes)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(frame + (size_t)srows[slot] * p.srow + off),
                                                 (lds_ptr_t)(base + (size_t)slot * p.row_pad + piece * 1024), 16, 0, PSD_DMA_AUX);
        }
    };
    auto frame_ptr = [&](int t) { return t < 0 ? p.prev : p.src + (size_t)t * p.sstride; };
    // the frame in front of the chunk only feeds the HSV carry
    const bool halo = HSV && (t0 > 0 || p.prev != nullptr);
    const int tb = halo ? t0 - 1 : t0;
    u32 ph[HSV ? G : 1], ps[HSV ? G : 1], pv[HSV ? G : 1];         // the previous frame's H, S, V of this thread's pixels
#pragma unroll
    for (int g = 0; g < (HSV ? G : 1); g++) ph[g] = ps[g] = pv[g] = 0;
    bool have_prev = false;
    // this wave's LDS-DMA instructions per frame (the same every frame): with three buffers the rows of frame t+1 may
    // still be in flight when frame t is taken (DMA completes in order, so "at most that many outstanding" means
    // frame t has landed)
    const int mine = (pieces * 2 * nrows - wave + RS_NW - 1) / RS_NW;
    const int ahead = p.depth - 1;             // frames in flight beyond the current one
    const u32 rep = (u32)(tid & (RS_REP - 1));
    if (tb < t1) issue(frame_ptr(tb), 0);
    if (ahead > 1 && tb + 1 < t1) issue(frame_ptr(tb + 1), 1);
    // Clip-start flag of the NEXT frame to be stepped.  It is requested behind that frame's DMA issue and turned into an SGPR at
    // the top of the frame's own step, right behind the wait that has covered it: hipcc waits for a loaded register where it is
    // first read and knows nothing of the kernel's own s_waitcnt, so a flag read behind the issue of the frame AFTER would put an
    // s_waitcnt vmcnt(0) there and drain the prefetch the wave has just started (psd_score_kernels.hip, round 5).
    u32 seg_next = 0;
    if constexpr (SEG) {
        if (tb < t1 && tb >= 0) seg_next = p.seg[tb];
    }
    // 16-byte store instructions of this wave per store_tile (uniform: thread tid takes the pieces tid, tid + 256, ...)
    const int n16 = (npx * ((STORE || VOUT) ? (VOUT ? 1 : 3) : 0)) >> 4;
    const int my_stores = n16 > 64 * wave ? (n16 - 64 * wave + RS_WG - 1) / RS_WG : 0;
    auto store_tile = [&](int t, int slot) {
        const uint4* from = reinterpret_cast<const uint4*>(obuf[slot]);
        constexpr int BPP = VOUT ? 1 : 3;          // bytes per pixel that leave: the V plane or the BGR frame
        uint4* to = reinterpret_cast<uint4*>(VOUT ? p.vout + ((size_t)t * p.dh + r0) * p.dw : p.dst + (size_t)t * p.dstride + (size_t)r0 * p.dw * 3);
        for (int i = tid; i < (npx * BPP) >> 4; i += RS_WG) to[i] = from[i];
    };
    auto frame_step = [&](const int t, const auto& ph, const auto& ps, const auto& pv, auto& nh, auto& ns, auto& nv) {
        const int step = t - tb;
        const int buf = ahead > 1 ? step % 3 : (step & 1);
        if (ahead > 1 && t + 1 < t1) {
            if (mine >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (mine >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (mine >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (mine >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (mine >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (mine >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (mine >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#if PSD_RS_STORE_WAIT
            // STORE / VOUT: the 16-byte stores of frame t-2's pixels went out in the step before this one BEHIND the DMA of frame t, and
            // vector memory operations complete in order: "at most my_stores outstanding" means frame t has landed, and the wave does
            // not sit out the write acknowledgements (round 6, fifth session)
            // (HIST: rs_hist_flush's stores of frame t-2's partial histogram as well -- wave 0 issues at least two, `bytes` and `over`)
            const int hist_stores = !HIST ? 0 : RS_HPACK8 ? (wave == 0 ? 2 : 0) : (wave < 2 ? 1 : 0);
            const int younger = t - t0 >= 2 ? (((STORE || VOUT) && p.store_vec) ? my_stores : 0) + hist_stores : 0;
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (HSV || LUMA || STORE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's hidden LDS increments / pixel bytes of frame t-1
        u32 flag_now = 0;
        if constexpr (SEG) {
            flag_now = __builtin_amdgcn_readfirstlane(seg_next);
            asm volatile("" ::"s"(flag_now) : "memory");
        }
        lds_barrier();                         // every wave's rows of frame t have landed; frame t-1 is fully consumed (LDS only: no vmcnt drain)
        if (t + ahead < t1) issue(frame_ptr(t + ahead), ahead > 1 ? (step + 2) % 3 : (buf ^ 1));
        if constexpr (SEG) {
            if (t + 1 < t1) seg_next = p.seg[t + 1];   // (t + 1 >= 0: the halo frame in front of the batch is t = -1 at the earliest)
        }
        const int rel = t - t0;                // chunk-relative frame index (-1 for the halo frame)
        if ((HSV || LUMA) && rel >= 8 && (rel & 7) == 0 && tid < 32) {
            // frames rel-8 .. rel-1 are complete (all waves added them before this frame's barrier)
            const int f = rel - 8 + (tid >> 2), q = tid & 3;
            u32* cell = sums[f & (RS_SLOTS - 1)][q];
            u32 v = 0;
#pragma unroll
            for (int k = 0; k < RS_REP; k += 4) {
                const uint4 w = *reinterpret_cast<const uint4*>(cell + k);
                v += w.x + w.y + w.z + w.w;
            }
            // (cleared with stores the compiler does not see: a visible LDS store would wait for the staging issued a few lines up)
#pragma unroll
            for (int k = 0; k < RS_REP; k += 2) asm volatile("ds_write_b64 %0, %1" ::"v"((u32)(uintptr_t)(cell + k)), "v"(0ull) : "memory");
            psd_frame_scores* rec = p.out + t0 + f;
            if (v) atomicAdd(q == 0 ? (unsigned long long*)&rec->sad_h : q == 1 ? (unsigned long long*)&rec->sad_s
                             : q == 2 ? (unsigned long long*)&rec->sad_v : (unsigned long long*)&rec->byte_sum, (unsigned long long)v);
        }
        if (HIST && rel >= 1) rs_hist_flush<true>(lhist[(rel - 1) & 1], p.hpart + (PSD_RS_HPART_TILE_MAJOR ? (size_t)tile * p.n + (t - 1) : (size_t)(t - 1) * p.n_tiles + tile) * p.hstride, tid);
        if ((STORE || VOUT) && rel >= 1) {
            if (p.store_vec) store_tile(t - 1, (rel - 1) & 1);       // frame t-1's pixels are complete (every wave wrote them before this barrier)
        }
        const uint8_t* base = rs_stage + (size_t)buf * buf_bytes;
        u32 sh = 0, ss = 0, sv = 0, bsum = 0;
        const bool count_luma = (HIST || VOUT) && t >= t0;   // (the halo frame only feeds the HSV carry; VOUT: neither is its V plane stored)
        u32* hcur = lhist[HIST ? (rel & 1) : 0] + (HIST ? (tid & (RS_HREP - 1)) * 256 : 0);
        const bool chain = have_prev && !(SEG && flag_now != 0);
        constexpr int C = G < 4 ? G : 4;        // pixels whose taps are in flight together (six dwords each)
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += C) {
            RsTaps taps[C];
            u32 addr[C];
#pragma unroll
            for (int i = 0; i < C; i++) addr[i] = ((u32)(uintptr_t)base + px[g0 + i].off) & ~3u;
            rs_taps_load<C>(addr, (u32)p.row_pad, taps);
#pragma unroll
            for (int i = 0; i < C; i++) {
                const int g = g0 + i;
                u32 c[3];
                // (the staging buffers start 16-byte aligned and are multiples of 1024 bytes long: the tap's byte shift is off & 3)
                if (p.area2) rs_interp<true>(taps[i], px[g].off, px[g], c);
                else rs_interp<false>(taps[i], px[g].off, px[g], c);
                if (STORE && t >= t0 && live[g]) {
                    if (p.store_vec) {
                        // (stores the compiler does not see, like rs_lds_add: a visible LDS store waits for the next frame's staging)
                        const u32 o = (u32)(uintptr_t)&obuf[rel & 1][(g * RS_WG + tid) * 3];
                        asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1\n\tds_write_b8 %0, %3 offset:2"
                                     ::"v"(o), "v"(c[0]), "v"(c[1]), "v"(c[2]) : "memory");
                    } else {
                        uint8_t* d = p.dst + (size_t)t * p.dstride + dst_off[g];
                        d[0] = (uint8_t)c[0]; d[1] = (uint8_t)c[1]; d[2] = (uint8_t)c[2];
                    }
                }
                if (LUMA) {
                    if (count_luma) {
                        // BT.601 luma in OpenCV's 14-bit fixed point (color_yuv.simd.hpp: 1868 B + 9617 G + 4899 R, + 8192 >> 14)
                        const u32 y = (c[0] * 1868u + c[1] * 9617u + c[2] * 4899u + 8192u) >> 14;
                        rs_lds_add(&hcur[y], live[g] ? 1u : 0u);
                        bsum += c[0] + c[1] + c[2];
                    }
                }
                if (HSV) {
                    u32 h, s, v;
                    rs_hsv(c[0], c[1], c[2], lut_s, lut_h, h, s, v);
                    if (chain) {
                        sh = __builtin_amdgcn_sad_u16(h, ph[g], sh);      // |h - ph| + sh on the 16-bit halves (the upper ones are 0)
                        ss = __builtin_amdgcn_sad_u16(s, ps[g], ss);
                        sv = __builtin_amdgcn_sad_u16(v, pv[g], sv);
                    }
                    nh[g] = h; ns[g] = s; nv[g] = v;
                    if (VOUT) {
                        if (count_luma) {
                            if (PSD_RS_VOUT_HIST) rs_lds_add(&hcur[v], live[g] ? 1u : 0u);
                            if (live[g]) {
                                if (p.store_vec) {
                                    asm volatile("ds_write_b8 %0, %1" ::"v"((u32)(uintptr_t)&obuf[rel & 1][g * RS_WG + tid]), "v"(v) : "memory");
                                } else {
                                    p.vout[((size_t)t * p.dh + r0) * p.dw + g * RS_WG + tid] = (uint8_t)v;
                                }
                            }
                        }
                    }
                }
            }
        }
        u32* s4 = &sums[rel & (RS_SLOTS - 1)][0][0] + rep;
        if (HSV) {
            if (chain) {
                rs_lds_add(s4, sh); rs_lds_add(s4 + RS_REP, ss); rs_lds_add(s4 + 2 * RS_REP, sv);
            }
            have_prev = true;
        }
        if (LUMA) {
            if (count_luma) rs_lds_add(s4 + 3 * RS_REP, bsum);
        }
    };
#if PSD_RS_SWAP
    // two steps per trip with the two H, S, V register sets swapped: "previous = current" is a renaming instead of 3 G v_mov_b32 per
    // frame (hipcc does not rotate registers across the back edge of a loop that holds inline asm and a barrier; psd_score_kernels.hip,
    // round 3)
    u32 qh[HSV ? G : 1], qs[HSV ? G : 1], qv[HSV ? G : 1];
#pragma unroll
    for (int g = 0; g < (HSV ? G : 1); g++) qh[g] = qs[g] = qv[g] = 0;
    for (int t = tb; t < t1; t += 2) {
        frame_step(t, ph, ps, pv, qh, qs, qv);
        if (t + 1 < t1) frame_step(t + 1, qh, qs, qv, ph, ps, pv);
    }
#else
    for (int t = tb; t < t1; t++) frame_step(t, ph, ps, pv, ph, ps, pv);
#endif
    if (STORE || VOUT) {
        if (p.store_vec && t1 > t0) {                  // the chunk's last frame
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            store_tile(t1 - 1, (t1 - 1 - t0) & 1);
        }
    }
    if (HSV || LUMA) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        // the periodic flush took frames [0, 8 * floor((done - 1) / 8)); at most 8 are left
        const int done = t1 - t0;
        const int f = (done > 0 ? ((done - 1) & ~7) : 0) + (tid >> 2), q = tid & 3;
        if (tid < 32 && f < done) {
            const u32* cell = sums[f & (RS_SLOTS - 1)][q];
            u32 v = 0;
#pragma unroll
            for (int k = 0; k < RS_REP; k++) v += cell[k];
            psd_frame_scores* rec = p.out + t0 + f;
            if (v) atomicAdd(q == 0 ? (unsigned long long*)&rec->sad_h : q == 1 ? (unsigned long long*)&rec->sad_s
                             : q == 2 ? (unsigned long long*)&rec->sad_v : (unsigned long long*)&rec->byte_sum, (unsigned long long)v);
        }
        if (HIST && done > 0) rs_hist_flush<false>(lhist[(done - 1) & 1], p.hpart + (PSD_RS_HPART_TILE_MAJOR ? (size_t)tile * p.n + (t1 - 1) : (size_t)(t1 - 1) * p.n_tiles + tile) * p.hstride, tid);   // the chunk's last frame
    }
}

// The V histograms of the VOUT instances' planes (numpy.median for Canny's thresholds, content_detector.py:229-233), round 6, fifth session.
// Until then the VOUT instance counted V with the luma instances' machinery -- an LDS increment per destination pixel, a 256-bin flush
// per tile and frame, hist_reduce_kernel behind it: 1.385 ms per 4096 x 1080p -> 256 x 144 against 1.03 ms for the instance that only
// scores (profiles/r06_aq_*).  The plane it writes is small (36 KB per frame at the default downscale, 1 / 170 of the source bytes), so
// reading it once more costs next to nothing (151 MB per 4096 frames, 37 us: profiles/r06_bb_pmc_*): one workgroup per frame (`parts` of them for large
// planes: global atomics then, the histogram zeroed by the caller), sixteen bytes per lane and step, a 16-fold replicated LDS
// histogram laid out [bin][copy] (lanes that meet in a bin -- flat regions -- land on sixteen banks), plain stores of the 256 sums.
// -DPSD_RS_VOUT_HIST=1 restores the count inside the downscale kernel (A/B).
__global__ __launch_bounds__(1024) void vplane_hist_kernel(const uint8_t* v, size_t npix, u32* hist, int parts)
{
    __shared__ __attribute__((aligned(16))) u32 h[256][16];
    const int tid = threadIdx.x, frame = blockIdx.x / parts, part = blockIdx.x - frame * parts;
    for (int i = tid; i < 256 * 16; i += 1024) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* src = v + (size_t)frame * npix;
    u32* mine = &h[0][tid & 15];
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const size_t chunks = npix / 16, c0 = chunks * part / parts, c1 = chunks * (part + 1) / parts;
        for (size_t c = c0 + tid; c < c1; c += 1024) {
            const uint4 w = reinterpret_cast<const uint4*>(src)[c];
            const u32 d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(mine + 16 * (d[k] & 255u), 1u); atomicAdd(mine + 16 * ((d[k] >> 8) & 255u), 1u);
                atomicAdd(mine + 16 * ((d[k] >> 16) & 255u), 1u); atomicAdd(mine + 16 * (d[k] >> 24), 1u);
            }
        }
        if (part == parts - 1 && (size_t)tid < npix - chunks * 16) atomicAdd(mine + 16 * src[chunks * 16 + tid], 1u);
    } else {
        // (a plane that does not start on 16 bytes: odd-sized frames behind the first; byte loads)
        const size_t b0 = npix * part / parts, b1 = npix * (part + 1) / parts;
        for (size_t b = b0 + tid; b < b1; b += 1024) atomicAdd(mine + 16 * src[b], 1u);
    }
    __syncthreads();
    if (tid < 256) {
        const uint4* q = reinterpret_cast<const uint4*>(&h[tid][0]);
        u32 sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint4 w = q[k]; sum += w.x + w.y + w.z + w.w; }
        if (parts == 1) hist[(size_t)frame * 256 + tid] = sum;
        else if (sum) atomicAdd(&hist[(size_t)frame * 256 + tid], sum);
    }
}

// rec[t].hist = sum over the tiles of frame t of their partial histograms (rs_hist_flush's layout).  grid = frames; W threads per group
// (W = 64: one per word of four low bytes; 128 in the 16-bit layout) or a multiple: the tiles are split over blockDim / W groups of threads
// and joined in LDS.  out_words + t * stride_words = the 256 bins of frame t (a record's `hist`, or a row of the edge term's V histograms).
// (tile-major partials, fifth session of round 6: frame t's tile k lies at hpart + (k * n_frames + t_base + t) * hstride -- the downscale kernel's
//  workgroups each write one contiguous run over their walk; these reads are the scattered side now, 272 bytes each, by a kernel that is not in a hurry)
__global__ __launch_bounds__(512) void hist_reduce_kernel(const u32* hpart, int n_tiles, int hstride, u32* out_words, size_t stride_words, int n_frames,
                                                          int t_base)
{
    constexpr int W = RS_HPACK8 ? 64 : 128, B = 256 / W;        // threads per group, bins per thread
    __shared__ u32 part[512 / W - 1][256];
    const int t = blockIdx.x, i = threadIdx.x % W, q = threadIdx.x / W, nq = blockDim.x / W;
    const u32* src = PSD_RS_HPART_TILE_MAJOR ? hpart + (size_t)(t_base + t) * hstride : hpart + (size_t)(t_base + t) * n_tiles * hstride;
    const size_t tile_step = PSD_RS_HPART_TILE_MAJOR ? (size_t)n_frames * hstride : (size_t)hstride;
    u32 c[B];
#pragma unroll
    for (int b = 0; b < B; b++) c[b] = 0;
    for (int k = q; k < n_tiles; k += nq) {
        const u32* tp = src + (size_t)k * tile_step;
        const u32 v = tp[i];
        if constexpr (RS_HPACK8) {
            c[0] += v & 255u; c[1] += (v >> 8) & 255u; c[2] += (v >> 16) & 255u; c[3] += v >> 24;
            const unsigned long long over = (unsigned long long)tp[64] | ((unsigned long long)tp[65] << 32);
            if ((over >> i) & 1ull) {
                const u32 carry = tp[66 + __popcll(over & ((1ull << i) - 1ull))];
                c[0] += (carry & 15u) << 8; c[1] += ((carry >> 4) & 15u) << 8; c[2] += ((carry >> 8) & 15u) << 8; c[3] += (carry >> 12) << 8;
            }
        } else {
            c[0] += v & 0xffffu; c[1] += v >> 16;
        }
    }
    if (q) {
#pragma unroll
        for (int b = 0; b < B; b++) part[q - 1][B * i + b] = c[b];
    }
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            for (int r = 0; r + 1 < nq; r++) c[b] += part[r][B * i + b];
            out_words[(size_t)t * stride_words + B * i + b] = c[b];
        }
    }
}

// Generic cv2.resize(INTER_LINEAR): any alignment / row stride.  grid = (ceil(dw/256), dh, n)
__global__ __launch_bounds__(256) void resize_linear_generic_kernel(const uint8_t* src, int sh, int sw, size_t srow, size_t sstride,
                                                                    uint8_t* dst, int dh, int dw, size_t dstride, const XTap* xt,
                                                                    const YTap* yt, int area2)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    const uint8_t* S = src + (size_t)blockIdx.z * sstride;
    uint8_t* D = dst + (size_t)blockIdx.z * dstride + ((size_t)dy * dw + dx) * 3;
    if (area2) {
        const uint8_t* q = S + (size_t)(2 * dy) * srow + (size_t)(2 * dx) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) D[c] = (uint8_t)((q[c] + q[3 + c] + q[srow + c] + q[srow + 3 + c] + 2) >> 2);
        return;
    }
    const XTap x = xt[dx];
    const YTap y = yt[dy];
    const uint8_t* ra = S + (size_t)y.s0 * srow;
    const uint8_t* rb = S + (size_t)y.s1 * srow;
    const int a0 = (short)(x.a & 0xffff), a1 = x.a >> 16, b0 = (short)(y.b & 0xffff), b1 = y.b >> 16;
#pragma unroll
    for (int c = 0; c < 3; c++) D[c] = (uint8_t)interp(ra[x.o0 + c], ra[x.o1 + c], rb[x.o0 + c], rb[x.o1 + c], a0, a1, b0, b1);
}

template <bool STORE, bool HSV, bool LUMA, bool SEG, bool VOUT>
static const void* walk_fn_g(int g)
{
    switch (g) {
    case 1: return reinterpret_cast<const void*>(&resize_walk_kernel<STORE, HSV, 1, LUMA, SEG, VOUT>);
    case 2: return reinterpret_cast<const void*>(&resize_walk_kernel<STORE, HSV, 2, LUMA, SEG, VOUT>);
    case 4: return reinterpret_cast<const void*>(&resize_walk_kernel<STORE, HSV, 4, LUMA, SEG, VOUT>);
    default: return reinterpret_cast<const void*>(&resize_walk_kernel<STORE, HSV, 8, LUMA, SEG, VOUT>);
    }
}

template <bool STORE, bool HSV, bool LUMA = false, bool VOUT = false>
static const void* walk_fn(bool seg, int g)
{
    // (the clip-start flags only matter to the HSV carry: instances without the HSV term have no SEG form)
    if constexpr (HSV) {
        if (seg) return walk_fn_g<STORE, HSV, LUMA, true, VOUT>(g);
    }
    return walk_fn_g<STORE, HSV, LUMA, false, VOUT>(g);
}

// Resident workgroups per CU of one instance with `lds` bytes of staging (the runtime's own occupancy figure: registers, LDS
// -- static + dynamic -- and wave slots), asked once per (instance, lds).
static int walk_blocks_per_cu(const void* fn, size_t lds)
{
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, int> seen;
    std::lock_guard<std::mutex> lk(mu);
    auto it = seen.find({fn, lds});
    if (it != seen.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, RS_WG, lds) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
    seen[{fn, lds}] = nb;
    return nb;
}

// cv2.resize(INTER_LINEAR) of n frames and / or the HSV term of the resized frames.
//   d_dst   resized frames out (packed rows, dst_frame_stride apart) or null
//   d_out   zero-initialised records to add the terms of the resized frames into, or null
//   terms   with d_out: PSD_SCORE_HSV_SAD (sad_h / sad_s / sad_v) and / or PSD_SCORE_LUMA_HIST | PSD_SCORE_BYTE_SUM (hist and
//           byte_sum, always both); the luma terms only without d_dst
//   d_prev  source-size frame preceding frame 0 (HSV carry), or null
//   d_vout / d_vhist (both or neither; with the HSV term alone, without d_dst): the resized frames' V planes (dst_h * dst_w bytes
//           per frame, packed) and V histograms (256 words per frame) for the edge term -- the VOUT instances
static int resize_linear_impl(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_row_stride,
                              size_t src_frame_stride, const uint8_t* d_prev, uint8_t* d_dst, int dst_h, int dst_w,
                              size_t dst_frame_stride, psd_frame_scores* d_out, hipStream_t stream, int* launches, const uint8_t* d_seg,
                              bool area_mode, uint32_t terms, uint8_t* d_vout, u32* d_vhist)
{
    const bool want_hsv = d_out && (terms & PSD_SCORE_HSV_SAD), want_luma = d_out && (terms & (PSD_SCORE_LUMA_HIST | PSD_SCORE_BYTE_SUM));
    const bool want_v = d_vout != nullptr;
    if (want_v && (!d_vhist || !want_hsv || want_luma || d_dst)) { psd_set_error("fused downscale: the V plane rides on the HSV term alone"); return PSD_ERR_INVALID; }
    if (d_out && !want_hsv && !want_luma) return PSD_OK;
    if (want_luma && d_dst) { psd_set_error("fused downscale: the luma terms do not combine with storing the resized frames"); return PSD_ERR_UNSUPPORTED; }
    if (n == 0 || (!d_dst && !d_out)) return PSD_OK;
    ResizeTabs tabs;
    int rc = get_tabs(e, src_h, src_w, dst_h, dst_w, stream, &tabs, area_mode);
    if (rc != PSD_OK) return rc;
    const int row_bytes = src_w * 3;
    const bool fast = (row_bytes % 16 == 0) && (src_row_stride % 16 == 0) && ((uintptr_t)d_src % 16 == 0) &&
                      (src_frame_stride % 16 == 0 || n == 1) && (!d_prev || (uintptr_t)d_prev % 16 == 0) && row_bytes <= 16384;
    if (!fast) {
        if (d_out) {
            psd_set_error("fused downscale + score needs 16-byte aligned packed source rows");
            return PSD_ERR_UNSUPPORTED;   // the caller resizes first and scores the small frames
        }
        hipLaunchKernelGGL(resize_linear_generic_kernel, dim3((dst_w + 255) / 256, dst_h, n), dim3(256), 0, stream, d_src, src_h, src_w,
                           src_row_stride, src_frame_stride, d_dst, dst_h, dst_w, dst_frame_stride, tabs.x, tabs.y, tabs.area2 ? 1 : 0);
        HIP_TRY(hipGetLastError());
        if (launches) *launches += 1;
        return PSD_OK;
    }
    RsParams p{};
    p.src = d_src; p.prev = d_prev; p.seg = d_seg; p.sstride = src_frame_stride; p.srow = src_row_stride;
    p.sh = src_h; p.sw = src_w; p.row_bytes = row_bytes;
    p.dst = d_dst; p.dstride = dst_frame_stride; p.dh = dst_h; p.dw = dst_w;
    p.out = d_out; p.lut = engine_lut(e); p.xt = tabs.x; p.yt = tabs.y; p.n = n; p.area2 = tabs.area2 ? 1 : 0;
    p.row_pad = (row_bytes + 1023) & ~1023;            // whole 1 KiB pieces (the tail lanes of the last piece stay idle)
    // staging depth: 2 buffers (one frame ahead).  PSD_RESIZE_DEPTH=3 keeps two frames in flight per workgroup; measured on
    // 4096 x 1080p -> 256x144 (kernel ms; rows per tile / depth): 2/2 1.29, 1/2 1.36, 3/2 1.44, 1/3 1.57, 2/3 1.75 -- the
    // extra LDS costs more residency than the deeper prefetch buys
    static const int depth_env = [] { const char* v = getenv("PSD_RESIZE_DEPTH"); return v ? atoi(v) : 0; }();
    p.depth = depth_env == 3 ? 3 : 2;
    // rows per tile: `depth` staging buffers of 2 R padded rows within ~24 KiB each, at most 8 pixels per thread
    int R = (int)((24u << 10) / (2u * (unsigned)p.row_pad));
    static const int rows_env = [] { const char* v = getenv("PSD_RESIZE_ROWS"); return v ? atoi(v) : 0; }();
    if (rows_env > 0) R = rows_env;   // experiments
    if (R < 1) R = 1;
    if (R > 16) R = 16;
    while (R > 1 && (R * dst_w + RS_WG - 1) / RS_WG > 8) R--;
    if (R > dst_h) R = dst_h;
    if (rows_env <= 0) {
        // pixel slots per thread come in powers of two (the G instances): six pixels per thread -- 640 x 360 -> 256 x 144, R = 6 --
        // ran the 8-slot instance with a quarter of its slots dead.  Take the largest R' in (R / 2, R] whose tile fills its slots.
        for (int r = R; 2 * r > R; r--) {
            const int px = r * dst_w;
            const int per = px / RS_WG;
            if (px % RS_WG == 0 && per >= 1 && per <= 8 && (per & (per - 1)) == 0) { R = r; break; }
        }
    }
    const int per_thread = (R * dst_w + RS_WG - 1) / RS_WG;
    if (per_thread > 8) {   // destination rows wider than 2048 pixels: not a downscale for scoring; plain kernel
        if (d_out) { psd_set_error("destination rows of %d pixels are too wide for the fused downscale + score", dst_w); return PSD_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL(resize_linear_generic_kernel, dim3((dst_w + 255) / 256, dst_h, n), dim3(256), 0, stream, d_src, src_h, src_w,
                           src_row_stride, src_frame_stride, d_dst, dst_h, dst_w, dst_frame_stride, tabs.x, tabs.y, tabs.area2 ? 1 : 0);
        HIP_TRY(hipGetLastError());
        if (launches) *launches += 1;
        return PSD_OK;
    }
    const int g = per_thread <= 1 ? 1 : per_thread <= 2 ? 2 : per_thread <= 4 ? 4 : 8;
    // (every tile is whole rows, so it is one run of the frame; 16-byte stores when every such run starts and ends on 16 bytes)
    static const int store_env = [] { const char* v = getenv("PSD_RESIZE_STORE_VEC"); return v ? atoi(v) : 1; }();
    p.store_vec = (d_dst && store_env && (dst_w * 3) % 16 == 0 && (uintptr_t)d_dst % 16 == 0 && (dst_frame_stride % 16 == 0 || n == 1)) ? 1 : 0;
    if (want_v) p.store_vec = (store_env && dst_w % 16 == 0 && (uintptr_t)d_vout % 16 == 0 && ((size_t)dst_h * dst_w % 16 == 0 || n == 1)) ? 1 : 0;
    p.vout = d_vout;
    p.rows_per_tile = R;
    p.n_tiles = (dst_h + R - 1) / R;
    const size_t lds = (size_t)p.depth * 2 * R * p.row_pad + 16;      // (+ 16: a tap read of 8 bytes at the end of the last staged row)
    const bool seg = d_seg != nullptr;
    const void* fn = want_v ? walk_fn<false, true, false, true>(seg, g) : (d_dst && want_hsv) ? walk_fn<true, true>(seg, g) : (want_hsv && want_luma) ? walk_fn<false, true, true>(seg, g)
                     : want_luma ? walk_fn<false, false, true>(seg, g) : want_hsv ? walk_fn<false, true>(seg, g) : walk_fn<true, false>(seg, g);
    // Time chunks.  The workgroups of a launch all take about the same time, so it runs in ROUNDS of as many workgroups as the
    // chip holds at once (CUs x the instance's occupancy), and a last round that is 2 % full takes as long as a full one.  Until
    // round 6 the count was "12 workgroups per CU" rounded UP to whole chunks per tile: 3096 workgroups for 3072 slots at 640 x
    // 360 (4 per CU, 3.02 rounds) and 6.05 rounds at 1080p (2 per CU) -- a nearly empty round at the end of every launch.  Now:
    // whole rounds, rounded DOWN to whole chunks per tile; walks long enough that the halo frame stays small.
    const int capacity = engine_num_cus(e) * walk_blocks_per_cu(fn, lds);
    static const int rounds_env = [] { const char* v = getenv("PSD_RESIZE_ROUNDS"); return v ? atoi(v) : -1; }();   // experiments; 0 = the old rule
    int chunks;
    if (rounds_env == 0) {
        const int target = engine_num_cus(e) * 12;
        chunks = (target + p.n_tiles - 1) / p.n_tiles;
    } else {
        // six rounds: 640 x 360 (4 workgroups per CU) 0.64-0.66 of peak under the old rule, 0.69-0.70 with 3 whole rounds, 0.71 with
        // 6, 0.72 with 8; 1080p (2 per CU) 0.79 / 0.80 / 0.82 / 0.80-0.81 (profiles/r06_r_ab_resize_whole_rounds.txt)
        const int rounds = rounds_env > 0 ? rounds_env : 6;
        chunks = (int)(((long)rounds * capacity) / p.n_tiles);
    }
    if (want_hsv) { const int by_walk = (n + 31) / 32; if (chunks > by_walk) chunks = by_walk; }
    if (chunks > n) chunks = n;
    if (chunks < 1) chunks = 1;
    p.frames_per_chunk = (n + chunks - 1) / chunks;
    chunks = (n + p.frames_per_chunk - 1) / p.frames_per_chunk;
    const int grid = p.n_tiles * chunks;
    if (want_hsv) note_walk_geometry(p.frames_per_chunk, p.n_tiles);
    const bool tile_hist = want_luma || (want_v && PSD_RS_VOUT_HIST);
    if (tile_hist) {
        void* scratch = nullptr;
        p.hstride = rs_hist_stride_words(R * dst_w);
        rc = engine_hist_scratch(e, (size_t)n * p.n_tiles * p.hstride * sizeof(u32), stream, &scratch);
        if (rc != PSD_OK) return rc;
        p.hpart = static_cast<u32*>(scratch);
    }
    {
        void* args[] = {&p};
        HIP_TRY(hipLaunchKernel(fn, dim3(grid), dim3(RS_WG), args, lds, stream));
    }
    HIP_TRY(hipGetLastError());
    if (launches) *launches += 1;
    if (want_v && !PSD_RS_VOUT_HIST) {
        // the planes' histograms from the planes (vplane_hist_kernel): one workgroup per frame up to 256 K pixels, more above
        const size_t npix = (size_t)dst_h * dst_w;
        const int parts = (int)std::min<size_t>(64, (npix + 262143) / 262144);
        if (parts > 1) HIP_TRY(hipMemsetAsync(d_vhist, 0, (size_t)n * 256 * sizeof(u32), stream));
        for (int t0 = 0; t0 < n; t0 += 32768)
            hipLaunchKernelGGL(vplane_hist_kernel, dim3(std::min(32768, n - t0) * parts), dim3(1024), 0, stream, d_vout + (size_t)t0 * npix, npix,
                               d_vhist + (size_t)t0 * 256, parts);
        HIP_TRY(hipGetLastError());
        if (launches) *launches += 1;
    }
    if (tile_hist) {
        // (one quarter-less pass of 128 threads per frame: in the pipelined flow the small workgroups slip in beside the next
        //  submission's kernels -- 1.29-1.30 ms per 4096 frames against 1.31-1.32 with 512 threads, which are faster alone)
        u32* words = want_v ? d_vhist : reinterpret_cast<u32*>(&d_out[0].hist[0]);
        const size_t stride = want_v ? 256 : sizeof(psd_frame_scores) / sizeof(u32);
        for (int t0 = 0; t0 < n; t0 += 65535)
            hipLaunchKernelGGL(hist_reduce_kernel, dim3(std::min(65535, n - t0)), dim3(128), 0, stream, p.hpart, p.n_tiles, p.hstride,
                               words + (size_t)t0 * stride, stride, n, t0);
        HIP_TRY(hipGetLastError());
        if (launches) *launches += 1;
    }
    return PSD_OK;
}

int resize_linear_score(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_row_stride,
                        size_t src_frame_stride, const uint8_t* d_prev, uint8_t* d_dst, int dst_h, int dst_w,
                        size_t dst_frame_stride, psd_frame_scores* d_out, hipStream_t stream, int* launches, const uint8_t* d_seg,
                        bool area_mode, uint32_t terms)
{
    return resize_linear_impl(e, d_src, n, src_h, src_w, src_row_stride, src_frame_stride, d_prev, d_dst, dst_h, dst_w, dst_frame_stride, d_out, stream,
                              launches, d_seg, area_mode, terms, nullptr, nullptr);
}

// cv2.resize(INTER_LINEAR) + the HSV term + the resized frames' V planes and V histograms, nothing else in memory: the front end of
// the edge term behind the default downscale (psd_edge_kernels.hip).  PSD_ERR_UNSUPPORTED where the fused kernel does not apply
// (source rows not 16-byte aligned, very wide targets): the caller resizes into a buffer instead.
int resize_linear_score_vplane(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_frame_stride, const uint8_t* d_prev,
                               int dst_h, int dst_w, psd_frame_scores* d_out, hipStream_t stream, int* launches, const uint8_t* d_seg,
                               uint8_t* d_vout, uint32_t* d_vhist)
{
    return resize_linear_impl(e, d_src, n, src_h, src_w, (size_t)src_w * 3, src_frame_stride, d_prev, nullptr, dst_h, dst_w, 0, d_out, stream, launches,
                              d_seg, false, PSD_SCORE_HSV_SAD, d_vout, d_vhist);
}

// whether resize_linear_score_vplane would take the fused kernel for this shape (the edge term decides its route up front)
bool resize_vplane_available(const uint8_t* d_src, int src_w, size_t src_frame_stride, const uint8_t* d_prev, int dst_w, int n)
{
    const int row_bytes = src_w * 3;
    return row_bytes % 16 == 0 && (uintptr_t)d_src % 16 == 0 && (src_frame_stride % 16 == 0 || n == 1) && (!d_prev || (uintptr_t)d_prev % 16 == 0) &&
           row_bytes <= 16384 && dst_w <= 2048;
}

// ---- cv2.resize(INTER_NEAREST) and cv2.resize(INTER_AREA), 8-bit, 3 channels ---------------------------------------
// The other two `Interpolation` modes SceneManager can be asked to downscale with (reference common.py:148-160,
// scene_manager.py:670-678).  One destination pixel per thread, 256-thread workgroups over the flattened small frame;
// offsets / run tables per (src, dst) shape come from the engine's table cache like the bilinear taps.

constexpr int kOtherWG = 256;

__global__ __launch_bounds__(kOtherWG) void resize_nearest_kernel(const uint8_t* src, int sw, size_t sstride, uint8_t* dst, int dh, int dw,
                                                                  size_t dstride, const int* xofs, const int* yofs)
{
    const int p = blockIdx.x * kOtherWG + threadIdx.x;
    if (p >= dh * dw) return;
    const int dy = p / dw, dx = p - dy * dw;
    const uint8_t* q = src + (size_t)blockIdx.y * sstride + ((size_t)yofs[dy] * sw + xofs[dx]) * 3;
    uint8_t* D = dst + (size_t)blockIdx.y * dstride + (size_t)p * 3;
    D[0] = q[0]; D[1] = q[1]; D[2] = q[2];
}

// mode 0: float run tables (ResizeArea_<uchar,float> accumulation order: left to right within a source row, rows
// top to bottom, every product and sum rounded separately); mode 1: integer box * (1.f/area); mode 2: 2x2 rounding shift
__global__ __launch_bounds__(kOtherWG) void resize_area_kernel(const uint8_t* src, int sw, size_t sstride, uint8_t* dst, int dh, int dw,
                                                               size_t dstride, const AreaRun* xtab, const AreaRun* ytab, int mode,
                                                               float inv_area)
{
    const int p = blockIdx.x * kOtherWG + threadIdx.x;
    if (p >= dh * dw) return;
    const int dy = p / dw, dx = p - dy * dw;
    const uint8_t* S = src + (size_t)blockIdx.y * sstride;
    uint8_t* D = dst + (size_t)blockIdx.y * dstride + (size_t)p * 3;
    const AreaRun xr = xtab[dx], yr = ytab[dy];
    if (mode == 0) {
        float sum[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < yr.count; j++) {
            const uint8_t* row = S + ((size_t)(yr.first + j) * sw + xr.first) * 3;
            const float beta = (j == 0 && yr.has_head) ? yr.a_head : (j == yr.count - 1 && yr.has_tail) ? yr.a_tail : yr.a_mid;
            float acc[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < xr.count; k++) {
                const float a = (k == 0 && xr.has_head) ? xr.a_head : (k == xr.count - 1 && xr.has_tail) ? xr.a_tail : xr.a_mid;
#pragma unroll
                for (int c = 0; c < 3; c++) acc[c] = __fadd_rn(acc[c], __fmul_rn((float)row[3 * k + c], a));
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float term = __fmul_rn(beta, acc[c]);
                sum[c] = j == 0 ? term : __fadd_rn(sum[c], term);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) D[c] = (uint8_t)min(255, max(0, __float2int_rn(sum[c])));
    } else {
        int sum[3] = {0, 0, 0};
        for (int j = 0; j < yr.count; j++) {
            const uint8_t* row = S + ((size_t)(yr.first + j) * sw + xr.first) * 3;
            for (int k = 0; k < xr.count; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) sum[c] += row[3 * k + c];
        }
#pragma unroll
        for (int c = 0; c < 3; c++)
            D[c] = mode == 2 ? (uint8_t)((sum[c] + 2) >> 2) : (uint8_t)min(255, max(0, __float2int_rn(__fmul_rn((float)sum[c], inv_area))));
    }
}

void area_table(int ssize, int dsize, AreaRun* tab)
{
    const double scale = 1. / ((double)dsize / ssize);
    for (int dx = 0; dx < dsize; dx++) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        AreaRun r;
        memset(&r, 0, sizeof r);
        r.first = sx1;
        if (sx1 - fsx1 > 1e-3) { r.has_head = 1; r.first = sx1 - 1; r.a_head = (float)((sx1 - fsx1) / cell); r.count++; }
        r.a_mid = (float)(1.0 / cell);
        r.count += sx2 - sx1;
        if (fsx2 - sx2 > 1e-3) {
            double tl = fsx2 - sx2 < 1. ? fsx2 - sx2 : 1.;
            if (tl > cell) tl = cell;
            r.has_tail = 1; r.a_tail = (float)(tl / cell); r.count++;
        }
        tab[dx] = r;
    }
}

void area_tables(int src_h, int src_w, int dst_h, int dst_w, AreaRun* tab, int* mode, float* inv_area)
{
    const double scale_x = 1. / ((double)dst_w / src_w), scale_y = 1. / ((double)dst_h / src_h);
    const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    const bool area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    *mode = 0;
    *inv_area = 0.f;
    if (area_fast) {   // whole boxes: integer sums (OpenCV's ResizeAreaFast)
        AreaRun r;
        memset(&r, 0, sizeof r);
        for (int x = 0; x < dst_w; x++) { r.first = x * iscale_x; r.count = iscale_x; tab[x] = r; }
        for (int y = 0; y < dst_h; y++) { r.first = y * iscale_y; r.count = iscale_y; tab[dst_w + y] = r; }
        *mode = (iscale_x == 2 && iscale_y == 2) ? 2 : 1;
        *inv_area = 1.f / (float)(iscale_x * iscale_y);
    } else {
        area_table(src_w, dst_w, tab);
        area_table(src_h, dst_h, tab + dst_w);
    }
}

static void nearest_offsets(int src_h, int src_w, int dst_h, int dst_w, int* ofs)
{
    const double scale_x = 1. / ((double)dst_w / src_w), scale_y = 1. / ((double)dst_h / src_h);
    for (int x = 0; x < dst_w; x++) ofs[x] = std::min((int)floor(x * scale_x), src_w - 1);
    for (int y = 0; y < dst_h; y++) ofs[dst_w + y] = std::min((int)floor(y * scale_y), src_h - 1);
}

// ---- cv2.resize(INTER_LANCZOS4), 8-bit (Interpolation.LANCZOS4, common.py:148-160) -----------------------------------------
// OpenCV's generic path for this filter is integer arithmetic behind the coefficient tables (resize.cpp: lanczos4_tab[CV_8U] =
// HResizeLanczos4<uchar, int, short> + VResizeLanczos4<..., FixedPtCast<int, uchar, 22>, VResizeNoVec>; no IPP branch, no SIMD
// vertical pass -- unlike INTER_CUBIC below, whose 8-bit result depends on the build), so the device result is the
// reference's: eight taps per axis at source positions s - 3 .. s + 4 (s = floor of (d + 0.5) * scale - 0.5 in float32), a tap
// outside the image replaced by the nearest pixel inside, 11-bit coefficients
//   D_k = sum_j S[row_k][col_j] * alpha[j]  (int32),   dst = saturate_u8((sum_k D_k * beta[k] + 2^21) >> 22)  (low 32 bits).
struct LzTap { int first; short a[8]; };      // first = s - 3 (not clamped)

// interpolateLanczos4 (OpenCV 4.x), the same float / double steps: the weights through the angle-sum table from one sin / cos pair
// in double, cast to float, 1e30 for a tap at distance < 1e-6, normalised by their float32 sum
static void lanczos4_coeffs(float x, float* coeffs)
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    const double pi = 3.1415926535897932384626433832795;
    float sum = 0;
    const double y0 = -(x + 3) * pi * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        const float d = (x + 3 - i);
        if (fabsf(d) >= 1e-6f) {
            const double y = -d * pi * 0.25;
            coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        } else {
            coeffs[i] = 1e30f;
        }
        sum += coeffs[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

static void lanczos4_taps(int ssize, int dsize, LzTap* t)
{
    const double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s = (int)floorf(f);
        f -= s;
        float cbuf[8];
        lanczos4_coeffs(f, cbuf);
        t[d].first = s - 3;
        for (int k = 0; k < 8; k++) t[d].a[k] = sat_s16_round(cbuf[k] * 2048);
    }
}

// one destination pixel per thread: 8 x 8 taps x 3 channels.  Behind a 7.5-fold downscale neighbouring pixels share no taps, so
// there is nothing to stage; the lanes of a wave read runs 22.5 bytes apart of the same eight source rows (whole cache lines).
__global__ __launch_bounds__(kOtherWG) void resize_lanczos4_kernel(const uint8_t* src, int sh, int sw, size_t sstride, uint8_t* dst, int dh, int dw,
                                                                   size_t dstride, const LzTap* xtab, const LzTap* ytab)
{
    const int p = blockIdx.x * kOtherWG + threadIdx.x;
    if (p >= dh * dw) return;
    const int dy = p / dw, dx = p - dy * dw;
    const LzTap X = xtab[dx], Y = ytab[dy];
    const uint8_t* S = src + (size_t)blockIdx.y * sstride;
    const bool inner = X.first >= 0 && X.first + 7 < sw;
    u32 acc[3] = {0u, 0u, 0u};      // (unsigned: the running sum may pass 2^31; OpenCV's int arithmetic keeps the low 32 bits as well)
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        const uint8_t* row = S + (size_t)min(max(Y.first + k, 0), sh - 1) * sw * 3;
        int h[3] = {0, 0, 0};
        if (inner) {
            const uint8_t* q = row + (size_t)X.first * 3;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                h[0] += (int)q[3 * j] * X.a[j]; h[1] += (int)q[3 * j + 1] * X.a[j]; h[2] += (int)q[3 * j + 2] * X.a[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint8_t* q = row + (size_t)min(max(X.first + j, 0), sw - 1) * 3;
                h[0] += (int)q[0] * X.a[j]; h[1] += (int)q[1] * X.a[j]; h[2] += (int)q[2] * X.a[j];
            }
        }
        const u32 b = (u32)(int)Y.a[k];
        acc[0] += (u32)h[0] * b; acc[1] += (u32)h[1] * b; acc[2] += (u32)h[2] * b;
    }
    uint8_t* D = dst + (size_t)blockIdx.y * dstride + (size_t)p * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int r = ((int)(acc[c] + (1u << 21))) >> 22;
        D[c] = (uint8_t)min(255, max(0, r));
    }
}

// ---- cv2.resize(INTER_CUBIC), 8-bit (Interpolation.CUBIC, common.py:148-160) -----------------------------------------------
// Four taps per axis at s - 1 .. s + 2 (s as for LANCZOS4), interpolateCubic's float32 weights (A = -0.75) as 11-bit shorts, the
// horizontal pass in int -- and a VERTICAL pass that OpenCV builds compute differently (resize.cpp: VResizeCubic<..., FixedPtCast<int,
// uchar, 22>, VResizeCubicVec_32s8u>).  PSD_CUBIC_FORM picks the build that is reproduced byte for byte:
//   sse (default)  OpenCV 4.x without IPP on an SSE2/SSE3 baseline: whole groups of 8 elements (v_int16x8) of a destination row go through
//                  float32 -- S0*b0 + (S1*b1 + (S2*b2 + S3*b3)), b_k = beta[k] * 2^-22, every product and every sum rounded, cvtps2dq
//                  (to nearest even), saturating packs -- and the width*3 % 8 elements at the end of a row through the scalar fixed point;
//   fma            the same with fused multiply-adds (v_muladd on aarch64 NEON / FMA baselines);
//   fixed          the scalar FixedPtCast everywhere: (sum_k D_k * beta[k] + 2^21) >> 22 (CV_SIMD off).
// The forms differ in about one byte per 50,000 (products that land within float32 rounding of .5).  x86-64 PyPI wheels hand 8-bit
// CUBIC to IPP, whose arithmetic is not published: no form claims to be theirs.
struct CbTap { int first; short a[4]; };      // first = s - 1 (not clamped)
enum CubicForm { kCubicSse = 0, kCubicFma = 1, kCubicFixed = 2 };

static int cubic_form()
{
    static const int form = [] {
        const char* v = getenv("PSD_CUBIC_FORM");
        if (!v || !strcmp(v, "sse")) return (int)kCubicSse;
        if (!strcmp(v, "fma")) return (int)kCubicFma;
        if (!strcmp(v, "fixed")) return (int)kCubicFixed;
        return -1;
    }();
    return form;
}

// interpolateCubic (OpenCV 4.x), float32 step by step (no contraction: an x86 baseline build has no FMA to contract into)
static void cubic_coeffs(float x, float* coeffs)
{
#pragma clang fp contract(off)
    const float A = -0.75f;
    coeffs[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    coeffs[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    coeffs[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    coeffs[3] = 1.f - coeffs[0] - coeffs[1] - coeffs[2];
}

static void cubic_taps(int ssize, int dsize, CbTap* t)
{
    const double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s = (int)floorf(f);
        f -= s;
        float cbuf[4];
        cubic_coeffs(f, cbuf);
        t[d].first = s - 1;
        for (int k = 0; k < 4; k++) t[d].a[k] = sat_s16_round(cbuf[k] * 2048);
    }
}

// one destination pixel per thread: 4 x 4 taps x 3 channels; vec_end = the elements of a destination row (3 per pixel) in front of the
// scalar tail, 0 for the fixed form
template <bool FMA>
__global__ __launch_bounds__(kOtherWG) void resize_cubic_kernel(const uint8_t* src, int sh, int sw, size_t sstride, uint8_t* dst, int dh, int dw,
                                                                size_t dstride, const CbTap* xtab, const CbTap* ytab, int vec_end)
{
    const int p = blockIdx.x * kOtherWG + threadIdx.x;
    if (p >= dh * dw) return;
    const int dy = p / dw, dx = p - dy * dw;
    const CbTap X = xtab[dx], Y = ytab[dy];
    const uint8_t* S = src + (size_t)blockIdx.y * sstride;
    const bool inner = X.first >= 0 && X.first + 3 < sw;
    int h[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint8_t* row = S + (size_t)min(max(Y.first + k, 0), sh - 1) * sw * 3;
        h[k][0] = h[k][1] = h[k][2] = 0;
        if (inner) {
            const uint8_t* q = row + (size_t)X.first * 3;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                h[k][0] += (int)q[3 * j] * X.a[j]; h[k][1] += (int)q[3 * j + 1] * X.a[j]; h[k][2] += (int)q[3 * j + 2] * X.a[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint8_t* q = row + (size_t)min(max(X.first + j, 0), sw - 1) * 3;
                h[k][0] += (int)q[0] * X.a[j]; h[k][1] += (int)q[1] * X.a[j]; h[k][2] += (int)q[2] * X.a[j];
            }
        }
    }
    const float scale = 1.f / (2048 * 2048);
    const float b0 = (float)Y.a[0] * scale, b1 = (float)Y.a[1] * scale, b2 = (float)Y.a[2] * scale, b3 = (float)Y.a[3] * scale;   // (exact)
    uint8_t* D = dst + (size_t)blockIdx.y * dstride + (size_t)p * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int r;
        if (dx * 3 + c < vec_end) {
            // |h| < 2^20: the conversions are exact; the _rn intrinsics keep hipcc from contracting the unfused form
            float t = __fmul_rn((float)h[3][c], b3);
            if (FMA) {
                t = __fmaf_rn((float)h[2][c], b2, t);
                t = __fmaf_rn((float)h[1][c], b1, t);
                t = __fmaf_rn((float)h[0][c], b0, t);
            } else {
                t = __fadd_rn(__fmul_rn((float)h[2][c], b2), t);
                t = __fadd_rn(__fmul_rn((float)h[1][c], b1), t);
                t = __fadd_rn(__fmul_rn((float)h[0][c], b0), t);
            }
            r = __float2int_rn(t);
        } else {
            const u32 acc = (u32)h[0][c] * (u32)(int)Y.a[0] + (u32)h[1][c] * (u32)(int)Y.a[1] + (u32)h[2][c] * (u32)(int)Y.a[2] +
                            (u32)h[3][c] * (u32)(int)Y.a[3];
            r = ((int)(acc + (1u << 21))) >> 22;
        }
        D[c] = (uint8_t)min(255, max(0, r));
    }
}

// interpolation: 0 = INTER_NEAREST, 2 = INTER_CUBIC, 3 = INTER_AREA that shrinks along both axes, 4 = INTER_LANCZOS4 (cv2's values)
int resize_other(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_frame_stride, uint8_t* d_dst, int dst_h,
                 int dst_w, size_t dst_frame_stride, int interpolation, hipStream_t stream)
{
    if (n == 0) return PSD_OK;
    if (interpolation != PSD_INTER_NEAREST && interpolation != PSD_INTER_AREA && interpolation != PSD_INTER_LANCZOS4 &&
        interpolation != PSD_INTER_CUBIC) {
        psd_set_error("interpolation %d is not one of cv2's filters the reference offers (0 = NEAREST, 1 = LINEAR, 2 = CUBIC, 3 = AREA, 4 = LANCZOS4)", interpolation);
        return PSD_ERR_UNSUPPORTED;
    }
    if (interpolation == PSD_INTER_AREA && (dst_w > src_w || dst_h > src_h)) {
        psd_set_error("INTER_AREA run tables are for decimation (%dx%d -> %dx%d)", src_w, src_h, dst_w, dst_h);
        return PSD_ERR_UNSUPPORTED;
    }
    if (interpolation == PSD_INTER_CUBIC && cubic_form() < 0) {
        psd_set_error("PSD_CUBIC_FORM=%s: the forms of OpenCV's 8-bit INTER_CUBIC are sse (default), fma and fixed", getenv("PSD_CUBIC_FORM"));
        return PSD_ERR_INVALID;
    }
    const int kind = interpolation == PSD_INTER_NEAREST ? kTabNearest : interpolation == PSD_INTER_LANCZOS4 ? kTabLanczos4 :
                     interpolation == PSD_INTER_CUBIC ? kTabCubic : kTabArea;
    DevTable tab;
    if (!table_find(e, kind, src_h, src_w, dst_h, dst_w, &tab)) {
        int rc;
        if (kind == kTabLanczos4) {
            std::vector<LzTap> t((size_t)dst_w + dst_h);
            lanczos4_taps(src_w, dst_w, t.data());
            lanczos4_taps(src_h, dst_h, t.data() + dst_w);
            rc = table_store(e, kind, src_h, src_w, dst_h, dst_w, t.data(), t.size() * sizeof(LzTap), 0, 0.f, &tab);
        } else if (kind == kTabCubic) {
            std::vector<CbTap> t((size_t)dst_w + dst_h);
            cubic_taps(src_w, dst_w, t.data());
            cubic_taps(src_h, dst_h, t.data() + dst_w);
            rc = table_store(e, kind, src_h, src_w, dst_h, dst_w, t.data(), t.size() * sizeof(CbTap), 0, 0.f, &tab);
        } else if (kind == kTabNearest) {
            std::vector<int> ofs((size_t)dst_w + dst_h);
            nearest_offsets(src_h, src_w, dst_h, dst_w, ofs.data());
            rc = table_store(e, kind, src_h, src_w, dst_h, dst_w, ofs.data(), ofs.size() * sizeof(int), 0, 0.f, &tab);
        } else {
            std::vector<AreaRun> t((size_t)dst_w + dst_h);
            int mode;
            float inv_area;
            area_tables(src_h, src_w, dst_h, dst_w, t.data(), &mode, &inv_area);
            rc = table_store(e, kind, src_h, src_w, dst_h, dst_w, t.data(), t.size() * sizeof(AreaRun), mode, inv_area, &tab);
        }
        if (rc != PSD_OK) return rc;
    }
    const int blocks = (int)(((long)dst_h * dst_w + kOtherWG - 1) / kOtherWG);
    for (int t0 = 0; t0 < n; t0 += 32768) {   // grid.y is limited to 65535
        const int cnt = n - t0 < 32768 ? n - t0 : 32768;
        const uint8_t* S = d_src + (size_t)t0 * src_frame_stride;
        uint8_t* D = d_dst + (size_t)t0 * dst_frame_stride;
        if (kind == kTabLanczos4)
            hipLaunchKernelGGL(resize_lanczos4_kernel, dim3(blocks, cnt), dim3(kOtherWG), 0, stream, S, src_h, src_w, src_frame_stride, D, dst_h,
                               dst_w, dst_frame_stride, (const LzTap*)tab.ptr, (const LzTap*)tab.ptr + dst_w);
        else if (kind == kTabCubic) {
            const int form = cubic_form(), vec_end = form == kCubicFixed ? 0 : dst_w * 3 - dst_w * 3 % 8;
            if (form == kCubicFma)
                hipLaunchKernelGGL(resize_cubic_kernel<true>, dim3(blocks, cnt), dim3(kOtherWG), 0, stream, S, src_h, src_w, src_frame_stride, D,
                                   dst_h, dst_w, dst_frame_stride, (const CbTap*)tab.ptr, (const CbTap*)tab.ptr + dst_w, vec_end);
            else
                hipLaunchKernelGGL(resize_cubic_kernel<false>, dim3(blocks, cnt), dim3(kOtherWG), 0, stream, S, src_h, src_w, src_frame_stride, D,
                                   dst_h, dst_w, dst_frame_stride, (const CbTap*)tab.ptr, (const CbTap*)tab.ptr + dst_w, vec_end);
        } else if (kind == kTabNearest)
            hipLaunchKernelGGL(resize_nearest_kernel, dim3(blocks, cnt), dim3(kOtherWG), 0, stream, S, src_w, src_frame_stride, D, dst_h,
                               dst_w, dst_frame_stride, (const int*)tab.ptr, (const int*)tab.ptr + dst_w);
        else
            hipLaunchKernelGGL(resize_area_kernel, dim3(blocks, cnt), dim3(kOtherWG), 0, stream, S, src_w, src_frame_stride, D, dst_h, dst_w,
                               dst_frame_stride, (const AreaRun*)tab.ptr, (const AreaRun*)tab.ptr + dst_w, tab.mode, tab.inv_area);
    }
    HIP_TRY(hipGetLastError());
    return PSD_OK;
}

// The source rows a downscale reads, ascending (psd_resize_source_rows): what a host feeder has to upload.
int resize_source_rows(int src_h, int src_w, int dst_h, int dst_w, int interpolation, int* rows, int* n_rows)
{
    std::vector<uint8_t> used((size_t)src_h, 0);
    const bool area_up = interpolation == PSD_INTER_AREA && (dst_w > src_w || dst_h > src_h);
    if (interpolation == PSD_INTER_LINEAR || area_up) {
        std::vector<XTap> xt;
        std::vector<YTap> yt;
        bool area2;
        linear_tabs_host(src_h, src_w, dst_h, dst_w, area_up, xt, yt, &area2);
        if (area2) std::fill(used.begin(), used.end(), 1);   // the exact 2x2 decimation averages every row
        else for (int dy = 0; dy < dst_h; dy++) { used[yt[dy].s0] = 1; used[yt[dy].s1] = 1; }
    } else if (interpolation == PSD_INTER_NEAREST) {
        std::vector<int> ofs((size_t)dst_w + dst_h);
        nearest_offsets(src_h, src_w, dst_h, dst_w, ofs.data());
        for (int dy = 0; dy < dst_h; dy++) used[ofs[dst_w + dy]] = 1;
    } else if (interpolation == PSD_INTER_AREA) {
        std::vector<AreaRun> t((size_t)dst_w + dst_h);
        int mode;
        float inv_area;
        area_tables(src_h, src_w, dst_h, dst_w, t.data(), &mode, &inv_area);
        for (int dy = 0; dy < dst_h; dy++)
            for (int j = 0; j < t[dst_w + dy].count; j++) used[std::min(src_h - 1, t[dst_w + dy].first + j)] = 1;
    } else if (interpolation == PSD_INTER_CUBIC) {
        std::vector<CbTap> t((size_t)dst_h);
        cubic_taps(src_h, dst_h, t.data());
        for (int dy = 0; dy < dst_h; dy++)
            for (int k = 0; k < 4; k++) used[std::min(src_h - 1, std::max(0, t[dy].first + k))] = 1;
    } else if (interpolation == PSD_INTER_LANCZOS4) {
        std::vector<LzTap> t((size_t)dst_h);
        lanczos4_taps(src_h, dst_h, t.data());
        for (int dy = 0; dy < dst_h; dy++)
            for (int k = 0; k < 8; k++) used[std::min(src_h - 1, std::max(0, t[dy].first + k))] = 1;
    } else {
        psd_set_error("interpolation %d is not one of cv2's filters the reference offers (0 = NEAREST, 1 = LINEAR, 2 = CUBIC, 3 = AREA, 4 = LANCZOS4)", interpolation);
        return PSD_ERR_UNSUPPORTED;
    }
    int k = 0;
    for (int y = 0; y < src_h; y++)
        if (used[y]) rows[k++] = y;
    *n_rows = k;
    return PSD_OK;
}

}  // namespace psd
