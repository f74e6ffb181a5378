// psd_edge_kernels.hip -- the edge term of ContentDetector and the NEAREST / AREA downscale modes, for gfx950.
//
// Edge term (reference scenedetect/detectors/content_detector.py:170-174,213-239):
//     edges_t   = cv2.dilate(cv2.Canny(V_t, low, high), ones(k,k))   with low/high from numpy.median(V_t)
//     delta_edges = mean |edges_t - edges_{t-1}| = 255 * popcount(edges_t XOR edges_{t-1}) / (H*W)
// Device pipeline per chunk of frames (everything integer, so the result is exact):
//   K1 value_plane_hist   V = max(B,G,R) -> u8 plane + per-frame 256-bin histogram (for the median); with the HSV term in
//                         the same call the HSV pass writes both (V mode, psd_score_kernels.hip)
//   K2 median_thresholds  exact numpy.median from the histogram -> (low, high) via a host-built table
//   K3 sobel_nms_bits     Sobel 3x3 (replicate border), |dx|+|dy|, non-maximum suppression with OpenCV's TG22 fixed point
//                         -> TWO BIT PLANES, strong (m > high) and weak (low < m <= high); LDS tiles with halo.  The weak
//                         plane is written TILE-MAJOR (a 64x64 hysteresis tile = 128 consecutive words), the strong plane
//                         both tile-major (for K4) and row-major (for K5)
//   K4 hysteresis_frame   8-connected growth of strong into weak: one wave per 64x64 tile, one row per lane, 64-bit
//                         word-parallel steps with carry-chain run filling, rows exchanged by whole-wave DPP shifts; one
//                         workgroup takes a frame to its fix point; reads the tile-major planes, stores promotions to both
//   K5 dilate_xor         k x k dilation of the (row-major) strong bits and XOR count against the previous frame's dilated
//                         bits, bands of rows walking the time axis in registers
// Algorithmic traffic is 5 B/px (3 read + edge map write + previous edge map read); the V plane (1 B/px out
// and in) and the bit planes (3 x 0.125 B/px) are implementation overhead.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "psd_internal.h"

extern "C" void psd_set_error(const char* fmt, ...);

namespace psd {

void** engine_edge_ws(psd_engine* e);
size_t* engine_edge_ws_bytes(psd_engine* e);
int engine_num_cus(psd_engine* e);
hipStream_t engine_stream(psd_engine* e);

typedef uint32_t u32;
typedef uint64_t u64;

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            psd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)hipGetLastError(); /* (the failure is reported here: do not leave it for the next launch check) */ \
            return PSD_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)

// ---- K1: V plane + histogram -------------------------------------------------------------------

struct EdgeGeom {
    int height, width;
    long npix;
    size_t row_stride, frame_stride;
    int words_per_row;  // ceil(width/32)
};

// virtual frame j of a chunk -> source pointer
struct ChunkSrc {
    const uint8_t* frames;  // batch base
    const uint8_t* prev;    // frame preceding the batch (may be null)
    int start;              // batch index of virtual frame `first_is_prev`
    int first_is_prev;      // virtual frame 0 is `prev`
};

__device__ __forceinline__ const uint8_t* chunk_frame(const ChunkSrc& c, size_t frame_stride, int j)
{
    if (c.first_is_prev && j == 0) return c.prev;
    return c.frames + (size_t)(c.start + j - c.first_is_prev) * frame_stride;
}

constexpr int VP_ITER = 16;  // 4-pixel packets per thread: amortises the histogram clear/flush of a block

// One thread = VP_ITER packets of 4 pixels.  grid = (ceil(npix/(1024*VP_ITER)), frames)
__global__ __launch_bounds__(256) void value_plane_hist_kernel(ChunkSrc src, EdgeGeom g, uint8_t* vplane, u32* hist)
{
    // 32 replicas, lane l uses replica l%32: the LDS atomics of a wave never collide, whatever the content
    __shared__ u32 lh[256 * 32];
    for (int i = threadIdx.x; i < 256 * 32; i += 256) lh[i] = 0;
    __syncthreads();
    const int j = blockIdx.y;
    const uint8_t* frame = chunk_frame(src, g.frame_stride, j);
    const int rep = threadIdx.x & 31;
    for (int it = 0; it < VP_ITER; it++) {
    const long p0 = (((long)blockIdx.x * VP_ITER + it) * 256 + threadIdx.x) * 4;
    if (p0 < g.npix) {
        u32 packed = 0;
        const bool packed_rows = g.row_stride == (size_t)g.width * 3;
        if (packed_rows && p0 + 4 <= g.npix && (((uintptr_t)frame) & 3) == 0) {
            // 4 pixels = 12 contiguous bytes = 3 aligned dwords: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
            const u32* w = reinterpret_cast<const u32*>(frame + (size_t)p0 * 3);
            const u32 w0 = w[0], w1 = w[1], w2 = w[2];
            const u32 v0 = max(max(w0 & 0xffu, (w0 >> 8) & 0xffu), (w0 >> 16) & 0xffu);
            const u32 v1 = max(max(w0 >> 24, w1 & 0xffu), (w1 >> 8) & 0xffu);
            const u32 v2 = max(max((w1 >> 16) & 0xffu, w1 >> 24), w2 & 0xffu);
            const u32 v3 = max(max((w2 >> 8) & 0xffu, (w2 >> 16) & 0xffu), w2 >> 24);
            packed = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
            atomicAdd(&lh[v0 * 32 + rep], 1u);
            atomicAdd(&lh[v1 * 32 + rep], 1u);
            atomicAdd(&lh[v2 * 32 + rep], 1u);
            atomicAdd(&lh[v3 * 32 + rep], 1u);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const long p = p0 + k;
                if (p < g.npix) {
                    const uint8_t* s;
                    if (packed_rows) s = frame + (size_t)p * 3;
                    else {
                        const int row = (int)(p / g.width), col = (int)(p - (long)row * g.width);
                        s = frame + (size_t)row * g.row_stride + (size_t)col * 3;
                    }
                    const u32 v = max(max((u32)s[0], (u32)s[1]), (u32)s[2]);
                    packed |= v << (8 * k);
                    atomicAdd(&lh[v * 32 + rep], 1u);
                }
            }
        }
        uint8_t* dst = vplane + (size_t)j * g.npix + p0;
        if (p0 + 4 <= g.npix && ((g.npix & 3) == 0)) *reinterpret_cast<u32*>(dst) = packed;
        else
            for (int k = 0; k < 4 && p0 + k < g.npix; k++) dst[k] = (uint8_t)(packed >> (8 * k));
    }
    }
    __syncthreads();
    {
        const int b = threadIdx.x;
        u32 s = 0;
#pragma unroll
        for (int r = 0; r < 32; r++) s += lh[b * 32 + ((r + b) & 31)];   // rotated: conflict-free across lanes
        if (s) atomicAdd(&hist[(size_t)j * 256 + b], s);
    }
}

// Same result as value_plane_hist_kernel for packed, 16-byte aligned frames whose pixel count is a multiple
// of 16: one 4-wave workgroup per (tile, frame) streams its tile HBM -> LDS with global_load_lds_dwordx4 into
// wave-private 3 KiB slots one step ahead (lane i moves bytes [16 i, 16 i + 16) of each 1 KiB piece, then reads
// its 16 pixels back as 48 contiguous bytes) -- the layout of the scoring kernels, which lifts the 4.1 TB/s
// ceiling of 12-byte-stride register loads.  grid = (tiles, frames).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
constexpr int VD_G = 2, VD_NW = 4, VD_STEP = VD_G * VD_NW * 64, VD_STEPS_PER_TILE = 8;

__global__ __launch_bounds__(256) void value_plane_hist_dma_kernel(ChunkSrc src, EdgeGeom g, int groups_per_tile, uint8_t* vplane,
                                                                   u32* hist)
{
    constexpr int AC = 16;
    __shared__ __attribute__((aligned(16))) u32 lh[256 * AC];
    __shared__ __attribute__((aligned(16))) uint8_t stage[VD_G * VD_NW * 3072];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 256 * AC; i += 256) lh[i] = 0;
    __syncthreads();
    const int j = blockIdx.y;
    const uint8_t* frame = chunk_frame(src, g.frame_stride, j);
    const int n_groups = (int)(g.npix >> 4);
    const int g0 = blockIdx.x * groups_per_tile, g1 = min(n_groups, g0 + groups_per_tile);
    const long limit = (long)g1 * 48;
    auto issue = [&](int sbase) {
#pragma unroll
        for (int k = 0; k < VD_G; k++) {
            const long gfirst = (long)sbase + (long)(k * VD_NW + wave) * 64;
            uint8_t* slot = stage + (size_t)(k * VD_NW + wave) * 3072;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const long off = gfirst * 48 + q * 1024 + lane * 16;
                if (off + 16 <= limit)
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(frame + off), (lds_ptr_t)(slot + q * 1024), 16, 0, PSD_DMA_AUX);
            }
        }
    };
    u32* my_h = lh + (tid & (AC - 1));
    uint8_t* vout = vplane + (size_t)j * g.npix;
    if (g0 < g1) issue(g0);
    for (int sbase = g0; sbase < g1; sbase += VD_STEP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32 w[VD_G][12];
#pragma unroll
        for (int k = 0; k < VD_G; k++) {
            const u32x4* s4 = reinterpret_cast<const u32x4*>(stage + (size_t)(k * VD_NW + wave) * 3072 + lane * 48);
            const u32x4 a = s4[0], b = s4[1], c = s4[2];
            w[k][0] = a.x; w[k][1] = a.y; w[k][2] = a.z; w[k][3] = a.w;
            w[k][4] = b.x; w[k][5] = b.y; w[k][6] = b.z; w[k][7] = b.w;
            w[k][8] = c.x; w[k][9] = c.y; w[k][10] = c.z; w[k][11] = c.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (sbase + VD_STEP < g1) issue(sbase + VD_STEP);
#pragma unroll
        for (int k = 0; k < VD_G; k++) {
            const int gidx = sbase + (k * VD_NW + wave) * 64 + lane;
            if (gidx < g1) {
                u32 o[4] = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int ib = 3 * i, ig = 3 * i + 1, ir = 3 * i + 2;
                    const u32 bb = (w[k][ib >> 2] >> ((ib & 3) * 8)) & 0xffu;
                    const u32 gg = (w[k][ig >> 2] >> ((ig & 3) * 8)) & 0xffu;
                    const u32 rr = (w[k][ir >> 2] >> ((ir & 3) * 8)) & 0xffu;
                    const u32 v = max(max(bb, gg), rr);
                    // (an LDS add the compiler does not see as an LDS store: it orders every LDS store behind all outstanding
                    //  LDS-DMA with s_waitcnt vmcnt(0), i.e. behind the NEXT step's staging -- psd_score_kernels.hip, lds_add_hidden)
                    asm volatile("ds_add_u32 %0, %1" ::"v"((u32)(uintptr_t)&my_h[v * AC]), "v"(1u) : "memory");
                    o[i >> 2] |= v << ((i & 3) * 8);
                }
                u32x4 pk;
                pk.x = o[0]; pk.y = o[1]; pk.z = o[2]; pk.w = o[3];
                *reinterpret_cast<u32x4*>(vout + (size_t)gidx * 16) = pk;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the hidden increments, before anyone reads the histogram
    __syncthreads();
    {
        const int b = tid;
        u32 sum = 0;
#pragma unroll
        for (int r = 0; r < AC; r++) sum += lh[b * AC + ((r + b) & (AC - 1))];
        if (sum) atomicAdd(&hist[(size_t)j * 256 + b], sum);
    }
}

// ---- K2: median -> Canny thresholds ------------------------------------------------------------

// numpy.median of the V plane: for an even count the mean of the two middle order statistics, so
// 2*median = a + b is an integer in [0, 510]; thr_tab[a+b] = (low, high) computed on the host with
// the reference's float64 expression (content_detector.py:229-233).
__global__ __launch_bounds__(256) void median_thresholds_kernel(const u32* hist, long npix, const int2* thr_tab, int2* thr)
{
    __shared__ u32 cum[256];
    const int j = blockIdx.x, b = threadIdx.x;
    cum[b] = hist[(size_t)j * 256 + b];
    __syncthreads();
    if (b == 0) {
        const long ka = (npix - 1) / 2, kb = npix / 2;  // 0-based ranks of the two middle elements
        long run = 0;
        int a = -1, bb = -1;
        for (int v = 0; v < 256; v++) {
            run += cum[v];
            if (a < 0 && run > ka) a = v;
            if (bb < 0 && run > kb) { bb = v; break; }
        }
        thr[j] = thr_tab[a + bb];
    }
}

// ---- K3: Sobel + non-maximum suppression of 128 x 32 tiles straight into BIT PLANES ------------------------------------
// (Tiles load whole dwords where the frame's rows are dword aligned -- width % 4 == 0: 1080p, 4K, 720p, 640x360, 256x144 ...;
//  other widths assemble theirs from clamped byte loads.  Round 4 retired the byte-map kernel of round 1 that those took.)
// The Canny map never exists as bytes: a pixel is "strong" (m > high), "weak" (low < m <= high) or nothing, so the kernel
// writes two bits per pixel -- strong[y][x >> 5] bit (x & 31) row-major, and strong / weak again in the tile-major layout the
// hysteresis reads (below) -- instead of a 1 B/px map, and everything behind it (hysteresis, dilation, XOR count) is word-parallel.
// Phase 1 computes only the magnitude |dx| + |dy| of the tile plus a 1-px ring, four pixels of a row per work item as two
// packed 16-bit pairs (even / odd columns), rolling down the rows of a 5-row segment; phase 2 suppresses non-maxima: a pixel
// whose magnitude is not above `low` is done after one read (almost all of a natural frame), the others recompute dx, dy
// from the 3x3 V neighbourhood, classify the direction with OpenCV's TG22 fixed point and compare with the two neighbours
// along it.
constexpr int HT_TILE = 64;   // = HT, the hysteresis tile edge (defined below)
constexpr int N2_W = 128, N2_H = 32, N2_SVW = N2_W + 8, N2_SEG = 5, N2_NSEG = 7, N2_NCG = N2_SVW / 4;
static_assert(N2_NCG * N2_NSEG <= 256 && N2_SEG * N2_NSEG >= N2_H + 2, "one pass of 256 threads covers the ring");

typedef short s16x2 __attribute__((ext_vector_type(2)));

// packed 16-bit max through clang's vector types (v_pk_max_i16 without inline asm, which hipcc would pad with s_nop and
// could not schedule across)
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }

__device__ __forceinline__ int mag_at(const uint2 (*smq)[N2_NCG], int row, int c)
{
    // column c of a magnitude row: quad c >> 2 holds {m0 | m2 << 16, m1 | m3 << 16}
    const unsigned short* q = reinterpret_cast<const unsigned short*>(&smq[row][c >> 2]);
    return q[((c & 1) << 1) | ((c >> 1) & 1)];
}

// PSD_SOBEL_P1_DPP (round 6, fifth session): the segments of ONE column group sit in neighbouring lanes (lane = 7 cg' + seg, nine
// column groups per wave, lane 63 idle), every work item runs the horizontal step on six rows -- its five and the first of the
// two it shares with the segment below -- and takes the seventh, the segment below's second row, from lane + 1 with four DPP
// moves (wave_shl:1) instead of an eleven-instruction step and three LDS reads: 66 + 4 + 65 instructions per work item where
// the rolling form has 77 + 65 (static count of the phase: 205 -> 193 VALU, 25 -> 18 LDS instructions).  The arithmetic per value
// is the same, so the magnitudes are; whole edge + HSV term +1 % on shot-like and object frames (profiles/r06_an_ab_sobel_p1_dpp.txt;
// spreading the V rows over the LDS banks for the new lane order -- a 212-byte row stride -- measured nothing).  (The bottom segment's seventh
// row -- another column group's row 1 -- only feeds magnitude row 34, which does not exist.)
#ifndef PSD_SOBEL_P1_DPP
#define PSD_SOBEL_P1_DPP 1
#endif
template <bool INNER>
__device__ __forceinline__ void sobel_phase1_quads(const uint8_t (*sv)[N2_SVW], uint2 (*smq)[N2_NCG], int tid, int x0, int y0, int H, int W)
{
#if PSD_SOBEL_P1_DPP
    static_assert(N2_NSEG == 7 && 9 * 4 >= N2_NCG && N2_SEG == 5, "nine column groups of seven segments per wave, four waves");
    const int lane_ = tid & 63, cgl = lane_ / N2_NSEG, seg = lane_ - cgl * N2_NSEG;
    const int cg_raw = (tid >> 6) * 9 + cgl;
    const bool live = lane_ < 63 && cg_raw < N2_NCG;      // (the others run along on a clamped column group and store nothing)
    const int cg = min(cg_raw, N2_NCG - 1);
#else
    if (tid >= N2_NCG * N2_NSEG) return;
    const int cg = tid % N2_NCG, seg = tid / N2_NCG;
#endif
    const int m0 = seg * N2_SEG, m1 = min(m0 + N2_SEG, N2_H + 2);
    const int cl = max(cg - 1, 0) * 4, cm = cg * 4, cr = min(cg + 1, N2_NCG - 1) * 4;   // clamped neighbours only feed unused columns
    u32 keepE = 0xffffffffu, keepO = 0xffffffffu;
    if (!INNER) {   // "magnitude outside the image is 0"
        const int xb = x0 - 4 + cm;
        keepE = ((xb >= 0 && xb < W) ? 0xffffu : 0u) | ((xb + 2 >= 0 && xb + 2 < W) ? 0xffff0000u : 0u);
        keepO = ((xb + 1 >= 0 && xb + 1 < W) ? 0xffffu : 0u) | ((xb + 3 >= 0 && xb + 3 < W) ? 0xffff0000u : 0u);
    }
    // The packed arithmetic runs on the fp16 pipe over the raw integers: a bit pattern n < 2048 read as a binary16 is
    // n * 2^-24 (subnormals and the first normal binade share that spacing; denormals are on: .amdhsa_float_denorm_mode_16_64
    // 3), every Sobel quantity is an integer of magnitude <= 2040, so sums, differences and the fused x2 are exact and the
    // result's bits ARE the integer -- in sign-magnitude, which makes |x| one full-rate v_and_b32 instead of a packed
    // negate + max, and a + 2 b one v_pk_fma_f16 instead of a packed shift + add (7 half-rate + 2 full-rate instructions per
    // pixel pair and row instead of 13 half-rate ones).  The magnitudes that leave are plain 16-bit integers.
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    auto H2 = [](u32 a) { return __builtin_bit_cast(h16x2, a); };
    auto U = [](h16x2 a) { return __builtin_bit_cast(u32, a); };
    const h16x2 two = {(_Float16)2.0f, (_Float16)2.0f};
#if PSD_SOBEL_P1_DPP
    h16x2 h1E[N2_SEG + 2], h1O[N2_SEG + 2], h2E[N2_SEG + 2], h2O[N2_SEG + 2];
#else
    h16x2 h1E[3], h1O[3], h2E[3], h2O[3];
#endif
    auto horiz = [&](int r, int slot) {
        const u32 wl = *reinterpret_cast<const u32*>(&sv[r][cl]), wm = *reinterpret_cast<const u32*>(&sv[r][cm]),
                  wr = *reinterpret_cast<const u32*>(&sv[r][cr]);
        // pixels a0 = wl.b3, a1..a4 = wm, a5 = wr.b0; pairs P0 = (a0,a2) P1 = (a1,a3) P2 = (a2,a4) P3 = (a3,a5)
        const h16x2 P1 = H2(wm & 0x00ff00ffu), P2 = H2((wm >> 8) & 0x00ff00ffu);
        const h16x2 P0 = H2(__builtin_amdgcn_perm(wm, wl, 0x0c050c03u));   // [wl.b3, 0, wm.b1, 0]
        const h16x2 P3 = H2(__builtin_amdgcn_perm(wr, wm, 0x0c040c02u));   // [wm.b2, 0, wr.b0, 0]
        h1E[slot] = P2 - P0; h1O[slot] = P3 - P1;                                        // right - left
        h2E[slot] = __builtin_elementwise_fma(P1, two, P0) + P2;                         // left + 2 mid + right
        h2O[slot] = __builtin_elementwise_fma(P2, two, P1) + P3;
    };
#if PSD_SOBEL_P1_DPP
#pragma unroll
    for (int k = 0; k <= N2_SEG; k++) horiz(m0 + k, k);      // rows m0 .. m0 + 5 (<= 35: inside the tile's 36 rows)
    {
        auto below = [&](h16x2 v) { return H2((u32)__builtin_amdgcn_update_dpp(0, (int)U(v), 0x130, 0xf, 0xf, false)); };   // lane i <- lane i + 1
        h1E[N2_SEG + 1] = below(h1E[1]); h1O[N2_SEG + 1] = below(h1O[1]); h2E[N2_SEG + 1] = below(h2E[1]); h2O[N2_SEG + 1] = below(h2O[1]);
    }
#else
    horiz(m0, 0);
    horiz(m0 + 1, 1);
#endif
#pragma unroll
    for (int k = 0; k < N2_SEG; k++) {
        const int my = m0 + k;
#if PSD_SOBEL_P1_DPP
        if (my < m1 && live) {
            const int a = k, b = k + 1, c = k + 2;
#else
        if (my < m1) {
            const int a = k % 3, b = (k + 1) % 3, c = (k + 2) % 3;
            horiz(my + 2, c);
#endif
            const h16x2 dxE = __builtin_elementwise_fma(h1E[b], two, h1E[a]) + h1E[c], dxO = __builtin_elementwise_fma(h1O[b], two, h1O[a]) + h1O[c];
            const h16x2 dyE = h2E[c] - h2E[a], dyO = h2O[c] - h2O[a];
            // |dx| + |dy|: both are non-negative integers <= 1020 per half by now, so the sum is a plain 32-bit add (full rate; no carry
            // can leave a half) -- the same bits the packed fp16 add of the two patterns gives
            u32 mE = (U(dxE) & 0x7fff7fffu) + (U(dyE) & 0x7fff7fffu), mO = (U(dxO) & 0x7fff7fffu) + (U(dyO) & 0x7fff7fffu);
            if (!INNER) {
                const int y = y0 - 1 + my;
                if (y < 0 || y >= H) { mE = 0; mO = 0; }
                mE &= keepE; mO &= keepO;
            }
            smq[my][cg] = make_uint2(mE, mO);
        }
    }
}

// grid = (tiles_x * tiles_y, frames).  Workgroups are dealt to the eight XCDs round-robin by their linear id, so with the tile
// taken straight from blockIdx.x every tile's neighbours -- which share its halo: 4 of 36 rows above / below, and the 128-byte
// lines left / right of which it needs four bytes -- sit in seven OTHER L2s and every halo line is fetched over the fabric
// again: FETCH_SIZE 3.17 GB per 1024 x 1080p frames for a 2.12 GB plane (profiles/hbm_traffic.json, round 5).  Round 6: the
// workgroups k = r, r + 8, r + 16 ... of a frame (one XCD: the frame's first workgroup has linear id tiles * frame, a constant
// rotation of r) take one CONTIGUOUS run of tiles, row-major, about four tile rows at 1080p, so halo lines hit in that XCD's
// L2.  (Round 3 had tried this for TIME and measured none -- 299.5 k vs 299-301 k frames/s: the kernel is instruction-bound
// and the Infinity Cache absorbs the re-fetches -- and did not count the traffic.)
// `dirty` (zeroed by the caller): one flag per 64x64 hysteresis tile, set where this tile leaves a weak pixel -- the first
// hysteresis launch then only looks at tiles that can change at all (a natural frame has few of them).
// tiles_x_magic = ceil(2^32 / tiles_x): tile / tiles_x as one multiply-high (exact for tile < 2^20).
// RAGGED: the instance for frames whose rows are not dword aligned (kept out of the common instance: its byte loads would
// double the code the interior tiles of every 1080p frame run past).
template <bool RAGGED>
__global__ __launch_bounds__(256) void sobel_nms_bits_kernel(const uint8_t* vplane, EdgeGeom g, const int2* thr, u32* strong, u32* strong_t,
                                                             u32* weak_t, uint8_t* dirty, int tiles_x, u32 tiles_x_magic, int htiles_x,
                                                             int htiles_per_frame, int ht_frame_words)
{
    __shared__ __attribute__((aligned(16))) uint8_t sv[N2_H + 8][N2_SVW];   // rows y0-2 .. y0+33, cols x0-4 .. x0+131 (+ 4 scratch rows)
    __shared__ __attribute__((aligned(16))) uint2 smq[N2_H + 2][N2_NCG];    // rows y0-1 .. y0+32, magnitudes as even / odd pairs
    __shared__ __attribute__((aligned(16))) uint8_t obits[2][N2_H][N2_W / 8]; // [strong | weak][row][8 pixels]
    __shared__ unsigned short cand[N2_H * N2_W / 4];                        // sparse tiles: the pixels above `low`, row << 7 | column (phase 2)
    static_assert(N2_H * (N2_W / 8) == 512, "the sparse / dense rule below counts (thread, step) pairs of eight pixels");
    __shared__ int ncand, npos;   // candidates of the tile; list positions handed out (two words: nobody resets one the others still read)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.y;
    // XCD-contiguous runs (above): residue class r = k % 8 holds ceil((T - r) / 8) workgroups and starts at r (T / 8) + min(r, T % 8)
    const int k = blockIdx.x, T = gridDim.x;
    const int tile = T >= 16 ? (k & 7) * (T >> 3) + min(k & 7, T & 7) + (k >> 3) : k;
    const int ty = tiles_x == 1 ? tile : (int)__umulhi((u32)tile, tiles_x_magic), tx = tile - ty * tiles_x;
    const uint8_t* V = vplane + (size_t)j * g.npix;
    const int x0 = tx * N2_W, y0 = ty * N2_H, H = g.height, W = g.width;
    const int2 lohi = thr[j];   // read here: behind the barriers its latency would sit in front of phase 2
    const bool inner = !RAGGED && x0 >= 4 && x0 + N2_W + 4 <= W && y0 >= 2 && y0 + N2_H + 2 <= H;
    if constexpr (RAGGED) {
        // any width: the 36 x 34 dwords of the tile from four byte loads each, row and column clamped per byte
        // (BORDER_REPLICATE); slow, exact, and only odd-sized frames come here
        auto ld4 = [&](int y, int xa) -> u32 {
            const uint8_t* row = V + (size_t)min(max(y, 0), H - 1) * W;
            u32 w = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) w |= (u32)row[min(max(xa + k, 0), W - 1)] << (8 * k);
            return w;
        };
        const int cw = lane & 31;
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int ly = 2 * (wave * 5 + it) + (lane >> 5);           // rows 36..39 of sv are scratch
            *reinterpret_cast<u32*>(&sv[ly][4 * cw]) = ld4(y0 + min(ly, N2_H + 3) - 2, x0 - 4 + 4 * cw);
        }
        const int tl = min(tid >> 1, N2_H + 3), c2 = 32 + (tid & 1);
        *reinterpret_cast<u32*>(&sv[tl][4 * c2]) = ld4(y0 + tl - 2, x0 - 4 + 4 * c2);
    } else if (inner) {
        // Interior tiles (four in five at 1080p): no clamping, so all six addresses are ONE per-lane 32-bit offset on top of
        // scalar row bases (global_load_dword v, v_off, s[base]) -- a dozen VALU instructions where the clamped 64-bit
        // addresses of the general form below took a hundred, a quarter of the kernel's arithmetic.
        const int cw = lane & 31, half = lane >> 5;
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);                    // (uniform: the row bases stay in SGPRs)
        const uint8_t* base = V + (size_t)(y0 - 2) * W + (x0 - 4);                  // row y0-2, column x0-4 of the frame
        const u32 off = (u32)(half * W + 4 * cw);
        u32 v[5];
#pragma unroll
        for (int it = 0; it < 5; it++) {
            // rows 2 (5 wave + it) + half; the last wave's surplus rows (36..39) re-read row 35 into the scratch rows
            const int row = min(2 * (wave_s * 5 + it), N2_H + 2);
            typedef const __attribute__((address_space(1))) uint8_t* gbl_u8_t;
            gbl_u8_t rb = (gbl_u8_t)(base + (size_t)row * W);
            asm volatile("" : "+s"(rb));   // (keeps the row base in an SGPR pair: hipcc would fold it into 64-bit VALU adds)
            v[it] = *reinterpret_cast<const __attribute__((address_space(1))) u32*>(rb + off);
        }
        const int tl = min(tid >> 1, N2_H + 3), c2 = 32 + (tid & 1);
        const u32 vt = *reinterpret_cast<const u32*>(base + (u32)(tl * W + 4 * c2));
#pragma unroll
        for (int it = 0; it < 5; it++) *reinterpret_cast<u32*>(&sv[2 * (wave * 5 + it) + half][4 * cw]) = v[it];
        *reinterpret_cast<u32*>(&sv[tl][4 * c2]) = vt;
    } else {
        // 36 rows x 34 dwords, BORDER_REPLICATE (cv2.Sobel inside cv2.Canny): two V rows per wave instruction (lanes 0-31 /
        // 32-63 take the first 32 dwords of a row, 72 threads the two dwords left over per row), the row and the dword
        // position clamped into the image and, where the position was clamped, the edge pixel replicated.  All six loads go
        // out before the first LDS write (unconditional on clamped coordinates; behind a branch hipcc waits for every load
        // before issuing the next -- six memory latencies per tile).
        const int cw = lane & 31;
        const int xa = x0 - 4 + 4 * cw;
        const uint8_t* col = V + min(max(xa, 0), W - 4);
        u32 v[5];
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int ly = min(2 * (wave * 5 + it) + (lane >> 5), N2_H + 3);
            v[it] = *reinterpret_cast<const u32*>(col + (size_t)min(max(y0 + ly - 2, 0), H - 1) * W);
        }
        const int tl = min(tid >> 1, N2_H + 3), c2 = 32 + (tid & 1);
        const int xb = x0 - 4 + 4 * c2;
        u32 vt = *reinterpret_cast<const u32*>(V + (size_t)min(max(y0 + tl - 2, 0), H - 1) * W + min(xb, W - 4));
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int ly = 2 * (wave * 5 + it) + (lane >> 5);
            u32 w = v[it];
            if (xa < 0) w = __builtin_amdgcn_perm(w, w, 0u);                    // V[y][0] four times
            else if (xa >= W) w = __builtin_amdgcn_perm(w, w, 0x03030303u);     // V[y][W-1] four times
            // (unconditional: rows 36..39 of sv are scratch for the last wave's surplus rows -- behind a predicate hipcc sinks
            //  the load into the branch and waits for it there)
            *reinterpret_cast<u32*>(&sv[ly][4 * cw]) = w;
        }
        // (unconditional as well: threads 72.. repeat what threads 70 / 71 load and write)
        *reinterpret_cast<u32*>(&sv[tl][4 * c2]) = xb >= W ? __builtin_amdgcn_perm(vt, vt, 0x03030303u) : vt;
    }
    __syncthreads();
    // (the output words and the candidate counters are cleared here, in front of phase 1's barrier: phase 1 does not touch them,
    //  and a barrier of their own cost every tile of every frame a few per cent)
    reinterpret_cast<u32*>(&obits[0][0][0])[tid] = 0;                       // 2 x 32 x 16 bytes = 256 words
    if (tid == 0) { ncand = 0; npos = 0; }
    if (inner) sobel_phase1_quads<true>(sv, smq, tid, x0, y0, H, W);
    else sobel_phase1_quads<false>(sv, smq, tid, x0, y0, H, W);
    __syncthreads();
    // ---- phase 2: non-maximum suppression.
    // 2a, eight pixels of a row per thread and step: which of them have a magnitude above `low` at all (the candidates)?
    // None in most tiles of a natural frame: such a tile is done here.  Where there are some they are usually a few per cent of
    // the tile, and handled in place -- the pixels of a thread one after the other, each behind its own branch (round 3) -- a
    // wave runs the whole candidate path for one or two live lanes, sixteen times over: +29 % wave instructions on frames with
    // objects for 3 % of the pixels.  So a SPARSE tile puts its candidates on a list in LDS (row << 7 | column) and 2b takes
    // the list densely, one candidate per lane; a DENSE tile (noise: every pixel a candidate, where the list would only add
    // traffic) keeps the in-place loop, whose lanes are all busy anyway.
    // Per candidate: dx, dy from the 3 x 3 neighbourhood, the direction in OpenCV's TG22 fixed point, the comparison with the
    // two neighbours along it.
    const int low = lohi.x, high = lohi.y;
    const u32 lowpk = (u32)low * 0x10001u;
    constexpr int N_IT = N2_H * (N2_W / 8) / 256;
    // how many (thread, step) pairs see a candidate at all: at most CAND_CAP / 8 of them -> sparse (the list cannot overflow)
    constexpr int CAND_CAP = N2_H * N2_W / 4;
    u32 anyx[N_IT];
#pragma unroll
    for (int it = 0; it < N_IT; it++) {
        const int i = tid + it * 256;
        const int ly = i >> 4, o = i & 15;
        const uint2 qa = smq[ly + 1][1 + 2 * o], qb = smq[ly + 1][2 + 2 * o];
        // a half above `low` survives the max: (max(m, low) ^ low) != 0   (magnitudes outside the image are 0: never above)
        anyx[it] = (pk_max(qa.x, lowpk) ^ lowpk) | (pk_max(qa.y, lowpk) ^ lowpk) | (pk_max(qb.x, lowpk) ^ lowpk) | (pk_max(qb.y, lowpk) ^ lowpk);
        const unsigned long long vote = __ballot(anyx[it] != 0);
        if (vote && lane == 0) atomicAdd(&ncand, __popcll(vote));
    }
    __syncthreads();
    const int n_any = ncand;
    // one candidate: is it a local maximum along its gradient, and how strong
    auto classify = [&](int ly, int cx, int m) -> int {      // m = its magnitude; 0 = nothing, 1 = weak, 2 = strong
        const int r = ly + 2, c = 4 + cx;                          // centre in sv; magnitude row ly + 1, column c
        const int tl = sv[r - 1][c - 1], tc = sv[r - 1][c], tr = sv[r - 1][c + 1], ml = sv[r][c - 1], mr = sv[r][c + 1],
                  bl = sv[r + 1][c - 1], bc = sv[r + 1][c], br = sv[r + 1][c + 1];
        const int xs = (tr + 2 * mr + br) - (tl + 2 * ml + bl), ys = (bl + 2 * bc + br) - (tl + 2 * tc + tr);
        const int ax = abs(xs), ay = abs(ys) << 15;
        const int tg22x = ax * 13573;  // TG22 = round(tan(22.5 deg) * 2^15)
        // the two neighbours along the gradient WITHOUT a branch per direction (round 3 had three arms, and a wave that held
        // all three directions -- any wave on a busy tile -- ran all three, two magnitude reads each): left / right, above /
        // below, or the diagonal the signs point along
        const bool horiz = ay < tg22x, vert = !horiz && ay > tg22x + (ax << 16);
        const int sgn = (xs ^ ys) < 0 ? -1 : 1;
        const int dr = horiz ? 0 : 1, dc = horiz ? 1 : vert ? 0 : sgn;
        const int m1 = mag_at(smq, ly + 1 - dr, c - dc), m2 = mag_at(smq, ly + 1 + dr, c + dc);
        // horizontal / vertical: m > first && m >= second; diagonal: strictly above both
        const bool is_max = m > m1 && m + ((horiz || vert) ? 1 : 0) > m2;
        return is_max ? (m > high ? 2 : 1) : 0;
    };
    if (n_any > CAND_CAP / 8) {
        // dense: in place, a thread's eight pixels one after the other (one copy of the eight in the code: the steps loop)
        static_assert(N_IT == 2, "");
#pragma unroll 1
        for (int it = 0; it < N_IT; it++) {
            const int i = tid + it * 256;
            const int ly = i >> 4, o = i & 15;
            u32 sb = 0, wb = 0;
            if (it == 0 ? anyx[0] : anyx[1]) {
                // The magnitudes of the 3 x 10 neighbourhood of the thread's eight pixels come into registers once -- three rows
                // of four quads (the two of the pixels and one each side) -- and a candidate picks its two neighbours along the
                // gradient from them with selects: the run-time LDS address of mag_at() cost eight VALU instructions and a read
                // per neighbour, and on a busy tile every pixel is a candidate.
                uint2 mq[3][4];
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int q = 0; q < 4; q++) mq[r][q] = smq[ly + r][2 * o + q];
                // magnitude at row r (0 = above), column `col` (-1 .. 8) relative to the thread's first pixel; both constants once
                // the loop over k is unrolled.  A quad holds {m0 | m2 << 16, m1 | m3 << 16}.
                auto M = [&](int r, int col) -> int {
                    const int cc = col + 4;
                    const uint2 q = mq[r][cc >> 2];
                    const u32 w = (cc & 1) ? q.y : q.x;
                    return (int)((cc & 2) ? (w >> 16) : (w & 0xffffu));
                };
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int m = M(1, k);
                    if (m > low) {
                        const int r = ly + 2, c = 4 + 8 * o + k;                  // centre in sv
                        const int tl = sv[r - 1][c - 1], tc = sv[r - 1][c], tr = sv[r - 1][c + 1], ml = sv[r][c - 1], mr = sv[r][c + 1],
                                  bl = sv[r + 1][c - 1], bc = sv[r + 1][c], br = sv[r + 1][c + 1];
                        const int xs = (tr + 2 * mr + br) - (tl + 2 * ml + bl), ys = (bl + 2 * bc + br) - (tl + 2 * tc + tr);
                        const int ax = abs(xs), ay = abs(ys) << 15;
                        const int tg22x = ax * 13573;  // TG22 = round(tan(22.5 deg) * 2^15)
                        // (bitwise, not short-circuit, logic and plain selects: hipcc turns && / nested ?: here into branches,
                        //  eight divergent ones per pixel)
                        const bool horiz = ay < tg22x, vert = (!horiz) & (ay > tg22x + (ax << 16));
                        const bool pos = (xs ^ ys) >= 0;                         // the diagonal the signs point along
                        const int d1 = pos ? M(0, k - 1) : M(0, k + 1), d2 = pos ? M(2, k + 1) : M(2, k - 1);
                        const int v1 = vert ? M(0, k) : d1, v2 = vert ? M(2, k) : d2;
                        const int m1 = horiz ? M(1, k - 1) : v1, m2 = horiz ? M(1, k + 1) : v2;
                        // horizontal / vertical: m > first && m >= second; diagonal: strictly above both
                        const u32 is_max = (u32)(m > m1) & (u32)(m + (int)(horiz | vert) > m2);
                        const u32 strong_px = (u32)(m > high);
                        sb |= (is_max & strong_px) << k;
                        wb |= (is_max & (strong_px ^ 1u)) << k;
                    }
                }
            }
            obits[0][ly][o] = (uint8_t)sb;
            obits[1][ly][o] = (uint8_t)wb;
            if (wb) {
                const int y = y0 + ly, xq = x0 + 8 * o;
                if (y < H && xq < W) dirty[(size_t)j * htiles_per_frame + (size_t)(y / HT_TILE) * htiles_x + xq / HT_TILE] = 1;
            }
        }
    } else if (n_any > 0) {
        // sparse: onto the list (positions claimed per thread with one LDS atomic), then one candidate per lane
#pragma unroll
        for (int it = 0; it < N_IT; it++) {
            if (anyx[it]) {
                const int i = tid + it * 256;
                const int ly = i >> 4, o = i & 15;
                const uint2 qa = smq[ly + 1][1 + 2 * o], qb = smq[ly + 1][2 + 2 * o];
                const u32 xa = pk_max(qa.x, lowpk) ^ lowpk, xb = pk_max(qa.y, lowpk) ^ lowpk, xc = pk_max(qb.x, lowpk) ^ lowpk, xd = pk_max(qb.y, lowpk) ^ lowpk;
                u32 cm = 0;
                cm |= (xa & 0xffffu) ? 1u : 0u;   cm |= (xb & 0xffffu) ? 2u : 0u;   cm |= (xa >> 16) ? 4u : 0u;    cm |= (xb >> 16) ? 8u : 0u;
                cm |= (xc & 0xffffu) ? 16u : 0u;  cm |= (xd & 0xffffu) ? 32u : 0u;  cm |= (xc >> 16) ? 64u : 0u;   cm |= (xd >> 16) ? 128u : 0u;
                int at = atomicAdd(&npos, __popc(cm));
                while (cm) {
                    const int k = __ffs((int)cm) - 1;
                    cm &= cm - 1;
                    cand[at++] = (unsigned short)((ly << 7) | (8 * o + k));
                }
            }
        }
        __syncthreads();
        const int n_cand = npos;
        for (int i = tid; i < n_cand; i += 256) {
            const int e = cand[i], ly = e >> 7, cx = e & 127;
            const int cls = classify(ly, cx, mag_at(smq, ly + 1, 4 + cx));
            if (cls) {
                atomicOr(reinterpret_cast<u32*>(&obits[cls == 2 ? 0 : 1][ly][0]) + (cx >> 5), 1u << (cx & 31));
                if (cls == 1) {
                    const int y = y0 + ly, xq = x0 + cx;
                    if (y < H && xq < W) dirty[(size_t)j * htiles_per_frame + (size_t)(y / HT_TILE) * htiles_x + xq / HT_TILE] = 1;
                }
            }
        }
    }
    if (n_any > 0) __syncthreads();   // (uniform; a tile without candidates stores the words cleared in front of phase 1's barrier)
    {
        // 2 planes x 32 rows x 4 words = 256 words, one per thread
        const int plane = tid >> 7, ly = (tid >> 2) & 31, wq = tid & 3;
        const int y = y0 + ly, w = (x0 >> 5) + wq;
        if (y < H && w < g.words_per_row) {
            const u32 bits = *reinterpret_cast<const u32*>(&obits[plane][ly][4 * wq]);
            // the hysteresis reads TILE-MAJOR copies (a 64 x 64 tile = 128 consecutive words, row r at 2 r: ht_word below); the
            // strong plane also goes out row-major, which is what the dilation walks.  This tile is two hysteresis tiles wide
            // and half of one high: 32 rows x 8 bytes, consecutive, per plane and hysteresis tile.
            const size_t t = (size_t)j * ht_frame_words + ((size_t)(y >> 6) * htiles_x + (w >> 1)) * 128 + (y & 63) * 2 + (w & 1);
            if (plane) weak_t[t] = bits;
            else { strong[((size_t)j * H + y) * g.words_per_row + w] = bits; strong_t[t] = bits; }
        }
    }
}

// ---- K4: hysteresis on the bit planes -------------------------------------------------------------------------------------

constexpr int HT = 64;  // hysteresis tile edge: 64 rows (one per lane of a wave) x 64 columns (two words, one 64-bit value per lane)
static_assert(HT == HT_TILE, "");

// Grows strong into 8-connected weak pixels inside 64x64 tiles until nothing changes, using the neighbouring tiles'
// current state as a read-only halo (OpenCV's stack flood fill computes 8-connected reachability from the strong pixels,
// which is order independent, so any schedule that reaches the fix point gives the same map).
// ONE WAVE PER TILE, one image row per lane: the row's 64 strong / weak bits are one 64-bit register each, the halo column
// left and right one bit each, the rows above / below come from the neighbouring lanes (whole-wave DPP shifts) and,
// for lanes 0 and 63, from the halo rows read with the tile.  One step is a dozen 64-bit operations per lane for all 4096
// pixels; horizontal runs of weak pixels are filled in ONE step by a carry chain (seed + weak ripples through the run, in
// both directions via a bit reversal).  No LDS, no barriers.
// Work list: a tile is only looked at when its flag says so (first round: the flags of the NMS kernel; later: a neighbour
// promoted a pixel that touches it); hysteresis_frame_kernel below drives the rounds of one frame.
__device__ __forceinline__ unsigned long long fill_runs_up(unsigned long long seeds, unsigned long long run)
{
    // every bit of `run` reachable from a seed bit (seeds subset of run) by walking towards the MSB through set bits of `run`
    return (((seeds + run) ^ run) & run) | seeds;
}

// strong words: plain loads between launches; inside the per-frame kernel, where ANOTHER wave of the workgroup may have
// stored them a round ago, loads that go to the L2 (stores are write-through and complete before the round's barrier)
template <bool COHERENT>
__device__ __forceinline__ u32 ld_s(const u32* p)
{
    if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COHERENT>
__device__ __forceinline__ unsigned long long ld_s2(const u32* p)   // two words, 8-byte aligned
{
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    if (COHERENT) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *q;
}

// One 64 x 64 tile (by, bx) of one frame's bit planes to its fix point, by one wave.  St / Kt: the frame's strong / weak plane
// TILE-MAJOR (tile t = 128 consecutive words, row r of it at 2 r; written by the NMS kernel, St updated in place here); S: the
// strong plane row-major, where the promotions are stored as well (the dilation reads that one).
// Round 4: the visit used to read the row-major planes -- every lane its own image row, i.e. its own cache line: 40 L2
// requests per load instruction, 14 GB of traffic per 1024 noise frames for 1.4 GB of tile data
// (profiles/r04_pqr_edge_kernels_ab_and_hysteresis_counters.txt).  Tile-major a visit reads 512 contiguous bytes per plane,
// one word column of the tiles left and right, and three words of the rows above and below.
// Returns (uniformly) WHICH neighbours have to look again, bit (dy + 1) * 3 + (dx + 1) for the tile at (by + dy, bx + dx): only
// those that touch a promoted pixel -- a promotion in the top row wakes the tile above, in a corner the diagonal one as well.
// (Round 3 woke all eight on any border promotion: on frames with objects most tile visits of the later rounds were
// neighbours with nothing new to see, each a round trip to memory.)
template <bool COHERENT>
__device__ __forceinline__ u32 hyst_tile(u32* S, u32* St, const u32* Kt, const EdgeGeom& g, int by, int bx, int tiles_x, int tiles_y, int lane)
{
    const int H = g.height, nw = g.words_per_row;
    const int w0 = 2 * bx, y0 = by * HT;
    // this lane's row, and (lanes 0 / 63) the halo rows above / below; words outside the image read as 0
    const int y = y0 + lane;
    const bool row_in = y < H;
    const bool has_hi = w0 + 1 < nw, has_l = w0 > 0, has_r = w0 + 2 < nw;
    const int tile = by * tiles_x + bx;
    const u32* own = St + (size_t)tile * 128 + lane * 2;
    // halo row of lane 0: row 63 of the tile above; of the other lanes (lane 63 uses it): row 0 of the tile below
    const bool halo_in = lane == 0 ? by > 0 : (lane == 63 && by + 1 < tiles_y);
    const int hby = lane == 0 ? max(by - 1, 0) : min(by + 1, tiles_y - 1);
    const u32* hrow = St + (size_t)(hby * tiles_x + bx) * 128 + (lane == 0 ? 126 : 0);
    // (unconditional loads on clamped tile indices -- they go out together --, masked afterwards)
    const int dl = has_l ? 128 : 0, dr = has_r ? 128 : 0;
    unsigned long long s64 = ld_s2<COHERENT>(own);
    unsigned long long k64 = *reinterpret_cast<const unsigned long long*>(Kt + (size_t)tile * 128 + lane * 2);
    u32 s_l = ld_s<COHERENT>(own - dl + 1), s_r = ld_s<COHERENT>(own + dr);
    unsigned long long h64 = ld_s2<COHERENT>(hrow);
    u32 h_l = ld_s<COHERENT>(hrow - dl + 1), h_r = ld_s<COHERENT>(hrow + dr);
    u32 s_lo = (u32)s64, s_hi = (u32)(s64 >> 32), k_lo = (u32)k64, k_hi = (u32)(k64 >> 32), h_lo = (u32)h64, h_hi = (u32)(h64 >> 32);
    if (!row_in) { s_lo = s_hi = s_l = s_r = k_lo = k_hi = 0; }
    if (!has_hi) { s_hi = 0; k_hi = 0; h_hi = 0; }
    if (!has_l) { s_l = 0; h_l = 0; }
    if (!has_r) { s_r = 0; h_r = 0; }
    if (!halo_in) { h_lo = h_hi = h_l = h_r = 0; }
    unsigned long long Sv = ((unsigned long long)s_hi << 32) | s_lo;
    const unsigned long long Kv = ((unsigned long long)k_hi << 32) | k_lo;
    const unsigned long long Kvr = __brevll(Kv);           // (the weak row mirrored, for the runs filled towards the LSB)
    const unsigned long long S0 = Sv;
    const u32 eL = s_l >> 31, eR = s_r & 1u;              // strong state of the pixels left / right of this row (fixed here)
    const unsigned long long Hv = ((unsigned long long)h_hi << 32) | h_lo;
    const u32 hL = h_l >> 31, hR = h_r & 1u;
    if (__ballot(Kv != 0) == 0) return 0u;                 // no weak pixel in the tile: nothing can change
    // Rows above / below come from the neighbouring lanes with ONE DPP move per 32-bit half (wave_shr:1 / wave_shl:1 shift the
    // whole wave by a lane; the lane without a source keeps `old`, which is where the halo row goes in: lane 0's row above is
    // the halo row it loaded, lane 63's row below likewise).  Round 4: these were ds_bpermute_b32 -- eight LDS round trips per
    // step of the fix point, four of them for the columns left / right of the tile, which do not change inside a visit.
    auto from_above = [](u32 halo, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)halo, (int)v, 0x138, 0xf, 0xf, false); };   // lane i <- lane i - 1
    auto from_below = [](u32 halo, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)halo, (int)v, 0x130, 0xf, 0xf, false); };   // lane i <- lane i + 1
    const u32 h_lo32 = (u32)Hv, h_hi32 = (u32)(Hv >> 32);
    // strong in the column left / right of the tile, rows y - 1 .. y + 1: fixed for this visit
    const unsigned long long nl = (u64)((from_above(hL, eL) | eL | from_below(hL, eL)) & 1u);
    const unsigned long long nr = (u64)((from_above(hR, eR) | eR | from_below(hR, eR)) & 1u) << 63;
    for (;;) {
        const u32 s_lo32 = (u32)Sv, s_hi32 = (u32)(Sv >> 32);
        const unsigned long long up = ((u64)from_above(h_hi32, s_hi32) << 32) | from_above(h_lo32, s_lo32);
        const unsigned long long dn = ((u64)from_below(h_hi32, s_hi32) << 32) | from_below(h_lo32, s_lo32);
        const unsigned long long n = up | Sv | dn;                                   // strong in the three rows, same column
        const unsigned long long near = n | (n << 1) | (n >> 1) | nl | nr;           // ... or a column next to it
        unsigned long long grown = Sv | (Kv & near);
        // horizontal runs of weak pixels in one go: promoted pixels are seeds inside the weak runs
        const unsigned long long seeds = grown & Kv;
        grown |= fill_runs_up(seeds, Kv);
        grown |= __brevll(fill_runs_up(__brevll(seeds), Kvr));
        const bool ch = grown != Sv;
        Sv = grown;
        if (__ballot(ch) == 0) break;
    }
    const unsigned long long added = Sv & ~S0;
    const unsigned long long any_added = __ballot(added != 0);
    if (any_added == 0) return 0u;
    if (added != 0 && row_in) {
        *reinterpret_cast<unsigned long long*>(St + (size_t)tile * 128 + lane * 2) = Sv;   // (bits beyond the image never get set: Kv is masked)
        if ((u32)added) S[(size_t)y * nw + w0] = (u32)Sv;
        if ((u32)(added >> 32) && has_hi) S[(size_t)y * nw + w0 + 1] = (u32)(Sv >> 32);
    }
    // promoted pixels on the tile border: the neighbours that touch them have to look again
    const unsigned long long west = __ballot((added & 1ull) != 0), east = __ballot((added >> 63) != 0);
    auto row_of = [&](int l) {   // `added` of lane l, into scalar registers
        return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(added >> 32), l) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)added, l);
    };
    const unsigned long long top = row_of(0), bottom = row_of(63);                 // rows 0 and 63 of the tile
    u32 wake = 0;
    if (top) wake |= 1u << 1;
    if (bottom) wake |= 1u << 7;
    if (west) wake |= 1u << 3;
    if (east) wake |= 1u << 5;
    if (top & 1ull) wake |= 1u << 0;
    if (top >> 63) wake |= 1u << 2;
    if (bottom & 1ull) wake |= 1u << 6;
    if (bottom >> 63) wake |= 1u << 8;
    return wake;
}

// The whole hysteresis of ONE FRAME in one workgroup and one launch: chains of weak pixels never leave their frame, so
// nothing has to be handed from launch to launch -- no convergence flags, no host round trips.  (Rounds 2-3 relaunched a
// tile-parallel kernel over dirty-tile lists: frames full of straight object edges needed fifty launches, an edge that runs
// along a tile border crosses it at every wiggle.)  The workgroup keeps two bit maps of the frame's tiles in LDS: the tiles to
// look at in this round (first round: the flags of the NMS kernel) and the ones woken for the next; its waves take the set
// bits of this round's map, a barrier, swap, until a round wakes nobody.  A wave that wakes a neighbour has stored its
// promotions before the round's barrier (write-through L1), and the strong words are read with loads that go to the L2, so
// the next round sees them whichever wave stored them.  grid = frames; dynamic LDS = 2 x ceil(tiles / 32) words.
// strong = the row-major plane (promotions are stored there too: the dilation reads it), strong_t / weak_t = the tile-major
// planes the visits read (see hyst_tile); dirty = one flag byte per tile, `dirty_stride` (a multiple of 32) bytes per frame.
// HF_WAVES waves per workgroup: as many as still let every frame of the chunk be resident at once -- frames whose chains
// take fifty rounds are bound by the latency of a round, not by throughput, so two waves of workgroups take twice as long.
template <int HF_WAVES>
__global__ __launch_bounds__(HF_WAVES * 64) void hysteresis_frame_kernel(u32* strong, u32* strong_t, const u32* weak_t, EdgeGeom g,
                                                                         const uint8_t* dirty, int dirty_stride, int tiles_x, int tiles_y)
{
    extern __shared__ u32 hf_maps[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = blockIdx.x, per_frame = tiles_x * tiles_y, nwords = (per_frame + 31) >> 5;
    u32* cur = hf_maps;
    u32* nxt = hf_maps + nwords;
    // the frame's tile flags (one byte each, the stride a multiple of 32 bytes, the padding zero): 32 of them per map word,
    // read as two 16-byte loads
    const uint4* D = reinterpret_cast<const uint4*>(dirty + (size_t)j * dirty_stride);
    for (int w = tid; w < nwords; w += HF_WAVES * 64) {
        const uint4 a = D[2 * w], b = D[2 * w + 1];
        const u32 f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        u32 m = 0;
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((f[q] >> (8 * k)) & 0xffu) m |= 1u << (4 * q + k);
        cur[w] = m;
        nxt[w] = 0;
    }
    __syncthreads();
    u32* S = strong + (size_t)j * g.height * g.words_per_row;
    u32* St = strong_t + (size_t)j * per_frame * 128;
    const u32* Kt = weak_t + (size_t)j * per_frame * 128;
    for (int round = 0;; round++) {
        for (int w = wave; w < nwords; w += HF_WAVES) {
            u32 bits = __builtin_amdgcn_readfirstlane(cur[w]);
            while (bits) {
                const int tile = w * 32 + __ffs((int)bits) - 1;
                bits &= bits - 1;
                const int by = tile / tiles_x, bx = tile - by * tiles_x;
                // (the first round -- on noisy frames most of the work -- reads through the L1: nothing of this frame can be
                //  stale in it yet, and what a neighbouring wave stores meanwhile is picked up in the round that wave triggers)
                const u32 wake = round == 0 ? hyst_tile<false>(S, St, Kt, g, by, bx, tiles_x, tiles_y, lane)
                                            : hyst_tile<true>(S, St, Kt, g, by, bx, tiles_x, tiles_y, lane);
                if (wake) {
                    if (lane < 9 && ((wake >> lane) & 1u)) {
                        const int ny = by + lane / 3 - 1, nx = bx + lane % 3 - 1;
                        if (ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x) {
                            const int nt = ny * tiles_x + nx;
                            atomicOr(&nxt[nt >> 5], 1u << (nt & 31));
                        }
                    }
                }
            }
        }
        __syncthreads();                       // this round's stores (vmcnt drained) and wake-ups are in
        u32 any = 0;
        for (int w = tid; w < nwords; w += HF_WAVES * 64) {
            const u32 v = nxt[w];
            cur[w] = v;
            nxt[w] = 0;
            any |= v;
        }
        if (!__syncthreads_or(any != 0)) break;
    }
}

// ---- K5 + K6: dilation and XOR count on the bit planes, walking the time axis ---------------------------------------------

// cv2.dilate(edges, ones(k, k)) (anchor at the centre, outside the image ignored) and the XOR count against the previous
// frame's dilated edges (content_detector.py:170-174, 239), word-parallel and entirely in registers.
// A WAVE owns DX_R image rows x a strip of DX_STRIP words (lane l holds word strip0 - 1 + l: the first and the last lane are
// the halo words of the horizontal window) and WALKS the frames of a chunk.  Per frame every lane reads the DX_R + k - 1 words
// of its column that the vertical window needs (one coalesced 256-byte row segment per load instruction, the next frame's
// loads in flight while this one is worked on), ORs them into the DX_R vertically dilated words -- a common core plus suffix /
// prefix ORs, about 4 ORs per output word for k = 13 instead of 12 --, takes the horizontal neighbours from the adjacent lanes
// (two wave shifts per word), ORs the k-window horizontally by doubling (x |= x >> 1, >> 2, >> 4 ..., then one shift for the
// remainder, on 32-bit halves with v_alignbit_b32), XORs with the previous frame's dilated words -- which never left the
// lane's registers -- and counts.  No LDS, no barriers; the dilated map goes to memory only when a caller wants to see it
// (psd_edge_map_device).  Per frame the pass reads 0.25 bit/px a few times over (L2 hits) and writes nothing (round 2: byte
// map 1 B/px in, dilated bits out and in again).  The frame in front of a walk is dilated once more as its halo (`carry` =
// the Canny bits of the frame before virtual frame 0, when there is one).
// K = the dilation size at compile time for the sizes the reference's _estimated_kernel_size produces up to 8K frames, 0 =
// any size at run time (slower: one compare + OR per (window row, output row)).
// grid = (ceil(H / (4 DX_R)) * strips, walks), 4 waves per workgroup (4 consecutive row groups).
constexpr int DX_R = 8, DX_STRIP = 62;

template <int K>
__global__ __launch_bounds__(256) void dilate_xor_kernel(const u32* strong, EdgeGeom g, int k_rt, int strips, int frames_per_walk,
                                                         int count, const u32* carry, unsigned long long* out_xor, u32* dil_out)
{
    constexpr int KMAX = K ? K : 63, NIN = DX_R + KMAX - 1;
    const int k = K ? K : k_rt;
    const int H = g.height, nw = g.words_per_row, W = g.width;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bi = blockIdx.x / strips, si = blockIdx.x - bi * strips;
    const int y0 = (bi * 4 + wave) * DX_R;
    if (y0 >= H) return;                                     // (whole waves; nothing below synchronises across waves)
    const int w = si * DX_STRIP - 1 + lane;                  // this lane's word; lanes 0 and 63 only feed their neighbours
    const bool w_in = w >= 0 && w < nw;
    const bool out_lane = lane >= 1 && lane <= DX_STRIP && w_in;
    const int up = k / 2, dn = k - 1 - k / 2;                // rows [y - up, y + dn], columns [x - left, x + right]
    const int left = up, right = dn;
    const int nin = DX_R + k - 1;
    const int t0 = blockIdx.y * frames_per_walk, t1 = min(count, t0 + frames_per_walk);
    const size_t words = (size_t)H * nw;
    const int wc = min(max(w, 0), nw - 1);
    u32 lastmask = 0xffffffffu;
    if (w_in && w * 32 + 32 > W) lastmask = (1u << (W - w * 32)) - 1u;
    auto src_of = [&](int t) { return t >= 0 ? strong + (size_t)t * words : carry; };
    u32 pre[NIN];
    auto fetch = [&](const u32* src) {
#pragma unroll
        for (int r = 0; r < NIN; r++)
            if (K || r < nin) pre[r] = src[(size_t)min(max(y0 - up + r, 0), H - 1) * nw + wc];   // unconditional, clamped
    };
    u32 prevd[DX_R];
#pragma unroll
    for (int i = 0; i < DX_R; i++) prevd[i] = 0;
    bool have_prev = false;
    // frame t0 - 1 first (halo of the walk: dilated, not counted; the very first frame has no predecessor), then t0 .. t1 - 1
    int t = t0 - 1;
    if (src_of(t) == nullptr) t++;
    if (t < t1) fetch(src_of(t));
    for (; t < t1; t++) {
        u32 h[NIN];
#pragma unroll
        for (int r = 0; r < NIN; r++) {
            const int y = y0 - up + r;
            h[r] = (K || r < nin) && w_in && y >= 0 && y < H ? pre[r] : 0u;
        }
        if (t + 1 < t1) fetch(src_of(t + 1));                // in flight while this frame is worked on
        // ---- vertical OR: out[i] = OR h[i .. i + k - 1]
        u32 v[DX_R];
        if constexpr (K >= DX_R) {
            u32 core = h[DX_R - 1];
#pragma unroll
            for (int r = DX_R; r <= K - 1; r++) core |= h[r];
            u32 suf[DX_R], pfx[DX_R];                        // suf[i] = OR h[i .. DX_R-2], pfx[i] = OR h[K .. i+K-1]
            suf[DX_R - 1] = 0;
#pragma unroll
            for (int i = DX_R - 2; i >= 0; i--) suf[i] = suf[i + 1] | h[i];
            pfx[0] = 0;
#pragma unroll
            for (int i = 1; i < DX_R; i++) pfx[i] = pfx[i - 1] | h[i + K - 1];
#pragma unroll
            for (int i = 0; i < DX_R; i++) v[i] = core | suf[i] | pfx[i];
        } else if constexpr (K > 0) {
#pragma unroll
            for (int i = 0; i < DX_R; i++) {
                u32 a = h[i];
#pragma unroll
                for (int r = 1; r < K; r++) a |= h[i + r];
                v[i] = a;
            }
        } else {
#pragma unroll
            for (int i = 0; i < DX_R; i++) v[i] = 0;
            for (int r = 0; r < nin; r++) {
                u32 hr = 0;
#pragma unroll
                for (int q = 0; q < NIN; q++) hr = q == r ? h[q] : hr;   // (register arrays cannot be indexed at run time)
#pragma unroll
                for (int i = 0; i < DX_R; i++) v[i] |= (r >= i && r < i + k) ? hr : 0u;
            }
        }
        // ---- horizontal OR over [x - left, x + right], XOR with the previous frame, count
        u32 diff = 0;
#pragma unroll
        for (int i = 0; i < DX_R; i++) {
            const u32 mid = v[i], lo = __shfl_up(mid, 1), hi = __shfl_down(mid, 1);
            // (windows reach at most one word across: k <= 63, enforced in psd_engine.cpp, keeps left and right below 32)
            u32 rl = mid, rh = hi, lh = mid, ll = lo;
            int cover = 1;
            for (; 2 * cover <= right + 1; cover *= 2) { rl |= __builtin_amdgcn_alignbit(rh, rl, cover); rh |= rh >> cover; }
            if (cover < right + 1) rl |= __builtin_amdgcn_alignbit(rh, rl, right + 1 - cover);
            cover = 1;
            for (; 2 * cover <= left + 1; cover *= 2) { lh |= __builtin_amdgcn_alignbit(lh, ll, 32 - cover); ll |= ll << cover; }
            if (cover < left + 1) lh |= __builtin_amdgcn_alignbit(lh, ll, 32 - (left + 1 - cover));
            const u32 d = (rl | lh) & lastmask;
            const bool row_in = y0 + i < H;
            if (have_prev && out_lane && row_in) diff += __popc(d ^ prevd[i]);
            prevd[i] = d;
            if (dil_out != nullptr && t >= t0 && out_lane && row_in) dil_out[(size_t)t * words + (size_t)(y0 + i) * nw + w] = d;
        }
        if (have_prev && t >= t0) {
            for (int o = 32; o > 0; o >>= 1) diff += __shfl_down(diff, o);
            if (lane == 0 && diff) atomicAdd(&out_xor[t], (unsigned long long)diff);
        }
        have_prev = true;
    }
}

// Copy the chunk's XOR counts into the batch records (virtual frame j -> batch frame first_t + j).
__global__ void store_xor_kernel(const unsigned long long* xr, int count, int skip_first, int first_has_pred,
                                 int first_t, psd_frame_scores* out, const uint8_t* seg)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    if (skip_first && j == 0) return;          // virtual frame 0 is `prev`, not part of the batch
    if (j == 0 && !first_has_pred) return;     // no predecessor: edge_xor stays 0
    if (seg && seg[first_t + j - skip_first]) return;   // first frame of a packed clip: no predecessor either
    out[first_t + j - skip_first].edge_xor = xr[j];
}

// ---- host orchestration ---------------------------------------------------------------------------

static int estimated_kernel_size(int width, int height)
{
    // 4 + round(sqrt(w*h)/192), made odd (content_detector.py:39-46; Python round = half to even)
    int size = 4 + (int)nearbyint(sqrt((double)width * (double)height) / 192.0);
    if (size % 2 == 0) size += 1;
    return size;
}

struct EdgeBuffers {
    uint8_t* vplane; u32* strong; u32* strong_t; u32* weak_t; u32* hist; int2* thr; u32* carry; u32* dil1;
    unsigned long long* xr; int2* thr_tab; uint8_t* dirty;
    int cap_frames; size_t tiles_per_frame; size_t ht_frame_words;   // (flags per frame, padded; words of a tile-major plane per frame)
    uint8_t* zero_begin; size_t zero_bytes;   // hist, xr and the tile flags: cleared by ONE memset per chunk
    bool fresh;                               // the workspace was (re)allocated by this call
};

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

static int edge_buffers(psd_engine* e, const EdgeGeom& g, int want_frames, EdgeBuffers* b)
{
    const size_t words = (size_t)g.height * g.words_per_row;
    // (one flag byte per 64 x 64 hysteresis tile; a frame's flags padded to whole 32-byte groups, which the per-frame kernel loads)
    const size_t ht_tiles = (size_t)((g.width + HT - 1) / HT) * ((g.height + HT - 1) / HT);
    const size_t tiles = (ht_tiles + 31) & ~(size_t)31;
    const size_t tm_bytes = ht_tiles * 128 * 4;   // one tile-major bit plane of a frame (whole tiles)
    const size_t per_frame = align_up((size_t)g.npix) + align_up(256 * 4) + align_up(sizeof(int2)) + align_up(words * 4) + align_up(tm_bytes) * 2 +
                             align_up(8) + align_up(tiles);
    // bound the workspace (default 8 GiB of the 288 GB: 2048 x 1080p frames in one chunk, +1.4 % over 4 GiB on such batches;
    // PSD_EDGE_WS_MB overrides) unless a single frame needs more
    static const size_t ws_cap = [] {
        const char* e = getenv("PSD_EDGE_WS_MB");
        const long mb = e ? atol(e) : 8192;
        return (size_t)(mb > 0 ? mb : 8192) << 20;
    }();
    int frames = (int)std::max<size_t>(1, std::min<size_t>((size_t)want_frames, ws_cap / per_frame));
    const size_t fixed = 2 * align_up(words * 4) + align_up(511 * sizeof(int2)) + 16 * 256;
    const size_t need = per_frame * (size_t)frames + fixed;
    void** ws = engine_edge_ws(e);
    size_t* ws_bytes = engine_edge_ws_bytes(e);
    b->fresh = false;
    if (*ws_bytes < need) {
        if (*ws) HIP_TRY(hipFree(*ws));
        *ws = nullptr; *ws_bytes = 0;
        HIP_TRY(hipMalloc(ws, need));
        *ws_bytes = need;
        b->fresh = true;
    }
    uint8_t* p = (uint8_t*)*ws;
    auto take = [&](size_t bytes) { uint8_t* r = p; p += align_up(bytes); return r; };
    b->thr_tab = (int2*)take(511 * sizeof(int2));   // first: its place does not depend on the geometry (uploaded once)
    b->vplane = take((size_t)g.npix * frames);
    b->thr = (int2*)take((size_t)frames * sizeof(int2));
    b->strong = (u32*)take(words * 4 * frames);
    b->strong_t = (u32*)take(tm_bytes * frames);
    b->weak_t = (u32*)take(tm_bytes * frames);
    b->ht_frame_words = ht_tiles * 128;
    b->carry = (u32*)take(words * 4);
    b->dil1 = (u32*)take(words * 4);
    b->zero_begin = p;
    b->hist = (u32*)take((size_t)frames * 256 * 4);
    b->xr = (unsigned long long*)take((size_t)frames * 8);
    b->dirty = take(tiles * frames);
    b->zero_bytes = (size_t)(p - b->zero_begin);
    b->tiles_per_frame = tiles;
    b->cap_frames = frames;
    if ((size_t)(p - (uint8_t*)*ws) > *ws_bytes) { psd_set_error("edge workspace layout overflow"); return PSD_ERR_NOMEM; }
    return PSD_OK;
}

static void threshold_table(int2* tab)
{
    // low = int(max(0, (1-sigma)*median)), high = int(min(255, (1+sigma)*median)), sigma = 1/3, in the
    // reference's float64 arithmetic (content_detector.py:229-233); median = m2/2 for m2 = a+b.
    const double sigma = 1.0 / 3.0;
    for (int m2 = 0; m2 <= 510; m2++) {
        const double median = m2 / 2.0;
        const double lo = (1.0 - sigma) * median, hi = (1.0 + sigma) * median;
        int low = (int)(lo > 0 ? lo : 0), high = (int)(hi < 255 ? hi : 255);
        // cv2.Canny swaps the thresholds if they are out of order and floors them (already ints here)
        if (low > high) { const int t = low; low = high; high = t; }
        tab[m2] = make_int2(low, high);
    }
}

// K1..K4 for `count` virtual frames; on return (stream-ordered) b.strong holds their final Canny edges as bit rows.
// hsv != nullptr: the chunk's real frames get their V plane and V histogram from the HSV pass in V mode (one read of the
// frames for both terms, psd_score_kernels.hip); only a predecessor frame standing in as virtual frame 0 still goes
// through the V-plane kernel.
// down != nullptr (with hsv: its `out` / `seg` of the chunk): the chunk's frames are FULL-SIZE frames behind the default downscale;
// the fused downscale kernel's VOUT instance is the front end (psd_resize_kernels.hip), for the predecessor frame as well.
static int edge_chunk(psd_engine* e, const EdgeGeom& g, const ChunkSrc& src, int count, const EdgeBuffers& b, hipStream_t stream,
                      const ScoreParams* hsv = nullptr, int target_blocks = 0, int* launches = nullptr, const DownSrc* down = nullptr)
{
    HIP_TRY(hipMemsetAsync(b.zero_begin, 0, b.zero_bytes, stream));   // histograms, XOR counters, tile flags
    // packed 16-byte aligned frames with a multiple of 16 pixels take the LDS-DMA streaming variant
    const bool vp_dma = g.row_stride == (size_t)g.width * 3 && (g.npix & 15) == 0 && (g.frame_stride & 15) == 0 &&
                        ((uintptr_t)src.frames & 15) == 0 && (!src.first_is_prev || ((uintptr_t)src.prev & 15) == 0);
    const int vp_count = down ? 0 : hsv ? (src.first_is_prev ? 1 : 0) : count;   // virtual frames the V-plane kernel converts
    if (down) {
        const int fp = src.first_is_prev;
        int rc = PSD_OK;
        // (the predecessor alone: no SADs -- it has no predecessor of its own -- only its V plane and histogram, virtual frame 0)
        if (fp) rc = resize_linear_score_vplane(e, down->prev, 1, down->src_h, down->src_w, down->frame_stride, nullptr, g.height, g.width, hsv->out,
                                                stream, launches, nullptr, b.vplane, b.hist);
        if (rc != PSD_OK) return rc;
        const int cnt = count - fp;
        if (cnt > 0) {
            const uint8_t* first = down->frames + (size_t)src.start * down->frame_stride;
            const uint8_t* before = src.start == 0 ? down->prev : first - down->frame_stride;
            rc = resize_linear_score_vplane(e, first, cnt, down->src_h, down->src_w, down->frame_stride, before, g.height, g.width, hsv->out, stream,
                                            launches, hsv->seg, b.vplane + (fp ? (size_t)g.npix : 0), b.hist + (fp ? 256 : 0));
            if (rc != PSD_OK) return rc;
        }
    } else if (vp_count > 0 && vp_dma) {
        const int n_groups = (int)(g.npix >> 4);
        const int cap = VD_STEP * (vp_count >= 32 ? VD_STEPS_PER_TILE : 1);
        const int tiles = (n_groups + cap - 1) / cap;
        const int groups_per_tile = (n_groups + tiles - 1) / tiles;
        hipLaunchKernelGGL(value_plane_hist_dma_kernel, dim3(tiles, vp_count), dim3(256), 0, stream, src, g, groups_per_tile, b.vplane, b.hist);
    } else if (vp_count > 0) {
        hipLaunchKernelGGL(value_plane_hist_kernel, dim3((unsigned)((g.npix + 1024 * VP_ITER - 1) / (1024 * VP_ITER)), vp_count), dim3(256), 0, stream, src, g,
                           b.vplane, b.hist);
    }
    if (hsv && !down) HIP_TRY(launch_score_frames(*hsv, true, false, true, target_blocks, stream, launches));
    hipLaunchKernelGGL(median_thresholds_kernel, dim3(count), dim3(256), 0, stream, b.hist, g.npix, b.thr_tab, b.thr);
    const int htx = (g.width + HT - 1) / HT, hty = (g.height + HT - 1) / HT;
    {
        const int tx = (g.width + N2_W - 1) / N2_W, ty = (g.height + N2_H - 1) / N2_H;
        const int per_frame = tx * ty;
        if (per_frame >= (1 << 20)) { psd_set_error("frame too large for the edge term"); return PSD_ERR_UNSUPPORTED; }
        const u32 magic = tx == 1 ? 0u : (u32)((0x100000000ull + (unsigned long long)tx - 1) / (unsigned long long)tx);   // ceil(2^32 / tx)
        for (int f0 = 0; f0 < count; f0 += 32768) {   // grid.y limit
            const int nf = std::min(32768, count - f0);
            // rows not dword aligned (or narrower than two dwords): the instance that assembles its tiles from byte loads
            const bool ragged = (g.width & 3) != 0 || g.width < 8;
            auto kernel = ragged ? sobel_nms_bits_kernel<true> : sobel_nms_bits_kernel<false>;
            hipLaunchKernelGGL(kernel, dim3(per_frame, nf), dim3(256), 0, stream,
                               b.vplane + (size_t)f0 * g.npix, g, b.thr + f0, b.strong + (size_t)f0 * g.height * g.words_per_row,
                               b.strong_t + (size_t)f0 * b.ht_frame_words, b.weak_t + (size_t)f0 * b.ht_frame_words,
                               b.dirty + (size_t)f0 * b.tiles_per_frame, tx, magic, htx, (int)b.tiles_per_frame, (int)b.ht_frame_words);
        }
    }
    HIP_TRY(hipGetLastError());
    // hysteresis to the fix point: one workgroup per frame, one launch.  As many waves per workgroup as still let every frame
    // of the chunk be resident at once (wave slots of THIS device: 32 per CU).
    {
        const long slots = (long)engine_num_cus(e) * 32;
        const size_t lds = 2 * (size_t)(((long)htx * hty + 31) / 32) * sizeof(u32);
        if (lds > 96 * 1024) { psd_set_error("frame too large for the edge term (%d x %d hysteresis tiles)", htx, hty); return PSD_ERR_UNSUPPORTED; }
        if ((long)count * 16 <= slots)
            hipLaunchKernelGGL(hysteresis_frame_kernel<16>, dim3(count), dim3(16 * 64), lds, stream, b.strong, b.strong_t, b.weak_t, g, b.dirty, (int)b.tiles_per_frame, htx, hty);
        else if ((long)count * 8 <= slots)
            hipLaunchKernelGGL(hysteresis_frame_kernel<8>, dim3(count), dim3(8 * 64), lds, stream, b.strong, b.strong_t, b.weak_t, g, b.dirty, (int)b.tiles_per_frame, htx, hty);
        else
            hipLaunchKernelGGL(hysteresis_frame_kernel<4>, dim3(count), dim3(4 * 64), lds, stream, b.strong, b.strong_t, b.weak_t, g, b.dirty, (int)b.tiles_per_frame, htx, hty);
    }
    HIP_TRY(hipGetLastError());
    return PSD_OK;
}

// dilation + XOR count of `count` virtual frames whose Canny bits are in b.strong; `carry` = the Canny bits of the frame
// before virtual frame 0 (or nullptr); dil_out = nullptr unless the dilated maps themselves are wanted
static int launch_dilate_xor(const EdgeGeom& g, int k, int count, const EdgeBuffers& b, const u32* carry, u32* dil_out, hipStream_t stream)
{
    const int strips = (g.words_per_row + DX_STRIP - 1) / DX_STRIP;
    const int bands = (g.height + 4 * DX_R - 1) / (4 * DX_R);
    // frames per walk: about 8 k workgroups per launch (a walk also dilates the frame in front of it, so not below 8 frames)
    int walk = (int)(((long)count * bands * strips + 8191) / 8192);
    walk = walk < 8 ? 8 : walk > 64 ? 64 : walk;
    const dim3 grid(bands * strips, (count + walk - 1) / walk);
#define PSD_DX_CASE(K) case K: hipLaunchKernelGGL(dilate_xor_kernel<K>, grid, dim3(256), 0, stream, b.strong, g, k, strips, walk, count, carry, b.xr, dil_out); break;
    switch (k) {
        PSD_DX_CASE(3) PSD_DX_CASE(5) PSD_DX_CASE(7) PSD_DX_CASE(9) PSD_DX_CASE(11) PSD_DX_CASE(13) PSD_DX_CASE(15) PSD_DX_CASE(17)
        PSD_DX_CASE(19) PSD_DX_CASE(21) PSD_DX_CASE(23) PSD_DX_CASE(25) PSD_DX_CASE(27) PSD_DX_CASE(29) PSD_DX_CASE(31) PSD_DX_CASE(33) PSD_DX_CASE(35)
        default: hipLaunchKernelGGL(dilate_xor_kernel<0>, grid, dim3(256), 0, stream, b.strong, g, k, strips, walk, count, carry, b.xr, dil_out);
    }
#undef PSD_DX_CASE
    HIP_TRY(hipGetLastError());
    return PSD_OK;
}

static EdgeGeom make_geom(int height, int width, size_t row_stride, size_t frame_stride)
{
    EdgeGeom g;
    g.height = height; g.width = width; g.npix = (long)height * width;
    g.row_stride = row_stride; g.frame_stride = frame_stride;
    g.words_per_row = (width + 31) / 32;
    return g;
}

// down != nullptr: d_frames / d_prev / row_stride / frame_stride are not used -- height x width is the RESIZED size, the frames are
// down->frames at full size, and the HSV term comes with the front end (d_out gets its SADs; d_seg as always).
int edges_score(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride,
                size_t frame_stride, const uint8_t* d_prev, int edge_kernel, psd_frame_scores* d_out,
                hipStream_t stream, const uint8_t* d_seg, const ScoreParams* hsv, int target_blocks, int* launches, const DownSrc* down)
{
    ScoreParams down_hsv{};
    if (down) {
        d_prev = down->prev;                      // (only its presence counts below)
        down_hsv.out = d_out; down_hsv.seg = d_seg;
        hsv = &down_hsv;
    }
    const EdgeGeom g = make_geom(height, width, row_stride, frame_stride);
    const int k = edge_kernel ? edge_kernel : estimated_kernel_size(width, height);
    const int total = n + (d_prev ? 1 : 0);
    EdgeBuffers b;
    int rc = edge_buffers(e, g, total, &b);
    if (rc != PSD_OK) return rc;
    // the 511 threshold pairs never change: built once, kept for the life of the process (no stack buffer behind an
    // asynchronous copy, no synchronisation on the call path)
    static const int2* tab = [] { int2* t = new int2[511]; threshold_table(t); return t; }();
    // ... and uploaded once per workspace: the table sits at its start, whatever the geometry
    static thread_local const void* tab_in_ws = nullptr;
    if (b.fresh || tab_in_ws != (const void*)b.thr_tab) {
        HIP_TRY(hipMemcpyAsync(b.thr_tab, tab, 511 * sizeof(int2), hipMemcpyHostToDevice, stream));
        tab_in_ws = b.thr_tab;
    }
    const size_t words = (size_t)g.height * g.words_per_row;
    int done = 0;          // virtual frames processed
    bool have_carry = false;
    while (done < total) {
        const int count = std::min(b.cap_frames, total - done);
        ChunkSrc src;
        src.frames = d_frames;
        src.prev = d_prev;
        src.first_is_prev = (done == 0 && d_prev) ? 1 : 0;
        src.start = done - (d_prev ? 1 : 0) + src.first_is_prev;  // batch index of the first non-prev virtual frame
        ScoreParams cp;
        if (hsv) {
            // the HSV term of this chunk's real frames, in V mode: [src.start, src.start + count - first_is_prev) of the batch
            cp = *hsv;
            cp.n = count - src.first_is_prev;
            cp.frames = hsv->frames + (size_t)src.start * frame_stride;
            cp.prev = src.start == 0 ? hsv->prev : hsv->frames + (size_t)(src.start - 1) * frame_stride;
            cp.out = hsv->out + src.start;
            cp.seg = hsv->seg ? hsv->seg + src.start : nullptr;
            cp.vout = b.vplane + (src.first_is_prev ? (size_t)g.npix : 0);
            // the 16-wave fused kernel counts the V histogram itself.  (Alternatives measured and dropped: the 4-wave HSV kernel
            // storing V and a separate kernel counting it, round 3, 2.7 % slower; the 4-wave HSV kernel counting V into per-tile
            // partial histograms, round 4: 2.03-2.06 ms + 0.1 ms for adding the tiles up against 2.0 ms per 1024 x 1080p --
            // the LDS increments cost the same wherever they sit; profiles/r04_f_vmode_front_end_experiments.txt.)
            cp.vhist = b.hist + (src.first_is_prev ? 256 : 0);
        }
        rc = edge_chunk(e, g, src, count, b, stream, hsv ? &cp : nullptr, target_blocks, launches, down);
        if (rc != PSD_OK) return rc;
        rc = launch_dilate_xor(g, k, count, b, have_carry ? b.carry : nullptr, nullptr, stream);
        if (rc != PSD_OK) return rc;
        hipLaunchKernelGGL(store_xor_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, b.xr, count, src.first_is_prev,
                           have_carry ? 1 : 0, src.start, d_out, d_seg);
        HIP_TRY(hipGetLastError());
        // the next chunk compares its first frame with this chunk's last: keep that frame's Canny bits
        HIP_TRY(hipMemcpyAsync(b.carry, b.strong + (size_t)(count - 1) * words, words * 4, hipMemcpyDeviceToDevice, stream));
        have_carry = true;
        done += count;
    }
    return PSD_OK;
}

int edges_map(psd_engine* e, const uint8_t* d_frame, int height, int width, size_t row_stride, int edge_kernel,
              uint8_t* h_edges)
{
    const EdgeGeom g = make_geom(height, width, row_stride, 0);
    const int k = edge_kernel ? edge_kernel : estimated_kernel_size(width, height);
    hipStream_t stream = engine_stream(e);
    EdgeBuffers b;
    int rc = edge_buffers(e, g, 1, &b);
    if (rc != PSD_OK) return rc;
    int2 tab[511];
    threshold_table(tab);
    HIP_TRY(hipMemcpyAsync(b.thr_tab, tab, sizeof(tab), hipMemcpyHostToDevice, stream));
    ChunkSrc src;
    src.frames = d_frame; src.prev = nullptr; src.start = 0; src.first_is_prev = 0;
    rc = edge_chunk(e, g, src, 1, b, stream);
    if (rc != PSD_OK) return rc;
    rc = launch_dilate_xor(g, k, 1, b, nullptr, b.dil1, stream);
    if (rc != PSD_OK) return rc;
    const size_t words = (size_t)g.height * g.words_per_row;
    std::vector<u32> bits(words);
    HIP_TRY(hipMemcpyAsync(bits.data(), b.dil1, words * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            h_edges[(size_t)y * width + x] = (bits[(size_t)y * g.words_per_row + (x >> 5)] >> (x & 31)) & 1u ? 255 : 0;
    return PSD_OK;
}

void edges_release(psd_engine* e)
{
    void** ws = engine_edge_ws(e);
    if (*ws) (void)hipFree(*ws);
    *ws = nullptr;
    *engine_edge_ws_bytes(e) = 0;
}

}  // namespace psd
