// psd_edge_kernels.hip -- the edge term of ContentDetector and the NEAREST / AREA downscale modes, for gfx950.
//
// Edge term (reference scenedetect/detectors/content_detector.py:170-174,213-239):
//     edges_t   = cv2.dilate(cv2.Canny(V_t, low, high), ones(k,k))   with low/high from numpy.median(V_t)
//     delta_edges = mean |edges_t - edges_{t-1}| = 255 * popcount(edges_t XOR edges_{t-1}) / (H*W)
// Device pipeline per chunk of frames (everything integer, so the result is exact):
//   K1 value_plane_hist   V = max(B,G,R) -> u8 plane + per-frame 256-bin histogram (for the median)
//   K2 median_thresholds  exact numpy.median from the histogram -> (low, high) via a host-built table
//   K3 sobel_nms          Sobel 3x3 (replicate border), |dx|+|dy|, non-maximum suppression with
//                         OpenCV's TG22 fixed point -> map {0 none, 1 weak, 2 strong}; LDS tiles with halo
//   K4 hysteresis         8-connected growth of strong into weak: in-LDS fix point per 64x64 tile,
//                         relaunched until no tile changes (the result is order independent)
//   K5 pack_hdilate       strong pixels -> bit rows, horizontal OR over the k-window (bit shifts)
//   K6 vdilate_xor        vertical OR over the k-window, XOR with the previous frame's dilated bits,
//                         popcount -> edge_xor; the dilated bits stay resident for the next frame
// Algorithmic traffic is 5 B/px (3 read + edge map write + previous edge map read);
// the intermediate planes are implementation overhead.
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

// ---- K3: Sobel + non-maximum suppression -------------------------------------------------------

constexpr int NT_W = 64, NT_H = 32;  // tile of the NMS kernel (256 threads x 8 px)

// grid = (tiles_x, tiles_y, frames)
__global__ __launch_bounds__(256) void sobel_nms_kernel(const uint8_t* vplane, EdgeGeom g, const int2* thr, uint8_t* map)
{
    constexpr int SVW = NT_W + 8;                       // V rows cover x0-4 .. x0+67 (dword aligned)
    __shared__ __attribute__((aligned(16))) uint8_t sv[NT_H + 4][SVW];   // rows y0-2 .. y0+NT_H+1
    __shared__ unsigned short smag[NT_H + 2][NT_W + 2 + 2];              // |dx|+|dy|, 1-px halo, 0 outside the image
    __shared__ u32 sdxy[NT_H][NT_W];                                     // (dy << 16) | (dx & 0xffff) of interior pixels
    const int j = blockIdx.z;
    const uint8_t* V = vplane + (size_t)j * g.npix;
    const int x0 = blockIdx.x * NT_W, y0 = blockIdx.y * NT_H;
    const int H = g.height, W = g.width;
    // Load V with the border replicated.  Interior tiles of frames whose width is a multiple of 4
    // use aligned dword loads, everything else clamps per byte.
    const bool inner = (W & 3) == 0 && ((g.npix & 3) == 0) && x0 >= 4 && x0 + NT_W + 4 <= W && y0 >= 2 && y0 + NT_H + 2 <= H;
    if (inner) {
        for (int i = threadIdx.x; i < (NT_H + 4) * (SVW / 4); i += 256) {
            const int ly = i / (SVW / 4), lw = i - ly * (SVW / 4);
            const u32 v = *reinterpret_cast<const u32*>(V + (size_t)(y0 + ly - 2) * W + x0 - 4 + lw * 4);
            *reinterpret_cast<u32*>(&sv[ly][lw * 4]) = v;
        }
    } else {
        for (int i = threadIdx.x; i < (NT_H + 4) * SVW; i += 256) {
            const int ly = i / SVW, lx = i - ly * SVW;
            const int y = min(max(y0 + ly - 2, 0), H - 1), x = min(max(x0 + lx - 4, 0), W - 1);
            sv[ly][lx] = V[(size_t)y * W + x];
        }
    }
    __syncthreads();
    // Separable Sobel down columns: per V row, h1 = right - left and h2 = left + 2*mid + right;
    // dx = h1[-1] + 2 h1[0] + h1[+1], dy = h2[+1] - h2[-1].  One work item = one column of the
    // (NT_W+2)-wide magnitude halo region x a third of its rows.
    // (three row segments: 66 x 3 = 198 work items fit one pass of the 256 threads; four would need a second
    //  pass for 8 stragglers)
    constexpr int MW = NT_W + 2, MH = NT_H + 2, NSEG = 3, QR = (MH + NSEG - 1) / NSEG;
    static_assert(MW * NSEG <= 256, "one pass");
    for (int item = threadIdx.x; item < MW * NSEG; item += 256) {
        const int lx = item % MW, q = item / MW;
        const int r0 = q * QR, r1 = min(MH, r0 + QR);          // magnitude rows [r0, r1)
        // magnitude (ly, lx) is centred on sv[ly + 1][lx + 3]
        const int cx = lx + 3;
        int h1a, h1b, h2a, h2b;
        {
            const int a = sv[r0][cx - 1], b = sv[r0][cx], c = sv[r0][cx + 1];
            h1a = c - a; h2a = a + 2 * b + c;
            const int a1 = sv[r0 + 1][cx - 1], b1 = sv[r0 + 1][cx], c1 = sv[r0 + 1][cx + 1];
            h1b = c1 - a1; h2b = a1 + 2 * b1 + c1;
        }
        const int x = x0 + lx - 1;
        for (int ly = r0; ly < r1; ly++) {
            const int a = sv[ly + 2][cx - 1], b = sv[ly + 2][cx], c = sv[ly + 2][cx + 1];
            const int h1c = c - a, h2c = a + 2 * b + c;
            const int dx = h1a + 2 * h1b + h1c, dy = h2c - h2a;
            const int y = y0 + ly - 1;
            unsigned short m = 0;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                m = (unsigned short)(abs(dx) + abs(dy));
                if (ly >= 1 && ly <= NT_H && lx >= 1 && lx <= NT_W) sdxy[ly - 1][lx - 1] = ((u32)dy << 16) | ((u32)dx & 0xffffu);
            }
            smag[ly][lx] = m;
            h1a = h1b; h1b = h1c; h2a = h2b; h2b = h2c;
        }
    }
    __syncthreads();
    const int low = thr[j].x, high = thr[j].y;
    for (int i = threadIdx.x; i < NT_H * NT_W; i += 256) {
        const int ly = i / NT_W, lx = i - ly * NT_W;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        const int m = smag[ly + 1][lx + 1];
        uint8_t out = 0;
        if (m > low) {
            const u32 pk = sdxy[ly][lx];
            const int xs = (int)(short)(pk & 0xffffu), ys = (int)pk >> 16;
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int tg22x = ax * 13573;  // TG22 = round(tan(22.5 deg) * 2^15)
            bool is_max;
            if (ay < tg22x) {
                is_max = m > smag[ly + 1][lx] && m >= smag[ly + 1][lx + 2];
            } else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) {
                    is_max = m > smag[ly][lx + 1] && m >= smag[ly + 2][lx + 1];
                } else {
                    const int s = (xs ^ ys) < 0 ? -1 : 1;
                    is_max = m > smag[ly][lx + 1 - s] && m > smag[ly + 2][lx + 1 + s];
                }
            }
            if (is_max) out = m > high ? 2 : 1;
        }
        map[(size_t)j * g.npix + (size_t)y * W + x] = out;
    }
}

// K3, dword form (frames whose width and pixel count are multiples of 4: 1080p, 4K, 720p, 256x144 ...).
// 128 x 32 tiles.  Phase 1 computes only the magnitude |dx| + |dy| of the tile plus a 1-px ring, four pixels of a row
// per work item as two packed 16-bit pairs (even / odd columns), rolling down the rows of a 5-row segment: per row three
// dword reads of V, two v_perm + two masks to get the four column pairs, packed adds / multiply-adds for the separable
// Sobel sums.  Phase 2 suppresses non-maxima: a pixel whose magnitude is not above `low` is done after one read (almost
// all of a natural frame); the others recompute dx, dy from the 3x3 V neighbourhood, classify the direction with OpenCV's
// TG22 fixed point and compare with the two neighbours along it.  Results leave as dwords.  Same integers as
// sobel_nms_kernel, a quarter of its instructions.
constexpr int HT_TILE = 64;   // = HT, the hysteresis tile edge (defined below)
constexpr int N2_W = 128, N2_H = 32, N2_SVW = N2_W + 8, N2_SEG = 5, N2_NSEG = 7, N2_NCG = N2_SVW / 4;
static_assert(N2_NCG * N2_NSEG <= 256 && N2_SEG * N2_NSEG >= N2_H + 2, "one pass of 256 threads covers the ring");

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// packed 16-bit arithmetic through clang's vector types (v_pk_add_u16 / v_pk_sub_i16 / v_pk_max_i16 without inline asm,
// which hipcc would pad with s_nop and could not schedule across)
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return __builtin_bit_cast(u32, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ u32 pk_sub(u32 a, u32 b) { return __builtin_bit_cast(u32, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
// a + 2 b on both halves
__device__ __forceinline__ u32 pk_add2(u32 a, u32 b) { return pk_add(a, pk_add(b, b)); }
__device__ __forceinline__ u32 pk_abs(u32 a) { return pk_max(a, pk_sub(0u, a)); }

// grid = (tiles_x, tiles_y, frames)
// `dirty` (zeroed by the caller): one flag per 64x64 hysteresis tile of the frame, set where this tile leaves a weak pixel --
// the first hysteresis launch then only looks at tiles that can change at all (a natural frame has few of them).
__global__ __launch_bounds__(256) void sobel_nms_dword_kernel(const uint8_t* vplane, EdgeGeom g, const int2* thr, uint8_t* map,
                                                              uint8_t* dirty, int htiles_x, int htiles_per_frame)
{
    __shared__ __attribute__((aligned(16))) uint8_t sv[N2_H + 4][N2_SVW];          // rows y0-2 .. y0+33, cols x0-4 .. x0+131
    __shared__ __attribute__((aligned(16))) unsigned short smag[N2_H + 2][N2_SVW]; // rows y0-1 .. y0+32, same columns
    const int j = blockIdx.z, tid = threadIdx.x;
    const uint8_t* V = vplane + (size_t)j * g.npix;
    const int x0 = blockIdx.x * N2_W, y0 = blockIdx.y * N2_H, H = g.height, W = g.width;
    const bool inner = x0 >= 4 && x0 + N2_W + 4 <= W && y0 >= 2 && y0 + N2_H + 2 <= H;
    if (inner) {
        // all five loads before the first LDS write (as a loop, each load is waited for before the next is issued)
        constexpr int NLD = ((N2_H + 4) * N2_NCG + 255) / 256;
        u32 v[NLD];
#pragma unroll
        for (int it = 0; it < NLD; it++) {
            const int i = min(tid + it * 256, (N2_H + 4) * N2_NCG - 1);
            const int ly = i / N2_NCG, lw = i - ly * N2_NCG;
            v[it] = *reinterpret_cast<const u32*>(V + (size_t)(y0 + ly - 2) * W + x0 - 4 + lw * 4);
        }
#pragma unroll
        for (int it = 0; it < NLD; it++) {
            const int i = tid + it * 256;
            const int ly = i / N2_NCG, lw = i - ly * N2_NCG;
            if (i < (N2_H + 4) * N2_NCG) *reinterpret_cast<u32*>(&sv[ly][lw * 4]) = v[it];
        }
    } else {   // border tiles: replicate (BORDER_REPLICATE of cv2.Sobel inside Canny)
        for (int i = tid; i < (N2_H + 4) * N2_SVW; i += 256) {
            const int ly = i / N2_SVW, lx = i - ly * N2_SVW;
            const int y = min(max(y0 + ly - 2, 0), H - 1), x = min(max(x0 + lx - 4, 0), W - 1);
            sv[ly][lx] = V[(size_t)y * W + x];
        }
    }
    __syncthreads();
    // ---- phase 1: magnitudes of smag rows [5 seg, 5 seg + 5), columns 4 cg .. 4 cg + 3
    if (tid < N2_NCG * N2_NSEG) {
        const int cg = tid % N2_NCG, seg = tid / N2_NCG;
        const int m0 = seg * N2_SEG, m1 = min(m0 + N2_SEG, N2_H + 2);
        const int cl = max(cg - 1, 0) * 4, cm = cg * 4, cr = min(cg + 1, N2_NCG - 1) * 4;   // clamped neighbours only feed unused columns
        // columns of the four outputs in the image, for the "magnitude outside the image is 0" rule
        const int xb = x0 - 4 + cm;
        const u32 keepE = ((xb >= 0 && xb < W) ? 0xffffu : 0u) | ((xb + 2 >= 0 && xb + 2 < W) ? 0xffff0000u : 0u);
        const u32 keepO = ((xb + 1 >= 0 && xb + 1 < W) ? 0xffffu : 0u) | ((xb + 3 >= 0 && xb + 3 < W) ? 0xffff0000u : 0u);
        u32 h1E[3], h1O[3], h2E[3], h2O[3];
        auto horiz = [&](int r, int slot) {
            const u32 wl = *reinterpret_cast<const u32*>(&sv[r][cl]), wm = *reinterpret_cast<const u32*>(&sv[r][cm]),
                      wr = *reinterpret_cast<const u32*>(&sv[r][cr]);
            // pixels a0 = wl.b3, a1..a4 = wm, a5 = wr.b0; pairs P0 = (a0,a2) P1 = (a1,a3) P2 = (a2,a4) P3 = (a3,a5)
            const u32 P1 = wm & 0x00ff00ffu, P2 = (wm >> 8) & 0x00ff00ffu;
            const u32 P0 = __builtin_amdgcn_perm(wm, wl, 0x0c050c03u);   // [wl.b3, 0, wm.b1, 0]
            const u32 P3 = __builtin_amdgcn_perm(wr, wm, 0x0c040c02u);   // [wm.b2, 0, wr.b0, 0]
            h1E[slot] = pk_sub(P2, P0); h1O[slot] = pk_sub(P3, P1);      // right - left
            h2E[slot] = pk_add(pk_add2(P0, P1), P2); h2O[slot] = pk_add(pk_add2(P1, P2), P3);   // left + 2 mid + right
        };
        horiz(m0, 0);
        horiz(m0 + 1, 1);
#pragma unroll
        for (int k = 0; k < N2_SEG; k++) {
            const int my = m0 + k;
            if (my < m1) {
                const int a = k % 3, b = (k + 1) % 3, c = (k + 2) % 3;
                horiz(my + 2, c);
                const u32 dxE = pk_add(pk_add2(h1E[a], h1E[b]), h1E[c]), dxO = pk_add(pk_add2(h1O[a], h1O[b]), h1O[c]);
                const u32 dyE = pk_sub(h2E[c], h2E[a]), dyO = pk_sub(h2O[c], h2O[a]);
                u32 mE = pk_add(pk_abs(dxE), pk_abs(dyE)), mO = pk_add(pk_abs(dxO), pk_abs(dyO));
                const int y = y0 - 1 + my;
                if (y < 0 || y >= H) { mE = 0; mO = 0; }
                mE &= keepE; mO &= keepO;
                uint2 out;
                out.x = __builtin_amdgcn_perm(mO, mE, 0x05040100u);    // mag[c], mag[c+1]
                out.y = __builtin_amdgcn_perm(mO, mE, 0x07060302u);    // mag[c+2], mag[c+3]
                *reinterpret_cast<uint2*>(&smag[my][cm]) = out;
            }
        }
    }
    __syncthreads();
    // ---- phase 2: non-maximum suppression, four pixels of a row per thread and step
    const int low = thr[j].x, high = thr[j].y;
    for (int i = tid; i < N2_H * (N2_W / 4); i += 256) {
        const int ly = i / (N2_W / 4), q = i - ly * (N2_W / 4);
        const int y = y0 + ly, xq = x0 + 4 * q;
        if (y >= H || xq >= W) continue;
        const uint2 m4 = *reinterpret_cast<const uint2*>(&smag[ly + 1][4 + 4 * q]);
        const int mm[4] = {(int)(m4.x & 0xffffu), (int)(m4.x >> 16), (int)(m4.y & 0xffffu), (int)(m4.y >> 16)};
        u32 packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int m = mm[k];
            if (m > low) {
                const int r = ly + 2, c = 4 + 4 * q + k;       // centre in sv; magnitude at smag[ly + 1][c]
                const int tl = sv[r - 1][c - 1], tc = sv[r - 1][c], tr = sv[r - 1][c + 1], ml = sv[r][c - 1], mr = sv[r][c + 1],
                          bl = sv[r + 1][c - 1], bc = sv[r + 1][c], br = sv[r + 1][c + 1];
                const int xs = (tr + 2 * mr + br) - (tl + 2 * ml + bl), ys = (bl + 2 * bc + br) - (tl + 2 * tc + tr);
                const int ax = abs(xs), ay = abs(ys) << 15;
                const int tg22x = ax * 13573;  // TG22 = round(tan(22.5 deg) * 2^15)
                bool is_max;
                if (ay < tg22x) {
                    is_max = m > smag[ly + 1][c - 1] && m >= smag[ly + 1][c + 1];
                } else if (ay > tg22x + (ax << 16)) {
                    is_max = m > smag[ly][c] && m >= smag[ly + 2][c];
                } else {
                    const int sgn = (xs ^ ys) < 0 ? -1 : 1;
                    is_max = m > smag[ly][c - sgn] && m > smag[ly + 2][c + sgn];
                }
                if (is_max) packed |= (m > high ? 2u : 1u) << (8 * k);
            }
        }
        *reinterpret_cast<u32*>(map + (size_t)j * g.npix + (size_t)y * W + xq) = packed;
        // a byte equal to 1 (weak): (packed ^ 0x01010101) has a zero byte there
        const u32 z = packed ^ 0x01010101u;
        if ((z - 0x01010101u) & ~z & 0x80808080u) dirty[(size_t)j * htiles_per_frame + (size_t)(y / HT_TILE) * htiles_x + xq / HT_TILE] = 1;
    }
}

// K3, full-width tiles (frames whose width is a multiple of the 128-px tile: 1080p, 4K, 720p, 640x360, 256x144 ...).
// The same arithmetic as sobel_nms_dword_kernel with the per-wave instruction count cut by a third (the kernel is VALU
// bound: 540 VALU instructions per wave, 85 % VALU-busy in profiles/r02_h_pmc_edges.txt):
//   * tile load: two V rows per wave instruction (lanes 0-31 / 32-63 take the 32 dwords of a row, 72 threads the two dwords
//     left over per row), row index clamped, the replicated column outside a left / right image edge made from the
//     neighbouring dword -- six loads with a handful of address instructions instead of five (interior) or nineteen
//     (border tiles: 18 % of a 1080p frame, byte by byte before) rounds of divide-by-34 indexing;
//   * phase 1 is specialised for interior tiles (no "outside the image" masks) and leaves the magnitudes as the even /
//     odd column pairs it computed them in (no re-interleave);
//   * phase 2 takes eight pixels per thread and decides "no candidate among them" with four packed max + xor.
__device__ __forceinline__ int mag_at(const uint2 (*smq)[N2_NCG], int row, int c)
{
    // column c of a magnitude row: quad c >> 2 holds {m0 | m2 << 16, m1 | m3 << 16}
    const unsigned short* q = reinterpret_cast<const unsigned short*>(&smq[row][c >> 2]);
    return q[((c & 1) << 1) | ((c >> 1) & 1)];
}

template <bool INNER>
__device__ __forceinline__ void sobel_phase1_quads(const uint8_t (*sv)[N2_SVW], uint2 (*smq)[N2_NCG], int tid, int x0, int y0, int H, int W)
{
    if (tid >= N2_NCG * N2_NSEG) return;
    const int cg = tid % N2_NCG, seg = tid / N2_NCG;
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
    // pixel pair and row instead of 13 half-rate ones).  The magnitudes that leave are the same 16-bit integers as before.
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    auto H2 = [](u32 a) { return __builtin_bit_cast(h16x2, a); };
    auto U = [](h16x2 a) { return __builtin_bit_cast(u32, a); };
    const h16x2 two = {(_Float16)2.0f, (_Float16)2.0f};
    h16x2 h1E[3], h1O[3], h2E[3], h2O[3];
    auto horiz = [&](int r, int slot) {
        const u32 wl = *reinterpret_cast<const u32*>(&sv[r][cl]), wm = *reinterpret_cast<const u32*>(&sv[r][cm]),
                  wr = *reinterpret_cast<const u32*>(&sv[r][cr]);
        const h16x2 P1 = H2(wm & 0x00ff00ffu), P2 = H2((wm >> 8) & 0x00ff00ffu);
        const h16x2 P0 = H2(__builtin_amdgcn_perm(wm, wl, 0x0c050c03u));   // [wl.b3, 0, wm.b1, 0]
        const h16x2 P3 = H2(__builtin_amdgcn_perm(wr, wm, 0x0c040c02u));   // [wm.b2, 0, wr.b0, 0]
        h1E[slot] = P2 - P0; h1O[slot] = P3 - P1;                                        // right - left
        h2E[slot] = __builtin_elementwise_fma(P1, two, P0) + P2;                         // left + 2 mid + right
        h2O[slot] = __builtin_elementwise_fma(P2, two, P1) + P3;
    };
    horiz(m0, 0);
    horiz(m0 + 1, 1);
#pragma unroll
    for (int k = 0; k < N2_SEG; k++) {
        const int my = m0 + k;
        if (my < m1) {
            const int a = k % 3, b = (k + 1) % 3, c = (k + 2) % 3;
            horiz(my + 2, c);
            const h16x2 dxE = __builtin_elementwise_fma(h1E[b], two, h1E[a]) + h1E[c], dxO = __builtin_elementwise_fma(h1O[b], two, h1O[a]) + h1O[c];
            const h16x2 dyE = h2E[c] - h2E[a], dyO = h2O[c] - h2O[a];
            u32 mE = U(H2(U(dxE) & 0x7fff7fffu) + H2(U(dyE) & 0x7fff7fffu)), mO = U(H2(U(dxO) & 0x7fff7fffu) + H2(U(dyO) & 0x7fff7fffu));
            if (!INNER) {
                const int y = y0 - 1 + my;
                if (y < 0 || y >= H) { mE = 0; mO = 0; }
                mE &= keepE; mO &= keepO;
            }
            smq[my][cg] = make_uint2(mE, mO);
        }
    }
}

// grid = (W / 128, tiles_y, frames); requires W % 128 == 0
__global__ __launch_bounds__(256) void sobel_nms_tile_kernel(const uint8_t* vplane, EdgeGeom g, const int2* thr, uint8_t* map,
                                                             uint8_t* dirty, int htiles_x, int htiles_per_frame)
{
    __shared__ __attribute__((aligned(16))) uint8_t sv[N2_H + 4][N2_SVW];   // rows y0-2 .. y0+33, cols x0-4 .. x0+131
    __shared__ __attribute__((aligned(16))) uint2 smq[N2_H + 2][N2_NCG];    // rows y0-1 .. y0+32, magnitudes as even / odd pairs
    const int j = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* V = vplane + (size_t)j * g.npix;
    const int x0 = blockIdx.x * N2_W, y0 = blockIdx.y * N2_H, H = g.height, W = g.width;
    const int2 lohi = thr[j];   // read here: behind the barriers its latency would sit in front of phase 2
    {
        const int cw = lane & 31;
        const int left = (x0 == 0 && cw == 0) ? 1 : 0;
        const uint8_t* col = V + x0 - 4 + 4 * (cw + left);
        // all six loads go out before the first LDS write (the loads are unconditional on a clamped row, only the writes are
        // predicated: behind a branch hipcc waits for every load before issuing the next -- six memory latencies per tile)
        u32 v[5];
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int ly = min(2 * (wave * 5 + it) + (lane >> 5), N2_H + 3);
            v[it] = *reinterpret_cast<const u32*>(col + (size_t)min(max(y0 + ly - 2, 0), H - 1) * W);
        }
        const int tl = min(tid >> 1, N2_H + 3), c2 = 32 + (tid & 1);
        const int right = (x0 + N2_W == W && c2 == 33) ? 1 : 0;
        u32 vt = *reinterpret_cast<const u32*>(V + (size_t)min(max(y0 + tl - 2, 0), H - 1) * W + x0 - 4 + 4 * (c2 - right));
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int ly = 2 * (wave * 5 + it) + (lane >> 5);
            if (ly < N2_H + 4) *reinterpret_cast<u32*>(&sv[ly][4 * cw]) = left ? __builtin_amdgcn_perm(v[it], v[it], 0u) : v[it];   // BORDER_REPLICATE: V[y][0] four times
        }
        if (tid < 2 * (N2_H + 4)) *reinterpret_cast<u32*>(&sv[tl][4 * c2]) = right ? __builtin_amdgcn_perm(vt, vt, 0x03030303u) : vt;   // V[y][W-1] four times
    }
    __syncthreads();
    const bool inner = x0 >= 4 && x0 + N2_W + 4 <= W && y0 >= 2 && y0 + N2_H + 2 <= H;
    if (inner) sobel_phase1_quads<true>(sv, smq, tid, x0, y0, H, W);
    else sobel_phase1_quads<false>(sv, smq, tid, x0, y0, H, W);
    __syncthreads();
    // ---- phase 2: non-maximum suppression, eight pixels of a row per thread and step
    const int low = lohi.x, high = lohi.y;
    const u32 lowpk = (u32)low * 0x10001u;
#pragma unroll
    for (int it = 0; it < N2_H * (N2_W / 8) / 256; it++) {
        const int i = tid + it * 256;
        const int ly = i >> 4, o = i & 15;
        const int y = y0 + ly;
        if (y >= H) continue;
        const uint2 qa = smq[ly + 1][1 + 2 * o], qb = smq[ly + 1][2 + 2 * o];
        // a half above `low` survives the max: (max(m, low) ^ low) != 0
        const u32 any = (pk_max(qa.x, lowpk) ^ lowpk) | (pk_max(qa.y, lowpk) ^ lowpk) | (pk_max(qb.x, lowpk) ^ lowpk) | (pk_max(qb.y, lowpk) ^ lowpk);
        u32 packed[2] = {0u, 0u};
        if (any) {
            const u32 w4[4] = {qa.x, qa.y, qb.x, qb.y};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int m = (int)((w4[2 * (k >> 2) + (k & 1)] >> (16 * ((k >> 1) & 1))) & 0xffffu);
                if (m > low) {
                    const int r = ly + 2, c = 4 + 8 * o + k;       // centre in sv; magnitude row ly + 1, column c
                    const int tl = sv[r - 1][c - 1], tc = sv[r - 1][c], tr = sv[r - 1][c + 1], ml = sv[r][c - 1], mr = sv[r][c + 1],
                              bl = sv[r + 1][c - 1], bc = sv[r + 1][c], br = sv[r + 1][c + 1];
                    const int xs = (tr + 2 * mr + br) - (tl + 2 * ml + bl), ys = (bl + 2 * bc + br) - (tl + 2 * tc + tr);
                    const int ax = abs(xs), ay = abs(ys) << 15;
                    const int tg22x = ax * 13573;  // TG22 = round(tan(22.5 deg) * 2^15)
                    bool is_max;
                    if (ay < tg22x) {
                        is_max = m > mag_at(smq, ly + 1, c - 1) && m >= mag_at(smq, ly + 1, c + 1);
                    } else if (ay > tg22x + (ax << 16)) {
                        is_max = m > mag_at(smq, ly, c) && m >= mag_at(smq, ly + 2, c);
                    } else {
                        const int sgn = (xs ^ ys) < 0 ? -1 : 1;
                        is_max = m > mag_at(smq, ly, c - sgn) && m > mag_at(smq, ly + 2, c + sgn);
                    }
                    if (is_max) packed[k >> 2] |= (m > high ? 2u : 1u) << (8 * (k & 3));
                }
            }
        }
        const int xq = x0 + 8 * o;
        *reinterpret_cast<uint2*>(map + (size_t)j * g.npix + (size_t)y * W + xq) = make_uint2(packed[0], packed[1]);
        if (any) {
            // a byte equal to 1 (weak): (packed ^ 0x01010101) has a zero byte there
            const u32 z0 = packed[0] ^ 0x01010101u, z1 = packed[1] ^ 0x01010101u;
            if ((((z0 - 0x01010101u) & ~z0) | ((z1 - 0x01010101u) & ~z1)) & 0x80808080u)
                dirty[(size_t)j * htiles_per_frame + (size_t)(y / HT_TILE) * htiles_x + xq / HT_TILE] = 1;
        }
    }
}

// ---- K4: hysteresis -----------------------------------------------------------------------------

constexpr int HT = 64;  // hysteresis tile edge
static_assert(HT == HT_TILE, "");

// grid = ceil(tiles / tpw) workgroups, each responsible for tpw <= 64 consecutive tiles of the (frame, tile_y, tile_x) order
// (16: sweep on 256 x 1080p, shot-like / all-dirty noise: 64 -> 1.27 / 4.94 ms per call, 32 -> 1.22 / 4.61, 16 -> 1.23 / 4.41,
// 8 -> 1.25 / 4.37; one workgroup per tile, as before: 1.30 / 4.36).
// Grows strong (2) into 8-connected weak (1) pixels inside a tile until nothing changes, using the neighbouring
// tiles' current state as a read-only halo.
// Work list: a tile is only looked at when `dirty_in` says it holds weak pixels (first launch: flags from the NMS kernel)
// or one of its neighbours promoted a pixel on the shared border in the previous launch; when it promotes border pixels
// itself it marks its neighbours in `dirty_out` and raises *changed.  A natural frame leaves a few per cent of the tiles
// dirty: one wave reads the flags of the workgroup's range at once and the workgroup walks the set bits -- a launch over
// 130 k tiles (256 x 1080p) is 8 k workgroups instead of 130 k that exit at once (30 us per launch, nine launches per call).
__global__ __launch_bounds__(256) void hysteresis_kernel(uint8_t* map, EdgeGeom g, uint8_t* dirty_in, uint8_t* dirty_out,
                                                         int* changed, int tiles_x, int tiles_y, long n_tiles, int tpw)
{
    // tile + 1-px halo; image column x0 sits at LDS column OX + 1 = 4 so interior rows are dword aligned
    constexpr int OX = 3;
    __shared__ __attribute__((aligned(16))) uint8_t t[HT + 2][HT + 8];
    __shared__ int any_weak, tile_changed, border_changed, round_changed;
    __shared__ unsigned long long todo_mask;
    const long first = (long)blockIdx.x * tpw;
    if (threadIdx.x < 64) {
        const long ti = first + threadIdx.x;
        const bool mine = (int)threadIdx.x < tpw && ti < n_tiles;
        const bool set = mine && dirty_in[ti] != 0;
        const unsigned long long m = __ballot(set);
        if (threadIdx.x == 0) todo_mask = m;
        // consumed: leave the list empty, it is the next launch's dirty_out (nobody else touches this range of dirty_in)
        if (set) dirty_in[ti] = 0;
    }
    __syncthreads();
    unsigned long long todo = todo_mask;
    const int H = g.height, W = g.width;
    const int per_frame = tiles_x * tiles_y;
  while (todo) {
    const int bit = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const long ti = first + bit;
    const int j = (int)(ti / per_frame), rem = (int)(ti - (long)j * per_frame);
    const int by = rem / tiles_x, bx = rem - by * tiles_x;
    const size_t tile_base = (size_t)j * per_frame;
    uint8_t* M = map + (size_t)j * g.npix;
    const int x0 = bx * HT, y0 = by * HT;
    __syncthreads();   // the previous tile of this workgroup is done with t[] and the flags
    if (threadIdx.x == 0) { any_weak = 0; tile_changed = 0; border_changed = 0; }
    __syncthreads();
    int weak_here = 0;
    // Tiles whose 64 columns are all inside a frame with 4-byte aligned rows load their interior as dwords
    // (16 per row) and only the two halo columns as bytes; everything else goes byte by byte.
    const bool wide = (W & 3) == 0 && (g.npix & 3) == 0 && x0 + HT <= W;
    if (wide) {
        // all loads of the tile (five dwords and one halo byte per thread) go out before the first LDS write: unconditional
        // on clamped coordinates, the out-of-image zero applied afterwards (a load behind a branch is waited for before the
        // next one is issued: six memory latencies per dirty tile)
        constexpr int NDW = ((HT + 2) * (HT / 4) + 255) / 256;
        static_assert((HT + 2) * 2 <= 256, "");
        u32 dw[NDW];
#pragma unroll
        for (int it = 0; it < NDW; it++) {
            const int i = min((int)threadIdx.x + it * 256, (HT + 2) * (HT / 4) - 1);
            const int ly = i / (HT / 4), lw = i - ly * (HT / 4);
            const int y = y0 + ly - 1;
            dw[it] = *reinterpret_cast<const u32*>(M + (size_t)min(max(y, 0), H - 1) * W + x0 + lw * 4);
        }
        const int hi = min((int)threadIdx.x, (HT + 2) * 2 - 1);
        const int hly = hi >> 1, hside = hi & 1;
        const int hy = y0 + hly - 1, hx = hside ? x0 + HT : x0 - 1;
        uint8_t hb = M[(size_t)min(max(hy, 0), H - 1) * W + min(max(hx, 0), W - 1)];
#pragma unroll
        for (int it = 0; it < NDW; it++) {
            const int i = (int)threadIdx.x + it * 256;
            if (i < (HT + 2) * (HT / 4)) {
                const int ly = i / (HT / 4), lw = i - ly * (HT / 4);
                const int y = y0 + ly - 1;
                const u32 v = (y >= 0 && y < H) ? dw[it] : 0u;
                *reinterpret_cast<u32*>(&t[ly][OX + 1 + lw * 4]) = v;
                // a byte equal to 1 inside the tile rows: (v ^ 0x01010101) has a zero byte there
                if (ly >= 1 && ly <= HT) {
                    const u32 z = v ^ 0x01010101u;
                    if ((z - 0x01010101u) & ~z & 0x80808080u) weak_here = 1;
                }
            }
        }
        if ((int)threadIdx.x < (HT + 2) * 2)
            t[hly][hside ? OX + HT + 1 : OX] = (hy >= 0 && hy < H && hx >= 0 && hx < W) ? hb : (uint8_t)0;
    } else {
        for (int i = threadIdx.x; i < (HT + 2) * (HT + 2); i += 256) {
            const int ly = i / (HT + 2), lx = i - ly * (HT + 2);
            const int y = y0 + ly - 1, x = x0 + lx - 1;
            uint8_t v = 0;
            if (y >= 0 && y < H && x >= 0 && x < W) v = M[(size_t)y * W + x];
            t[ly][OX + lx] = v;
            if (v == 1 && ly >= 1 && ly <= HT && lx >= 1 && lx <= HT) weak_here = 1;
        }
    }
    if (weak_here) any_weak = 1;
    __syncthreads();
    if (!any_weak) continue;
    // each thread owns a 4x4 patch of the 64x64 tile
    const int py = (threadIdx.x >> 4) * 4 + 1, px = (threadIdx.x & 15) * 4 + 1 + OX;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) round_changed = 0;
        __syncthreads();
        int ch = 0, bch = 0;
#pragma unroll
        for (int dy = 0; dy < 4; dy++)
#pragma unroll
            for (int dx = 0; dx < 4; dx++) {
                const int y = py + dy, x = px + dx;
                if (t[y][x] == 1) {
                    const int s = (t[y - 1][x - 1] | t[y - 1][x] | t[y - 1][x + 1] | t[y][x - 1] | t[y][x + 1] |
                                   t[y + 1][x - 1] | t[y + 1][x] | t[y + 1][x + 1]) & 2;
                    if (s) {
                        t[y][x] = 2;  // racing writers only ever store 2: benign
                        ch = 1;
                        if (y == 1 || y == HT || x == OX + 1 || x == OX + HT) bch = 1;
                    }
                }
            }
        if (ch) round_changed = 1;
        if (bch) border_changed = 1;
        __syncthreads();
        if (!round_changed) break;
        if (threadIdx.x == 0) tile_changed = 1;
    }
    __syncthreads();
    if (tile_changed) {
        for (int i = threadIdx.x; i < HT * HT; i += 256) {
            const int ly = i / HT, lx = i - ly * HT;
            const int y = y0 + ly, x = x0 + lx;
            if (y < H && x < W && t[ly + 1][OX + lx + 1] == 2) M[(size_t)y * W + x] = 2;
        }
        if (border_changed && threadIdx.x < 9 && threadIdx.x != 4) {
            const int ny = by + (int)threadIdx.x / 3 - 1, nx = bx + (int)threadIdx.x % 3 - 1;
            if (ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x) dirty_out[tile_base + (size_t)ny * tiles_x + nx] = 1;
        }
        if (border_changed && threadIdx.x == 0) atomicOr(changed, 1);
    }
  }
}

// ---- K5: pack strong pixels into bit rows + horizontal dilation ---------------------------------

// grid = (height, frames); one block per image row.  anchor = k/2; window [x-anchor, x-anchor+k-1].
__global__ __launch_bounds__(256) void pack_hdilate_kernel(const uint8_t* map, EdgeGeom g, int k, u32* hbits)
{
    extern __shared__ u32 rowbits[];  // words_per_row + 2 (one zero word of padding on each side)
    const int j = blockIdx.y, y = blockIdx.x, W = g.width, nw = g.words_per_row;
    const uint8_t* M = map + (size_t)j * g.npix + (size_t)y * W;
    const bool wide = (((uintptr_t)M) & 15) == 0;
    for (int w = threadIdx.x; w < nw + 2; w += 256) {
        u32 bits = 0;
        if (w >= 1 && w <= nw) {
            const int xb = (w - 1) * 32;
            if (wide && xb + 32 <= W) {
                // 32 map bytes (values 0,1,2): bit 1 marks "strong"; gather one bit per byte with a multiply
                const uint4 a = *reinterpret_cast<const uint4*>(M + xb), c = *reinterpret_cast<const uint4*>(M + xb + 16);
                const u32 q[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
                for (int i = 0; i < 8; i++) bits |= ((((q[i] >> 1) & 0x01010101u) * 0x01020408u) >> 24) << (4 * i);
            } else {
#pragma unroll 8
                for (int b = 0; b < 32; b++) {
                    const int x = xb + b;
                    if (x < W && M[x] == 2) bits |= 1u << b;
                }
            }
        }
        rowbits[w] = bits;
    }
    __syncthreads();
    const int left = k / 2, right = k - 1 - k / 2;  // output x is set if any input in [x-left, x+right] is set
    for (int w = threadIdx.x; w < nw; w += 256) {
        const u32 lo = rowbits[w], mid = rowbits[w + 1], hi = rowbits[w + 2];
        u32 out = mid;
        // input at x+s (s>0) contributes: shift the (mid,hi) pair right by s
        for (int s = 1; s <= right; s++) out |= s < 32 ? (u32)((((u64)hi << 32) | mid) >> s) : (hi >> (s - 32));
        // input at x-s contributes: shift the (lo,mid) pair left by s
        for (int s = 1; s <= left; s++) out |= s < 32 ? (u32)(((((u64)mid << 32) | lo) << s) >> 32) : (lo << (s - 32));
        const int xb = w * 32;
        if (xb + 32 > W) out &= (W - xb >= 32) ? 0xffffffffu : ((1u << (W - xb)) - 1u);
        hbits[((size_t)j * g.height + y) * nw + w] = out;
    }
}

// ---- K6: vertical dilation, then XOR count ------------------------------------------------------

// grid = (ceil(words_per_frame/256), frames).  dil[j] = OR of hbits rows [y-anchor, y-anchor+k-1].
__global__ __launch_bounds__(256) void vdilate_kernel(const u32* hbits, EdgeGeom g, int k, u32* dil)
{
    const int j = blockIdx.y, nw = g.words_per_row;
    const long words = (long)g.height * nw;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= words) return;
    const int y = (int)(i / nw), w = (int)(i - (long)y * nw);
    const int ytop = max(y - k / 2, 0), ybot = min(y - k / 2 + k - 1, g.height - 1);
    const u32* hb = hbits + (size_t)j * words;
    u32 v = 0;
    for (int yy = ytop; yy <= ybot; yy++) v |= hb[(size_t)yy * nw + w];
    dil[(size_t)j * words + i] = v;
}

// K5 + K6 in one kernel: a workgroup takes a band of `band` image rows, packs the strong pixels of the rows it needs
// (band + the k/2 rows above and below) into bit rows in LDS, dilates them horizontally (second LDS array), then ORs
// the k-window vertically and writes the band's dilated bit rows.  The bit rows never travel through HBM and a frame is
// ~17-34 workgroups instead of one per image row.  grid = (bands, frames); dynamic LDS = 2 * (band + k - 1) * (nw + 2) words.
__global__ __launch_bounds__(256) void pack_dilate_kernel(const uint8_t* map, EdgeGeom g, int k, int band, u32* dil)
{
    extern __shared__ u32 pd_lds[];
    const int j = blockIdx.y, W = g.width, H = g.height, nw = g.words_per_row, pitch = nw + 2;
    const int y0 = blockIdx.x * band, y1 = min(H, y0 + band);
    const int up = k / 2, dn = k - 1 - k / 2;
    const int r0 = max(0, y0 - up), r1 = min(H, y1 + dn);          // rows [r0, r1) are needed
    const int nrows = r1 - r0;
    u32* raw = pd_lds;                                             // [nrows][pitch], one zero word on each side
    u32* hd = pd_lds + (size_t)(band + k - 1) * pitch;             // [nrows][pitch] horizontally dilated
    const uint8_t* M = map + (size_t)j * g.npix;
    // rows by thread group, words by lane within the group (a power of two >= the row pitch, at most a wave): no division by
    // the run-time pitch in the three loops
    const int sh = pitch <= 8 ? 3 : pitch <= 16 ? 4 : pitch <= 32 ? 5 : 6;
    const int lane = threadIdx.x & ((1 << sh) - 1), wave = threadIdx.x >> sh, nwave = 256 >> sh, lstep = 1 << sh;
    // four rows per step with all eight 16-byte loads issued before the first use: the loop is bound by load latency
    // (one row per step: 18 dependent round trips per thread, 43 us per workgroup)
    for (int rr0 = wave; rr0 < nrows; rr0 += 4 * nwave)
    for (int w = lane; w < pitch; w += lstep) {
        const int xb = (w - 1) * 32;
        const bool inside = w >= 1 && w <= nw;
        uint4 qa[4], qc[4];
        bool fastp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int rr = rr0 + u * nwave;
            const uint8_t* row = M + (size_t)(r0 + min(rr, nrows - 1)) * W;
            fastp[u] = inside && rr < nrows && xb + 32 <= W && ((((uintptr_t)row) + xb) & 15) == 0;
            // unconditional loads (a lane without a fast word reads the start of the map, always mapped and aligned): behind a
            // branch hipcc puts an s_waitcnt vmcnt(0) in front of every pair and the eight loads go out one pair at a time
            const uint8_t* src = fastp[u] ? row + xb : map;
            qa[u] = *reinterpret_cast<const uint4*>(src);
            qc[u] = *reinterpret_cast<const uint4*>(src + 16);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int rr = rr0 + u * nwave;
            if (rr >= nrows) break;
            u32 bits = 0;
            if (fastp[u]) {
                // 32 map bytes (values 0,1,2): bit 1 marks "strong".  v_dot4_u32_u8 gathers the bits: the bytes (0 or 2 after
                // the mask) of two dwords times the weights 1..128 sum to twice the output byte (a 32-bit multiply per dword,
                // the usual gather, runs at quarter rate here)
                const u32 q[8] = {qa[u].x, qa[u].y, qa[u].z, qa[u].w, qc[u].x, qc[u].y, qc[u].z, qc[u].w};
                u32 d[4];
#pragma unroll
                for (int t = 0; t < 4; t++)
                    d[t] = __builtin_amdgcn_udot4(q[2 * t + 1] & 0x02020202u, 0x80402010u,
                                                  __builtin_amdgcn_udot4(q[2 * t] & 0x02020202u, 0x08040201u, 0u, false), false);
                // every d is even and <= 510: its bit 8 falls on the (zero) bit 0 of the next one, d[3] << 23 stays below 2^32
                bits = ((d[0] | (d[1] << 8) | (d[2] << 16)) >> 1) | (d[3] << 23);
            } else if (inside) {
                const uint8_t* row = M + (size_t)(r0 + rr) * W;
                for (int b = 0; b < 32; b++) {
                    const int x = xb + b;
                    if (x < W && row[x] == 2) bits |= 1u << b;
                }
            }
            raw[rr * pitch + w] = bits;
        }
    }
    __syncthreads();
    const int left = k / 2, right = k - 1 - k / 2;  // output x is set if any input in [x-left, x+right] is set
    for (int rr = wave; rr < nrows; rr += nwave)
    for (int w = lane; w < nw; w += lstep) {
        const u32 lo = raw[rr * pitch + w], mid = raw[rr * pitch + w + 1], hi = raw[rr * pitch + w + 2];
        // OR of the shifts 0..n by doubling (x |= x >> 1, >> 2, >> 4 ... then one shift for the remainder) on the 64-bit
        // pairs (hi:mid) and (mid:lo): 3 + 3 steps for k = 13 instead of 12 funnel shifts; windows reach at most one word
        // across (k <= 63, enforced in psd_engine.cpp, keeps left and right below 32)
        // (as two 32-bit halves with v_alignbit_b32 funnel shifts: 64-bit shifts run at quarter rate)
        u32 rl = mid, rh = hi, lh = mid, ll = lo;
        int cover = 1;
        for (; 2 * cover <= right + 1; cover *= 2) { rl |= __builtin_amdgcn_alignbit(rh, rl, cover); rh |= rh >> cover; }
        if (cover < right + 1) rl |= __builtin_amdgcn_alignbit(rh, rl, right + 1 - cover);
        cover = 1;
        for (; 2 * cover <= left + 1; cover *= 2) { lh |= __builtin_amdgcn_alignbit(lh, ll, 32 - cover); ll |= ll << cover; }
        if (cover < left + 1) lh |= __builtin_amdgcn_alignbit(lh, ll, 32 - (left + 1 - cover));
        u32 out = rl | lh;
        const int xb = w * 32;
        if (xb + 32 > W) out &= (1u << (W - xb)) - 1u;
        hd[rr * pitch + w] = out;
    }
    __syncthreads();
    for (int ly = wave; ly < y1 - y0; ly += nwave)
    for (int w = lane; w < nw; w += lstep) {
        const int y = y0 + ly;
        const int ytop = max(y - up, 0), ybot = min(y + dn, H - 1);
        // four independent partial ORs: a single chain waits for one LDS round trip per row of the window
        u32 v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        const u32* col = hd + (ytop - r0) * pitch + w;
        const int nwin = ybot - ytop + 1;
        int q = 0;
        for (; q + 4 <= nwin; q += 4) {
            v0 |= col[q * pitch]; v1 |= col[(q + 1) * pitch]; v2 |= col[(q + 2) * pitch]; v3 |= col[(q + 3) * pitch];
        }
        for (; q < nwin; q++) v0 |= col[q * pitch];
        dil[((size_t)j * H + y) * nw + w] = (v0 | v1) | (v2 | v3);
    }
}

// Number of pixels whose dilated edge bit differs between virtual frame j and its predecessor
// (dil[j-1], or `carry` = last frame of the previous chunk for j == 0).
__global__ __launch_bounds__(256) void xor_count_kernel(const u32* dil, long words, const u32* carry, int have_carry,
                                                         unsigned long long* out_xor)
{
    const int j = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    u32 diff = 0;
    if (i < words) {
        const u32 v = dil[(size_t)j * words + i];
        if (j > 0) diff = __popc(v ^ dil[(size_t)(j - 1) * words + i]);
        else if (have_carry) diff = __popc(v ^ carry[i]);
    }
    __shared__ u32 red[4];
    for (int o = 32; o > 0; o >>= 1) diff += __shfl_down(diff, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = diff;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 s = red[0] + red[1] + red[2] + red[3];
        if (s) atomicAdd(&out_xor[j], (unsigned long long)s);
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
    uint8_t* vplane; uint8_t* map; u32* hist; int2* thr; u32* hbits; u32* dil; u32* carry;
    unsigned long long* xr; int* flags; int2* thr_tab; uint8_t* dirty[2];
    int cap_frames; size_t tiles_per_frame;
    uint8_t* zero_begin; size_t zero_bytes;   // hist, xr, both tile lists and the flags: cleared by ONE memset per chunk
    bool fresh;                               // the workspace was (re)allocated by this call
};

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

static int edge_buffers(psd_engine* e, const EdgeGeom& g, int want_frames, EdgeBuffers* b)
{
    const size_t words = (size_t)g.height * g.words_per_row;
    const size_t tiles = (size_t)((g.width + HT - 1) / HT) * ((g.height + HT - 1) / HT);
    const size_t per_frame = align_up((size_t)g.npix) * 2 + align_up(256 * 4) + align_up(sizeof(int2)) +
                             align_up(words * 4) * 2 + align_up(8) + 2 * align_up(tiles);
    // bound the workspace (default 4 GiB of the 288 GB; PSD_EDGE_WS_MB overrides) unless a single frame needs
    // more: longer chunks amortise the host round trips of the hysteresis rounds
    static const size_t ws_cap = [] {
        const char* e = getenv("PSD_EDGE_WS_MB");
        const long mb = e ? atol(e) : 4096;
        return (size_t)(mb > 0 ? mb : 4096) << 20;
    }();
    int frames = (int)std::max<size_t>(1, std::min<size_t>((size_t)want_frames, ws_cap / per_frame));
    const size_t fixed = align_up(words * 4) + align_up(64 * sizeof(int)) + align_up(511 * sizeof(int2));
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
    } else {
        // the cached workspace may fit more frames than computed for `want_frames`; keep `frames`
    }
    uint8_t* p = (uint8_t*)*ws;
    auto take = [&](size_t bytes) { uint8_t* r = p; p += align_up(bytes); return r; };
    b->thr_tab = (int2*)take(511 * sizeof(int2));   // first: its place does not depend on the geometry (uploaded once)
    b->vplane = take((size_t)g.npix * frames);
    b->map = take((size_t)g.npix * frames);
    b->thr = (int2*)take((size_t)frames * sizeof(int2));
    b->hbits = (u32*)take(words * 4 * frames);
    b->dil = (u32*)take(words * 4 * frames);
    b->carry = (u32*)take(words * 4);
    b->zero_begin = p;
    b->hist = (u32*)take((size_t)frames * 256 * 4);
    b->xr = (unsigned long long*)take((size_t)frames * 8);
    b->dirty[0] = take(tiles * frames);
    b->dirty[1] = take(tiles * frames);
    b->flags = (int*)take(64 * sizeof(int));
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

// after the fixed rounds of the speculative path: did the last launch still pass something on?
__global__ void note_unconverged_kernel(const int* last_flag, int* unconverged)
{
    if (*last_flag) atomicOr(unconverged, 1);
}

constexpr int HYST_SPEC_LAUNCHES = 6;   // hysteresis launches of the speculative path: chains that cross up to 5 tile borders

// Runs K1..K6a for `count` virtual frames; on return (stream-ordered) b.dil holds their dilated bits.
// d_unconverged == nullptr: the hysteresis is relaunched until a launch changes nothing, the host reading the flags
// after every round (exact, blocks the calling thread).  Otherwise: HYST_SPEC_LAUNCHES launches are enqueued without
// any host round trip and *d_unconverged is set if the last one still promoted pixels on a tile border -- the caller
// then repeats the work on the exact path (weak-edge chains that long are rare: tests/test_gpu_fullsize.py builds one).
// hsv != nullptr: the chunk's real frames get their V plane and V histogram from the HSV pass in V mode (one read of the
// frames for both terms, psd_score_kernels.hip); only a predecessor frame standing in as virtual frame 0 still goes
// through the V-plane kernel.
static int edge_chunk(const EdgeGeom& g, const ChunkSrc& src, int count, int k, const EdgeBuffers& b, hipStream_t stream,
                      int* d_unconverged = nullptr, const ScoreParams* hsv = nullptr, int target_blocks = 0, int* launches = nullptr)
{
    const size_t words = (size_t)g.height * g.words_per_row;
    HIP_TRY(hipMemsetAsync(b.zero_begin, 0, b.zero_bytes, stream));   // histograms, XOR counters, both tile lists, flags
    // packed 16-byte aligned frames with a multiple of 16 pixels take the LDS-DMA streaming variant
    static const bool vp_direct = [] { const char* e = getenv("PSD_EDGE_VP_DIRECT"); return e && atoi(e) != 0; }();
    const bool vp_dma = !vp_direct && g.row_stride == (size_t)g.width * 3 && (g.npix & 15) == 0 && (g.frame_stride & 15) == 0 &&
                        ((uintptr_t)src.frames & 15) == 0 && (!src.first_is_prev || ((uintptr_t)src.prev & 15) == 0);
    const int vp_count = hsv ? (src.first_is_prev ? 1 : 0) : count;   // virtual frames the V-plane kernel converts
    if (vp_count > 0 && vp_dma) {
        const int n_groups = (int)(g.npix >> 4);
        const int cap = VD_STEP * (vp_count >= 32 ? VD_STEPS_PER_TILE : 1);
        const int tiles = (n_groups + cap - 1) / cap;
        const int groups_per_tile = (n_groups + tiles - 1) / tiles;
        hipLaunchKernelGGL(value_plane_hist_dma_kernel, dim3(tiles, vp_count), dim3(256), 0, stream, src, g, groups_per_tile, b.vplane, b.hist);
    } else if (vp_count > 0) {
        hipLaunchKernelGGL(value_plane_hist_kernel, dim3((unsigned)((g.npix + 1024 * VP_ITER - 1) / (1024 * VP_ITER)), vp_count), dim3(256), 0, stream, src, g,
                           b.vplane, b.hist);
    }
    if (hsv) HIP_TRY(launch_score_frames(*hsv, true, false, true, target_blocks, stream, launches));
    hipLaunchKernelGGL(median_thresholds_kernel, dim3(count), dim3(256), 0, stream, b.hist, g.npix, b.thr_tab, b.thr);
    static const bool nms_bytes = [] { const char* e = getenv("PSD_EDGE_NMS_BYTES"); return e && atoi(e) != 0; }();
    const dim3 hgrid((g.width + HT - 1) / HT, (g.height + HT - 1) / HT, count);
    const size_t dirty_bytes = b.tiles_per_frame * (size_t)count;
    const long hyst_tiles = (long)hgrid.x * hgrid.y * count;
    static const int hyst_tpw = [] { const char* e = getenv("PSD_EDGE_HYST_TPW"); const int v = e ? atoi(e) : 16; return v >= 1 && v <= 64 ? v : 16; }();
    const dim3 hyst_grid((unsigned)((hyst_tiles + hyst_tpw - 1) / hyst_tpw));
    if ((g.width & 3) == 0 && (g.npix & 3) == 0 && !nms_bytes) {
        // the dword kernel reports which hysteresis tiles hold weak pixels: only those are looked at
        static const bool no_tile = [] { const char* e = getenv("PSD_EDGE_NMS_DWORD"); return e && atoi(e) != 0; }();
        if (g.width % N2_W == 0 && !no_tile)
            hipLaunchKernelGGL(sobel_nms_tile_kernel, dim3(g.width / N2_W, (g.height + N2_H - 1) / N2_H, count), dim3(256), 0, stream,
                               b.vplane, g, b.thr, b.map, b.dirty[0], (int)hgrid.x, (int)b.tiles_per_frame);
        else
        hipLaunchKernelGGL(sobel_nms_dword_kernel, dim3((g.width + N2_W - 1) / N2_W, (g.height + N2_H - 1) / N2_H, count), dim3(256), 0,
                           stream, b.vplane, g, b.thr, b.map, b.dirty[0], (int)hgrid.x, (int)b.tiles_per_frame);
    } else {
        HIP_TRY(hipMemsetAsync(b.dirty[0], 1, dirty_bytes, stream));
        hipLaunchKernelGGL(sobel_nms_kernel, dim3((g.width + NT_W - 1) / NT_W, (g.height + NT_H - 1) / NT_H, count), dim3(256), 0,
                           stream, b.vplane, g, b.thr, b.map);
    }
    HIP_TRY(hipGetLastError());
    // hysteresis to the fix point: rounds of R launches (ping-pong dirty-tile lists, one flag per
    // launch); done when a launch promoted nothing on any tile border.
    // every launch empties the list it read, so the two lists only need the clearing at the top of the chunk
    int launch = 0;
    if (d_unconverged) {
        for (; launch < HYST_SPEC_LAUNCHES; launch++) {
            hipLaunchKernelGGL(hysteresis_kernel, hyst_grid, dim3(256), 0, stream, b.map, g, b.dirty[launch & 1], b.dirty[(launch + 1) & 1],
                               b.flags + launch, (int)hgrid.x, (int)hgrid.y, hyst_tiles, hyst_tpw);
        }
        hipLaunchKernelGGL(note_unconverged_kernel, dim3(1), dim3(1), 0, stream, b.flags + HYST_SPEC_LAUNCHES - 1, d_unconverged);
    } else {
        bool converged = false;
        for (int round = 0; round < 4096 && !converged; round++) {
            constexpr int R = 3;
            // a fresh flag per launch out of the 64 cleared at the top of the chunk (re-cleared every 21 rounds)
            const int fbase = (round % (64 / R)) * R;
            if (round > 0 && fbase == 0) HIP_TRY(hipMemsetAsync(b.flags, 0, 64 * sizeof(int), stream));
            for (int i = 0; i < R; i++, launch++) {
                hipLaunchKernelGGL(hysteresis_kernel, hyst_grid, dim3(256), 0, stream, b.map, g, b.dirty[launch & 1], b.dirty[(launch + 1) & 1],
                                   b.flags + fbase + i, (int)hgrid.x, (int)hgrid.y, hyst_tiles, hyst_tpw);
            }
            int flags[R];
            HIP_TRY(hipMemcpyAsync(flags, b.flags + fbase, sizeof(flags), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            for (int i = 0; i < R; i++) converged = converged || !flags[i];   // a launch with nothing to pass on ends it
        }
        if (!converged) {
            psd_set_error("edge hysteresis did not converge within %d launches", launch);
            return PSD_ERR_HIP;
        }
    }
    // band height of the fused pack + dilate kernel: as tall as two LDS arrays of (band + k - 1) bit rows allow within ~56 KiB
    const size_t pitch = (size_t)g.words_per_row + 2;
    int band = (int)((56u << 10) / (2 * pitch * sizeof(u32))) - (k - 1);
    static const bool split_dilate = [] { const char* e = getenv("PSD_EDGE_SPLIT_DILATE"); return e && atoi(e) != 0; }();
    if (band >= 8 && !split_dilate) {
        if (band > 64) band = 64;
        if (band > g.height) band = g.height;
        hipLaunchKernelGGL(pack_dilate_kernel, dim3((g.height + band - 1) / band, count), dim3(256),
                           2 * (size_t)(band + k - 1) * pitch * sizeof(u32), stream, b.map, g, k, band, b.dil);
    } else {
        hipLaunchKernelGGL(pack_hdilate_kernel, dim3(g.height, count), dim3(256), (g.words_per_row + 2) * sizeof(u32), stream, b.map, g,
                           k, b.hbits);
        hipLaunchKernelGGL(vdilate_kernel, dim3((unsigned)((words + 255) / 256), count), dim3(256), 0, stream, b.hbits, g, k, b.dil);
    }
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

int edges_score(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride,
                size_t frame_stride, const uint8_t* d_prev, int edge_kernel, psd_frame_scores* d_out,
                hipStream_t stream, const uint8_t* d_seg, int* d_unconverged, const ScoreParams* hsv, int target_blocks,
                int* launches)
{
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
            cp.vhist = b.hist + (src.first_is_prev ? 256 : 0);
        }
        rc = edge_chunk(g, src, count, k, b, stream, d_unconverged, hsv ? &cp : nullptr, target_blocks, launches);
        if (rc != PSD_OK) return rc;
        hipLaunchKernelGGL(xor_count_kernel, dim3((unsigned)((words + 255) / 256), count), dim3(256), 0, stream, b.dil, (long)words,
                           b.carry, have_carry ? 1 : 0, b.xr);
        hipLaunchKernelGGL(store_xor_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, b.xr, count, src.first_is_prev,
                           have_carry ? 1 : 0, src.start, d_out, d_seg);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(b.carry, b.dil + (size_t)(count - 1) * words, words * 4, hipMemcpyDeviceToDevice, stream));
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
    rc = edge_chunk(g, src, 1, k, b, stream);
    if (rc != PSD_OK) return rc;
    const size_t words = (size_t)g.height * g.words_per_row;
    std::vector<u32> bits(words);
    HIP_TRY(hipMemcpyAsync(bits.data(), b.dil, words * 4, hipMemcpyDeviceToHost, stream));
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

// ---- cv2.resize(INTER_NEAREST) and cv2.resize(INTER_AREA), 8-bit, 3 channels ---------------------------
// The other two `Interpolation` modes SceneManager can be asked to downscale with (reference common.py:148-160,
// scene_manager.py:670-678).  One destination pixel per thread: correctness paths, not tuned (the default,
// INTER_LINEAR, lives in psd_resize_kernels.hip).

__global__ __launch_bounds__(64) void resize_nearest_kernel(const uint8_t* src, int sh, int sw, size_t sstride, uint8_t* dst, int dh,
                                                            int dw, size_t dstride, const int* xofs, const int* yofs)
{
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    const uint8_t* p = src + (size_t)blockIdx.z * sstride + ((size_t)yofs[dy] * sw + xofs[dx]) * 3;
    uint8_t* D = dst + (size_t)blockIdx.z * dstride + ((size_t)dy * dw + dx) * 3;
    D[0] = p[0]; D[1] = p[1]; D[2] = p[2];
}

struct AreaRun3 {   // same run-length form as psd_hash_kernels.hip
    int first, count, has_head, has_tail;
    float a_head, a_mid, a_tail;
    int pad;
};

// mode 0: float run tables (ResizeArea_<uchar,float> accumulation order: left to right within a source row, rows
// top to bottom, every product and sum rounded separately); mode 1: integer box * (1.f/area); mode 2: 2x2 rounding shift
__global__ __launch_bounds__(64) void resize_area_kernel(const uint8_t* src, int sw, size_t sstride, uint8_t* dst, int dh, int dw,
                                                         size_t dstride, const AreaRun3* xtab, const AreaRun3* ytab, int mode,
                                                         float inv_area)
{
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    const uint8_t* S = src + (size_t)blockIdx.z * sstride;
    uint8_t* D = dst + (size_t)blockIdx.z * dstride + ((size_t)dy * dw + dx) * 3;
    const AreaRun3 xr = xtab[dx], yr = ytab[dy];
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

static void area_table3(int ssize, int dsize, AreaRun3* tab)
{
    const double scale = 1. / ((double)dsize / ssize);
    for (int dx = 0; dx < dsize; dx++) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        AreaRun3 r;
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

// interpolation: 0 = INTER_NEAREST, 3 = INTER_AREA (cv2's values)
int resize_other(const uint8_t* d_src, int n, int src_h, int src_w, size_t src_frame_stride, uint8_t* d_dst, int dst_h,
                 int dst_w, size_t dst_frame_stride, int interpolation, hipStream_t stream)
{
    if (n == 0) return PSD_OK;
    const dim3 grid((dst_w + 63) / 64, dst_h, n);
    const double scale_x = 1. / ((double)dst_w / src_w), scale_y = 1. / ((double)dst_h / src_h);
    uint8_t* tabs = nullptr;
    hipError_t err = hipSuccess;
    if (interpolation == 0) {
        std::vector<int> ofs((size_t)dst_w + dst_h);
        for (int x = 0; x < dst_w; x++) ofs[x] = std::min((int)floor(x * scale_x), src_w - 1);
        for (int y = 0; y < dst_h; y++) ofs[dst_w + y] = std::min((int)floor(y * scale_y), src_h - 1);
        HIP_TRY(hipMalloc((void**)&tabs, ofs.size() * sizeof(int)));
        err = hipMemcpyAsync(tabs, ofs.data(), ofs.size() * sizeof(int), hipMemcpyHostToDevice, stream);
        if (err == hipSuccess) {
            hipLaunchKernelGGL(resize_nearest_kernel, grid, dim3(64), 0, stream, d_src, src_h, src_w, src_frame_stride, d_dst, dst_h,
                               dst_w, dst_frame_stride, (const int*)tabs, (const int*)tabs + dst_w);
            err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipStreamSynchronize(stream);   // `ofs` and `tabs` go away below
    } else if (interpolation == 3) {
        if (dst_w > src_w || dst_h > src_h) {
            psd_set_error("INTER_AREA is implemented for decimation only (%dx%d -> %dx%d)", src_w, src_h, dst_w, dst_h);
            return PSD_ERR_UNSUPPORTED;
        }
        const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
        const bool area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
        std::vector<AreaRun3> t((size_t)dst_w + dst_h);
        int mode = 0;
        float inv_area = 0.f;
        if (area_fast) {
            for (int x = 0; x < dst_w; x++) { AreaRun3 r; memset(&r, 0, sizeof r); r.first = x * iscale_x; r.count = iscale_x; t[x] = r; }
            for (int y = 0; y < dst_h; y++) { AreaRun3 r; memset(&r, 0, sizeof r); r.first = y * iscale_y; r.count = iscale_y; t[dst_w + y] = r; }
            mode = (iscale_x == 2 && iscale_y == 2) ? 2 : 1;
            inv_area = 1.f / (float)(iscale_x * iscale_y);
        } else {
            area_table3(src_w, dst_w, t.data());
            area_table3(src_h, dst_h, t.data() + dst_w);
        }
        HIP_TRY(hipMalloc((void**)&tabs, t.size() * sizeof(AreaRun3)));
        err = hipMemcpyAsync(tabs, t.data(), t.size() * sizeof(AreaRun3), hipMemcpyHostToDevice, stream);
        if (err == hipSuccess) {
            hipLaunchKernelGGL(resize_area_kernel, grid, dim3(64), 0, stream, d_src, src_w, src_frame_stride, d_dst, dst_h, dst_w,
                               dst_frame_stride, (const AreaRun3*)tabs, (const AreaRun3*)tabs + dst_w, mode, inv_area);
            err = hipGetLastError();
        }
        if (err == hipSuccess) err = hipStreamSynchronize(stream);
    } else {
        psd_set_error("interpolation %d is not implemented on the device (0 = NEAREST, 1 = LINEAR, 3 = AREA)", interpolation);
        return PSD_ERR_UNSUPPORTED;
    }
    if (tabs) (void)hipFree(tabs);
    if (err != hipSuccess) { psd_set_error("resize failed: %s", hipGetErrorString(err)); return PSD_ERR_HIP; }
    return PSD_OK;
}

}  // namespace psd
