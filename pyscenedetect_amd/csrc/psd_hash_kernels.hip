// psd_hash_kernels.hip -- front half of HashDetector.hash_frame for gfx950 (MI355X, CDNA4).
//
// Reference (paths relative to its tree): scenedetect/detectors/hash_detector.py:125-129
//     gray    = cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY)
//     resized = cv2.resize(gray, (imsize, imsize), interpolation=cv2.INTER_AREA)
// One fused pass reads every BGR byte once (3 B/px algorithmic) and leaves an imsize x imsize 8-bit
// thumbnail per frame; the float DCT / median / Hamming distance of 1 KiB per frame is host work
// (psd_epilogue_hash_*).
//
// Arithmetic follows OpenCV's 8-bit paths exactly:
//   grey      (3735 B + 19235 G + 9798 R + 2^14) >> 15                (RGB2Gray<uchar>, 15-bit coefficients)
//   INTER_AREA, integer scale in both directions: integer box sums, 2x2 -> (s + 2) >> 2, otherwise
//             saturate_cast<uchar>(sum * (1.f / area))                   (ResizeAreaFast)
//   INTER_AREA, fractional scale: float32 run tables (computeResizeAreaTab); per source row
//             buf[dx] = sum_k S[sx_k] * alpha_k accumulated left to right, per destination row
//             sum[dx] = beta_0 buf_0 + beta_1 buf_1 + ... top to bottom, every product and sum rounded
//             to float32 separately, result rounded half-to-even                (ResizeArea_<uchar, float>)
// The float sums are evaluated in exactly that order (no reassociation, no fused multiply-add), so the
// thumbnails are bit-identical to the CPU oracle's.
//
// Mapping: one workgroup per (frame, destination row).  It walks the ~H/imsize source rows of its cell row
// in batches of R rows: (A) all threads load the batch with 16-byte loads, convert to grey and park it in
// LDS; (B) thread (r, dx) reduces its run of one grey row; (C) threads dx < imsize fold the batch's row
// partials into the running column sums in row order.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "psd_internal.h"

extern "C" void psd_set_error(const char* fmt, ...);

// Every float product and sum below must round separately, as the x86 code it mirrors does.
#pragma clang fp contract(off)

namespace psd {

typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct HashGeom {
    const uint8_t* frames;
    size_t frame_stride, row_stride;
    int height, width, size;
    int mode;            // 0 = float run tables, 1 = integer box, 2 = 2x2 box
    float inv_area;      // mode 1
    int rows_per_batch;  // R
    int wpad;            // LDS row pitch in bytes (multiple of 16)
};

constexpr int kHashWG = 256;

__device__ __forceinline__ u32 gray_of(u32 b, u32 g, u32 r)
{
    return (__umul24(b, 3735u) + __umul24(g, 19235u) + __umul24(r, 9798u) + (1u << 14)) >> 15;
}

// 16 packed BGR pixels (12 dwords) -> 16 grey bytes (4 dwords)
__device__ __forceinline__ u32x4 gray16(const u32 (&w)[12])
{
    u32 o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int ib = 3 * i, ig = 3 * i + 1, ir = 3 * i + 2;
        const u32 b = (w[ib >> 2] >> ((ib & 3) * 8)) & 0xffu;
        const u32 g = (w[ig >> 2] >> ((ig & 3) * 8)) & 0xffu;
        const u32 r = (w[ir >> 2] >> ((ir & 3) * 8)) & 0xffu;
        o[i >> 2] |= gray_of(b, g, r) << ((i & 3) * 8);
    }
    u32x4 v;
    v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
    return v;
}

template <bool FAST>
__global__ __launch_bounds__(kHashWG) void gray_area_kernel(const HashGeom g, const AreaRun* __restrict__ xtab,
                                                            const AreaRun* __restrict__ ytab,
                                                            uint8_t* __restrict__ thumbs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int R = g.rows_per_batch, S = g.size, tid = threadIdx.x;
    uint8_t* rows = smem;                                               // [R][wpad] grey
    float* part = reinterpret_cast<float*>(smem + (size_t)R * g.wpad);  // [R][S] row partials
    AreaRun* xr_lds = reinterpret_cast<AreaRun*>(part + R * S);         // [S]
    const int dy = blockIdx.x, t = blockIdx.y;
    const uint8_t* frame = g.frames + (size_t)t * g.frame_stride;
    for (int i = tid; i < S; i += kHashWG) xr_lds[i] = xtab[i];
    const AreaRun yr = ytab[dy];
    float fsum = 0.f;
    int isum = 0;
    const int groups_per_row = g.width >> 4;
    for (int j0 = 0; j0 < yr.count; j0 += R) {
        const int nrows = min(R, yr.count - j0);
        // (A) grey rows of this batch -> LDS
        if (FAST) {
            for (int item = tid; item < nrows * groups_per_row; item += kHashWG) {
                const int r = item / groups_per_row, gi = item - r * groups_per_row;
                const u32x4* src = reinterpret_cast<const u32x4*>(frame + (size_t)(yr.first + j0 + r) * g.row_stride + (size_t)gi * 48);
                const u32x4 a = __builtin_nontemporal_load(src), b = __builtin_nontemporal_load(src + 1),
                            c = __builtin_nontemporal_load(src + 2);
                const u32 w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
                *reinterpret_cast<u32x4*>(rows + (size_t)r * g.wpad + gi * 16) = gray16(w);
            }
        } else {
            for (int item = tid; item < nrows * g.width; item += kHashWG) {
                const int r = item / g.width, x = item - r * g.width;
                const uint8_t* s = frame + (size_t)(yr.first + j0 + r) * g.row_stride + (size_t)x * 3;
                rows[(size_t)r * g.wpad + x] = (uint8_t)gray_of(s[0], s[1], s[2]);
            }
        }
        __syncthreads();
        // (B) one run of one row per item, accumulated left to right
        for (int item = tid; item < nrows * S; item += kHashWG) {
            const int r = item / S, dx = item - r * S;
            const AreaRun xr = xr_lds[dx];
            const uint8_t* p = rows + (size_t)r * g.wpad + xr.first;
            if (g.mode == 0) {
                float acc = 0.f;
                int k = 0;
                const int end = xr.count - xr.has_tail;
                if (xr.has_head) { acc = __fmul_rn((float)p[0], xr.a_head); k = 1; }
                for (; k + 4 <= end; k += 4) {  // loads first, then the (ordered) chain
                    const float f0 = (float)p[k], f1 = (float)p[k + 1], f2 = (float)p[k + 2], f3 = (float)p[k + 3];
                    acc = __fadd_rn(acc, __fmul_rn(f0, xr.a_mid));
                    acc = __fadd_rn(acc, __fmul_rn(f1, xr.a_mid));
                    acc = __fadd_rn(acc, __fmul_rn(f2, xr.a_mid));
                    acc = __fadd_rn(acc, __fmul_rn(f3, xr.a_mid));
                }
                for (; k < end; k++) acc = __fadd_rn(acc, __fmul_rn((float)p[k], xr.a_mid));
                if (xr.has_tail) acc = __fadd_rn(acc, __fmul_rn((float)p[end], xr.a_tail));
                part[r * S + dx] = acc;
            } else {
                int acc = 0;
                for (int k = 0; k < xr.count; k++) acc += p[k];
                part[r * S + dx] = __int_as_float(acc);
            }
        }
        __syncthreads();
        // (C) fold the batch into the column sums, top to bottom
        if (tid < S) {
            if (g.mode == 0) {
                for (int r = 0; r < nrows; r++) {
                    const int j = j0 + r;
                    const float beta = (j == 0 && yr.has_head) ? yr.a_head : (j == yr.count - 1 && yr.has_tail) ? yr.a_tail : yr.a_mid;
                    const float term = __fmul_rn(beta, part[r * S + tid]);
                    fsum = j == 0 ? term : __fadd_rn(fsum, term);
                }
            } else {
                for (int r = 0; r < nrows; r++) isum += __float_as_int(part[r * S + tid]);
            }
        }
        __syncthreads();
    }
    if (tid < S) {
        int v;
        if (g.mode == 0) v = __float2int_rn(fsum);
        else if (g.mode == 2) v = (isum + 2) >> 2;
        else v = __float2int_rn(__fmul_rn((float)isum, g.inv_area));
        thumbs[((size_t)t * S + dy) * S + tid] = (uint8_t)min(255, max(0, v));
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Packed rows (row_stride == 3*width, width % 16 == 0): the source rows of a cell row are ONE contiguous byte
// range, so phase (A) can stream it HBM -> LDS with global_load_lds_dwordx4 exactly like the scoring kernel:
// lane i of a wave moves bytes [16 i, 16 i + 16) of each 1 KiB piece (every 128-byte line is touched by one
// instruction; the 48-byte lane stride of the register path tops out near 4.4 TB/s), then picks up its 16
// pixels with three ds_read_b128 at a 12-dword stride.  Each wave owns G slots of 3 KiB and refills them one
// step ahead -- also across the (B)/(C) phases, so the next batch is in flight while this one is reduced.
template <int G>
__global__ __launch_bounds__(kHashWG) void gray_area_dma_kernel(const HashGeom g, const AreaRun* __restrict__ xtab,
                                                                const AreaRun* __restrict__ ytab,
                                                                uint8_t* __restrict__ thumbs)
{
    constexpr int NW = kHashWG / 64;
    constexpr int STEP_GROUPS = G * NW * 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int R = g.rows_per_batch, S = g.size, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* stage = smem;                                              // [G*NW][3072] raw BGR
    uint8_t* rows = smem + G * NW * 3072;                               // [R][width] grey
    float* part = reinterpret_cast<float*>(rows + (size_t)R * g.wpad);  // [R][S] row partials
    AreaRun* xr_lds = reinterpret_cast<AreaRun*>(part + R * S);         // [S]
    const int dy = blockIdx.x, t = blockIdx.y;
    const uint8_t* frame = g.frames + (size_t)t * g.frame_stride;
    for (int i = tid; i < S; i += kHashWG) xr_lds[i] = xtab[i];
    const AreaRun yr = ytab[dy];
    const int gpr = g.width >> 4;                 // 16-pixel groups per row
    const size_t row_bytes = (size_t)g.width * 3;
    float fsum = 0.f;
    int isum = 0;

    // the stream of DMA steps: (batch first row, step within the batch), issued one ahead of its use
    int ij0 = 0, istep = 0;
    bool more = yr.count > 0;
    auto issue_next = [&]() {
        const int nrows = min(R, yr.count - ij0);
        const long limit = (long)nrows * gpr * 48;
        const uint8_t* base = frame + (size_t)(yr.first + ij0) * row_bytes;
#pragma unroll
        for (int k = 0; k < G; k++) {
            const long gfirst = (long)istep * STEP_GROUPS + (long)(k * NW + wave) * 64;
            uint8_t* slot = stage + (size_t)(k * NW + wave) * 3072;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const long off = gfirst * 48 + j * 1024 + lane * 16;
                if (off + 16 <= limit)
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + off), (lds_ptr_t)(slot + j * 1024), 16, 0, PSD_DMA_AUX);
            }
        }
        istep++;
        if ((long)istep * STEP_GROUPS >= (long)nrows * gpr) { istep = 0; ij0 += R; more = ij0 < yr.count; }
    };
    if (more) issue_next();

    for (int j0 = 0; j0 < yr.count; j0 += R) {
        const int nrows = min(R, yr.count - j0);
        const int ngroups = nrows * gpr;
        // (A) staged BGR -> grey rows
        for (int sbase = 0; sbase < ngroups; sbase += STEP_GROUPS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            u32 w[G][12];
#pragma unroll
            for (int k = 0; k < G; k++) {
                const u32x4* src = reinterpret_cast<const u32x4*>(stage + (size_t)(k * NW + wave) * 3072 + lane * 48);
                const u32x4 a = src[0], b = src[1], c = src[2];
                w[k][0] = a.x; w[k][1] = a.y; w[k][2] = a.z; w[k][3] = a.w;
                w[k][4] = b.x; w[k][5] = b.y; w[k][6] = b.z; w[k][7] = b.w;
                w[k][8] = c.x; w[k][9] = c.y; w[k][10] = c.z; w[k][11] = c.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (more) issue_next();
#pragma unroll
            for (int k = 0; k < G; k++) {
                const int gidx = sbase + (k * NW + wave) * 64 + lane;
                if (gidx < ngroups) *reinterpret_cast<u32x4*>(rows + (size_t)gidx * 16) = gray16(w[k]);
            }
        }
        lds_barrier();      // (not __syncthreads(): its vmcnt(0) would drain the next batch's staging, issued a few lines up -- the
                            //  stream that is meant to run under phases (B) and (C))
        // (B) one run of one row per item, accumulated left to right
        for (int item = tid; item < nrows * S; item += kHashWG) {
            const int r = item / S, dx = item - r * S;
            const AreaRun xr = xr_lds[dx];
            const uint8_t* p = rows + (size_t)r * g.wpad + xr.first;
            if (g.mode == 0) {
                float acc = 0.f;
                int k = 0;
                const int end = xr.count - xr.has_tail;
                if (xr.has_head) { acc = __fmul_rn((float)p[0], xr.a_head); k = 1; }
                for (; k + 4 <= end; k += 4) {  // loads first, then the (ordered) chain
                    const float f0 = (float)p[k], f1 = (float)p[k + 1], f2 = (float)p[k + 2], f3 = (float)p[k + 3];
                    acc = __fadd_rn(acc, __fmul_rn(f0, xr.a_mid));
                    acc = __fadd_rn(acc, __fmul_rn(f1, xr.a_mid));
                    acc = __fadd_rn(acc, __fmul_rn(f2, xr.a_mid));
                    acc = __fadd_rn(acc, __fmul_rn(f3, xr.a_mid));
                }
                for (; k < end; k++) acc = __fadd_rn(acc, __fmul_rn((float)p[k], xr.a_mid));
                if (xr.has_tail) acc = __fadd_rn(acc, __fmul_rn((float)p[end], xr.a_tail));
                part[r * S + dx] = acc;
            } else {
                int acc = 0;
                for (int k = 0; k < xr.count; k++) acc += p[k];
                part[r * S + dx] = __int_as_float(acc);
            }
        }
        lds_barrier();
        // (C) fold the batch into the column sums, top to bottom
        if (tid < S) {
            if (g.mode == 0) {
                for (int r = 0; r < nrows; r++) {
                    const int j = j0 + r;
                    const float beta = (j == 0 && yr.has_head) ? yr.a_head : (j == yr.count - 1 && yr.has_tail) ? yr.a_tail : yr.a_mid;
                    const float term = __fmul_rn(beta, part[r * S + tid]);
                    fsum = j == 0 ? term : __fadd_rn(fsum, term);
                }
            } else {
                for (int r = 0; r < nrows; r++) isum += __float_as_int(part[r * S + tid]);
            }
        }
        lds_barrier();
    }
    if (tid < S) {
        int v;
        if (g.mode == 0) v = __float2int_rn(fsum);
        else if (g.mode == 2) v = (isum + 2) >> 2;
        else v = __float2int_rn(__fmul_rn((float)isum, g.inv_area));
        thumbs[((size_t)t * S + dy) * S + tid] = (uint8_t)min(255, max(0, v));
    }
}

// 16-pixel groups per lane and step of the DMA kernel.  Round 6: 1 (was 2) -- half the staging (12 KiB), 12 fewer registers, and with
// the row buffer sized for it SIX workgroups per CU instead of three: 4.58-4.68 -> 4.03-4.05 ms per 4096 x 1080p (0.68-0.69 -> 0.79
// of peak; 4 groups: 16 ms, the registers spill) -- profiles/r06_aa_ab_hash_kernel_occupancy.txt
constexpr int kHashDmaG = 1;

// ---- frames SMALLER than the thumbnail along an axis (a 24-row frame, 32 x 32 thumbnails) -----------------------------------
// cv2.resize(INTER_AREA) that does not shrink along both axes is OpenCV's bilinear kernel with area-mode coefficients
// (resize.cpp), which the engine's cv2.resize path implements for BGR frames (psd_resize_kernels.hip, `area_mode`).  A grey
// image resized on its own equals any channel of the frame (grey, grey, grey) resized, so: grey into all three channels of a
// scratch frame, that resize, channel 0 out.  Three tiny kernels on frames of a few kilobytes; no video has them, the reference
// accepts them.
int resize_linear_score(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_row_stride,
                        size_t src_frame_stride, const uint8_t* d_prev, uint8_t* d_dst, int dst_h, int dst_w,
                        size_t dst_frame_stride, psd_frame_scores* d_out, hipStream_t stream, int* launches, const uint8_t* d_seg,
                        bool area_mode, uint32_t terms);

__global__ __launch_bounds__(256) void gray3_kernel(const uint8_t* __restrict__ frames, size_t frame_stride, size_t row_stride, int h, int w,
                                                    uint8_t* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const uint8_t* s = frames + (size_t)t * frame_stride + (size_t)y * row_stride + (size_t)x * 3;
    const uint8_t g = (uint8_t)gray_of(s[0], s[1], s[2]);
    uint8_t* d = out + ((size_t)t * h * w + i) * 3;
    d[0] = g; d[1] = g; d[2] = g;
}

__global__ __launch_bounds__(256) void first_channel_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = in[i * 3];
}

static int hash_thumbs_enlarging(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride,
                                 size_t frame_stride, int size, uint8_t* d_thumbs, hipStream_t stream, hipEvent_t ev_start)
{
    uint8_t *grey3 = nullptr, *small3 = nullptr;
    const size_t a = (size_t)n * height * width * 3, b = (size_t)n * size * size * 3;
    hipError_t err = hipMalloc((void**)&grey3, a);
    if (err == hipSuccess) err = hipMalloc((void**)&small3, b);
    if (err != hipSuccess) {
        if (grey3) (void)hipFree(grey3);
        psd_set_error("hash thumbnails: scratch of %zu bytes: %s", a + b, hipGetErrorString(err));
        (void)hipGetLastError();
        return err == hipErrorOutOfMemory ? PSD_ERR_NOMEM : PSD_ERR_HIP;
    }
    if (ev_start) (void)hipEventRecord(ev_start, stream);
    int rc = PSD_OK;
    for (int t0 = 0; t0 < n && rc == PSD_OK; t0 += 32768) {
        const int cnt = n - t0 < 32768 ? n - t0 : 32768;
        hipLaunchKernelGGL(gray3_kernel, dim3((height * width + 255) / 256, cnt), dim3(256), 0, stream, d_frames + (size_t)t0 * frame_stride,
                           frame_stride, row_stride, height, width, grey3 + (size_t)t0 * height * width * 3);
    }
    if (hipGetLastError() != hipSuccess) { psd_set_error("gray3_kernel launch failed"); rc = PSD_ERR_HIP; }
    if (rc == PSD_OK)
        rc = resize_linear_score(e, grey3, n, height, width, (size_t)width * 3, (size_t)height * width * 3, nullptr, small3, size, size,
                                 (size_t)size * size * 3, nullptr, stream, nullptr, nullptr, true, 0);
    if (rc == PSD_OK) {
        const long total = (long)n * size * size;
        hipLaunchKernelGGL(first_channel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, small3, d_thumbs, total);
        if (hipGetLastError() != hipSuccess) { psd_set_error("first_channel_kernel launch failed"); rc = PSD_ERR_HIP; }
    }
    (void)hipStreamSynchronize(stream);     // (the scratch goes away here: a path for toy frames, not a hot one)
    (void)hipFree(grey3);
    (void)hipFree(small3);
    return rc;
}

// d_thumbs: device buffer of n*size*size bytes.  The run tables of a (frame shape, size) pair are built once and stay in
// the engine's table cache.
int hash_thumbs(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride, size_t frame_stride,
                int size, uint8_t* d_thumbs, hipStream_t stream, hipEvent_t ev_start)
{
    if (size > width || size > height)
        return hash_thumbs_enlarging(e, d_frames, n, height, width, row_stride, frame_stride, size, d_thumbs, stream, ev_start);
    if (size > kHashWG) {
        psd_set_error("hash thumbnails: size*lowpass = %d exceeds %d", size, kHashWG);
        return PSD_ERR_UNSUPPORTED;
    }
    DevTable tab;
    if (!table_find(e, kTabHashArea, height, width, size, size, &tab)) {
        std::vector<AreaRun> tabs(2 * (size_t)size);
        int mode;
        float inv_area;
        area_tables(height, width, size, size, tabs.data(), &mode, &inv_area);
        const int rc = table_store(e, kTabHashArea, height, width, size, size, tabs.data(), tabs.size() * sizeof(AreaRun), mode, inv_area, &tab);
        if (rc != PSD_OK) return rc;
    }
    const AreaRun* d_tabs = static_cast<const AreaRun*>(tab.ptr);
    HashGeom g;
    memset(&g, 0, sizeof g);
    g.mode = tab.mode;
    g.inv_area = tab.inv_area;
    g.frames = d_frames; g.frame_stride = frame_stride; g.row_stride = row_stride;
    g.height = height; g.width = width; g.size = size;
    g.wpad = (width + 15) & ~15;
    const bool fast = (width % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0) && ((uintptr_t)d_frames % 16 == 0);
    static const bool no_dma = [] { const char* e = getenv("PSD_HASH_DIRECT"); return e && atoi(e) != 0; }();
    const bool dma = fast && row_stride == (size_t)width * 3 && !no_dma;
    // LDS per workgroup: staging (DMA path) + R grey rows + R*size partials + the x run table.  The register path has no
    // staging and may use up to 60 KiB.
    const size_t fixed = (size_t)size * sizeof(AreaRun) + (dma ? (size_t)kHashDmaG * (kHashWG / 64) * 3072 : 0);
    // (DMA path: six workgroups per CU -- 6 rows of a 1080p frame per batch; eight rows, five workgroups: 0.78; five rows: 0.75)
    const size_t budget = dma ? 26 * 1024 : 60 * 1024;
    int R = kHashWG / size;
    while (R > 1 && (size_t)R * g.wpad + (size_t)R * size * 4 + fixed > budget) R--;
    const size_t lds = (size_t)R * g.wpad + (size_t)R * size * 4 + fixed;
    if (lds > 64 * 1024) {
        psd_set_error("hash thumbnails: frame width %d too large for the LDS row buffer", width);
        return PSD_ERR_UNSUPPORTED;
    }
    g.rows_per_batch = R;
    if (ev_start) (void)hipEventRecord(ev_start, stream);  // the timed region is the kernel launches only
    // grid.y is limited to 65535: split long batches
    for (int t0 = 0; t0 < n; t0 += 32768) {
        const int cnt = n - t0 < 32768 ? n - t0 : 32768;
        HashGeom gg = g;
        gg.frames = d_frames + (size_t)t0 * frame_stride;
        uint8_t* out = d_thumbs + (size_t)t0 * size * size;
        const dim3 grid(size, cnt);
        if (dma) hipLaunchKernelGGL(gray_area_dma_kernel<kHashDmaG>, grid, dim3(kHashWG), lds, stream, gg, d_tabs, d_tabs + size, out);
        else if (fast) hipLaunchKernelGGL(gray_area_kernel<true>, grid, dim3(kHashWG), lds, stream, gg, d_tabs, d_tabs + size, out);
        else hipLaunchKernelGGL(gray_area_kernel<false>, grid, dim3(kHashWG), lds, stream, gg, d_tabs, d_tabs + size, out);
    }
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) { psd_set_error("gray_area_kernel launch: %s", hipGetErrorString(err)); return PSD_ERR_HIP; }
    return PSD_OK;
}

// ---- HashDetector.hash_frame, back half, on the device (hash_detector.py:131-151) -------------------------------------------
// One workgroup per frame takes the S x S grey thumbnail where gray_area_dma_kernel left it and produces the K x K hash bits:
// scale by the maximum (float32 division, as numpy's), the orthonormal 2-D DCT-II restricted to the K lowest frequencies per
// axis in FLOAT64 -- the very sums of psd_epilogue_hash_bits (psd_epilogue.cpp), in the same order, every product and sum rounded
// on its own (-ffp-contract=off: v_mul_f64 + v_add_f64), so the bits equal the host epilogue's -- one rounding to float32, the
// float32 median numpy.median computes (middle element, or the float32 mean of the two middle ones) by rank counting, and
// bit = coefficient > median.  268 M double multiply-adds per 4096 frames: nothing beside the 25 GB the thumbnails were made
// from, and the step no longer ends in 1.1 ms of host arithmetic on sixteen threads.
// Dynamic LDS: x float[S * S] | c double[K * S] | tmp double[K * S] | low float[K * K] | scratch.
constexpr int kBitsWG = 256;

__global__ __launch_bounds__(kBitsWG) void hash_bits_kernel(const uint8_t* __restrict__ thumbs, const double* __restrict__ basis, int S, int K,
                                                            uint8_t* __restrict__ bits)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t hb_lds[];
    double* c = reinterpret_cast<double*>(hb_lds);                       // [K][S]
    double* tmp = c + (size_t)K * S;                                      // [K][S]
    float* x = reinterpret_cast<float*>(tmp + (size_t)K * S);             // [S][S]
    float* low = x + (size_t)S * S;                                       // [K][K]
    __shared__ int s_max;
    __shared__ float s_mid[2];
    const int tid = threadIdx.x, t = blockIdx.x;
    const uint8_t* th = thumbs + (size_t)t * S * S;
    if (tid == 0) s_max = 0;
    for (int i = tid; i < K * S; i += kBitsWG) c[i] = basis[i];
    __syncthreads();
    int mx = 0;
    for (int i = tid; i < S * S; i += kBitsWG) mx = max(mx, (int)th[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) atomicMax(&s_max, mx);
    __syncthreads();
    const float fmx = (float)(s_max == 0 ? 1 : s_max);                    // hash_detector.py:132-135
    for (int i = tid; i < S * S; i += kBitsWG) x[i] = __fdiv_rn((float)th[i], fmx);
    __syncthreads();
    // tmp[u][xx] = sum over y (ascending) of c[u][y] * x[y][xx]
    for (int i = tid; i < K * S; i += kBitsWG) {
        const int u = i / S, xx = i - u * S;
        double acc = 0.0;
        for (int y = 0; y < S; y++) acc = __dadd_rn(acc, __dmul_rn(c[(size_t)u * S + y], (double)x[(size_t)y * S + xx]));
        tmp[i] = acc;
    }
    __syncthreads();
    // low[u][v] = (float) sum over xx (ascending) of tmp[u][xx] * c[v][xx]
    for (int i = tid; i < K * K; i += kBitsWG) {
        const int u = i / K, v = i - u * K;
        double acc = 0.0;
        for (int xx = 0; xx < S; xx++) acc = __dadd_rn(acc, __dmul_rn(tmp[(size_t)u * S + xx], c[(size_t)v * S + xx]));
        low[i] = (float)acc;
    }
    __syncthreads();
    // the two middle order statistics by rank counting: value low[i] occupies the sorted positions [#less, #less-or-equal)
    const int m = K * K, ia = (m - 1) / 2, ib = m / 2;
    for (int i = tid; i < m; i += kBitsWG) {
        const float v = low[i];
        int lt = 0, le = 0;
        for (int j = 0; j < m; j++) { const float w = low[j]; lt += w < v; le += w <= v; }
        if (lt <= ia && ia < le) s_mid[0] = v;      // (every thread whose value sits there writes the same value)
        if (lt <= ib && ib < le) s_mid[1] = v;
    }
    __syncthreads();
    const float med = (m & 1) ? s_mid[1] : __fdiv_rn(__fadd_rn(s_mid[0], s_mid[1]), 2.0f);
    uint8_t* b = bits + (size_t)t * m;
    for (int i = tid; i < m; i += kBitsWG) b[i] = low[i] > med ? 1 : 0;
}

// LDS the kernel needs for thumbnails of S x S and a K x K hash, or 0 if that exceeds what a workgroup may have
size_t hash_bits_lds(int S, int K)
{
    const size_t need = 2 * (size_t)K * S * sizeof(double) + (size_t)S * S * sizeof(float) + (size_t)K * K * sizeof(float);
    return need <= 60 * 1024 ? need : 0;
}

int hash_bits(const uint8_t* d_thumbs, int n, int S, int K, const double* d_basis, uint8_t* d_bits, hipStream_t stream)
{
    const size_t lds = hash_bits_lds(S, K);
    if (!lds) { psd_set_error("hash bits on the device: a %d x %d transform of a %d x %d thumbnail does not fit a workgroup's LDS", K, K, S, S); return PSD_ERR_UNSUPPORTED; }
    for (int t0 = 0; t0 < n; t0 += 65535)
        hipLaunchKernelGGL(hash_bits_kernel, dim3(std::min(65535, n - t0)), dim3(kBitsWG), lds, stream, d_thumbs + (size_t)t0 * S * S, d_basis, S, K,
                           d_bits + (size_t)t0 * K * K);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) { psd_set_error("hash_bits_kernel launch: %s", hipGetErrorString(err)); return PSD_ERR_HIP; }
    return PSD_OK;
}

}  // namespace psd
