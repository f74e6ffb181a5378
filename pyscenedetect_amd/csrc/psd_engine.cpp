// psd_engine.cpp -- C-ABI engine around the HIP scoring kernels (include/psd_engine.h).
//
// Owns: one HIP device, one stream, the fixed-point HSV tables in device memory, a small ring
// of record slots (device buffer + pinned host mirror + timing events) so submissions can be
// pipelined, and a pair of device staging buffers for callers that hand over host frames.
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "psd_internal.h"

namespace {
thread_local char g_err[512] = "";
thread_local int g_walk[2] = {0, 0};   // frames per chunk, spatial tiles of the calling thread's last time-walking launch
}

namespace psd {
void note_walk_geometry(int frames_per_chunk, int n_tiles) { g_walk[0] = frames_per_chunk; g_walk[1] = n_tiles; }
}

extern "C" void psd_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            psd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)hipGetLastError(); /* (the failure is reported here: do not leave it for the next launch check) */ \
            return PSD_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

namespace psd {
// psd_edge_kernels.hip
int edges_score(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride,
                size_t frame_stride, const uint8_t* d_prev, int edge_kernel, psd_frame_scores* d_out,
                hipStream_t stream, const uint8_t* d_seg, const ScoreParams* hsv = nullptr, int target_blocks = 0,
                int* launches = nullptr, const DownSrc* down = nullptr);
int edges_map(psd_engine* e, const uint8_t* d_frame, int height, int width, size_t row_stride,
              int edge_kernel, uint8_t* h_edges);
void edges_release(psd_engine* e);
// psd_resize_kernels.hip
int resize_linear_score(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_row_stride,
                        size_t src_frame_stride, const uint8_t* d_prev, uint8_t* d_dst, int dst_h, int dst_w,
                        size_t dst_frame_stride, psd_frame_scores* d_out, hipStream_t stream, int* launches, const uint8_t* d_seg,
                        bool area_mode = false, uint32_t terms = PSD_SCORE_HSV_SAD);
void resize_release(psd_engine* e);
int resize_other(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_frame_stride, uint8_t* d_dst, int dst_h,
                 int dst_w, size_t dst_frame_stride, int interpolation, hipStream_t stream);
int resize_source_rows(int src_h, int src_w, int dst_h, int dst_w, int interpolation, int* rows, int* n_rows);
// psd_feed.cpp
void feed_release(psd_engine* e);
// psd_hash_kernels.hip
int hash_thumbs(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride, size_t frame_stride,
                int size, uint8_t* d_thumbs, hipStream_t stream, hipEvent_t ev_start);
int hash_bits(const uint8_t* d_thumbs, int n, int S, int K, const double* d_basis, uint8_t* d_bits, hipStream_t stream);
size_t hash_bits_lds(int S, int K);
void hash_dct_basis(int size, int keep, double* c);
bool table_find(psd_engine* e, int kind, int sh, int sw, int dh, int dw, DevTable* out);
int table_store(psd_engine* e, int kind, int sh, int sw, int dh, int dw, const void* host, size_t bytes, int mode, float inv_area, DevTable* out);
}  // namespace psd

struct psd_feed;   // psd_feed.cpp

struct psd_slot {
    psd_frame_scores* d_recs = nullptr;
    psd_frame_sums* d_heads = nullptr;   // the records' 40-byte heads packed back to back (heads_only submissions; cap entries)
    uint8_t* h_recs = nullptr;           // pinned mirror: n records, or n 40-byte heads (psd_frame_sums) when heads_only
    size_t h_bytes = 0;
    bool heads_only = false;             // the submission has no luma histogram: only the sums travel to the host
    int cap = 0;
    int n = 0;
    bool pending = false;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_done = nullptr;
    int launches = 0;
    uint8_t* d_seg = nullptr;   // clip-start flags of a segmented submission (device + pinned mirror)
    uint8_t* h_seg = nullptr;
    int seg_cap = 0;
};

struct psd_engine {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    uint32_t* d_lut = nullptr;
    psd_slot slots[PSD_MAX_INFLIGHT];
    int head = 0;   // next slot to submit into
    int tail = 0;   // next slot to collect
    int last_slot = -1;   // most recently collected slot (psd_last_records_device)
    int pending = 0;
    float last_ms = 0.f;
    int last_launches = 0;
    uint8_t* d_stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    void* edge_ws = nullptr;  // owned by psd_edge_kernels.hip
    size_t edge_ws_bytes = 0;
    uint8_t* d_hash = nullptr;  // thumbnails of psd_hash_thumbs*
    size_t hash_bytes = 0;
    uint8_t* d_hbits = nullptr; // hash bits of psd_hash_bits_device
    size_t hbits_bytes = 0;
    uint8_t* d_hdiff = nullptr; // psd_hist_diff_device: hist_diff (double[n])
    size_t hdiff_bytes = 0;
    hipEvent_t ev_hash[2] = {nullptr, nullptr};
    void* resize_cache = nullptr;   // coefficient tables per (src, dst) shape of every resize mode + the hash thumbnails, owned by psd_resize_kernels.hip
    uint8_t* d_small = nullptr;     // resized frames of psd_score_downscaled_* when the terms need them in memory
    size_t small_bytes = 0;
    void* d_hpart = nullptr;        // per (frame, tile) partial luma histograms of the fused downscale + luma pass
    size_t hpart_bytes = 0;
    hipStream_t copy_stream = nullptr;   // psd_upload_async: host -> device copies that overlap the scoring stream
    hipEvent_t ev_copy = nullptr;
    psd_feed* feed = nullptr;     // psd_upload_rows_batch: gather threads + ring of page-locked / device staging segments
};

static void fill_tables(int32_t* sdiv, int32_t* hdiv)
{
    // OpenCV RGB2HSV_b tables: saturate_cast<int>(double) rounds half to even (lrint).
    sdiv[0] = hdiv[0] = 0;
    for (int i = 1; i < 256; i++) {
        sdiv[i] = (int32_t)lrint((255 << 12) / (1. * i));
        hdiv[i] = (int32_t)lrint((180 << 12) / (6. * i));
    }
}

static int ensure_slot(psd_engine* e, psd_slot& s, int n, bool heads_only)
{
    (void)e;
    if (s.cap < n) {
        const int cap = n < 64 ? 64 : n;
        if (s.d_recs) HIP_TRY(hipFree(s.d_recs));
        if (s.d_heads) HIP_TRY(hipFree(s.d_heads));
        s.d_recs = nullptr; s.d_heads = nullptr; s.cap = 0;
        HIP_TRY(hipMalloc((void**)&s.d_recs, (size_t)cap * sizeof(psd_frame_scores)));
        HIP_TRY(hipMalloc((void**)&s.d_heads, (size_t)cap * sizeof(psd_frame_sums)));
        s.cap = cap;
    }
    // the pinned mirror grows with what actually travels: 40 bytes per frame without the histogram, 1064 with it
    // (page-locking 78 MB for 73 k records took longer than scoring them)
    const size_t need = (size_t)(n < 64 ? 64 : n) * (heads_only ? sizeof(psd_frame_sums) : sizeof(psd_frame_scores));
    if (s.h_bytes < need) {
        if (s.h_recs) HIP_TRY(hipHostFree(s.h_recs));
        s.h_recs = nullptr; s.h_bytes = 0;
        HIP_TRY(hipHostMalloc((void**)&s.h_recs, need, hipHostMallocDefault));
        s.h_bytes = need;
    }
    return PSD_OK;
}

extern "C" {

int psd_abi_version(void) { return PSD_ABI_VERSION; }

const char* psd_last_error(void) { return g_err; }

int psd_device_count(int* count)
{
    if (!count) { psd_set_error("psd_device_count: null argument"); return PSD_ERR_INVALID; }
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) {
        *count = 0;
        psd_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(err));
        return PSD_ERR_NO_DEVICE;
    }
    *count = n;
    return PSD_OK;
}

int psd_hsv_tables(int32_t sdiv[256], int32_t hdiv180[256])
{
    if (!sdiv || !hdiv180) { psd_set_error("psd_hsv_tables: null argument"); return PSD_ERR_INVALID; }
    fill_tables(sdiv, hdiv180);
    return PSD_OK;
}

int psd_create(int device, psd_engine** out)
{
    if (!out) { psd_set_error("psd_create: null out"); return PSD_ERR_INVALID; }
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        psd_set_error("psd_create: no HIP device available");
        return PSD_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        psd_set_error("psd_create: device %d out of range (have %d)", device, count);
        return PSD_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device));
    psd_engine* e = new (std::nothrow) psd_engine();
    if (!e) { psd_set_error("psd_create: out of memory"); return PSD_ERR_NOMEM; }
    e->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) e->num_cus = prop.multiProcessorCount;
    int rc = PSD_OK;
    do {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { rc = PSD_ERR_HIP; break; }
        if (hipMalloc((void**)&e->d_lut, 1024 * sizeof(uint32_t)) != hipSuccess) { rc = PSD_ERR_HIP; break; }
        int32_t tab[512];
        fill_tables(tab, tab + 256);
        // The kernels keep both tables pre-shifted by 4 so that S and H land 16-bit aligned in the
        // products (see pixel<> in psd_score_kernels.hip).
        // The HSV-only pass computes in float32 (psd_score_kernels.hip, PSD_HSV_FP32): sdiv / 4096 bumped by one ulp so
        // that the exact .5 ties of diff * sdiv / 4096 round up like (x + 2048) >> 12, and hdiv180 / 4096 (both exact).
        float tabf[512];
        for (int i = 0; i < 256; i++) {
            tabf[i] = tab[i] ? nextafterf((float)tab[i] / 4096.0f, INFINITY) : 0.0f;
            tabf[256 + i] = (float)tab[256 + i] / 4096.0f;
        }
        for (int i = 0; i < 512; i++) tab[i] <<= 4;
        if (hipMemcpy(e->d_lut, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) { rc = PSD_ERR_HIP; break; }
        if (hipMemcpy(e->d_lut + 512, tabf, sizeof(tabf), hipMemcpyHostToDevice) != hipSuccess) { rc = PSD_ERR_HIP; break; }
        for (auto& s : e->slots) {
            if (hipEventCreate(&s.ev_start) != hipSuccess || hipEventCreate(&s.ev_stop) != hipSuccess ||
                hipEventCreate(&s.ev_done) != hipSuccess) { rc = PSD_ERR_HIP; break; }
        }
    } while (0);
    if (rc != PSD_OK) {
        psd_set_error("psd_create: HIP resource creation failed: %s", hipGetErrorString(hipGetLastError()));
        psd_destroy(e);
        return rc;
    }
    *out = e;
    return PSD_OK;
}

void psd_destroy(psd_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    psd::feed_release(e);      // (joins the gather threads, waits for the copy stream's last batch)
    psd::edges_release(e);
    psd::resize_release(e);
    if (e->d_small) (void)hipFree(e->d_small);
    if (e->d_hpart) (void)hipFree(e->d_hpart);
    for (auto& s : e->slots) {
        if (s.d_recs) (void)hipFree(s.d_recs);
        if (s.d_heads) (void)hipFree(s.d_heads);
        if (s.h_recs) (void)hipHostFree(s.h_recs);
        if (s.ev_start) (void)hipEventDestroy(s.ev_start);
        if (s.ev_stop) (void)hipEventDestroy(s.ev_stop);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.d_seg) (void)hipFree(s.d_seg);
        if (s.h_seg) (void)hipHostFree(s.h_seg);
    }
    for (auto& d : e->d_stage) if (d) (void)hipFree(d);
    if (e->d_hash) (void)hipFree(e->d_hash);
    if (e->d_hbits) (void)hipFree(e->d_hbits);
    if (e->d_hdiff) (void)hipFree(e->d_hdiff);
    for (auto& ev : e->ev_hash) if (ev) (void)hipEventDestroy(ev);
    if (e->d_lut) (void)hipFree(e->d_lut);
    if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
    if (e->ev_copy) (void)hipEventDestroy(e->ev_copy);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

static int validate(const void* frames, int n, int height, int width, size_t row_stride, size_t frame_stride,
                    uint32_t flags, int edge_kernel)
{
    if (n < 0 || height <= 0 || width <= 0) {
        psd_set_error("invalid batch shape n=%d height=%d width=%d", n, height, width);
        return PSD_ERR_INVALID;
    }
    if (n > 0 && !frames) { psd_set_error("frames pointer is null"); return PSD_ERR_INVALID; }
    if (row_stride < (size_t)width * 3) {
        psd_set_error("row_stride %zu smaller than width*3 = %zu", row_stride, (size_t)width * 3);
        return PSD_ERR_INVALID;
    }
    if (n > 1 && frame_stride < (size_t)(height - 1) * row_stride + (size_t)width * 3) {
        psd_set_error("frame_stride %zu smaller than one frame", frame_stride);
        return PSD_ERR_INVALID;
    }
    if ((flags & ~(uint32_t)PSD_SCORE_ALL) || flags == 0) {
        psd_set_error("invalid flags 0x%x", flags);
        return PSD_ERR_INVALID;
    }
    if ((flags & PSD_SCORE_EDGES) && edge_kernel != 0 && (edge_kernel < 3 || edge_kernel % 2 == 0)) {
        psd_set_error("kernel_size must be odd integer >= 3");
        return PSD_ERR_INVALID;
    }
    if ((flags & PSD_SCORE_EDGES) && edge_kernel > 63) {
        psd_set_error("kernel_size %d not supported on the device (max 63)", edge_kernel);
        return PSD_ERR_UNSUPPORTED;
    }
    if ((long long)height * width > 0x7fffffffLL / 4) {
        psd_set_error("frame too large");
        return PSD_ERR_UNSUPPORTED;
    }
    return PSD_OK;
}

// One submission = a record slot of the ring: zeroed records, timing events around the kernels, the copy of the
// records into the pinned mirror.  submit_begin / submit_end bracket whatever fills the records.
static int submit_begin(psd_engine* e, int n, uint32_t flags, hipStream_t stream, psd_slot** out)
{
    if (e->pending >= PSD_MAX_INFLIGHT) {
        psd_set_error("too many submissions in flight (max %d); call psd_score_collect", PSD_MAX_INFLIGHT);
        return PSD_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(e->device));
    psd_slot& s = e->slots[e->head];
    // the luma pass fills the histogram whenever it runs (for the byte sum alone too); without it the histogram stays zero
    // and only the 40-byte heads of the records go to the host
    const bool heads_only = !(flags & (PSD_SCORE_LUMA_HIST | PSD_SCORE_BYTE_SUM));
    int rc = ensure_slot(e, s, n, heads_only);
    if (rc != PSD_OK) return rc;
    s.heads_only = heads_only;
    s.n = n;
    s.launches = 0;
    if (n > 0) {
        HIP_TRY(hipMemsetAsync(s.d_recs, 0, (size_t)n * sizeof(psd_frame_scores), stream));
        HIP_TRY(hipEventRecord(s.ev_start, stream));
    }
    *out = &s;
    return PSD_OK;
}

// The 40-byte heads of n records (1064 bytes apart) packed back to back: one thread per 8-byte word.
static_assert(sizeof(psd_frame_sums) == 40 && sizeof(psd_frame_scores) % 8 == 0, "");
__global__ __launch_bounds__(256) void gather_heads_kernel(const psd_frame_scores* __restrict__ recs, unsigned long long* __restrict__ heads, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 5) return;
    const int t = i / 5, k = i - t * 5;
    heads[i] = reinterpret_cast<const unsigned long long*>(recs + t)[k];
}

// whole records of a small submission, word by word (the destination is the pinned mirror: see copy_records_to_host)
__global__ __launch_bounds__(256) void copy_words_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int n_words)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_words) dst[i] = src[i];
}

// device records -> pinned mirror: whole records, or only their 40-byte heads when no histogram was asked for (packed on the
// device first and moved by ONE contiguous copy: the strided hipMemcpy2DAsync this replaces took 134 us for the 29.5 k records
// of the BBC stand-in -- a blit of 40-byte rows -- on the critical path between the kernel and the decisions; now 3 + 25 us)
static int copy_records_to_host(psd_slot& s, hipStream_t stream)
{
    // A few records (the per-frame API submits ONE): PSD_SMALL_COPY = 0 the kernel that packs the heads stores them straight into
    // the pinned mirror (page-locked host memory is device-addressable: no copy engine, no second call; default), 1 one strided
    // hipMemcpy2DAsync (rounds 4-5), 2 the packed form of large submissions.
    static const int small_copy = [] { const char* v = getenv("PSD_SMALL_COPY"); return v ? atoi(v) : 0; }();
    if (s.heads_only && s.n < 256 && small_copy == 0) {
        hipLaunchKernelGGL(gather_heads_kernel, dim3((s.n * 5 + 255) / 256), dim3(256), 0, stream, s.d_recs, (unsigned long long*)s.h_recs, s.n);
        HIP_TRY(hipGetLastError());
    } else if (s.heads_only && s.n < 256 && small_copy == 1) {
        HIP_TRY(hipMemcpy2DAsync(s.h_recs, sizeof(psd_frame_sums), s.d_recs, sizeof(psd_frame_scores), sizeof(psd_frame_sums), (size_t)s.n,
                                 hipMemcpyDeviceToHost, stream));
    } else if (s.heads_only) {
        hipLaunchKernelGGL(gather_heads_kernel, dim3((s.n * 5 + 255) / 256), dim3(256), 0, stream, s.d_recs, (unsigned long long*)s.d_heads, s.n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(s.h_recs, s.d_heads, (size_t)s.n * sizeof(psd_frame_sums), hipMemcpyDeviceToHost, stream));
    } else if (s.n < 256 && small_copy == 0) {
        const int words = s.n * (int)(sizeof(psd_frame_scores) / 8);
        hipLaunchKernelGGL(copy_words_kernel, dim3((words + 255) / 256), dim3(256), 0, stream, (const unsigned long long*)s.d_recs,
                           (unsigned long long*)s.h_recs, words);
        HIP_TRY(hipGetLastError());
    } else
        HIP_TRY(hipMemcpyAsync(s.h_recs, s.d_recs, (size_t)s.n * sizeof(psd_frame_scores), hipMemcpyDeviceToHost, stream));
    return PSD_OK;
}

static int submit_end(psd_engine* e, psd_slot& s, hipStream_t stream)
{
    if (s.n > 0) {
        HIP_TRY(hipEventRecord(s.ev_stop, stream));
        int rc = copy_records_to_host(s, stream);
        if (rc != PSD_OK) return rc;
    }
    HIP_TRY(hipEventRecord(s.ev_done, stream));
    s.pending = true;
    e->head = (e->head + 1) % PSD_MAX_INFLIGHT;
    e->pending++;
    return PSD_OK;
}

// the scoring terms of `flags` for n resident frames, added into the slot's records
static int score_terms(psd_engine* e, psd_slot& s, const uint8_t* d_frames, int n, int height, int width, size_t row_stride,
                       size_t frame_stride, const uint8_t* d_prev, uint32_t flags, int edge_kernel, hipStream_t stream,
                       const uint8_t* d_seg = nullptr)
{
    const bool hsv = flags & PSD_SCORE_HSV_SAD;
    const bool luma = flags & (PSD_SCORE_LUMA_HIST | PSD_SCORE_BYTE_SUM);
    psd::ScoreParams p{};
    bool hsv_with_edges = false;   // the HSV term is computed by the edge term's front end (V mode: one read of the frames)
    if (hsv || luma) {
        p.frames = d_frames;
        p.prev = d_prev;
        p.seg = d_seg;
        p.out = s.d_recs;
        p.lut = e->d_lut;
        p.lutf = e->d_lut + 512;
        p.frame_stride = frame_stride;
        p.row_stride = row_stride;
        p.npix = (long)height * width;
        p.width = width;
        p.n = n;
        const bool fast = row_stride == (size_t)width * 3 && ((uintptr_t)d_frames % 16 == 0) &&
                          (frame_stride % 16 == 0 || n == 1) && (!d_prev || (uintptr_t)d_prev % 16 == 0);
        // ContentDetector with weights.delta_edges > 0 (or a StatsManager): the HSV pass also writes the V plane and the V
        // histogram the edge term starts from (PSD_EDGE_FUSE_HSV=0: two separate reads of the frames)
        static const bool fuse_env = [] { const char* v = getenv("PSD_EDGE_FUSE_HSV"); return !v || atoi(v) != 0; }();
        hsv_with_edges = hsv && (flags & PSD_SCORE_EDGES) && fast && fuse_env && psd::score_v_mode_available(p.npix);
        // ~8 workgroups per CU over the launch keeps the tail short (one 1024-thread WG per CU).
        if (!hsv_with_edges) HIP_TRY(psd::launch_score_frames(p, hsv, luma, fast, e->num_cus * 8, stream, &s.launches));
        else if (luma) HIP_TRY(psd::launch_score_frames(p, false, true, fast, e->num_cus * 8, stream, &s.launches));
    }
    if (flags & PSD_SCORE_EDGES) {
        // (the hysteresis reaches its fix point inside one launch, one workgroup per frame: nothing to check at collect)
        int rc = psd::edges_score(e, d_frames, n, height, width, row_stride, frame_stride, d_prev, edge_kernel, s.d_recs, stream, d_seg,
                                  hsv_with_edges ? &p : nullptr, e->num_cus * 8, &s.launches);
        if (rc != PSD_OK) return rc;
    }
    return PSD_OK;
}

int psd_score_submit_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width,
                            size_t row_stride, size_t frame_stride, const uint8_t* d_prev,
                            uint32_t flags, int edge_kernel, void* stream_)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    int rc = validate(d_frames, n, height, width, row_stride, frame_stride, flags, edge_kernel);
    if (rc != PSD_OK) return rc;
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : e->stream;
    psd_slot* s = nullptr;
    rc = submit_begin(e, n, flags, stream, &s);
    if (rc != PSD_OK) return rc;
    if (n > 0) {
        rc = score_terms(e, *s, d_frames, n, height, width, row_stride, frame_stride, d_prev, flags, edge_kernel, stream, nullptr);
        if (rc != PSD_OK) return rc;
    }
    return submit_end(e, *s, stream);
}

static int check_segments(const int32_t* seg_first, int n_seg, int n);

// clip-start flags of a packed batch -> the slot's device array (through its pinned mirror, on the launch stream)
static int upload_segments(psd_slot& s, int n, const int32_t* seg_first, int n_seg, hipStream_t stream)
{
    if (s.seg_cap < n) {
        const int cap = n < 64 ? 64 : n;
        if (s.d_seg) HIP_TRY(hipFree(s.d_seg));
        if (s.h_seg) HIP_TRY(hipHostFree(s.h_seg));
        s.d_seg = nullptr; s.h_seg = nullptr; s.seg_cap = 0;
        HIP_TRY(hipMalloc((void**)&s.d_seg, (size_t)cap));
        HIP_TRY(hipHostMalloc((void**)&s.h_seg, (size_t)cap, hipHostMallocDefault));
        s.seg_cap = cap;
    }
    memset(s.h_seg, 0, (size_t)n);
    for (int i = 0; i < n_seg; i++) s.h_seg[seg_first[i]] = 1;
    HIP_TRY(hipMemcpyAsync(s.d_seg, s.h_seg, (size_t)n, hipMemcpyHostToDevice, stream));
    return PSD_OK;
}

int psd_score_segments_submit_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride,
                                     size_t frame_stride, const int32_t* seg_first, int n_seg, uint32_t flags, int edge_kernel,
                                     void* stream_)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    int rc = validate(d_frames, n, height, width, row_stride, frame_stride, flags, edge_kernel);
    if (rc != PSD_OK) return rc;
    rc = check_segments(seg_first, n_seg, n);
    if (rc != PSD_OK) return rc;
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : e->stream;
    psd_slot* s = nullptr;
    rc = submit_begin(e, n, flags, stream, &s);
    if (rc != PSD_OK) return rc;
    if (n > 0) {
        rc = upload_segments(*s, n, seg_first, n_seg, stream);
        if (rc != PSD_OK) return rc;
        rc = score_terms(e, *s, d_frames, n, height, width, row_stride, frame_stride, nullptr, flags, edge_kernel, stream, s->d_seg);
        if (rc != PSD_OK) return rc;
    }
    return submit_end(e, *s, stream);
}

int psd_score_segments_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride,
                              size_t frame_stride, const int32_t* seg_first, int n_seg, uint32_t flags, int edge_kernel,
                              psd_frame_scores* out, void* stream)
{
    if (e && e->pending != 0) {
        psd_set_error("psd_score_segments_device: asynchronous submissions are still pending");
        return PSD_ERR_INVALID;
    }
    int rc = psd_score_segments_submit_device(e, d_frames, n, height, width, row_stride, frame_stride, seg_first, n_seg, flags,
                                              edge_kernel, stream);
    if (rc != PSD_OK) return rc;
    return psd_score_collect(e, out, n);
}

static int resize_any(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_frame_stride, uint8_t* d_dst,
                      int dst_h, int dst_w, size_t dst_frame_stride, int interpolation, hipStream_t stream)
{
    // INTER_AREA that does not shrink along both axes is, in OpenCV, the bilinear kernel with other coefficients (resize.cpp)
    const bool area_up = interpolation == PSD_INTER_AREA && (dst_w > src_w || dst_h > src_h);
    if (interpolation == PSD_INTER_LINEAR || area_up)
        return psd::resize_linear_score(e, d_src, n, src_h, src_w, (size_t)src_w * 3, src_frame_stride, nullptr, d_dst, dst_h, dst_w,
                                        dst_frame_stride, nullptr, stream, nullptr, nullptr, area_up);
    return psd::resize_other(e, d_src, n, src_h, src_w, src_frame_stride, d_dst, dst_h, dst_w, dst_frame_stride, interpolation, stream);
}

static int check_segments(const int32_t* seg_first, int n_seg, int n)
{
    if (n_seg < 0 || (n_seg > 0 && !seg_first)) { psd_set_error("invalid segment table"); return PSD_ERR_INVALID; }
    for (int i = 0; i < n_seg; i++) {
        if (seg_first[i] < 0 || seg_first[i] >= n || (i > 0 && seg_first[i] <= seg_first[i - 1])) {
            psd_set_error("segment table must hold ascending frame indices inside the batch (entry %d = %d, n = %d)", i, seg_first[i], n);
            return PSD_ERR_INVALID;
        }
    }
    return PSD_OK;
}

// cv2.resize in front of the terms of `flags` (scene_manager.py:666-678 + the detectors' process_frame), one submission.
// `segmented`: the batch is several clips back to back (seg_first: their first frames); d_prev is then null.
static int downscaled_submit(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w, size_t frame_stride,
                             const uint8_t* d_prev, bool segmented, const int32_t* seg_first, int n_seg, int dst_h, int dst_w,
                             int interpolation, uint32_t flags, int edge_kernel, void* stream_)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    int rc = validate(d_frames, n, src_h, src_w, (size_t)src_w * 3, frame_stride, flags, edge_kernel);
    if (rc != PSD_OK) return rc;
    if (dst_h <= 0 || dst_w <= 0) { psd_set_error("invalid target size %dx%d", dst_w, dst_h); return PSD_ERR_INVALID; }
    if (segmented) {
        rc = check_segments(seg_first, n_seg, n);
        if (rc != PSD_OK) return rc;
    }
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : e->stream;
    psd_slot* s = nullptr;
    rc = submit_begin(e, n, flags, stream, &s);
    if (rc != PSD_OK) return rc;
    if (n > 0) {
        const uint8_t* d_seg = nullptr;
        if (segmented) {
            rc = upload_segments(*s, n, seg_first, n_seg, stream);
            if (rc != PSD_OK) return rc;
            d_seg = s->d_seg;
        }
        bool done = false;
        if (interpolation == PSD_INTER_LINEAR && !(flags & PSD_SCORE_EDGES)) {
            // Content / Adaptive / Histogram / Threshold detectors, any set of them, behind the default downscale: the resized
            // frame never leaves the CU (only the edge term needs it in memory)
            rc = psd::resize_linear_score(e, d_frames, n, src_h, src_w, (size_t)src_w * 3, frame_stride, d_prev, nullptr, dst_h, dst_w, 0,
                                          s->d_recs, stream, &s->launches, d_seg, false, flags);
            if (rc == PSD_OK) done = true;
            else if (rc != PSD_ERR_UNSUPPORTED) return rc;
        }
        static const bool fuse_edges = [] { const char* v = getenv("PSD_EDGE_FUSE_DOWNSCALE"); return !v || atoi(v) != 0; }();
        if (!done && interpolation == PSD_INTER_LINEAR && flags == (PSD_SCORE_HSV_SAD | PSD_SCORE_EDGES) && fuse_edges &&
            !(dst_h == src_h && dst_w == src_w) && psd::resize_vplane_available(d_frames, src_w, frame_stride, d_prev, dst_w, n)) {
            // ContentDetector with the edge term (weights.delta_edges or a StatsManager) behind the default downscale: the fused
            // kernel's VOUT instance is the edge term's front end -- HSV SADs, and of the resized frame only its V plane and V
            // histogram in memory (round 6; until then the frame was resized into the buffer below and read again by the V-mode
            // pass: 453 MB of scattered writes per 4096 frames that cost 0.44 ms, psd_resize_kernels.hip)
            const psd::DownSrc down{d_frames, d_prev, src_h, src_w, frame_stride};
            rc = psd::edges_score(e, nullptr, n, dst_h, dst_w, (size_t)dst_w * 3, (size_t)dst_h * dst_w * 3, nullptr, edge_kernel, s->d_recs, stream, d_seg,
                                  nullptr, e->num_cus * 8, &s->launches, &down);
            if (rc == PSD_OK) done = true;
            else if (rc != PSD_ERR_UNSUPPORTED) return rc;
        }
        if (!done) {
            // resize [prev,] frames into the engine's small-frame buffer, then score them there
            const size_t sstride = (((size_t)dst_h * dst_w * 3) + 15) & ~(size_t)15;
            const size_t need = sstride * ((size_t)n + 1);
            if (e->small_bytes < need) {
                HIP_TRY(hipStreamSynchronize(stream));
                HIP_TRY(hipStreamSynchronize(e->stream));
                if (e->d_small) HIP_TRY(hipFree(e->d_small));
                e->d_small = nullptr; e->small_bytes = 0;
                hipError_t err = hipMalloc((void**)&e->d_small, need);
                if (err != hipSuccess) {
                    psd_set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(err));
            (void)hipGetLastError();
                    return err == hipErrorOutOfMemory ? PSD_ERR_NOMEM : PSD_ERR_HIP;
                }
                e->small_bytes = need;
            }
            if (d_prev) {
                rc = resize_any(e, d_prev, 1, src_h, src_w, frame_stride, e->d_small, dst_h, dst_w, sstride, interpolation, stream);
                if (rc != PSD_OK) return rc;
            }
            rc = resize_any(e, d_frames, n, src_h, src_w, frame_stride, e->d_small + sstride, dst_h, dst_w, sstride, interpolation, stream);
            if (rc != PSD_OK) return rc;
            s->launches += 1;
            rc = score_terms(e, *s, e->d_small + sstride, n, dst_h, dst_w, (size_t)dst_w * 3, sstride, d_prev ? e->d_small : nullptr, flags,
                             edge_kernel, stream, d_seg);
            if (rc != PSD_OK) return rc;
        }
    }
    return submit_end(e, *s, stream);
}

int psd_score_downscaled_submit_device(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w, size_t frame_stride,
                                       const uint8_t* d_prev, int dst_h, int dst_w, int interpolation, uint32_t flags,
                                       int edge_kernel, void* stream)
{
    return downscaled_submit(e, d_frames, n, src_h, src_w, frame_stride, d_prev, false, nullptr, 0, dst_h, dst_w, interpolation, flags,
                             edge_kernel, stream);
}

int psd_score_segments_downscaled_submit_device(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w,
                                                size_t frame_stride, const int32_t* seg_first, int n_seg, int dst_h, int dst_w,
                                                int interpolation, uint32_t flags, int edge_kernel, void* stream)
{
    return downscaled_submit(e, d_frames, n, src_h, src_w, frame_stride, nullptr, true, seg_first, n_seg, dst_h, dst_w, interpolation,
                             flags, edge_kernel, stream);
}

int psd_score_segments_downscaled_device(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w, size_t frame_stride,
                                         const int32_t* seg_first, int n_seg, int dst_h, int dst_w, int interpolation,
                                         uint32_t flags, int edge_kernel, psd_frame_scores* out, void* stream)
{
    if (e && e->pending != 0) {
        psd_set_error("psd_score_segments_downscaled_device: asynchronous submissions are still pending");
        return PSD_ERR_INVALID;
    }
    int rc = psd_score_segments_downscaled_submit_device(e, d_frames, n, src_h, src_w, frame_stride, seg_first, n_seg, dst_h, dst_w,
                                                         interpolation, flags, edge_kernel, stream);
    if (rc != PSD_OK) return rc;
    return psd_score_collect(e, out, n);
}

int psd_score_downscaled_device(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w, size_t frame_stride,
                                const uint8_t* d_prev, int dst_h, int dst_w, int interpolation, uint32_t flags, int edge_kernel,
                                psd_frame_scores* out, void* stream)
{
    if (e && e->pending != 0) {
        psd_set_error("psd_score_downscaled_device: asynchronous submissions are still pending");
        return PSD_ERR_INVALID;
    }
    int rc = psd_score_downscaled_submit_device(e, d_frames, n, src_h, src_w, frame_stride, d_prev, dst_h, dst_w, interpolation, flags,
                                                edge_kernel, stream);
    if (rc != PSD_OK) return rc;
    return psd_score_collect(e, out, n);
}

// Completion of a SMALL submission (the per-frame plug-in API scores one frame per call): hipEventSynchronize spins only
// briefly and then sleeps on an interrupt, which on some hosts costs 80-90 us per call -- score_device(n = 1) read 46 us on
// one box and 134 us on the next, process_frame() 170 vs 253 us (profiles/r05_q_*).  Poll the event instead for as long as
// such a submission can plausibly take, then fall back to the runtime's wait.  PSD_SPIN_US overrides the budget (0: never poll).
static hipError_t wait_done(hipEvent_t ev, int n)
{
    static const int spin_us = [] { const char* v = getenv("PSD_SPIN_US"); return v ? atoi(v) : 400; }();
    static const int spin_n = [] { const char* v = getenv("PSD_SPIN_N"); return v ? atoi(v) : 16; }();
    if (n <= spin_n && spin_us > 0) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
        bool polled_not_ready = false;
        do {
            const hipError_t q = hipEventQuery(ev);
            if (q != hipErrorNotReady) {
                // hipErrorNotReady lands in HIP's last-error slot like any failure: clear it, or the next launch check reports it --
                // but ONLY when this loop put it there (an unrelated error pending on this thread is not ours to discard)
                if (polled_not_ready) (void)hipGetLastError();
                return q;
            }
            polled_not_ready = true;
            // (a pause between two queries: the poll shares its core's issue slots with whatever else the host runs -- many engines
            //  or threads on the per-frame API -- and 400 us of back-to-back driver calls per collect is what it would otherwise burn)
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        } while (std::chrono::steady_clock::now() < until);
        (void)hipGetLastError();      // (only reached behind at least one hipErrorNotReady)
    }
    return hipEventSynchronize(ev);
}

// waits for the oldest submission and retires it; *slot_out = its slot (records in the pinned mirror)
static int collect_wait(psd_engine* e, const void* out, int n, const char* who, psd_slot** slot_out)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    if (e->pending <= 0) { psd_set_error("%s: nothing submitted", who); return PSD_ERR_INVALID; }
    psd_slot& s = e->slots[e->tail];
    if (n != s.n || (n > 0 && !out)) {
        psd_set_error("%s: expected n=%d records, caller asked for %d", who, s.n, n);
        return PSD_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(e->device));
    hipError_t err = wait_done(s.ev_done, s.n);
    s.pending = false;
    e->last_slot = e->tail;
    e->tail = (e->tail + 1) % PSD_MAX_INFLIGHT;
    e->pending--;
    if (err != hipSuccess) {
        psd_set_error("scoring failed on device: %s", hipGetErrorString(err));
        return PSD_ERR_HIP;
    }
    if (n > 0) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.ev_start, s.ev_stop) == hipSuccess) e->last_ms = ms;
        e->last_launches = s.launches;
    } else {
        e->last_ms = 0.f;
        e->last_launches = 0;
    }
    *slot_out = &s;
    return PSD_OK;
}

int psd_score_collect(psd_engine* e, psd_frame_scores* out, int n)
{
    psd_slot* s = nullptr;
    int rc = collect_wait(e, out, n, "psd_score_collect", &s);
    if (rc != PSD_OK || n == 0) return rc;
    if (!s->heads_only) {
        memcpy(out, s->h_recs, (size_t)n * sizeof(psd_frame_scores));
    } else {
        const psd_frame_sums* heads = (const psd_frame_sums*)s->h_recs;
        for (int i = 0; i < n; i++) {
            memset(&out[i], 0, sizeof(psd_frame_scores));
            memcpy(&out[i], &heads[i], sizeof(psd_frame_sums));
        }
    }
    return PSD_OK;
}

int psd_score_collect_sums(psd_engine* e, psd_frame_sums* out, int n)
{
    psd_slot* s = nullptr;
    int rc = collect_wait(e, out, n, "psd_score_collect_sums", &s);
    if (rc != PSD_OK || n == 0) return rc;
    if (s->heads_only) {
        memcpy(out, s->h_recs, (size_t)n * sizeof(psd_frame_sums));
    } else {
        const psd_frame_scores* recs = (const psd_frame_scores*)s->h_recs;
        for (int i = 0; i < n; i++) memcpy(&out[i], &recs[i], sizeof(psd_frame_sums));
    }
    return PSD_OK;
}

int psd_score_batch_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width,
                           size_t row_stride, size_t frame_stride, const uint8_t* d_prev,
                           uint32_t flags, int edge_kernel, psd_frame_scores* out, void* stream)
{
    if (e && e->pending != 0) {
        psd_set_error("psd_score_batch_device: asynchronous submissions are still pending");
        return PSD_ERR_INVALID;
    }
    int rc = psd_score_submit_device(e, d_frames, n, height, width, row_stride, frame_stride, d_prev, flags,
                                     edge_kernel, stream);
    if (rc != PSD_OK) return rc;
    return psd_score_collect(e, out, n);
}

int psd_last_records_device(psd_engine* e, const psd_frame_scores** d_recs, int* n)
{
    if (!e || !d_recs) { psd_set_error("psd_last_records_device: null argument"); return PSD_ERR_INVALID; }
    if (e->last_slot < 0) { psd_set_error("psd_last_records_device: nothing collected yet"); return PSD_ERR_INVALID; }
    *d_recs = e->slots[e->last_slot].d_recs;
    if (n) *n = e->slots[e->last_slot].n;
    return PSD_OK;
}

int psd_last_walk_geometry(psd_engine* e, int* frames_per_chunk, int* n_tiles)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    if (frames_per_chunk) *frames_per_chunk = g_walk[0];
    if (n_tiles) *n_tiles = g_walk[1];
    return PSD_OK;
}

int psd_last_kernel_ms(psd_engine* e, float* ms, int* launches)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    if (ms) *ms = e->last_ms;
    if (launches) *launches = e->last_launches;
    return PSD_OK;
}

}  // extern "C"

// Feeds host frames to `run(d_buf, cnt, packed_row, dstride, d_prev, done)` in bounded chunks through the engine's two device
// staging buffers (packed rows, 16-byte aligned frames); d_prev is the device copy of the frame preceding
// the chunk (h_prev for the first one, may be null).
template <typename Run>
static int for_each_host_chunk(psd_engine* e, const uint8_t* h_frames, int n, int height, int width, size_t row_stride,
                               size_t frame_stride, const uint8_t* h_prev, Run run)
{
    const size_t packed_row = (size_t)width * 3;
    const size_t dstride = ((size_t)height * packed_row + 15) & ~(size_t)15;  // 16-B aligned frames
    // Bounded staging: chunks of at most ~256 MiB (and at least one frame) per buffer, plus the
    // halo slot [0] that keeps the previous chunk's last frame.
    size_t per_chunk = (size_t)(256u << 20) / dstride;
    if (per_chunk < 1) per_chunk = 1;
    if (per_chunk > (size_t)n) per_chunk = (size_t)n;
    const size_t need = (per_chunk + 1) * dstride;
    if (e->stage_bytes < need) {
        for (auto& d : e->d_stage) { if (d) HIP_TRY(hipFree(d)); d = nullptr; }
        e->stage_bytes = 0;
        for (auto& d : e->d_stage) HIP_TRY(hipMalloc((void**)&d, need));
        e->stage_bytes = need;
    }
    auto upload = [&](uint8_t* dst, const uint8_t* src, size_t count) -> int {
        // frames [count] from host (row_stride/frame_stride) to packed device layout
        if (row_stride == packed_row && frame_stride == dstride) {
            HIP_TRY(hipMemcpyAsync(dst, src, count * dstride, hipMemcpyHostToDevice, e->stream));
        } else {
            for (size_t i = 0; i < count; i++)
                HIP_TRY(hipMemcpy2DAsync(dst + i * dstride, packed_row, src + i * frame_stride, row_stride,
                                         packed_row, (size_t)height, hipMemcpyHostToDevice, e->stream));
        }
        return PSD_OK;
    };
    int done = 0, chunk_idx = 0, rc;
    const uint8_t* d_prev = nullptr;
    if (h_prev) {
        rc = upload(e->d_stage[1], h_prev, 1);  // park the halo in the *other* buffer's slot 0
        if (rc != PSD_OK) return rc;
        d_prev = e->d_stage[1];
    }
    while (done < n) {
        const int cnt = (int)((size_t)(n - done) < per_chunk ? (size_t)(n - done) : per_chunk);
        uint8_t* buf = e->d_stage[chunk_idx & 1] + dstride;  // slot 0 is reserved for a halo copy
        rc = upload(buf, h_frames + (size_t)done * frame_stride, (size_t)cnt);
        if (rc != PSD_OK) return rc;
        rc = run(buf, cnt, packed_row, dstride, d_prev, done);
        if (rc != PSD_OK) return rc;
        d_prev = buf + (size_t)(cnt - 1) * dstride;
        done += cnt;
        chunk_idx++;
    }
    return PSD_OK;
}

namespace {
struct RowCopy { int first, len, step, count; };   // `count` groups of `len` consecutive rows, `step` rows apart, from row `first`

// Groups of consecutive rows (one contiguous piece each when the host rows are packed), then strided copies: from a group
// on, the layout is tried as p interleaved progressions (p = 1 .. 8) of equally long groups at one common distance -- a
// decimation by 7.5 alternates steps of 7 and 8 rows: two progressions of step 15 -- and the period that covers most
// groups per copy is taken.  The default 1080p -> 256 x 144 pipeline is two copies per frame.
std::vector<RowCopy> plan_row_copies(const int* rows, int n_rows, bool packed)
{
    std::vector<std::pair<int, int>> groups;   // first row, rows
    for (int i = 0; i < n_rows; i++) {
        if (packed && !groups.empty() && rows[i] == groups.back().first + groups.back().second) groups.back().second++;
        else groups.emplace_back(rows[i], 1);
    }
    std::vector<RowCopy> plan;
    const size_t G = groups.size();
    for (size_t g = 0; g < G;) {
        size_t best_p = 1, best_cover = 1;
        int best_step = 0;
        for (size_t p = 1; p <= 8 && g + p < G; p++) {
            const int step = groups[g + p].first - groups[g].first;
            size_t end = g + p;
            while (end < G && groups[end].second == groups[end - p].second && groups[end].first - groups[end - p].first == step) end++;
            const size_t cover = end - g;
            if (cover > p && cover * best_p > best_cover * p) { best_p = p; best_cover = cover; best_step = step; }
        }
        for (size_t j = 0; j < best_p && j < best_cover; j++)
            plan.push_back(RowCopy{groups[g + j].first, groups[g + j].second, best_step, (int)((best_cover - j + best_p - 1) / best_p)});
        g += best_cover;
    }
    return plan;
}

bool rows_ascending(const int* rows, int n_rows)
{
    for (int i = 0; i < n_rows; i++)
        if (rows[i] < 0 || (i && rows[i] <= rows[i - 1])) return false;
    return true;
}
}  // namespace

extern "C" {

int psd_score_batch(psd_engine* e, const uint8_t* h_frames, int n, int height, int width,
                    size_t row_stride, size_t frame_stride, const uint8_t* h_prev, uint32_t flags,
                    int edge_kernel, psd_frame_scores* out)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    int rc = validate(h_frames, n, height, width, row_stride, frame_stride, flags, edge_kernel);
    if (rc != PSD_OK) return rc;
    if (e->pending != 0) { psd_set_error("psd_score_batch: asynchronous submissions are still pending"); return PSD_ERR_INVALID; }
    if (n == 0) return PSD_OK;
    HIP_TRY(hipSetDevice(e->device));
    return for_each_host_chunk(e, h_frames, n, height, width, row_stride, frame_stride, h_prev,
                               [&](const uint8_t* buf, int cnt, size_t packed_row, size_t dstride, const uint8_t* d_prev, int done) {
                                   return psd_score_batch_device(e, buf, cnt, height, width, packed_row, dstride, d_prev, flags,
                                                                 edge_kernel, out + done, nullptr);
                               });
}

static int validate_hash(const void* frames, int n, int height, int width, size_t row_stride, size_t frame_stride,
                         int size, const void* out)
{
    int rc = validate(frames, n, height, width, row_stride, frame_stride, PSD_SCORE_BYTE_SUM, 0);
    if (rc != PSD_OK) return rc;
    if (size <= 0) { psd_set_error("hash thumbnail size must be positive, got %d", size); return PSD_ERR_INVALID; }
    if (n > 0 && !out) { psd_set_error("thumbnail output pointer is null"); return PSD_ERR_INVALID; }
    return PSD_OK;
}

int psd_hash_thumbs_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width,
                           size_t row_stride, size_t frame_stride, int size, uint8_t* h_thumbs)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    int rc = validate_hash(d_frames, n, height, width, row_stride, frame_stride, size, h_thumbs);
    if (rc != PSD_OK) return rc;
    if (n == 0) return PSD_OK;
    HIP_TRY(hipSetDevice(e->device));
    const size_t need = (((size_t)n * size * size) + 255) & ~(size_t)255;
    if (e->hash_bytes < need) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->d_hash) HIP_TRY(hipFree(e->d_hash));
        e->d_hash = nullptr; e->hash_bytes = 0;
        hipError_t err = hipMalloc((void**)&e->d_hash, need);
        if (err != hipSuccess) {
            psd_set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(err));
            (void)hipGetLastError();
            return err == hipErrorOutOfMemory ? PSD_ERR_NOMEM : PSD_ERR_HIP;
        }
        e->hash_bytes = need;
    }
    if (!e->ev_hash[0]) { HIP_TRY(hipEventCreate(&e->ev_hash[0])); HIP_TRY(hipEventCreate(&e->ev_hash[1])); }
    // ev_hash[0] is recorded inside, once the run tables are known to be on the device
    rc = psd::hash_thumbs(e, d_frames, n, height, width, row_stride, frame_stride, size, e->d_hash, e->stream, e->ev_hash[0]);
    if (rc != PSD_OK) return rc;
    HIP_TRY(hipEventRecord(e->ev_hash[1], e->stream));
    HIP_TRY(hipMemcpyAsync(h_thumbs, e->d_hash, (size_t)n * size * size, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipEventElapsedTime(&e->last_ms, e->ev_hash[0], e->ev_hash[1]));
    e->last_launches = (n + 32767) / 32768;
    return PSD_OK;
}

// grows one of the engine's device buffers on demand (the stream is drained first: queued kernels may still use the old one)
static int grow_buffer(psd_engine* e, uint8_t** buf, size_t* have, size_t need)
{
    need = (need + 255) & ~(size_t)255;
    if (*have >= need) return PSD_OK;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr; *have = 0;
    hipError_t err = hipMalloc((void**)buf, need);
    if (err != hipSuccess) {
        psd_set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(err));
        (void)hipGetLastError();
        return err == hipErrorOutOfMemory ? PSD_ERR_NOMEM : PSD_ERR_HIP;
    }
    *have = need;
    return PSD_OK;
}

int psd_hash_bits_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride, size_t frame_stride,
                         int size, int hash_size, uint8_t* h_bits, uint8_t* h_thumbs)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    int rc = validate_hash(d_frames, n, height, width, row_stride, frame_stride, size, h_bits);
    if (rc != PSD_OK) return rc;
    if (hash_size <= 0 || hash_size > size) { psd_set_error("hash_size must be in 1 .. %d, got %d", size, hash_size); return PSD_ERR_INVALID; }
    if (!psd::hash_bits_lds(size, hash_size)) {
        psd_set_error("hash bits on the device: a %d x %d transform of a %d x %d thumbnail does not fit a workgroup's LDS "
                      "(psd_hash_thumbs_device + psd_epilogue_hash_bits take any size)", hash_size, hash_size, size, size);
        return PSD_ERR_UNSUPPORTED;
    }
    if (n == 0) return PSD_OK;
    HIP_TRY(hipSetDevice(e->device));
    rc = grow_buffer(e, &e->d_hash, &e->hash_bytes, (size_t)n * size * size);
    if (rc != PSD_OK) return rc;
    rc = grow_buffer(e, &e->d_hbits, &e->hbits_bytes, (size_t)n * hash_size * hash_size);
    if (rc != PSD_OK) return rc;
    psd::DevTable basis;
    if (!psd::table_find(e, psd::kTabHashBasis, size, size, hash_size, hash_size, &basis)) {
        std::vector<double> c((size_t)hash_size * size);
        psd::hash_dct_basis(size, hash_size, c.data());      // (the host epilogue's own table: the same doubles)
        rc = psd::table_store(e, psd::kTabHashBasis, size, size, hash_size, hash_size, c.data(), c.size() * sizeof(double), 0, 0.f, &basis);
        if (rc != PSD_OK) return rc;
    }
    if (!e->ev_hash[0]) { HIP_TRY(hipEventCreate(&e->ev_hash[0])); HIP_TRY(hipEventCreate(&e->ev_hash[1])); }
    rc = psd::hash_thumbs(e, d_frames, n, height, width, row_stride, frame_stride, size, e->d_hash, e->stream, e->ev_hash[0]);
    if (rc != PSD_OK) return rc;
    rc = psd::hash_bits(e->d_hash, n, size, hash_size, static_cast<const double*>(basis.ptr), e->d_hbits, e->stream);
    if (rc != PSD_OK) return rc;
    HIP_TRY(hipEventRecord(e->ev_hash[1], e->stream));
    HIP_TRY(hipMemcpyAsync(h_bits, e->d_hbits, (size_t)n * hash_size * hash_size, hipMemcpyDeviceToHost, e->stream));
    if (h_thumbs) HIP_TRY(hipMemcpyAsync(h_thumbs, e->d_hash, (size_t)n * size * size, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipEventElapsedTime(&e->last_ms, e->ev_hash[0], e->ev_hash[1]));
    e->last_launches = (n + 32767) / 32768 + (n + 65534) / 65535;
    return PSD_OK;
}

// ---- HistogramDetector's hist_diff on the device (psd_hist_diff_device) -------------------------------------------------------
// psd_epilogue_hist_cuts spends 0.2 us per frame and core on calculate_histogram's tail and compareHist (histogram_detector.py:98,
// 156-163): re-bin, cv2.normalize(NORM_L2), cv2.compareHist(CORREL) in float64.  A frame pair depends on nothing but its two
// records, so for records that are still in HBM the pairs go to waves of the device, every sum in
// the order of the host epilogue (psd_epilogue.cpp: normalized_hist, correl), IEEE float32 / float64 arithmetic without contraction
// (the file is built with -ffp-contract=off; f64 division and square root are correctly rounded): the same bits, and what travels to
// the host is 8 bytes per frame instead of the 1 KiB histogram.
static int ensure_copy_stream(psd_engine* e);
struct HistRuns { int start[257]; };     // bin i of `bins` takes luma values start[i] .. start[i + 1] - 1 (cv2.calcHist's LUT is monotone)

// One wave per frame pair (t - 1, t).  What is elementwise is spread over the lanes (re-bin, int -> float32, the scaling); what the host
// sums in an order keeps that order: lane k adds the squares of histogram k one after the other (normalized_hist: exact in any order while
// every bin is below 2^24, sequential beyond -- sequential is both), lanes 0 / 1 ARE the two float64 lanes of OpenCV's compareHist loop.
__global__ __launch_bounds__(64) void hist_diff_kernel(const psd_frame_scores* __restrict__ recs, int n, int bins, HistRuns runs,
                                                       double* __restrict__ diff)
{
    __shared__ float f[2][256];
    __shared__ float sc[2];
    __shared__ double red[2][5];
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= n) return;
    if (t == 0) { if (lane == 0) diff[0] = NAN; return; }       // (no predecessor inside the batch; the caller marks clip starts)
    for (int k = 0; k < 2; k++) {
        const uint32_t* h = recs[t - 1 + k].hist;
        for (int i = lane; i < bins; i += 64) {
            unsigned long long c = 0;
            for (int j = runs.start[i]; j < runs.start[i + 1]; j++) c += h[j];
            f[k][i] = (float)(long long)c;                       // (exact below 2^24, one rounding above: as the host's cvtsi2ss)
        }
    }
    __syncthreads();
    if (lane < 2) {
        double ss = 0.0;
        for (int i = 0; i < bins; i++) { const double v = f[lane][i]; ss += v * v; }
        const double nrm = sqrt(ss);
        sc[lane] = (float)(nrm > DBL_EPSILON ? 1.0 / nrm : 0.0);
    }
    __syncthreads();
    for (int i = lane; i < bins; i += 64) { f[0][i] = f[0][i] * sc[0]; f[1][i] = f[1][i] * sc[1]; }
    __syncthreads();
    const int body = bins >= 4 ? (bins & ~3) : 0;                // OpenCV's vector body: whole groups of four
    if (lane < 2) {
        double s1 = 0, s2 = 0, s11 = 0, s12 = 0, s22 = 0;
        for (int j = lane; j < body; j += 2) {
            const double a = f[0][j], b = f[1][j];
            s12 += a * b; s11 += a * a; s22 += b * b; s1 += a; s2 += b;
        }
        red[lane][0] = s1; red[lane][1] = s2; red[lane][2] = s11; red[lane][3] = s12; red[lane][4] = s22;
    }
    __syncthreads();
    if (lane == 0) {
        double S1 = red[0][0] + red[1][0], S2 = red[0][1] + red[1][1], S11 = red[0][2] + red[1][2], S12 = red[0][3] + red[1][3],
               S22 = red[0][4] + red[1][4];
        for (int j = body; j < bins; j++) {
            const double a = f[0][j], b = f[1][j];
            S12 += a * b; S1 += a; S11 += a * a; S2 += b; S22 += b * b;
        }
        const double scale = 1. / bins;
        const double num = S12 - S1 * S2 * scale;
        const double denom2 = (S11 - S1 * S1 * scale) * (S22 - S2 * S2 * scale);
        diff[t] = fabs(denom2) > DBL_EPSILON ? num / sqrt(denom2) : 1.;
    }
}

int psd_hist_diff_device(psd_engine* e, const psd_frame_scores* d_recs, int n, int bins, double* h_diff, void* stream_)
{
    if (!e || n < 0 || (n > 0 && (!d_recs || !h_diff)) || bins < 1 || bins > 256) {
        psd_set_error("psd_hist_diff_device: invalid argument (n = %d, bins = %d)", n, bins);
        return PSD_ERR_INVALID;
    }
    if (n == 0) return PSD_OK;
    HIP_TRY(hipSetDevice(e->device));
    // (no stream given: a side stream of the engine's, NOT its scoring stream -- the records are complete, and behind the scoring stream this
    //  small kernel would wait for whatever submission is in flight there: in the packed flow, for the next piece's whole pass)
    int rc = stream_ ? PSD_OK : ensure_copy_stream(e);
    if (rc != PSD_OK) return rc;
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : e->copy_stream;
    rc = grow_buffer(e, &e->d_hdiff, &e->hdiff_bytes, (size_t)n * sizeof(double));
    if (rc != PSD_OK) return rc;
    HistRuns runs;
    for (int i = 0; i <= 256; i++) runs.start[i] = 256;
    {
        // cv2.calcHist's bin of luma value j, as psd_epilogue_hist_cuts builds it: floor(j * (bins / 256.0)), clamped
        int prev = -1;
        for (int j = 0; j < 256; j++) {
            int idx = (int)std::floor(j * (bins / 256.0));
            idx = idx < 0 ? 0 : (idx > bins - 1 ? bins - 1 : idx);
            for (int b = prev + 1; b <= idx; b++) runs.start[b] = j;       // (monotone: bins between two values stay empty runs)
            prev = idx;
        }
        for (int b = prev + 1; b <= bins; b++) runs.start[b] = 256;
    }
    double* diff = reinterpret_cast<double*>(e->d_hdiff);
    hipLaunchKernelGGL(hist_diff_kernel, dim3(n), dim3(64), 0, stream, d_recs, n, bins, runs, diff);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_diff, diff, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return PSD_OK;
}

int psd_hash_thumbs(psd_engine* e, const uint8_t* h_frames, int n, int height, int width,
                    size_t row_stride, size_t frame_stride, int size, uint8_t* h_thumbs)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    int rc = validate_hash(h_frames, n, height, width, row_stride, frame_stride, size, h_thumbs);
    if (rc != PSD_OK) return rc;
    if (e->pending != 0) { psd_set_error("psd_hash_thumbs: asynchronous submissions are still pending"); return PSD_ERR_INVALID; }
    if (n == 0) return PSD_OK;
    HIP_TRY(hipSetDevice(e->device));
    return for_each_host_chunk(e, h_frames, n, height, width, row_stride, frame_stride, nullptr,
                               [&](const uint8_t* buf, int cnt, size_t packed_row, size_t dstride, const uint8_t*, int done) {
                                   return psd_hash_thumbs_device(e, buf, cnt, height, width, packed_row, dstride, size,
                                                                 h_thumbs + (size_t)done * size * size);
                               });
}

int psd_device_alloc(psd_engine* e, size_t bytes, void** d_ptr)
{
    if (!e || !d_ptr) { psd_set_error("psd_device_alloc: null argument"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    *d_ptr = nullptr;
    hipError_t err = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (err != hipSuccess) {
        psd_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
        (void)hipGetLastError();   // (reported: the next kernel launch check must not see it again)
        return err == hipErrorOutOfMemory ? PSD_ERR_NOMEM : PSD_ERR_HIP;
    }
    return PSD_OK;
}

int psd_device_free(psd_engine* e, void* d_ptr)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return PSD_OK;
}

int psd_memcpy_h2d(psd_engine* e, void* d_dst, const void* h_src, size_t bytes)
{
    if (!e || (bytes && (!d_dst || !h_src))) { psd_set_error("psd_memcpy_h2d: null argument"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));  // order against work queued on the engine's (non-blocking) stream
    HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return PSD_OK;
}

int psd_host_alloc(psd_engine* e, size_t bytes, void** h_ptr)
{
    if (!e || !h_ptr) { psd_set_error("psd_host_alloc: null argument"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    *h_ptr = nullptr;
    hipError_t err = hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (err != hipSuccess) {
        psd_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
        (void)hipGetLastError();
        return err == hipErrorOutOfMemory ? PSD_ERR_NOMEM : PSD_ERR_HIP;
    }
    return PSD_OK;
}

int psd_host_free(psd_engine* e, void* h_ptr)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    if (h_ptr) HIP_TRY(hipHostFree(h_ptr));
    return PSD_OK;
}

int psd_upload(psd_engine* e, void* d_dst, const void* h_src, size_t bytes)
{
    if (!e || (bytes && (!d_dst || !h_src))) { psd_set_error("psd_upload: null argument"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    // blocking for the CALLER only: no engine state is touched and the engine's stream is not waited for, so a decode
    // thread can fill one device batch while the engine scores another
    HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return PSD_OK;
}

int psd_resize_source_rows(int src_h, int src_w, int dst_h, int dst_w, int interpolation, int* rows, int* n_rows)
{
    if (src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || !rows || !n_rows) {
        psd_set_error("psd_resize_source_rows: invalid argument");
        return PSD_ERR_INVALID;
    }
    return psd::resize_source_rows(src_h, src_w, dst_h, dst_w, interpolation, rows, n_rows);
}

int psd_upload_rows_plan(const int* rows, int n_rows, int packed, int* copies, int max_copies, int* n_copies)
{
    if (n_rows < 0 || (n_rows && !rows) || !n_copies || max_copies < 0 || (max_copies && !copies) || !rows_ascending(rows, n_rows)) {
        psd_set_error("psd_upload_rows_plan: invalid argument (rows must be ascending and >= 0)");
        return PSD_ERR_INVALID;
    }
    const std::vector<RowCopy> plan = plan_row_copies(rows, n_rows, packed != 0);
    *n_copies = (int)plan.size();
    for (size_t i = 0; i < plan.size() && i < (size_t)max_copies; i++) {
        copies[4 * i] = plan[i].first; copies[4 * i + 1] = plan[i].len; copies[4 * i + 2] = plan[i].step; copies[4 * i + 3] = plan[i].count;
    }
    return PSD_OK;
}

int psd_upload_rows(psd_engine* e, void* d_frame, const void* h_frame, size_t row_bytes, size_t h_row_stride,
                    const int* rows, int n_rows)
{
    if (!e || n_rows < 0 || (n_rows && (!d_frame || !h_frame || !rows)) || row_bytes == 0 || h_row_stride < row_bytes) {
        psd_set_error("psd_upload_rows: invalid argument");
        return PSD_ERR_INVALID;
    }
    if (!rows_ascending(rows, n_rows)) { psd_set_error("psd_upload_rows: rows must be ascending and >= 0"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    const bool packed = h_row_stride == row_bytes;
    for (const RowCopy& c : plan_row_copies(rows, n_rows, packed)) {
        const uint8_t* src = static_cast<const uint8_t*>(h_frame) + (size_t)c.first * h_row_stride;
        uint8_t* dst = static_cast<uint8_t*>(d_frame) + (size_t)c.first * row_bytes;
        if (c.count == 1)   // packed: one piece; otherwise the group is a single row
            HIP_TRY(hipMemcpy(dst, src, (size_t)c.len * row_bytes, hipMemcpyHostToDevice));
        else
            HIP_TRY(hipMemcpy2D(dst, (size_t)c.step * row_bytes, src, (size_t)c.step * h_row_stride, (size_t)c.len * row_bytes, c.count,
                                hipMemcpyHostToDevice));
    }
    return PSD_OK;
}

static int ensure_copy_stream(psd_engine* e)
{
    if (!e->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_copy, hipEventDisableTiming));
    }
    return PSD_OK;
}

int psd_upload_async(psd_engine* e, void* d_dst, const void* h_src, size_t bytes)
{
    if (!e || (bytes && (!d_dst || !h_src))) { psd_set_error("psd_upload_async: null argument"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    int rc = ensure_copy_stream(e);
    if (rc != PSD_OK) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, e->copy_stream));
    return PSD_OK;
}

int psd_upload_fence(psd_engine* e, int wait_on_host)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    if (!e->copy_stream) return PSD_OK;   // nothing was ever uploaded asynchronously
    HIP_TRY(hipSetDevice(e->device));
    if (wait_on_host) {
        HIP_TRY(hipStreamSynchronize(e->copy_stream));
    } else {
        HIP_TRY(hipEventRecord(e->ev_copy, e->copy_stream));
        HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_copy, 0));
    }
    return PSD_OK;
}

int psd_synchronize(psd_engine* e)
{
    if (!e) { psd_set_error("null engine"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return PSD_OK;
}

int psd_memcpy_d2d(psd_engine* e, void* d_dst, const void* d_src, size_t bytes)
{
    if (!e || (bytes && (!d_dst || !d_src))) { psd_set_error("psd_memcpy_d2d: null argument"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, e->stream));
    return PSD_OK;
}

int psd_memcpy_d2h(psd_engine* e, void* h_dst, const void* d_src, size_t bytes)
{
    if (!e || (bytes && (!h_dst || !d_src))) { psd_set_error("psd_memcpy_d2h: null argument"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return PSD_OK;
}

int psd_edge_map_device(psd_engine* e, const uint8_t* d_frame, int height, int width, size_t row_stride,
                        int edge_kernel, uint8_t* h_edges)
{
    if (!e || !d_frame || !h_edges) { psd_set_error("psd_edge_map_device: null argument"); return PSD_ERR_INVALID; }
    int rc = validate(d_frame, 1, height, width, row_stride, 0, PSD_SCORE_EDGES, edge_kernel);
    if (rc != PSD_OK) return rc;
    HIP_TRY(hipSetDevice(e->device));
    return psd::edges_map(e, d_frame, height, width, row_stride, edge_kernel, h_edges);
}

int psd_resize_device(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_frame_stride,
                      uint8_t* d_dst, int dst_h, int dst_w, size_t dst_frame_stride, int interpolation, void* stream)
{
    if (!e || n < 0 || (n > 0 && (!d_src || !d_dst)) || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) {
        psd_set_error("psd_resize_device: invalid argument");
        return PSD_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    return resize_any(e, d_src, n, src_h, src_w, src_frame_stride, d_dst, dst_h, dst_w, dst_frame_stride, interpolation, s);
}

int psd_resize_linear_device(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w,
                             size_t src_frame_stride, uint8_t* d_dst, int dst_h, int dst_w,
                             size_t dst_frame_stride, void* stream)
{
    return psd_resize_device(e, d_src, n, src_h, src_w, src_frame_stride, d_dst, dst_h, dst_w, dst_frame_stride,
                             PSD_INTER_LINEAR, stream);
}

}  // extern "C"

// Accessors used by psd_edge_kernels.hip (keeps the struct private to this file).
namespace psd {
void** engine_edge_ws(psd_engine* e) { return &e->edge_ws; }
size_t* engine_edge_ws_bytes(psd_engine* e) { return &e->edge_ws_bytes; }
int engine_num_cus(psd_engine* e) { return e->num_cus; }
hipStream_t engine_stream(psd_engine* e) { return e->stream; }
int engine_device(psd_engine* e) { return e->device; }
void** engine_resize_cache(psd_engine* e) { return &e->resize_cache; }
psd_feed** engine_feed_slot(psd_engine* e) { return &e->feed; }
int engine_hist_scratch(psd_engine* e, size_t bytes, hipStream_t stream, void** out)
{
    if (e->hpart_bytes < bytes) {
        // (growing: earlier submissions on either stream may still read the old buffer)
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->d_hpart) HIP_TRY(hipFree(e->d_hpart));
        e->d_hpart = nullptr; e->hpart_bytes = 0;
        const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        hipError_t err = hipMalloc(&e->d_hpart, cap);
        if (err != hipSuccess) {
            psd_set_error("hipMalloc(%zu) failed: %s", cap, hipGetErrorString(err));
            (void)hipGetLastError();
            return err == hipErrorOutOfMemory ? PSD_ERR_NOMEM : PSD_ERR_HIP;
        }
        e->hpart_bytes = cap;
    }
    *out = e->d_hpart;
    return PSD_OK;
}
int engine_copy_stream(psd_engine* e, hipStream_t* out)
{
    int rc = ensure_copy_stream(e);
    *out = e->copy_stream;
    return rc;
}
const uint32_t* engine_lut(psd_engine* e) { return e->d_lut; }
}  // namespace psd
