// psd_feed.cpp -- psd_upload_rows_batch: the host side of feeding decoded frames to the default (downscaled) pipeline.
//
// The reference's decode thread resizes every frame on the CPU and queues the small frame (scene_manager.py:625-710,
// 666-678).  Here the downscale runs on the device, so what has to cross PCIe per frame is the set of source rows that
// carry taps (psd_resize_source_rows: 288 of 1080 rows for 1080p -> 256 x 144, 1.66 MB).  Uploading them frame by frame
// with blocking strided copies (psd_upload_rows) spends as long setting copies up as moving bytes: 19-20 k frames/s where
// the link carries 34 k (profiles/r03_as_host_feed_rates.json).  This path instead
//   1. GATHERS the rows of many frames (separately allocated, pageable: what a decoder hands out) into one page-locked
//      segment, compactly, with a small pool of worker threads -- a frame's rows are a 1.66 MB memcpy, one thread would be the
//      bottleneck at 6-9 k frames/s;
//   2. moves the segment with ONE asynchronous contiguous copy on the engine's copy stream into device staging memory;
//   3. SCATTERS the rows to their places in the full-size device frames with a small kernel on the same stream (device
//      bandwidth: nothing next to PCIe) -- the row list travels in the segment's header, so it is one copy per batch in all.
// Segments form a ring of three (host + device + event): the gather of batch k+1 overlaps the DMA of batch k.
// NUMA: on a two-socket host the page-locked segments must sit on the GPU's own node -- with the staging memory (or the
// gather threads) on the other socket the same code moves 21 k instead of 31 k 1080p frames/s (profiles/r04_l_*).  The
// segments are allocated, and the helper threads run, on the CPUs of the node the GPU's PCI function reports
// (/sys/bus/pci/devices/<bdf>/numa_node); PSD_FEED_NUMA=0 leaves placement to the OS.
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "psd_internal.h"

extern "C" void psd_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            psd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)hipGetLastError(); /* (the failure is reported here: do not leave it for the next launch check) */ \
            return PSD_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

struct psd_feed;

namespace psd {
psd_feed** engine_feed_slot(psd_engine* e);
int engine_copy_stream(psd_engine* e, hipStream_t* out);
int engine_device(psd_engine* e);
}  // namespace psd

namespace {

constexpr int kSegments = 3;
constexpr size_t kHeaderAlign = 256;

struct Segment {
    uint8_t* h = nullptr;     // page-locked: [row list, padded to 256 B][n_frames * n_rows rows, compact]
    uint8_t* d = nullptr;     // device staging, same layout
    size_t bytes = 0;
    hipEvent_t ev = nullptr;  // recorded behind the scatter kernel: host and device halves are free again
    bool busy = false;
};

struct Job {
    const void* const* frames = nullptr;
    const int* rows = nullptr;
    int n_frames = 0, n_rows = 0, rows_per_unit = 1, units_per_frame = 0;
    size_t row_bytes = 0, h_row_stride = 0;
    uint8_t* dst = nullptr;
};

}  // namespace

struct psd_feed {
    Segment seg[kSegments];
    int next = 0;
    cpu_set_t near_gpu;            // CPUs of the GPU's NUMA node (within this process's affinity mask)
    bool have_near = false;
    // gather pool: the caller of psd_upload_rows_batch works too, `threads` helpers wait on cv_work between batches
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    Job job;
    std::atomic<long> next_unit{0};
    long total_units = 0;
    unsigned generation = 0;
    int working = 0;
    bool stop = false;
};

namespace {

// rows [u * rows_per_unit, ...) of frame u / units_per_frame: consecutive source rows that are also consecutive in the frame
// travel as one memcpy
void gather_unit(const Job& j, long u)
{
    const int f = (int)(u / j.units_per_frame), part = (int)(u - (long)f * j.units_per_frame);
    const int r0 = part * j.rows_per_unit, r1 = r0 + j.rows_per_unit < j.n_rows ? r0 + j.rows_per_unit : j.n_rows;
    const uint8_t* src = static_cast<const uint8_t*>(j.frames[f]);
    uint8_t* dst = j.dst + ((size_t)f * j.n_rows + r0) * j.row_bytes;
    const bool packed = j.h_row_stride == j.row_bytes;
    for (int r = r0; r < r1;) {
        int run = 1;
        if (packed)
            while (r + run < r1 && j.rows[r + run] == j.rows[r] + run) run++;
        memcpy(dst, src + (size_t)j.rows[r] * j.h_row_stride, (size_t)run * j.row_bytes);
        dst += (size_t)run * j.row_bytes;
        r += run;
    }
}

void drain(psd_feed* f)
{
    for (;;) {
        const long u = f->next_unit.fetch_add(1, std::memory_order_relaxed);
        if (u >= f->total_units) return;
        gather_unit(f->job, u);
    }
}

void worker(psd_feed* f)
{
    if (f->have_near) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &f->near_gpu);
    unsigned seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(f->m);
            f->cv_work.wait(lk, [&] { return f->stop || f->generation != seen; });
            if (f->stop) return;
            seen = f->generation;
        }
        drain(f);
        {
            std::lock_guard<std::mutex> lk(f->m);
            if (--f->working == 0) f->cv_done.notify_one();
        }
    }
}

int feed_threads()
{
    static const int n = [] {
        const char* v = getenv("PSD_FEED_THREADS");
        int t = v ? atoi(v) : 16;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && t > (int)hw) t = (int)hw;
        return t < 1 ? 1 : (t > 64 ? 64 : t);
    }();
    return n;
}

// compact rows -> their places in the full-size frames.  grid = (n_rows, n_frames); the row list sits in the segment header.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const uint8_t* seg, size_t header, uint8_t* d_first, size_t d_frame_stride,
                                                           int n_rows, unsigned row_bytes)
{
    const int r = blockIdx.x, f = blockIdx.y;
    const int row = reinterpret_cast<const int*>(seg)[r];
    const uint8_t* src = seg + header + ((size_t)f * n_rows + r) * row_bytes;
    uint8_t* dst = d_first + (size_t)f * d_frame_stride + (size_t)row * row_bytes;
    if ((row_bytes & 15u) == 0 && ((uintptr_t)dst & 15) == 0 && ((uintptr_t)src & 15) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (unsigned i = threadIdx.x; i < row_bytes / 16; i += 256) d4[i] = s4[i];
    } else {
        for (unsigned i = threadIdx.x; i < row_bytes; i += 256) dst[i] = src[i];
    }
}

// "0-63,128-191" -> cpu set; false if the text is not a cpu list
bool parse_cpulist(const char* text, cpu_set_t* out)
{
    CPU_ZERO(out);
    int n = 0;
    for (const char* p = text; *p && *p != '\n';) {
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0) return false;
        long b = a;
        p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); if (end == p + 1 || b < a) return false; p = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, out); n++; }
        if (*p == ',') p++;
    }
    return n > 0;
}

// the CPUs next to the GPU, restricted to what this thread may run on; false: unknown, one node, or switched off
bool cpus_near_gpu(int device, cpu_set_t* out)
{
    static const bool enabled = [] { const char* v = getenv("PSD_FEED_NUMA"); return !v || atoi(v) != 0; }();
    if (!enabled) return false;
    char bdf[64] = "";
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (char* c = bdf; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[160], line[4096];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE* fp = fopen(path, "r");
    if (!fp) return false;
    int node = -1;
    const bool got = fscanf(fp, "%d", &node) == 1;
    fclose(fp);
    if (!got || node < 0) return false;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    fp = fopen(path, "r");
    if (!fp) return false;
    const bool read = fgets(line, sizeof line, fp) != nullptr;
    fclose(fp);
    cpu_set_t node_cpus, mine;
    if (!read || !parse_cpulist(line, &node_cpus)) return false;
    if (sched_getaffinity(0, sizeof mine, &mine) != 0) return false;
    CPU_AND(out, &node_cpus, &mine);
    return CPU_COUNT(out) > 0 && CPU_COUNT(out) < CPU_COUNT(&mine);   // (all of them anyway: nothing to steer)
}

int segment_reserve(Segment& s, size_t bytes, const cpu_set_t* near_gpu)
{
    if (s.busy) {
        HIP_TRY(hipEventSynchronize(s.ev));
        s.busy = false;
    }
    if (!s.ev) HIP_TRY(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    if (s.bytes >= bytes) return PSD_OK;
    size_t cap = (bytes + ((size_t)4 << 20) - 1) & ~(((size_t)4 << 20) - 1);
    if (s.h) HIP_TRY(hipHostFree(s.h));
    if (s.d) HIP_TRY(hipFree(s.d));
    s.h = nullptr; s.d = nullptr; s.bytes = 0;
    // page-locked memory lands on the node of the thread that allocates (and first touches) it: be on the GPU's node meanwhile
    cpu_set_t before;
    const bool moved = near_gpu && sched_getaffinity(0, sizeof before, &before) == 0 && sched_setaffinity(0, sizeof(cpu_set_t), near_gpu) == 0;
    hipError_t err = hipHostMalloc((void**)&s.h, cap, hipHostMallocDefault);
    if (err == hipSuccess) memset(s.h, 0, cap);
    if (moved) (void)sched_setaffinity(0, sizeof before, &before);
    if (err == hipSuccess) err = hipMalloc((void**)&s.d, cap);
    if (err != hipSuccess) {
        if (s.h) (void)hipHostFree(s.h);
        s.h = nullptr;
        psd_set_error("psd_upload_rows_batch: staging allocation of %zu bytes failed: %s", cap, hipGetErrorString(err));
        (void)hipGetLastError();
        return err == hipErrorOutOfMemory ? PSD_ERR_NOMEM : PSD_ERR_HIP;
    }
    s.bytes = cap;
    return PSD_OK;
}

}  // namespace

namespace psd {
void feed_release(psd_engine* e)
{
    psd_feed** slot = engine_feed_slot(e);
    psd_feed* f = *slot;
    if (!f) return;
    {
        std::lock_guard<std::mutex> lk(f->m);
        f->stop = true;
    }
    f->cv_work.notify_all();
    for (auto& t : f->threads) t.join();
    for (auto& s : f->seg) {
        if (s.busy) (void)hipEventSynchronize(s.ev);
        if (s.ev) (void)hipEventDestroy(s.ev);
        if (s.h) (void)hipHostFree(s.h);
        if (s.d) (void)hipFree(s.d);
    }
    delete f;
    *slot = nullptr;
}
}  // namespace psd

extern "C" int psd_upload_rows_batch(psd_engine* e, void* d_first_frame, size_t d_frame_stride, const void* const* h_frames, int n_frames,
                                     size_t row_bytes, size_t h_row_stride, const int* rows, int n_rows)
{
    if (!e || n_frames < 0 || n_rows < 0 || row_bytes == 0 || h_row_stride < row_bytes || row_bytes > 0xffffffffull ||
        ((n_frames && n_rows) && (!d_first_frame || !h_frames || !rows))) {
        psd_set_error("psd_upload_rows_batch: invalid argument");
        return PSD_ERR_INVALID;
    }
    for (int i = 0; i < n_rows; i++)
        if (rows[i] < 0 || (i && rows[i] <= rows[i - 1])) { psd_set_error("psd_upload_rows_batch: rows must be ascending and >= 0"); return PSD_ERR_INVALID; }
    for (int i = 0; i < n_frames; i++)
        if (!h_frames[i]) { psd_set_error("psd_upload_rows_batch: frame %d is null", i); return PSD_ERR_INVALID; }
    if (n_frames == 0 || n_rows == 0) return PSD_OK;
    if (n_frames > 65535 || n_rows > 0x7fffffff / 2) { psd_set_error("psd_upload_rows_batch: batch too large"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(psd::engine_device(e)));
    hipStream_t copy_stream = nullptr;
    int rc = psd::engine_copy_stream(e, &copy_stream);
    if (rc != PSD_OK) return rc;
    psd_feed** slot = psd::engine_feed_slot(e);
    if (!*slot) {
        psd_feed* f = new (std::nothrow) psd_feed();
        if (!f) { psd_set_error("psd_upload_rows_batch: out of memory"); return PSD_ERR_NOMEM; }
        f->have_near = cpus_near_gpu(psd::engine_device(e), &f->near_gpu);
        const int helpers = feed_threads() - 1;
        try {
            for (int i = 0; i < helpers; i++) f->threads.emplace_back(worker, f);
        } catch (...) {   // fewer helpers than asked for: the caller gathers the rest itself
        }
        *slot = f;
    }
    psd_feed* f = *slot;
    const size_t header = ((size_t)n_rows * sizeof(int) + kHeaderAlign - 1) & ~(kHeaderAlign - 1);
    const size_t payload = (size_t)n_frames * n_rows * row_bytes;
    Segment& s = f->seg[f->next];
    f->next = (f->next + 1) % kSegments;
    rc = segment_reserve(s, header + payload, f->have_near ? &f->near_gpu : nullptr);
    if (rc != PSD_OK) return rc;
    memcpy(s.h, rows, (size_t)n_rows * sizeof(int));
    // ---- gather: units of about 128 KiB, handed out through one atomic counter
    Job& j = f->job;
    j.frames = h_frames; j.rows = rows; j.n_frames = n_frames; j.n_rows = n_rows;
    j.row_bytes = row_bytes; j.h_row_stride = h_row_stride; j.dst = s.h + header;
    int rpu = (int)(((size_t)128 << 10) / row_bytes);
    j.rows_per_unit = rpu < 1 ? 1 : rpu;
    j.units_per_frame = (n_rows + j.rows_per_unit - 1) / j.rows_per_unit;
    const int helpers = (int)f->threads.size();
    {
        std::lock_guard<std::mutex> lk(f->m);
        f->total_units = (long)j.units_per_frame * n_frames;
        f->next_unit.store(0, std::memory_order_relaxed);
        f->working = helpers;
        f->generation++;
    }
    if (helpers) f->cv_work.notify_all();
    drain(f);
    if (helpers) {
        std::unique_lock<std::mutex> lk(f->m);
        f->cv_done.wait(lk, [&] { return f->working == 0; });
    }
    // ---- one contiguous copy, then the rows to their places
    HIP_TRY(hipMemcpyAsync(s.d, s.h, header + payload, hipMemcpyHostToDevice, copy_stream));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)n_rows, (unsigned)n_frames), dim3(256), 0, copy_stream, s.d, header,
                       static_cast<uint8_t*>(d_first_frame), d_frame_stride, n_rows, (unsigned)row_bytes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s.ev, copy_stream));
    s.busy = true;
    return PSD_OK;
}

extern "C" int psd_cpus_near_device(psd_engine* e, int* cpus, int capacity, int* n_cpus)
{
    if (!e || !n_cpus || capacity < 0 || (capacity > 0 && !cpus)) { psd_set_error("psd_cpus_near_device: invalid argument"); return PSD_ERR_INVALID; }
    *n_cpus = 0;
    cpu_set_t near;
    CPU_ZERO(&near);
    if (!cpus_near_gpu(psd::engine_device(e), &near)) return PSD_OK;
    int n = 0;
    for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &near)) {
            if (n < capacity) cpus[n] = c;
            n++;
        }
    *n_cpus = n;
    return PSD_OK;
}
