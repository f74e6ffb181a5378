/*
 * abi_cpu.c -- the entry points of include/psd_engine.h that a per-frame binding touches (psd_create, psd_destroy,
 * psd_score_batch, psd_hash_thumbs, psd_last_error, psd_abi_version, and for a previous frame kept "on the device": psd_device_alloc,
 * psd_device_free, psd_memcpy_h2d, psd_score_batch_device -- device memory is host memory here), implemented on the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/.  It exists so that the reference-side binding of
 * INTEGRATION.md B (integration/scenedetect_amd.py) can be EXECUTED against the unmodified reference in the build
 * container, where /root/reference is present but no GPU is: tests/test_reference_binding.py loads this library
 * instead of libpsd_hip.so.  The product never loads it and has no CPU fallback.
 * Edge term: numpy.median + cv2.Canny + cv2.dilate + _mean_pixel_distance(edges) as
 * scenedetect/detectors/content_detector.py:170-174,213-239 composes them.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/psd_engine.h"

typedef struct {
    uint64_t sad_h, sad_s, sad_v, edge_xor, byte_sum;
    uint32_t hist[256];
} orc_frame_scores;
void orc_score_batch_flags(const uint8_t* frames, int n, int h, int w, size_t row_stride, size_t frame_stride,
                           const uint8_t* prev, orc_frame_scores* out, unsigned flags);
void orc_bgr2hsv_planes(const uint8_t* src, size_t src_step, uint8_t* hp, uint8_t* sp, uint8_t* vp, int h, int w);
void orc_canny(const uint8_t* src, size_t step, int h, int w, double low_thresh, double high_thresh, uint8_t* dst);
void orc_dilate_rect(const uint8_t* src, size_t step, int h, int w, int kh, int kw, uint8_t* dst);
void orc_bgr2gray(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int h, int w);
int orc_resize_area_u8(const uint8_t* src, size_t sstep, int sh, int sw, uint8_t* dst, size_t dstep, int dh, int dw);
void orc_resize_area_upscale_u8(const uint8_t* src, size_t sstep, int sh, int sw, int cn, uint8_t* dst, size_t dstep, int dh, int dw);

struct psd_engine { int unused; };
static __thread char g_err[256];

int psd_abi_version(void) { return PSD_ABI_VERSION; }
const char* psd_last_error(void) { return g_err; }

int psd_create(int device, psd_engine** out)
{
    (void)device;
    if (!out) { snprintf(g_err, sizeof g_err, "psd_create: null out"); return PSD_ERR_INVALID; }
    *out = (psd_engine*)calloc(1, sizeof(psd_engine));
    return *out ? PSD_OK : PSD_ERR_NOMEM;
}

void psd_destroy(psd_engine* e) { free(e); }

static int estimated_kernel_size(int width, int height)
{
    int size = 4 + (int)nearbyint(sqrt((double)width * (double)height) / 192.0);   /* Python round: half to even */
    return size % 2 == 0 ? size + 1 : size;
}

/* dilated Canny edge map of one frame (0 / 255 per pixel) */
static void edge_map(const uint8_t* frame, int h, int w, size_t row_stride, int k, uint8_t* out)
{
    const size_t np = (size_t)h * w;
    uint8_t* planes = (uint8_t*)malloc(np * 3);
    uint8_t* lum = planes + 2 * np;
    orc_bgr2hsv_planes(frame, row_stride, planes, planes + np, lum, h, w);
    /* numpy.median of the V plane: mean of the two middle order statistics */
    size_t hist[256] = {0};
    for (size_t i = 0; i < np; i++) hist[lum[i]]++;
    const size_t ka = (np - 1) / 2, kb = np / 2;
    size_t run = 0;
    int a = -1, b = -1;
    for (int v = 0; v < 256 && b < 0; v++) {
        run += hist[v];
        if (a < 0 && run > ka) a = v;
        if (run > kb) b = v;
    }
    const double median = (a + b) / 2.0, sigma = 1.0 / 3.0;
    const double lo = (1.0 - sigma) * median, hi = (1.0 + sigma) * median;
    const int low = (int)(lo > 0 ? lo : 0), high = (int)(hi < 255 ? hi : 255);
    uint8_t* edges = (uint8_t*)malloc(np);
    orc_canny(lum, (size_t)w, h, w, (double)low, (double)high, edges);
    orc_dilate_rect(edges, (size_t)w, h, w, k, k, out);
    free(edges);
    free(planes);
}

int psd_score_batch(psd_engine* e, const uint8_t* h_frames, int n, int height, int width, size_t row_stride,
                    size_t frame_stride, const uint8_t* h_prev, uint32_t flags, int edge_kernel, psd_frame_scores* out)
{
    if (!e || n < 0 || height <= 0 || width <= 0 || (n > 0 && (!h_frames || !out)) || row_stride < (size_t)width * 3 ||
        flags == 0 || (flags & ~(uint32_t)PSD_SCORE_ALL)) {
        snprintf(g_err, sizeof g_err, "invalid argument");
        return PSD_ERR_INVALID;
    }
    if ((flags & PSD_SCORE_EDGES) && edge_kernel != 0 && (edge_kernel < 3 || edge_kernel % 2 == 0)) {
        snprintf(g_err, sizeof g_err, "kernel_size must be odd integer >= 3");
        return PSD_ERR_INVALID;
    }
    if (n == 0) return PSD_OK;
    orc_score_batch_flags(h_frames, n, height, width, row_stride, frame_stride, h_prev, (orc_frame_scores*)out, flags & 7u);
    if (flags & PSD_SCORE_EDGES) {
        const size_t np = (size_t)height * width;
        const int k = edge_kernel ? edge_kernel : estimated_kernel_size(width, height);
        uint8_t* cur = (uint8_t*)malloc(np);
        uint8_t* last = (uint8_t*)malloc(np);
        int have_last = 0;
        if (h_prev) { edge_map(h_prev, height, width, row_stride, k, last); have_last = 1; }
        for (int t = 0; t < n; t++) {
            edge_map(h_frames + (size_t)t * frame_stride, height, width, row_stride, k, cur);
            uint64_t x = 0;
            if (have_last)
                for (size_t i = 0; i < np; i++) x += cur[i] != last[i];
            out[t].edge_xor = x;
            uint8_t* sw = cur; cur = last; last = sw;
            have_last = 1;
        }
        free(cur); free(last);
    }
    return PSD_OK;
}

/* "Device" memory of this build is host memory: the binding keeps a detector's previous frame in it. */
int psd_device_alloc(psd_engine* e, size_t bytes, void** d_ptr)
{
    if (!e || !d_ptr) { snprintf(g_err, sizeof g_err, "psd_device_alloc: null argument"); return PSD_ERR_INVALID; }
    *d_ptr = malloc(bytes ? bytes : 1);
    if (!*d_ptr) { snprintf(g_err, sizeof g_err, "malloc(%zu) failed", bytes); return PSD_ERR_NOMEM; }
    return PSD_OK;
}

int psd_device_free(psd_engine* e, void* d_ptr)
{
    if (!e) { snprintf(g_err, sizeof g_err, "null engine"); return PSD_ERR_INVALID; }
    free(d_ptr);
    return PSD_OK;
}

int psd_memcpy_h2d(psd_engine* e, void* d_dst, const void* h_src, size_t bytes)
{
    if (!e || (bytes && (!d_dst || !h_src))) { snprintf(g_err, sizeof g_err, "psd_memcpy_h2d: null argument"); return PSD_ERR_INVALID; }
    memcpy(d_dst, h_src, bytes);
    return PSD_OK;
}

int psd_score_batch_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride,
                           size_t frame_stride, const uint8_t* d_prev, uint32_t flags, int edge_kernel, psd_frame_scores* out,
                           void* stream)
{
    (void)stream;
    return psd_score_batch(e, d_frames, n, height, width, row_stride, frame_stride, d_prev, flags, edge_kernel, out);
}

/* HashDetector's front half: cv2.cvtColor(BGR2GRAY) + cv2.resize((size, size), INTER_AREA) per frame
 * (scenedetect/detectors/hash_detector.py:125-129), like oracle/lib.py's hash_thumbs. */
int psd_hash_thumbs(psd_engine* e, const uint8_t* h_frames, int n, int height, int width, size_t row_stride, size_t frame_stride,
                    int size, uint8_t* h_thumbs)
{
    if (!e || n < 0 || height <= 0 || width <= 0 || size <= 0 || (n > 0 && (!h_frames || !h_thumbs)) || row_stride < (size_t)width * 3) {
        snprintf(g_err, sizeof g_err, "psd_hash_thumbs: bad arguments");
        return PSD_ERR_INVALID;
    }
    uint8_t* gray = (uint8_t*)malloc((size_t)height * width);
    if (!gray) return PSD_ERR_NOMEM;
    for (int t = 0; t < n; ++t) {
        uint8_t* out = h_thumbs + (size_t)t * size * size;
        orc_bgr2gray(h_frames + (size_t)t * frame_stride, row_stride, gray, (size_t)width, height, width);
        if (size > width || size > height)
            orc_resize_area_upscale_u8(gray, (size_t)width, height, width, 1, out, (size_t)size, size, size);
        else if (orc_resize_area_u8(gray, (size_t)width, height, width, out, (size_t)size, size, size) != 0) {
            free(gray);
            snprintf(g_err, sizeof g_err, "psd_hash_thumbs: unsupported size");
            return PSD_ERR_UNSUPPORTED;
        }
    }
    free(gray);
    return PSD_OK;
}

/* ---- the exchange step (psd_comm_* / psd_allgather_host / psd_allgather_scores) without RCCL ----------------------------------------
 * A stand-in communicator over POSIX shared memory, so that the code path a multi-GPU run takes -- distributed.native_comm_for ->
 * NativeComm -> psd_allgather_host, instead of torch.distributed -- can be executed at world sizes 2 ... 8 in the CPU suite
 * (tests/test_distributed.py).  Same contract as the RCCL implementation in pyscenedetect_amd/csrc/psd_comm.cpp: counts identical on
 * every rank, ragged blocks padded to the largest, a rank with a local argument error still takes part (zero-filled) and reports
 * afterwards.  The unique id names the segment; a two-phase counter barrier (arrive / leave) orders writes and reads. */
#include <fcntl.h>
#include <stdatomic.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define PSD_SHM_BYTES ((size_t)256 << 20)      /* sparse: only touched pages exist */

typedef struct {
    _Atomic int arrived, left;
    _Atomic unsigned generation;
} shm_head;

struct psd_comm {
    int n_ranks, rank, fd;
    char name[64];
    uint8_t* base;
};

int psd_comm_unique_id(void* id128)
{
    if (!id128) { snprintf(g_err, sizeof g_err, "psd_comm_unique_id: null argument"); return PSD_ERR_INVALID; }
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id128, 0, 128);
    snprintf((char*)id128, 64, "/psd_comm_%d_%ld_%ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
    return PSD_OK;
}

int psd_comm_create(psd_engine* e, int n_ranks, int rank, const void* id128, psd_comm** out)
{
    if (!e || !id128 || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        snprintf(g_err, sizeof g_err, "psd_comm_create: invalid argument");
        return PSD_ERR_INVALID;
    }
    psd_comm* c = (psd_comm*)calloc(1, sizeof *c);
    if (!c) return PSD_ERR_NOMEM;
    c->n_ranks = n_ranks; c->rank = rank;
    memcpy(c->name, id128, 63);
    c->fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (c->fd < 0 || ftruncate(c->fd, (off_t)PSD_SHM_BYTES) != 0) {
        snprintf(g_err, sizeof g_err, "psd_comm_create: shared memory segment %s unavailable", c->name);
        if (c->fd >= 0) close(c->fd);
        free(c);
        return PSD_ERR_HIP;
    }
    c->base = (uint8_t*)mmap(NULL, PSD_SHM_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (c->base == MAP_FAILED) { close(c->fd); free(c); snprintf(g_err, sizeof g_err, "psd_comm_create: mmap failed"); return PSD_ERR_HIP; }
    *out = c;      /* (a fresh segment is zero: the counters start at 0 whoever arrives first) */
    return PSD_OK;
}

void psd_comm_destroy(psd_comm* c)
{
    if (!c) return;
    munmap(c->base, PSD_SHM_BYTES);
    close(c->fd);
    if (c->rank == 0) shm_unlink(c->name);
    free(c);
}

static void shm_barrier(psd_comm* c, _Atomic int* counter)
{
    shm_head* h = (shm_head*)c->base;
    const unsigned gen = atomic_load(&h->generation);
    if (atomic_fetch_add(counter, 1) + 1 == c->n_ranks) {
        atomic_store(counter, 0);
        atomic_fetch_add(&h->generation, 1);
    } else {
        while (atomic_load(&h->generation) == gen) { struct timespec ts = {0, 50000}; nanosleep(&ts, NULL); }
    }
}

int psd_allgather_host(psd_comm* c, const void* h_local, int n_local, size_t elem_bytes, const int* counts, void* h_all)
{
    if (!c || !counts || elem_bytes == 0) { snprintf(g_err, sizeof g_err, "psd_allgather_host: invalid argument"); return PSD_ERR_INVALID; }
    char local_msg[160] = "";
    int local_error = 0;
    if (n_local < 0 || (n_local > 0 && !h_local)) {
        snprintf(local_msg, sizeof local_msg, "psd_allgather_host: invalid local records (n_local = %d)", n_local);
        local_error = 1; n_local = 0;
    } else if (counts[c->rank] != n_local) {
        snprintf(local_msg, sizeof local_msg, "psd_allgather_host: counts[%d] = %d but this rank contributes %d records", c->rank,
                 counts[c->rank], n_local);
        local_error = 1;
        if (counts[c->rank] >= 0 && n_local > counts[c->rank]) n_local = counts[c->rank];
    }
    size_t cap = 1, total = 0;
    for (int i = 0; i < c->n_ranks; i++) {
        if (counts[i] < 0) { snprintf(g_err, sizeof g_err, "psd_allgather_host: negative count"); return PSD_ERR_INVALID; }
        if ((size_t)counts[i] > cap) cap = (size_t)counts[i];
        total += (size_t)counts[i];
    }
    const size_t block = cap * elem_bytes;
    if (total > 0 && !h_all) { snprintf(g_err, sizeof g_err, "psd_allgather_host: null output"); return PSD_ERR_INVALID; }
    if (4096 + block * (size_t)c->n_ranks > PSD_SHM_BYTES) { snprintf(g_err, sizeof g_err, "psd_allgather_host: stand-in segment too small"); return PSD_ERR_NOMEM; }
    shm_head* h = (shm_head*)c->base;
    uint8_t* data = c->base + 4096;
    memset(data + (size_t)c->rank * block, 0, block);
    if (n_local > 0) memcpy(data + (size_t)c->rank * block, h_local, (size_t)n_local * elem_bytes);
    shm_barrier(c, &h->arrived);                 /* every block is written */
    size_t off = 0;
    for (int i = 0; i < c->n_ranks && !local_error; i++) {
        memcpy((uint8_t*)h_all + off, data + (size_t)i * block, (size_t)counts[i] * elem_bytes);
        off += (size_t)counts[i] * elem_bytes;
    }
    shm_barrier(c, &h->left);                    /* every block is read: the segment may be reused */
    if (local_error) { snprintf(g_err, sizeof g_err, "%s (the collective was completed with zero-filled records)", local_msg); return PSD_ERR_INVALID; }
    return PSD_OK;
}

/* records that sit "on the device" (host memory here) */
int psd_allgather_scores(psd_comm* c, const psd_frame_scores* d_local, int n_local, const int* counts, psd_frame_scores* h_all)
{
    return psd_allgather_host(c, d_local, n_local, sizeof(psd_frame_scores), counts, h_all);
}

/* ---- many clips packed in one batch behind the downscale (psd_score_segments_downscaled_device), on the oracle --------------------
 * cv2.resize(frame, (dst_w, dst_h), INTER_LINEAR) per frame (oracle restatement), then the terms per clip with no predecessor for
 * a clip's first frame: what n_seg SceneManagers with auto_downscale compute (scene_manager.py:666-678 + the detectors).  The CPU
 * build of the entry point the batch front end of INTEGRATION.md B binds; INTER_LINEAR only (the reference's default). */
void orc_resize_linear_u8(const uint8_t* src, size_t sstep, int sh, int sw, int cn, uint8_t* dst, size_t dstep, int dh, int dw);

int psd_score_segments_downscaled_device(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w, size_t frame_stride,
                                         const int32_t* seg_first, int n_seg, int dst_h, int dst_w, int interpolation, uint32_t flags,
                                         int edge_kernel, psd_frame_scores* out, void* stream)
{
    (void)stream;
    if (!e || n < 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || (n > 0 && (!d_frames || !out)) || n_seg < 0 ||
        (n_seg > 0 && !seg_first) || flags == 0 || (flags & ~(uint32_t)PSD_SCORE_ALL)) {
        snprintf(g_err, sizeof g_err, "psd_score_segments_downscaled_device: invalid argument");
        return PSD_ERR_INVALID;
    }
    for (int i = 0; i < n_seg; i++)
        if (seg_first[i] < 0 || seg_first[i] >= n || (i > 0 && seg_first[i] <= seg_first[i - 1])) {
            snprintf(g_err, sizeof g_err, "segment table must hold ascending frame indices inside the batch (entry %d = %d, n = %d)", i, seg_first[i], n);
            return PSD_ERR_INVALID;
        }
    if (interpolation != PSD_INTER_LINEAR) { snprintf(g_err, sizeof g_err, "the CPU build resizes with INTER_LINEAR only"); return PSD_ERR_UNSUPPORTED; }
    if (n == 0) return PSD_OK;
    const size_t small = (size_t)dst_h * dst_w * 3;
    uint8_t* buf = (uint8_t*)malloc(small * (size_t)n);
    if (!buf) return PSD_ERR_NOMEM;
    for (int t = 0; t < n; t++)
        orc_resize_linear_u8(d_frames + (size_t)t * frame_stride, (size_t)src_w * 3, src_h, src_w, 3, buf + (size_t)t * small, (size_t)dst_w * 3, dst_h, dst_w);
    int rc = PSD_OK, a = 0;
    for (int i = 0; i <= n_seg && rc == PSD_OK; i++) {      /* clips: [0, first[0]), [first[0], first[1]), ... (frame 0 starts one either way) */
        const int b = i < n_seg ? seg_first[i] : n;
        if (b > a) rc = psd_score_batch(e, buf + (size_t)a * small, b - a, dst_h, dst_w, (size_t)dst_w * 3, small, NULL, flags, edge_kernel, out + a);
        a = b;
    }
    free(buf);
    return rc;
}
