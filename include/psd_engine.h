/*
 * psd_engine.h -- C-ABI of the MI355X-native per-frame scoring engine for PySceneDetect.
 *
 * The reference (pure Python) has no FFI of its own: its hot path is the four
 * process_frame() loops that call cv2/numpy.  Each entry point below replaces the pixel
 * arithmetic of one or more reference call sites (paths relative to the reference tree):
 *
 *   psd_score_batch / psd_score_batch_device   (flags select the terms)
 *     PSD_SCORE_HSV_SAD   cv2.cvtColor(BGR2HSV)+split and 3x _mean_pixel_distance
 *                         scenedetect/detectors/content_detector.py:29-36,155,166-169
 *     PSD_SCORE_LUMA_HIST cv2.cvtColor(BGR2YUV)+split and cv2.calcHist
 *                         scenedetect/detectors/histogram_detector.py:156-159
 *     PSD_SCORE_BYTE_SUM  numpy.mean(frame_img)
 *                         scenedetect/detectors/threshold_detector.py:127
 *     PSD_SCORE_EDGES     numpy.median + cv2.Canny + cv2.dilate + _mean_pixel_distance(edges)
 *                         scenedetect/detectors/content_detector.py:170-174,213-239
 *   psd_resize_device / psd_score_downscaled_device   cv2.resize(..., interpolation) in front of the detectors
 *                         scenedetect/scene_manager.py:110,123-140,666-678
 *   psd_epilogue_*                             the O(1)-per-frame decisions that follow the pixel work
 *                         content_detector.py:177-180,192-211 + detector.py:106-224 (FlashFilter)
 *                         adaptive_detector.py:100-143
 *                         histogram_detector.py:98-116,163 (normalize + compareHist + decision)
 *                         threshold_detector.py:100-191
 *
 * All device results are exact integers; every floating-point operation of the reference
 * happens in the epilogues (or in the Python host mirror) in the reference's own order.
 *
 * Conventions: plain C, caller-owned buffers, no exceptions across the ABI.  Every function
 * returning int returns PSD_OK (0) or a negative psd_status; psd_last_error() then holds a
 * thread-local message.  A psd_engine is bound to one HIP device; calls on one engine must be
 * serialised by the caller, distinct engines may be used from distinct threads.
 */
#ifndef PSD_ENGINE_H
#define PSD_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSD_ABI_VERSION 8 /* 2: additive over 1 (downscaled / segmented scoring, host feed, RCCL exchange, device-resident records);
                            * 3: additive over 2 (psd_frame_sums: records without the histogram, psd_score_collect_sums, psd_epilogue_*_sums);
                            * 4: additive over 3 (psd_resize_source_rows, psd_upload_rows: a host feeder uploads only the rows a downscale reads);
                            * 5: additive over 4 (psd_upload_rows_batch: many frames' rows gathered by worker threads into page-locked
                            *    memory and uploaded asynchronously; psd_last_walk_geometry);
                            * 6: additive over 5 (psd_cpus_near_device: the CPUs of the GPU's NUMA node, for hosts that place their decode
                            *    threads and frame buffers themselves);
                            * 7: additive over 6 (psd_score_segments_downscaled_device: MANY clips packed into one batch behind the
                            *    reference's default downscale -- what `detect(path, detector_cls())` of benchmark/__main__.py:44-61
                            *    computes per video, for a whole shard of videos in one launch; psd_allgather_host: the exchange step for
                            *    records a rank already holds on the host; psd_hash_bits_device: HashDetector's DCT / median on the device);
                            * 8: additive over 7 (psd_hist_diff_device + psd_epilogue_hist_cuts_from_diff: HistogramDetector's normalisation and
                            *    compareHist for records that are still in HBM -- 8 bytes per frame travel instead of the 1 KiB histogram) */

typedef enum psd_status {
    PSD_OK = 0,
    PSD_ERR_INVALID = -1,     /* bad argument (maps to ValueError) */
    PSD_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    PSD_ERR_HIP = -3,         /* HIP runtime error; message in psd_last_error() */
    PSD_ERR_UNSUPPORTED = -4, /* valid request the engine does not implement */
    PSD_ERR_NOMEM = -5
} psd_status;

enum {
    PSD_SCORE_HSV_SAD = 1u,
    PSD_SCORE_LUMA_HIST = 2u,
    PSD_SCORE_BYTE_SUM = 4u,
    PSD_SCORE_EDGES = 8u,
    PSD_SCORE_ALL = 15u
};

/* One record per frame.  sad_* / edge_xor are relative to the previous frame (the last frame
 * of the previous call when `prev` is given); they are 0 for a frame with no predecessor
 * (content_detector.py:161-164).  edge_xor = number of pixels whose dilated edge map differs
 * (the reference's delta_edges is 255*edge_xor/(H*W)).  hist is always the full 256-bin
 * histogram of Y; cv2.calcHist with `bins` uniform bins over [0,256) is bin[j*bins/256]. */
typedef struct psd_frame_scores {
    uint64_t sad_h, sad_s, sad_v;
    uint64_t edge_xor;
    uint64_t byte_sum;
    uint32_t hist[256];
} psd_frame_scores; /* 1064 bytes */

/* The head of a record: everything ContentDetector / AdaptiveDetector (content_detector.py:177-190,
 * adaptive_detector.py:100-143) and ThresholdDetector (threshold_detector.py:127) decide from.  A corpus run moves 40
 * instead of 1064 bytes per frame to the host with it (psd_score_collect_sums). */
typedef struct psd_frame_sums {
    uint64_t sad_h, sad_s, sad_v;
    uint64_t edge_xor;
    uint64_t byte_sum;
} psd_frame_sums; /* 40 bytes, the first 40 of psd_frame_scores */

typedef struct psd_engine psd_engine;

int psd_abi_version(void);
const char* psd_last_error(void);
int psd_device_count(int* count);

int psd_create(int device, psd_engine** out);
void psd_destroy(psd_engine* e);

/* Score `n` frames that are already resident in device memory (HBM).
 *   d_frames     device pointer, frame t at d_frames + t*frame_stride, rows at row_stride,
 *                pixels packed B,G,R uint8 (detector.py:56).
 *   d_prev       device pointer to the frame preceding frame 0, or NULL.
 *   edge_kernel  dilation size for PSD_SCORE_EDGES (odd >= 3), or 0 for the reference's
 *                _estimated_kernel_size(width,height) (content_detector.py:39-46).
 *   out          host array of n records (written when the call returns).
 *   stream       hipStream_t to launch on, or NULL for the engine's own stream.
 * Fast path: 16-byte aligned base pointers, frame_stride % 16 == 0, row_stride == 3*width.
 * Anything else takes a slower generic path with identical results.
 * PSD_SCORE_HSV_SAD | PSD_SCORE_EDGES together (ContentDetector with weights.delta_edges > 0 or a
 * StatsManager: content_detector.py:155-174 computes both from the same frame) read the frames once on
 * the fast path when the frame holds whole 16-pixel groups: the HSV pass also emits V = max(B,G,R) and
 * its histogram, which is what numpy.median / cv2.Canny start from (content_detector.py:213-239). */
int psd_score_batch_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width,
                           size_t row_stride, size_t frame_stride, const uint8_t* d_prev,
                           uint32_t flags, int edge_kernel, psd_frame_scores* out, void* stream);

/* Asynchronous form: enqueue the work and the device->pinned-host copy of the records on
 * `stream` and return.  psd_score_collect() waits for it and copies the records to `out`.
 * At most PSD_MAX_INFLIGHT submissions may be pending; they complete in submission order.  The frames (and d_prev) must
 * stay valid and unchanged until the submission has been collected. */
#define PSD_MAX_INFLIGHT 4
int psd_score_submit_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width,
                            size_t row_stride, size_t frame_stride, const uint8_t* d_prev,
                            uint32_t flags, int edge_kernel, void* stream);
int psd_score_collect(psd_engine* e, psd_frame_scores* out, int n);
/* Same, but only the five sums of every record.  A submission that asked for neither PSD_SCORE_LUMA_HIST nor
 * PSD_SCORE_BYTE_SUM (the luma pass did not run: hist is all zero) moves only these 40 bytes per frame from the device to
 * the host in the first place; psd_score_collect() then returns them with a zero histogram, as before. */
int psd_score_collect_sums(psd_engine* e, psd_frame_sums* out, int n);

/* MANY clips of one resolution packed into ONE batch (north_star: "frames from many videos are packed into one device
 * batch"; the reference scores one video per SceneManager, scene_manager.py:578-597, and benchmark/sweep.py:142-187 runs
 * them one after the other): seg_first[0..n_seg) are the batch indices of the first frame of every clip, ascending.
 * Such a frame has no predecessor -- its sad_* / edge_xor stay 0 exactly as for frame 0 of a separate call
 * (content_detector.py:161-164) -- so the records equal those of n_seg separate psd_score_batch_device calls, from one
 * launch per term instead of n_seg.  Frame 0 never has a predecessor here (list it or not).  The submit form pairs with
 * psd_score_collect(). */
int psd_score_segments_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width,
                              size_t row_stride, size_t frame_stride, const int32_t* seg_first, int n_seg,
                              uint32_t flags, int edge_kernel, psd_frame_scores* out, void* stream);
int psd_score_segments_submit_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width,
                                     size_t row_stride, size_t frame_stride, const int32_t* seg_first, int n_seg,
                                     uint32_t flags, int edge_kernel, void* stream);

/* Same for frames in host memory: the engine stages them through its own pinned/device
 * buffers in bounded chunks and carries the one-frame halo between chunks itself (copy and kernels share the
 * engine's stream; hosts that want the copy of batch k+1 to overlap the scoring of batch k use psd_upload* below).
 * h_prev may be NULL. */
int psd_score_batch(psd_engine* e, const uint8_t* h_frames, int n, int height, int width,
                    size_t row_stride, size_t frame_stride, const uint8_t* h_prev, uint32_t flags,
                    int edge_kernel, psd_frame_scores* out);

/* The records of the most recently COLLECTED submission where they were computed, in device memory (n of them):
 * the per-frame score vectors can go straight into a collective (RCCL all-gather over xGMI) or
 * another kernel without the round trip through the host.  The buffer belongs to the engine's ring of
 * PSD_MAX_INFLIGHT record slots: it is overwritten by the PSD_MAX_INFLIGHT-th submission after the one it belongs to. */
int psd_last_records_device(psd_engine* e, const psd_frame_scores** d_recs, int* n);

/* ---- multi-GPU: the exchange step ------------------------------------------------------------------------------------
 * One process per GPU; the pixel work needs no collective (clips are independent, a frame range plus a one-frame halo is
 * self-contained).  What is exchanged is the per-frame score records, so that every rank can run the deterministic
 * epilogues for every clip: one RCCL all-gather over xGMI.  RCCL is loaded at run time
 * (librccl.so; a copy already in the process, e.g. PyTorch's, is reused); PSD_ERR_UNSUPPORTED if there is none.
 *   psd_comm_unique_id   rank 0 makes the 128-byte id and hands it to the other ranks by the host's own means;
 *   psd_comm_create      collective over all ranks of the node (ncclCommInitRank) on the engine's device;
 *   psd_allgather_scores every rank contributes n_local records that sit in device memory (psd_last_records_device) and
 *                        receives everybody's in rank order: counts[r] records of rank r (counts must be the same array
 *                        on every rank, counts[rank] == n_local), h_all = host array of sum(counts) records. */
typedef struct psd_comm psd_comm;
int psd_comm_unique_id(void* id128);
int psd_comm_create(psd_engine* e, int n_ranks, int rank, const void* id128, psd_comm** out);
void psd_comm_destroy(psd_comm* c);
int psd_allgather_scores(psd_comm* c, const psd_frame_scores* d_local, int n_local, const int* counts,
                         psd_frame_scores* h_all);
/* The same exchange for records that are already on the HOST: after a corpus pass a rank holds the records of its clips from
 * several submissions (one per resolution, pieces of long runs, collected as they finished), usually as psd_frame_sums.
 * n_local elements of elem_bytes each (sizeof(psd_frame_sums) or sizeof(psd_frame_scores)) -> page-locked staging -> ONE
 * ncclAllGather of padded per-rank blocks -> h_all: counts[r] elements of every rank r in rank order.  counts as above: the
 * same array on every rank, counts[rank] == n_local -- the sharded flow derives it from the plan every rank computes alike
 * (clips greedy longest-first), so the exchange is this one collective and nothing else.  A rank with a local argument error
 * still takes part (zero-filled) and reports afterwards, like psd_allgather_scores. */
int psd_allgather_host(psd_comm* c, const void* h_local, int n_local, size_t elem_bytes, const int* counts, void* h_all);

/* Device time (ms, HIP events on the launch stream) spent in the scoring kernels of the most
 * recently *collected* submission, and the number of kernel launches it took. */
int psd_last_kernel_ms(psd_engine* e, float* ms, int* launches);
/* How the most recent time-walking launch issued on the CALLING thread (the HSV pass, the fused all-detectors pass, the
 * V-mode pass of the edge term) cut its batch: every workgroup walked `frames_per_chunk` consecutive frames of one of
 * `n_tiles` spatial tiles, so frames k * frames_per_chunk are where a walk starts from a re-read halo frame.  Parity
 * checks sample the oracle at exactly these boundaries (tests/test_gpu_headline_geometry.py, bench.py).  Zeros before the
 * first such launch. */
int psd_last_walk_geometry(psd_engine* e, int* frames_per_chunk, int* n_tiles);

/* Raw device buffer helpers so hosts without their own allocator (plain C, ctypes) can keep
 * batches resident in HBM. */
int psd_device_alloc(psd_engine* e, size_t bytes, void** d_ptr);
int psd_device_free(psd_engine* e, void* d_ptr);
int psd_memcpy_h2d(psd_engine* e, void* d_dst, const void* h_src, size_t bytes);
int psd_memcpy_d2h(psd_engine* e, void* h_dst, const void* d_src, size_t bytes);

/* Feeding frames from the host (the decoder side of scene_manager.py:625-710, backends/pyav.py:322-363):
 *   psd_host_alloc / psd_host_free   page-locked host memory: frames decoded INTO it cross PCIe by DMA at full rate and
 *                                    asynchronously;
 *   psd_upload        blocking host -> device copy that neither touches engine state nor waits for the engine's stream:
 *                     safe from a decode thread while another thread scores (the destination must not be in use);
 *   psd_upload_async  enqueue the copy on the engine's copy stream (page-locked source: truly asynchronous);
 *   psd_upload_fence  wait_on_host = 0: later work on the engine's stream waits for every copy enqueued so far;
 *                     wait_on_host = 1: the calling thread waits for them;
 *   psd_memcpy_d2d    device -> device copy in the order of the engine's stream (e.g. keeping a batch's last frame as the
 *                     next batch's predecessor);
 *   psd_synchronize   the calling thread waits until the engine's stream is idle. */
int psd_host_alloc(psd_engine* e, size_t bytes, void** h_ptr);
int psd_host_free(psd_engine* e, void* h_ptr);
int psd_upload(psd_engine* e, void* d_dst, const void* h_src, size_t bytes);
/* A downscale in front of the detectors (scene_manager.py:110, 123-140, 666-678) reads few of a frame's rows: 2 * dst_h
 * of src_h for INTER_LINEAR, dst_h for INTER_NEAREST (every row for INTER_AREA and the exact 2x2 case).
 *   psd_resize_source_rows  the rows the device downscale of this shape and mode reads, ascending (`rows` has room for
 *                           src_h entries); no engine, no device;
 *   psd_upload_rows         psd_upload for a frame of which only those rows are needed: rows[0 .. n_rows) (ascending) of the
 *                           packed host frame (row pitch h_row_stride >= row_bytes) go to the SAME rows of the packed device
 *                           frame (row pitch row_bytes); the rows in between keep whatever the buffer held (the caller
 *                           vouches that every listed row lies inside both frames).  Runs of equally
 *                           spaced row groups travel as one strided copy each.  Everything the engine computes from a
 *                           downscaled frame (records, thumbnails, the small frame itself) is the same as after a full upload. */
int psd_resize_source_rows(int src_h, int src_w, int dst_h, int dst_w, int interpolation, int* rows, int* n_rows);
int psd_upload_rows(psd_engine* e, void* d_frame, const void* h_frame, size_t row_bytes, size_t h_row_stride,
                    const int* rows, int n_rows);
/* The strided copies psd_upload_rows issues for a row list, for hosts that drive their own copy engine: copy i moves
 * copies[4i+3] groups of copies[4i+1] consecutive rows, copies[4i+2] rows apart, starting at row copies[4i].  `packed`:
 * the host frame's rows are contiguous (otherwise every group is one row).  *n_copies is the full count even when it
 * exceeds max_copies.  No engine, no device. */
int psd_upload_rows_plan(const int* rows, int n_rows, int packed, int* copies, int max_copies, int* n_copies);
/* psd_upload_rows for n_frames separately allocated host frames at once (what a decoder hands out: backends/pyav.py:322-363
 * returns a fresh array per frame): frame i's rows go to the packed device frame at d_first_frame + i * d_frame_stride.
 * The rows are gathered by the engine's worker threads (PSD_FEED_THREADS, default 16) into a ring of page-locked segments,
 * frame after frame without gaps, and travel from there as asynchronous strided copies on the engine's copy stream --
 * the call returns once they are enqueued, the host frames may be reused at once, and psd_upload_fence() orders them like
 * psd_upload_async's.  One call per 8 .. 32 frames keeps PCIe busy where one blocking psd_upload_rows per frame reaches
 * about 60 % of the link (two ~10 us copy set-ups per 1.7 MB).  Call from ONE thread at a time per engine (the decode
 * thread); it neither touches the scoring stream nor the record slots. */
int psd_upload_rows_batch(psd_engine* e, void* d_first_frame, size_t d_frame_stride, const void* const* h_frames, int n_frames,
                          size_t row_bytes, size_t h_row_stride, const int* rows, int n_rows);
/* The CPUs of the NUMA node the engine's GPU hangs off (/sys/bus/pci/devices/<bdf>/numa_node), restricted to the calling
 * thread's affinity mask: cpus[0 .. min(*n_cpus, capacity)) in ascending order, *n_cpus the full count -- 0 when the host has
 * one node, the node is unknown, the mask already lies inside it, or PSD_FEED_NUMA=0.  This is where psd_upload_rows_batch
 * puts its page-locked segments and worker threads; a host that decodes on threads of its own (the reference's decode thread,
 * scene_manager.py:565-572, 625-710) gets the same effect for the frames it allocates by running those threads here: with
 * the frames on the other socket the same feed moves 21 k instead of 31 k 1080p frames/s (profiles/r04_l_feed_numa.txt).
 * The engine itself never changes the affinity of a thread it did not create.  No device work. */
int psd_cpus_near_device(psd_engine* e, int* cpus, int capacity, int* n_cpus);
int psd_upload_async(psd_engine* e, void* d_dst, const void* h_src, size_t bytes);
int psd_upload_fence(psd_engine* e, int wait_on_host);
int psd_memcpy_d2d(psd_engine* e, void* d_dst, const void* d_src, size_t bytes);
int psd_synchronize(psd_engine* e);

/* The device's fixed-point HSV tables (for validation against the oracle). */
int psd_hsv_tables(int32_t sdiv[256], int32_t hdiv180[256]);

/* Debug/validation: dilated Canny edge map (0/255 per pixel, uint8[H*W], host) of one
 * device-resident frame, exactly as PSD_SCORE_EDGES computes it. */
int psd_edge_map_device(psd_engine* e, const uint8_t* d_frame, int height, int width,
                        size_t row_stride, int edge_kernel, uint8_t* h_edges);

/* cv2.resize(src, (dst_w, dst_h), interpolation) for n device-resident BGR frames (packed rows): the downscale
 * SceneManager applies in front of the detectors (scene_manager.py:666-678).  `interpolation` takes cv2's values
 * as the reference's Interpolation enum does (common.py:148-160), and all five are implemented: NEAREST, LINEAR (the default),
 * AREA (when it does not shrink along both axes it is, as in OpenCV, bilinear with area-mode coefficients), LANCZOS4 (OpenCV's
 * integer path for 8-bit images -- HResizeLanczos4 / VResizeLanczos4 with FixedPtCast<int, uchar, 22> and no vector pass -- so it
 * has one result, reproduced here byte for byte) and CUBIC.
 * CUBIC has no single 8-bit result in OpenCV: its vertical pass (VResizeCubicVec_32s8u) runs in float32 on groups of 8 elements
 * with a fixed-point scalar tail, fused or unfused by build, is absent from builds without CV_SIMD, and x86-64 PyPI wheels hand the
 * filter to IPP.  The environment variable PSD_CUBIC_FORM (read once per process) names the form reproduced byte for byte:
 * "sse" (default: OpenCV 4.x without IPP on an SSE2/SSE3 baseline, every product and sum rounded to float32), "fma" (fused
 * multiply-adds: aarch64, FMA baselines), "fixed" (FixedPtCast everywhere).  They differ in about one byte per 50,000.  Any other
 * value: PSD_ERR_INVALID from the first CUBIC call.  An `interpolation` outside 0..4: PSD_ERR_UNSUPPORTED. */
enum psd_interpolation { PSD_INTER_NEAREST = 0, PSD_INTER_LINEAR = 1, PSD_INTER_CUBIC = 2, PSD_INTER_AREA = 3, PSD_INTER_LANCZOS4 = 4 };
int psd_resize_device(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w,
                      size_t src_frame_stride, uint8_t* d_dst, int dst_h, int dst_w,
                      size_t dst_frame_stride, int interpolation, void* stream);
/* The reference's default pipeline in one call (scene_manager.py:110,123-140,666-678 followed by the detectors'
 * process_frame): cv2.resize(frame, (dst_w, dst_h), interpolation) and then the terms of `flags` on the RESIZED frames.
 * Source frames: packed rows (row stride 3*src_w), frame t at d_frames + t*frame_stride; d_prev is the SOURCE-size
 * frame preceding frame 0 or NULL.  Records are those psd_score_batch_device would return for the resized frames.
 * With INTER_LINEAR and any set of the HSV, luma-histogram and byte-sum terms (Content / Adaptive / Histogram / Threshold
 * detectors) on 16-byte aligned frames the resized frame never exists in memory: one kernel reads the 2*dst_h source
 * rows that carry taps, interpolates and scores.  The edge term and the other interpolation modes resize into an
 * engine-owned buffer first.
 * The submit form pairs with psd_score_collect() like psd_score_submit_device(). */
int psd_score_downscaled_device(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w,
                                size_t frame_stride, const uint8_t* d_prev, int dst_h, int dst_w,
                                int interpolation, uint32_t flags, int edge_kernel, psd_frame_scores* out,
                                void* stream);
int psd_score_downscaled_submit_device(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w,
                                       size_t frame_stride, const uint8_t* d_prev, int dst_h, int dst_w,
                                       int interpolation, uint32_t flags, int edge_kernel, void* stream);

/* psd_score_segments_device behind the downscale: MANY clips of one resolution packed into ONE batch, every frame resized as
 * SceneManager does by default (auto_downscale: scene_manager.py:110,123-140, applied at :666-678) and then scored.  This is the
 * arithmetic of the reference's own benchmark -- detect(video, detector_cls()) per video, benchmark/__main__.py:44-61, i.e. a
 * SceneManager with auto_downscale=True in front of every detector -- for a whole shard of videos in one launch
 * (north_star: "frames from many videos are packed into one device batch and sharded").
 *   seg_first[0..n_seg)  batch indices of the first frame of every clip, ascending; such a frame has no predecessor, so the
 *                        records equal those of n_seg separate psd_score_downscaled_device calls (and of n_seg SceneManagers).
 *   dst_h, dst_w         max(1, round(src / factor)) per axis, factor = compute_downscale_factor(max(src_w, src_h)).
 * Everything else as psd_score_downscaled_device: INTER_LINEAR with any set of the HSV, luma-histogram and byte-sum terms is ONE
 * fused kernel (the resized frames never exist in memory; the clip-start flag is an SGPR read in front of the next frame's DMA
 * issue); the edge term and INTER_NEAREST / INTER_AREA resize into an engine-owned buffer first.  The submit form pairs with
 * psd_score_collect() / psd_score_collect_sums(). */
int psd_score_segments_downscaled_device(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w,
                                         size_t frame_stride, const int32_t* seg_first, int n_seg, int dst_h, int dst_w,
                                         int interpolation, uint32_t flags, int edge_kernel, psd_frame_scores* out,
                                         void* stream);
int psd_score_segments_downscaled_submit_device(psd_engine* e, const uint8_t* d_frames, int n, int src_h, int src_w,
                                                size_t frame_stride, const int32_t* seg_first, int n_seg, int dst_h,
                                                int dst_w, int interpolation, uint32_t flags, int edge_kernel,
                                                void* stream);

/* psd_resize_device(..., PSD_INTER_LINEAR, ...) */
int psd_resize_linear_device(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w,
                             size_t src_frame_stride, uint8_t* d_dst, int dst_h, int dst_w,
                             size_t dst_frame_stride, void* stream);

/* HashDetector.hash_frame, front half (hash_detector.py:125-129): cv2.cvtColor(BGR2GRAY) followed by
 * cv2.resize(gray, (size, size), INTER_AREA) for n frames; `size` is the detector's size*lowpass.
 * h_thumbs: host array uint8[n][size][size].  Decimation only (size <= width, height), size <= 256.
 * The *_device form takes frames resident in HBM (same layout rules as psd_score_batch_device) and, like
 * it, reports the kernel time through psd_last_kernel_ms(). */
int psd_hash_thumbs_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width,
                           size_t row_stride, size_t frame_stride, int size, uint8_t* h_thumbs);
int psd_hash_thumbs(psd_engine* e, const uint8_t* h_frames, int n, int height, int width,
                    size_t row_stride, size_t frame_stride, int size, uint8_t* h_thumbs);
/* HashDetector.hash_frame as a whole for n device-resident frames (hash_detector.py:125-151): the thumbnails above and then, still
 * on the device, psd_epilogue_hash_bits -- scaling by the maximum, the 2-D DCT-II in float64 (the same sums in the same order),
 * one rounding to float32, the float32 median, bits = coefficient > median.  h_bits: uint8[n][hash_size^2] of 0 / 1, identical to
 * psd_epilogue_hash_bits(thumbs); h_thumbs: uint8[n][size][size] as well, or NULL.  256 instead of 1024 bytes per frame leave the
 * device and the step does not end in host arithmetic.  size <= 64 or so (the transform works in a workgroup's LDS): larger ones
 * return PSD_ERR_UNSUPPORTED -- take psd_hash_thumbs_device + psd_epilogue_hash_bits.  Kernel time: psd_last_kernel_ms(). */
int psd_hash_bits_device(psd_engine* e, const uint8_t* d_frames, int n, int height, int width, size_t row_stride,
                         size_t frame_stride, int size, int hash_size, uint8_t* h_bits, uint8_t* h_thumbs);

/* ---- host epilogues (no device involved) -------------------------------------------------
 * Frame positions are frame numbers first_frame .. first_frame+n-1 at a constant frame rate
 * fps_num/fps_den.  min_scene_len is given either in frames (min_len_frames >= 0 and
 * min_len_secs < 0) or in seconds (min_len_secs >= 0), like the reference's TimecodeLike.
 * Cut frame numbers are written to cuts[0..*n_cuts) (capacity n+1).  Constant frame rate only: positions that are
 * presentation timestamps (variable frame rate sources) are decided frame by frame by the host-language detectors. */

typedef struct psd_content_params {
    double threshold;        /* ContentDetector(threshold=27.0) */
    double weights[4];       /* Components(delta_hue, delta_sat, delta_lum, delta_edges) */
    int filter_mode;         /* 0 = MERGE, 1 = SUPPRESS (detector.py:109-115) */
    int64_t min_len_frames;
    double min_len_secs;
} psd_content_params;

/* content_val[t] and the four components (any output pointer may be NULL).
 * first_has_prev: 0 if recs[0] is the first frame of the video (score 0.0). */
int psd_epilogue_content_scores(const psd_frame_scores* recs, int n, int height, int width,
                                const double weights[4], int first_has_prev, double* content_val,
                                double* delta_hue, double* delta_sat, double* delta_lum,
                                double* delta_edges);

/* Same from sums: element t sits at (const char*)sums + t * stride_bytes -- sizeof(psd_frame_sums) for what
 * psd_score_collect_sums returned, sizeof(psd_frame_scores) to read the heads of full records in place. */
int psd_epilogue_content_scores_sums(const psd_frame_sums* sums, size_t stride_bytes, int n, int height, int width,
                                     const double weights[4], int first_has_prev, double* content_val,
                                     double* delta_hue, double* delta_sat, double* delta_lum,
                                     double* delta_edges);

int psd_epilogue_content_cuts(const double* content_val, int n, int64_t first_frame, int64_t fps_num,
                              int64_t fps_den, const psd_content_params* p, int64_t* cuts,
                              int* n_cuts);

typedef struct psd_adaptive_params {
    double adaptive_threshold; /* 3.0 */
    double min_content_val;    /* 15.0 */
    int window_width;          /* 2 */
    int64_t min_len_frames;
    double min_len_secs;
} psd_adaptive_params;

int psd_epilogue_adaptive_cuts(const double* content_val, int n, int64_t first_frame, int64_t fps_num,
                               int64_t fps_den, const psd_adaptive_params* p, double* adaptive_ratio,
                               int64_t* cuts, int* n_cuts);

typedef struct psd_hist_params {
    double threshold; /* HistogramDetector(threshold=0.20): cut iff correl <= 1-threshold */
    int bins;         /* 128 */
    int64_t min_len_frames;
    double min_len_secs;
} psd_hist_params;

/* hist_diff[t] (t>=1, or t>=0 when prev_rec != NULL) = compareHist(CORREL) of the L2-normalised
 * `bins`-bin histograms of consecutive frames. */
int psd_epilogue_hist_cuts(const psd_frame_scores* recs, int n, const psd_frame_scores* prev_rec,
                           int64_t first_frame, int64_t fps_num, int64_t fps_den,
                           const psd_hist_params* p, double* hist_diff, int64_t* cuts, int* n_cuts);

/* The same values for n records that are still in DEVICE memory (psd_last_records_device of a submission with PSD_SCORE_LUMA_HIST;
 * histogram_detector.py:98,156-163 = calcHist's re-binning, cv2.normalize(NORM_L2) in float32, cv2.compareHist(CORREL) in float64),
 * frame pairs in parallel on the device with every sum in the order psd_epilogue_hist_cuts keeps: bit for bit its hist_diff.
 * h_diff[0] = NaN (no predecessor inside the batch); the caller overwrites h_diff[f] with NaN where frame f starts a clip.  With it
 * the histograms need not travel: collect the sums (psd_score_collect_sums, 40 bytes per frame) and these 8 bytes per frame.
 * The records must be COMPLETE (those of a collected submission): the call does not order itself behind the engine's scoring stream --
 * `stream` = NULL runs it on a side stream of the engine's, beside whatever submission is in flight.  Synchronous. */
int psd_hist_diff_device(psd_engine* e, const psd_frame_scores* d_recs, int n, int bins, double* h_diff, void* stream);
/* ... and the decision of psd_epilogue_hist_cuts over such values (NaN = no predecessor: never a cut). */
int psd_epilogue_hist_cuts_from_diff(const double* hist_diff, int n, int64_t first_frame, int64_t fps_num, int64_t fps_den,
                                     const psd_hist_params* p, int64_t* cuts, int* n_cuts);

/* The two per-frame steps of the above, for hosts that decide frame by frame (HistogramDetector.process_frame):
 * cv2.calcHist(bins) + cv2.normalize (L2, float32) of one 256-bin luma histogram, and cv2.compareHist(CORREL). */
int psd_epilogue_hist_normalize(const uint32_t hist256[256], int bins, float* out);
int psd_epilogue_hist_correl(const float* h1, const float* h2, int bins, double* out);

typedef struct psd_threshold_params {
    int threshold;       /* int(threshold), default 12 */
    int method;          /* 0 = FLOOR, 1 = CEILING */
    double fade_bias;    /* 0.0 */
    int add_final_scene; /* 0 */
    int64_t min_len_frames;
    double min_len_secs;
} psd_threshold_params;

/* Runs process_frame over all n frames and then post_process(last frame). */
int psd_epilogue_threshold_cuts(const psd_frame_scores* recs, int n, int height, int width,
                                int64_t first_frame, int64_t fps_num, int64_t fps_den,
                                const psd_threshold_params* p, double* average_rgb, int64_t* cuts,
                                int* n_cuts);
int psd_epilogue_threshold_cuts_sums(const psd_frame_sums* sums, size_t stride_bytes, int n, int height, int width,
                                     int64_t first_frame, int64_t fps_num, int64_t fps_den,
                                     const psd_threshold_params* p, double* average_rgb, int64_t* cuts,
                                     int* n_cuts);

/* HashDetector.hash_frame, back half (hash_detector.py:131-151): scale by the maximum, 2-D DCT-II, keep
 * the hash_size x hash_size low frequencies, threshold at their median.  bits: uint8[n][hash_size^2] of 0/1.
 * The DCT is evaluated in float64 and rounded once to float32 (cv2.dct's own float32 operation order
 * depends on the OpenCV build and is not restatable). */
int psd_epilogue_hash_bits(const uint8_t* thumbs, int n, int size, int hash_size, uint8_t* bits);

typedef struct psd_hash_params {
    double threshold; /* HashDetector(threshold=0.35): cut iff hamming/hash_size^2 >= threshold */
    int hash_size;    /* 8 */
    int64_t min_len_frames;
    double min_len_secs;
} psd_hash_params;

/* hash_dist[t] (t>=1, or t>=0 when prev_bits != NULL) = normalised Hamming distance of consecutive hashes
 * (hash_detector.py:97-116). */
int psd_epilogue_hash_cuts(const uint8_t* bits, int n, const uint8_t* prev_bits, int64_t first_frame,
                           int64_t fps_num, int64_t fps_den, const psd_hash_params* p,
                           double* hash_dist, int64_t* cuts, int* n_cuts);

#ifdef __cplusplus
}
#endif
#endif /* PSD_ENGINE_H */
