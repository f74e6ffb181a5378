// psd_internal.h -- shared declarations between the kernels and the C-ABI engine.
#ifndef PSD_INTERNAL_H
#define PSD_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "psd_engine.h"

// Cache policy of the LDS-DMA frame stream (aux operand of global_load_lds): 2 = nt.  Every frame byte is read once,
// by one CU; a plain 16-byte-load microbenchmark streams 7.1 TB/s with nt and 6.1 TB/s with the default policy
// (profiles/r01_k_ubench_stream_read.txt).
#ifndef PSD_DMA_AUX
#define PSD_DMA_AUX 2
#endif

namespace psd {

// Workgroup barrier of kernels whose waves share nothing but LDS: it waits for this wave's LDS operations and not for its
// global ones.  __syncthreads() carries a workgroup-scope fence, for which hipcc also emits s_waitcnt vmcnt(0) -- and that drains
// the LDS-DMA prefetch (global_load_lds) a wave keeps in flight across the barrier, which is the point of issuing it early.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kScoreWG = 1024;  // threads per workgroup of the scoring kernels (16 waves) ...
#ifndef PSD_HSV_WG
#define PSD_HSV_WG 256
#endif
constexpr int kHsvWG = PSD_HSV_WG;  // ... except the staged HSV-only pass: 4-wave workgroups, four per CU
constexpr int kScoreG = 2;      // 16-pixel groups per lane per frame

struct ScoreParams {
    const uint8_t* frames;   // device, frame t at frames + t*frame_stride
    const uint8_t* prev;     // device, frame preceding frame 0, or nullptr
    psd_frame_scores* out;   // device, n records, zero-initialised
    const uint32_t* lut;     // device, [0..255] = sdiv << 4, [256..511] = hdiv180 << 4
    const uint32_t* lutf;    // device, float32 bit patterns: [0..255] = nextafter(sdiv / 4096), [256..511] = hdiv180 / 4096
    const uint8_t* seg;      // device, n flags: frame t starts a clip (no predecessor), or nullptr
    uint8_t* vout;           // V mode (HSV term + the edge term's front end in one pass): V plane of frame t at vout + t*npix,
    uint32_t* vhist;         //   its 256-bin histogram at vhist + t*256 (zeroed by the caller); nullptr otherwise
    size_t frame_stride;
    size_t row_stride;
    long npix;               // height*width
    int width;
    int n;
    // filled in by the launcher
    int group_begin;         // 16-px groups [group_begin, group_end) of every frame
    int group_end;
    int n_tiles;
    int groups_per_tile;
    int frames_per_chunk;
};

hipError_t launch_score_frames(ScoreParams p, bool hsv, bool luma, bool fast, int target_blocks,
                               hipStream_t stream, int* launches);
// V mode (ScoreParams::vout) needs the staged kernel and frames made of whole 16-pixel groups
bool score_v_mode_available(long npix);
// every launcher of a time-walking kernel leaves how it cut the batch (psd_last_walk_geometry; thread-local, psd_engine.cpp)
void note_walk_geometry(int frames_per_chunk, int n_tiles);

// The edge term behind the default downscale (psd_resize_kernels.hip): cv2.resize(INTER_LINEAR) + the HSV term + the resized
// frames' V planes and V histograms from one pass over the full-size frames -- the resized frame itself never exists.
// DownSrc: where the full-size frames of a submission are (edges_score takes it in place of small frames).
struct DownSrc {
    const uint8_t* frames;   // n full-size packed BGR frames, frame_stride apart
    const uint8_t* prev;     // full-size frame preceding frame 0, or null
    int src_h, src_w;
    size_t frame_stride;
};
int resize_linear_score_vplane(psd_engine* e, const uint8_t* d_src, int n, int src_h, int src_w, size_t src_frame_stride, const uint8_t* d_prev,
                               int dst_h, int dst_w, psd_frame_scores* d_out, hipStream_t stream, int* launches, const uint8_t* d_seg,
                               uint8_t* d_vout, uint32_t* d_vhist);
bool resize_vplane_available(const uint8_t* d_src, int src_w, size_t src_frame_stride, const uint8_t* d_prev, int dst_w, int n);

// OpenCV's computeResizeAreaTab for one destination index in run-length form: `count` consecutive source cells from
// `first`, the first / last of them with their own weight (hash thumbnails and cv2.resize(INTER_AREA) share it).
struct AreaRun {
    int first;       // source index of the first contributing cell
    int count;       // number of contributing cells (consecutive)
    int has_head, has_tail;
    float a_head, a_mid, a_tail;
    int pad;
};
void area_table(int ssize, int dsize, AreaRun* tab);   // double arithmetic on the host, like OpenCV
// fills `tab[0 .. dst_w)` (x) and `tab[dst_w .. dst_w + dst_h)` (y); mode 0 = float run tables, 1 = integer box, 2 = 2x2 box
void area_tables(int src_h, int src_w, int dst_h, int dst_w, AreaRun* tab, int* mode, float* inv_area);

// Coefficient tables stay on the device for as long as the engine does, one per (kind, src shape, dst shape): built and
// uploaded on the first call with a shape, found again afterwards (no allocation or synchronisation on the call path).
enum TableKind { kTabNearest = 2, kTabArea = 3, kTabHashArea = 4, kTabHashBasis = 5, kTabLanczos4 = 6, kTabCubic = 7 };
struct DevTable {
    const void* ptr = nullptr;
    int mode = 0;
    float inv_area = 0.f;
};
bool table_find(psd_engine* e, int kind, int sh, int sw, int dh, int dw, DevTable* out);
int table_store(psd_engine* e, int kind, int sh, int sw, int dh, int dw, const void* host, size_t bytes, int mode, float inv_area,
                DevTable* out);

}  // namespace psd

#endif
