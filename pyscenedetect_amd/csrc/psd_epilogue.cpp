// psd_epilogue.cpp -- host epilogues: the O(1)-per-frame decisions that follow the pixel work.
//
// These replay, over whole clips of per-frame integer records, the same sequential logic the
// reference runs inside process_frame() (paths relative to the reference tree):
//   content   content_detector.py:177-180,192-211 and FlashFilter detector.py:106-224
//   adaptive  adaptive_detector.py:100-143
//   histogram histogram_detector.py:59-120,122-165 (cv2.normalize + cv2.compareHist restated)
//   threshold threshold_detector.py:100-191
// All floating point is IEEE double (float where OpenCV uses float) in the reference's order, so
// metrics are bit-identical to the Python host mirror in pyscenedetect_amd/detectors/.
#include <cfloat>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "psd_engine.h"

extern "C" void psd_set_error(const char* fmt, ...);

namespace {

// `(a - b) >= min_scene_len` on frame-number timecodes (common.py:627-638,700-755):
// an int length compares frames; a float/str length is seconds -> round(secs * fps) frames
// (Python round = half-to-even, as nearbyint in the default rounding mode).
struct MinLen {
    int64_t frames;
    explicit MinLen(int64_t min_len_frames, double min_len_secs, double fps)
    {
        if (min_len_secs >= 0.0) frames = (int64_t)std::nearbyint(min_len_secs * fps);
        else frames = min_len_frames;
    }
    bool met(int64_t a, int64_t b) const
    {
        int64_t d = a - b;
        if (d < 0) d = 0;
        return d >= frames;
    }
};

double fps_of(int64_t num, int64_t den) { return (double)num / (double)den; }

bool check_fps(int64_t num, int64_t den)
{
    if (num <= 0 || den <= 0) {
        psd_set_error("frame rate must be positive (got %lld/%lld)", (long long)num, (long long)den);
        return false;
    }
    return true;
}

}  // namespace

extern "C" {

int psd_epilogue_content_scores_sums(const psd_frame_sums* sums, size_t stride_bytes, int n, int height, int width,
                                     const double weights[4], int first_has_prev, double* content_val,
                                     double* delta_hue, double* delta_sat, double* delta_lum,
                                     double* delta_edges)
{
    if (!sums || stride_bytes < sizeof(psd_frame_sums) || n < 0 || height <= 0 || width <= 0 || !weights) {
        psd_set_error("psd_epilogue_content_scores_sums: invalid argument");
        return PSD_ERR_INVALID;
    }
    const double num_pixels = (double)((int64_t)height * (int64_t)width);
    const double wsum = std::fabs(weights[0]) + std::fabs(weights[1]) + std::fabs(weights[2]) + std::fabs(weights[3]);
    for (int t = 0; t < n; t++) {
        double dh = 0, ds = 0, dl = 0, de = 0, score = 0;
        if (t > 0 || first_has_prev) {
            const psd_frame_sums& r = *(const psd_frame_sums*)((const char*)sums + (size_t)t * stride_bytes);
            dh = (double)r.sad_h / num_pixels;
            ds = (double)r.sad_s / num_pixels;
            dl = (double)r.sad_v / num_pixels;
            de = (double)(255 * r.edge_xor) / num_pixels;
            // sum(component*weight ...) starts from int 0 and adds left to right.
            score = (((0.0 + dh * weights[0]) + ds * weights[1]) + dl * weights[2]) + de * weights[3];
            score /= wsum;
        }
        if (content_val) content_val[t] = score;
        if (delta_hue) delta_hue[t] = dh;
        if (delta_sat) delta_sat[t] = ds;
        if (delta_lum) delta_lum[t] = dl;
        if (delta_edges) delta_edges[t] = de;
    }
    return PSD_OK;
}

int psd_epilogue_content_scores(const psd_frame_scores* recs, int n, int height, int width,
                                const double weights[4], int first_has_prev, double* content_val,
                                double* delta_hue, double* delta_sat, double* delta_lum,
                                double* delta_edges)
{
    if (!recs) { psd_set_error("psd_epilogue_content_scores: invalid argument"); return PSD_ERR_INVALID; }
    return psd_epilogue_content_scores_sums((const psd_frame_sums*)recs, sizeof(psd_frame_scores), n, height, width, weights, first_has_prev,
                                            content_val, delta_hue, delta_sat, delta_lum, delta_edges);
}

int psd_epilogue_content_cuts(const double* content_val, int n, int64_t first_frame, int64_t fps_num,
                              int64_t fps_den, const psd_content_params* p, int64_t* cuts, int* n_cuts)
{
    if (!content_val || n < 0 || !p || !cuts || !n_cuts || !check_fps(fps_num, fps_den)) {
        if (!p || !cuts || !n_cuts || !content_val) psd_set_error("psd_epilogue_content_cuts: invalid argument");
        return PSD_ERR_INVALID;
    }
    const double fps = fps_of(fps_num, fps_den);
    int nc = 0;
    // FlashFilter state (detector.py:127-143)
    const bool secs_given = p->min_len_secs >= 0.0;
    const bool disabled = secs_given ? (p->min_len_secs <= 0.0) : (p->min_len_frames <= 0);
    // An int length is converted to seconds on the first frame (detector.py:176-177,194-195).
    const double filter_secs = secs_given ? p->min_len_secs : (double)p->min_len_frames / fps;
    const MinLen len(0, filter_secs, fps);
    bool have_last_above = false, merge_enabled = false, merge_triggered = false;
    int64_t last_above = 0, merge_start = 0;

    for (int t = 0; t < n; t++) {
        const int64_t tc = first_frame + t;
        const bool above = content_val[t] >= p->threshold;
        if (disabled) {
            if (above) cuts[nc++] = tc;
            continue;
        }
        if (!have_last_above) { last_above = tc; have_last_above = true; }
        if (p->filter_mode == 1) {  // SUPPRESS
            const bool min_length_met = len.met(tc, last_above);
            if (above && min_length_met) { last_above = tc; cuts[nc++] = tc; }
            continue;
        }
        // MERGE
        const bool min_length_met = len.met(tc, last_above);
        if (above) last_above = tc;
        if (merge_triggered) {
            if (min_length_met && !above && len.met(last_above, merge_start)) {
                merge_triggered = false;
                cuts[nc++] = last_above;
            }
            continue;
        }
        if (!above) continue;
        if (min_length_met) { merge_enabled = true; cuts[nc++] = tc; continue; }
        if (merge_enabled) { merge_triggered = true; merge_start = tc; }
    }
    *n_cuts = nc;
    return PSD_OK;
}

int psd_epilogue_adaptive_cuts(const double* content_val, int n, int64_t first_frame, int64_t fps_num,
                               int64_t fps_den, const psd_adaptive_params* p, double* adaptive_ratio,
                               int64_t* cuts, int* n_cuts)
{
    if (!content_val || n < 0 || !p || !cuts || !n_cuts || p->window_width < 1 || !check_fps(fps_num, fps_den)) {
        psd_set_error("psd_epilogue_adaptive_cuts: invalid argument");
        return PSD_ERR_INVALID;
    }
    const double fps = fps_of(fps_num, fps_den);
    const MinLen len(p->min_len_frames, p->min_len_secs, fps);
    const int w = p->window_width;
    const int required = 1 + 2 * w;
    int nc = 0;
    if (adaptive_ratio)
        for (int t = 0; t < n; t++) adaptive_ratio[t] = NAN;  // no metric for the first/last w frames
    int64_t last_cut = first_frame;
    for (int t = 0; t < n; t++) {
        if (t + 1 < required) continue;
        const int lo = t - required + 1;  // buffer = content_val[lo .. t]
        const int target = lo + w;
        const double target_score = content_val[target];
        double sum = 0.0;  // Python sum(): int 0 then left to right
        for (int i = lo; i <= t; i++)
            if (i != target) sum = sum + content_val[i];
        const double avg = sum / (2.0 * w);
        const bool average_is_zero = std::fabs(avg) < 0.00001;
        double ratio = 0.0;
        if (!average_is_zero) ratio = std::fmin(target_score / avg, 255.0);
        else if (target_score >= p->min_content_val) ratio = 255.0;
        if (adaptive_ratio) adaptive_ratio[target] = ratio;
        const bool threshold_met = ratio >= p->adaptive_threshold && target_score >= p->min_content_val;
        const bool min_length_met = len.met(first_frame + t, last_cut);
        if (threshold_met && min_length_met) {
            last_cut = first_frame + target;
            cuts[nc++] = first_frame + target;
        }
    }
    *n_cuts = nc;
    return PSD_OK;
}

// cv2.calcHist LUT + cv2.normalize(NORM_L2) of one record's luma histogram.
static void normalized_hist(const uint32_t* hist256, int bins, const int* lut, float* out)
{
    uint64_t cnt[256];
    if (bins > 0 && (bins & (bins - 1)) == 0) {
        // power-of-two bin counts (the reference's default 128): floor(j * bins / 256) = j >> shift, consecutive runs
        const int per = 256 / bins;
        for (int i = 0; i < bins; i++) {
            uint64_t c = 0;
            for (int k = 0; k < per; k++) c += hist256[i * per + k];
            cnt[i] = c;
        }
    } else {
        for (int i = 0; i < bins; i++) cnt[i] = 0;
        for (int j = 0; j < 256; j++) cnt[lut[j]] += hist256[j];
    }
    // The squares are integers below 2^48 and their sum stays below 2^53 (a frame has fewer than 2^26 pixels where it matters:
    // counts are exact in float32 up to 2^24), so the double sum is exact in ANY order: four independent partial sums.
    double ss4[4] = {0.0, 0.0, 0.0, 0.0};
    bool exact = true;
    for (int i = 0; i < bins; i++) {
        out[i] = (float)(int64_t)cnt[i];  // exact below 2^24, a single rounding above (as OpenCV's int->f32); signed: one cvtsi2ss
        exact &= cnt[i] < (1ull << 24);
        ss4[i & 3] += (double)out[i] * (double)out[i];
    }
    double ss = (ss4[0] + ss4[1]) + (ss4[2] + ss4[3]);
    if (!exact) {                // (frames beyond 16.7 M pixels in one bin: OpenCV's sequential order)
        ss = 0.0;
        for (int i = 0; i < bins; i++) ss += (double)out[i] * (double)out[i];
    }
    const double nrm = std::sqrt(ss);
    const float scale = (float)(nrm > DBL_EPSILON ? 1.0 / nrm : 0.0);
    for (int i = 0; i < bins; i++) out[i] = out[i] * scale;
}

// cv2.compareHist(HISTCMP_CORREL): double sums in the lane order of OpenCV's 2-lane f64 SIMD
// (x86-64 baseline): even and odd elements accumulate separately over the multiple-of-4 body.
// (Written over pairs -- lane l of every sum takes the elements of parity l, in order: the same additions in the same order as
//  OpenCV's loop.)
static double correl(const float* h1, const float* h2, int n)
{
    double s1[2] = {0, 0}, s2[2] = {0, 0}, s11[2] = {0, 0}, s12[2] = {0, 0}, s22[2] = {0, 0};
    const int body = n >= 4 ? (n & ~3) : 0;      // OpenCV's vector body: whole groups of four
    int j = 0;
    for (; j < body; j += 2)
        for (int l = 0; l < 2; l++) {
            const double a = h1[j + l], b = h2[j + l];
            s12[l] += a * b; s11[l] += a * a; s22[l] += b * b; s1[l] += a; s2[l] += b;
        }
    double S1 = s1[0] + s1[1], S2 = s2[0] + s2[1], S11 = s11[0] + s11[1], S12 = s12[0] + s12[1],
           S22 = s22[0] + s22[1];
    for (; j < n; j++) {
        const double a = h1[j], b = h2[j];
        S12 += a * b; S1 += a; S11 += a * a; S2 += b; S22 += b * b;
    }
    const double scale = 1. / n;
    const double num = S12 - S1 * S2 * scale;
    const double denom2 = (S11 - S1 * S1 * scale) * (S22 - S2 * S2 * scale);
    return std::fabs(denom2) > DBL_EPSILON ? num / std::sqrt(denom2) : 1.;
}

int psd_epilogue_hist_cuts(const psd_frame_scores* recs, int n, const psd_frame_scores* prev_rec,
                           int64_t first_frame, int64_t fps_num, int64_t fps_den,
                           const psd_hist_params* p, double* hist_diff, int64_t* cuts, int* n_cuts)
{
    if (!recs || n < 0 || !p || !cuts || !n_cuts || p->bins < 1 || p->bins > 256 || !check_fps(fps_num, fps_den)) {
        psd_set_error("psd_epilogue_hist_cuts: invalid argument");
        return PSD_ERR_INVALID;
    }
    const double fps = fps_of(fps_num, fps_den);
    const MinLen len(p->min_len_frames, p->min_len_secs, fps);
    const double thr = std::fmax(0.0, std::fmin(1.0, 1.0 - p->threshold));
    int lut[256];
    for (int j = 0; j < 256; j++) {
        int idx = (int)std::floor(j * (p->bins / 256.0));
        lut[j] = idx < 0 ? 0 : (idx > p->bins - 1 ? p->bins - 1 : idx);
    }
    // hist_diff[t] of every frame first -- the normalisation and the correlation of a frame pair depend on nothing but the two
    // records, so frame ranges go to a few threads (round 6: behind the default downscale the packed flows' kernels take 2 ms per
    // pass and this loop, 0.22 us per frame on one core, had become as long as they are) -- then the cuts in order.
    std::vector<double> own;
    double* diff = hist_diff;
    if (!diff) { own.resize((size_t)n); diff = own.data(); }
    const int bins = p->bins;
    auto range = [&](int t_begin, int t_end) {
        std::vector<float> a((size_t)bins), b((size_t)bins);
        float* last = a.data();
        float* cur = b.data();
        bool have_last = false;
        if (t_begin > 0) { normalized_hist(recs[t_begin - 1].hist, bins, lut, last); have_last = true; }
        else if (prev_rec) { normalized_hist(prev_rec->hist, bins, lut, last); have_last = true; }
        for (int t = t_begin; t < t_end; t++) {
            normalized_hist(recs[t].hist, bins, lut, cur);
            diff[t] = have_last ? correl(last, cur, bins) : NAN;
            float* tmp = last; last = cur; cur = tmp;
            have_last = true;
        }
    };
    static const int max_threads = [] {
        const char* v = getenv("PSD_EPILOGUE_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        const int t = v ? atoi(v) : (hw > 16 ? 8 : hw > 1 ? hw / 2 : 1);
        return t < 1 ? 1 : t > 64 ? 64 : t;
    }();
    const int nt = std::min(max_threads, n / 1024);     // (a thread costs ~0.1 ms to start and join: not for short clips)
    if (nt > 1) {
        std::vector<std::thread> pool;
        pool.reserve((size_t)nt - 1);
        for (int k = 1; k < nt; k++) pool.emplace_back(range, (int)((long)n * k / nt), (int)((long)n * (k + 1) / nt));
        range(0, (int)((long)n / nt));
        for (auto& th : pool) th.join();
    } else {
        range(0, n);
    }
    int nc = 0;
    int64_t last_cut = first_frame - (prev_rec ? 1 : 0);
    for (int t = 0; t < n; t++) {
        const int64_t tc = first_frame + t;
        const double d = diff[t];
        if ((t > 0 || prev_rec) && d <= thr && len.met(tc, last_cut)) { cuts[nc++] = tc; last_cut = tc; }
    }
    *n_cuts = nc;
    return PSD_OK;
}

int psd_epilogue_hist_cuts_from_diff(const double* hist_diff, int n, int64_t first_frame, int64_t fps_num, int64_t fps_den,
                                     const psd_hist_params* p, int64_t* cuts, int* n_cuts)
{
    if ((!hist_diff && n > 0) || n < 0 || !p || !cuts || !n_cuts || !check_fps(fps_num, fps_den)) {
        psd_set_error("psd_epilogue_hist_cuts_from_diff: invalid argument");
        return PSD_ERR_INVALID;
    }
    // (the decision loop of psd_epilogue_hist_cuts over values computed elsewhere -- psd_hist_diff_device: a frame without a
    //  predecessor carries NaN, which is not <= anything)
    const MinLen len(p->min_len_frames, p->min_len_secs, fps_of(fps_num, fps_den));
    const double thr = std::fmax(0.0, std::fmin(1.0, 1.0 - p->threshold));
    int nc = 0;
    int64_t last_cut = first_frame;
    for (int t = 0; t < n; t++) {
        const int64_t tc = first_frame + t;
        if (hist_diff[t] <= thr && len.met(tc, last_cut)) { cuts[nc++] = tc; last_cut = tc; }
    }
    *n_cuts = nc;
    return PSD_OK;
}

int psd_epilogue_hist_normalize(const uint32_t hist256[256], int bins, float* out)
{
    if (!hist256 || !out || bins < 1 || bins > 256) {
        psd_set_error("psd_epilogue_hist_normalize: invalid argument");
        return PSD_ERR_INVALID;
    }
    int lut[256];
    for (int j = 0; j < 256; j++) {
        int idx = (int)std::floor(j * (bins / 256.0));
        lut[j] = idx < 0 ? 0 : (idx > bins - 1 ? bins - 1 : idx);
    }
    normalized_hist(hist256, bins, lut, out);
    return PSD_OK;
}

int psd_epilogue_hist_correl(const float* h1, const float* h2, int bins, double* out)
{
    if (!h1 || !h2 || !out || bins < 1) {
        psd_set_error("psd_epilogue_hist_correl: invalid argument");
        return PSD_ERR_INVALID;
    }
    *out = correl(h1, h2, bins);
    return PSD_OK;
}

int psd_epilogue_threshold_cuts_sums(const psd_frame_sums* sums, size_t stride_bytes, int n, int height, int width,
                                     int64_t first_frame, int64_t fps_num, int64_t fps_den,
                                     const psd_threshold_params* p, double* average_rgb, int64_t* cuts,
                                     int* n_cuts)
{
    if (!sums || stride_bytes < sizeof(psd_frame_sums) || n < 0 || height <= 0 || width <= 0 || !p || !cuts || !n_cuts || !check_fps(fps_num, fps_den)) {
        psd_set_error("psd_epilogue_threshold_cuts_sums: invalid argument");
        return PSD_ERR_INVALID;
    }
    const double fps = fps_of(fps_num, fps_den);
    const MinLen len(p->min_len_frames, p->min_len_secs, fps);
    const double count = (double)((int64_t)height * (int64_t)width * 3);
    const double thr = (double)p->threshold;
    int nc = 0;
    bool processed = false;
    int fade_type = -1;  // 0 = in, 1 = out
    int64_t fade_frame = 0, last_scene_cut = first_frame;
    for (int t = 0; t < n; t++) {
        const int64_t tc = first_frame + t;
        const double avg = (double)((const psd_frame_sums*)((const char*)sums + (size_t)t * stride_bytes))->byte_sum / count;  // numpy.mean: exact sum, one divide
        if (average_rgb) average_rgb[t] = avg;
        const bool below = avg < thr;
        if (processed) {
            const bool out_now = (p->method == 0) ? below : !below;
            if (fade_type == 0 && out_now) {
                fade_type = 1; fade_frame = tc;
            } else if (fade_type == 1 && !out_now) {
                if (len.met(tc, last_scene_cut)) {
                    const int64_t duration = tc - fade_frame;
                    const int64_t split = fade_frame + (int64_t)std::nearbyint((double)duration * (1.0 + p->fade_bias) / 2.0);
                    cuts[nc++] = split;
                    last_scene_cut = tc;
                }
                fade_type = 0; fade_frame = tc;
            }
        } else {
            fade_frame = tc;
            fade_type = below ? 1 : 0;  // first frame always compares with `<` (threshold_detector.py:161-165)
        }
        processed = true;
    }
    // post_process(last position)
    if (n > 0 && fade_type == 1 && p->add_final_scene) {
        const int64_t tc = first_frame + n - 1;
        if (len.met(tc, last_scene_cut)) cuts[nc++] = fade_frame;
    }
    *n_cuts = nc;
    return PSD_OK;
}

int psd_epilogue_threshold_cuts(const psd_frame_scores* recs, int n, int height, int width,
                                int64_t first_frame, int64_t fps_num, int64_t fps_den,
                                const psd_threshold_params* p, double* average_rgb, int64_t* cuts,
                                int* n_cuts)
{
    if (!recs) { psd_set_error("psd_epilogue_threshold_cuts: invalid argument"); return PSD_ERR_INVALID; }
    return psd_epilogue_threshold_cuts_sums((const psd_frame_sums*)recs, sizeof(psd_frame_scores), n, height, width, first_frame, fps_num,
                                            fps_den, p, average_rgb, cuts, n_cuts);
}

extern "C++" {
namespace psd {
// orthonormal DCT-II basis, rows 0 .. keep-1 of the size-point transform: c[k][j] (shared with the device form, psd_hash_bits_device)
void hash_dct_basis(int size, int keep, double* c)
{
    for (int k = 0; k < keep; k++)
        for (int j = 0; j < size; j++)
            c[(size_t)k * size + j] = k == 0 ? std::sqrt(1.0 / size)
                                             : std::sqrt(2.0 / size) * std::cos(3.14159265358979323846 * (2 * j + 1) * k / (2.0 * size));
}
}  // namespace psd
}  // extern "C++"

int psd_epilogue_hash_bits(const uint8_t* thumbs, int n, int size, int hash_size, uint8_t* bits)
{
    if (n < 0 || size <= 0 || hash_size <= 0 || hash_size > size || (n > 0 && (!thumbs || !bits))) {
        psd_set_error("psd_epilogue_hash_bits: invalid argument");
        return PSD_ERR_INVALID;
    }
    const int keep = hash_size;
    std::vector<double> c((size_t)keep * size);
    psd::hash_dct_basis(size, keep, c.data());
    // frames are independent: split long clips over a few host threads (results do not depend on the split)
    auto work = [&](int t_begin, int t_end) {
        std::vector<double> tmp((size_t)keep * size);
        std::vector<float> x((size_t)size * size), low((size_t)keep * keep), sorted((size_t)keep * keep);
        for (int t = t_begin; t < t_end; t++) {
            const uint8_t* th = thumbs + (size_t)t * size * size;
            int mx = 0;
            for (int i = 0; i < size * size; i++) mx = th[i] > mx ? th[i] : mx;
            if (mx == 0) mx = 1;  // hash_detector.py:132-135
            const float fmx = (float)mx;
            for (int i = 0; i < size * size; i++) x[i] = (float)th[i] / fmx;
            for (int u = 0; u < keep; u++)
                for (int xx = 0; xx < size; xx++) {
                    double acc = 0.0;
                    for (int y = 0; y < size; y++) acc += c[(size_t)u * size + y] * (double)x[(size_t)y * size + xx];
                    tmp[(size_t)u * size + xx] = acc;
                }
            for (int u = 0; u < keep; u++)
                for (int v = 0; v < keep; v++) {
                    double acc = 0.0;
                    for (int xx = 0; xx < size; xx++) acc += tmp[(size_t)u * size + xx] * c[(size_t)v * size + xx];
                    low[(size_t)u * keep + v] = (float)acc;
                }
            // numpy.median of float32: middle element, or the float32 mean of the two middle elements
            sorted = low;
            std::sort(sorted.begin(), sorted.end());
            const size_t m = sorted.size();
            float med;
            if (m % 2) med = sorted[m / 2];
            else {
                const float two = sorted[m / 2 - 1] + sorted[m / 2];
                med = two / 2.0f;
            }
            uint8_t* b = bits + (size_t)t * keep * keep;
            for (size_t i = 0; i < m; i++) b[i] = low[i] > med ? 1 : 0;
        }
    };
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads > 16) nthreads = 16;
    if (n < 256 || nthreads < 2) {
        work(0, n);
    } else {
        std::vector<std::thread> pool;
        const int per = (n + (int)nthreads - 1) / (int)nthreads;
        for (int t0 = 0; t0 < n; t0 += per) pool.emplace_back(work, t0, std::min(n, t0 + per));
        for (auto& th : pool) th.join();
    }
    return PSD_OK;
}

int psd_epilogue_hash_cuts(const uint8_t* bits, int n, const uint8_t* prev_bits, int64_t first_frame,
                           int64_t fps_num, int64_t fps_den, const psd_hash_params* p,
                           double* hash_dist, int64_t* cuts, int* n_cuts)
{
    if (n < 0 || (n > 0 && !bits) || !p || !cuts || !n_cuts || p->hash_size <= 0 || !check_fps(fps_num, fps_den)) {
        psd_set_error("psd_epilogue_hash_cuts: invalid argument");
        return PSD_ERR_INVALID;
    }
    const double fps = fps_of(fps_num, fps_den);
    const MinLen len(p->min_len_frames, p->min_len_secs, fps);
    const size_t nb = (size_t)p->hash_size * p->hash_size;
    const double size_sq = (double)nb;
    int nc = 0;
    int64_t last_cut = first_frame - (prev_bits ? 1 : 0);
    const uint8_t* last = prev_bits;
    for (int t = 0; t < n; t++) {
        const int64_t tc = first_frame + t;
        const uint8_t* cur = bits + (size_t)t * nb;
        double d = NAN;
        if (last) {
            int diff = 0;
            for (size_t i = 0; i < nb; i++) diff += cur[i] != last[i];
            d = (double)diff / size_sq;
            if (d >= p->threshold && len.met(tc, last_cut)) { cuts[nc++] = tc; last_cut = tc; }
        }
        if (hash_dist) hash_dist[t] = d;
        last = cur;
    }
    *n_cuts = nc;
    return PSD_OK;
}

}  // extern "C"
