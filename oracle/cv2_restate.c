/*
 * oracle/cv2_restate.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the OpenCV primitives that PySceneDetect's
 * process_frame() hot path calls.  OpenCV is a third-party, un-vendored,
 * un-pinned dependency of the reference (pyproject.toml:43-57; the only pin
 * in-tree is opencv-python-headless==5.0.0.93 in
 * packaging/windows/requirements.txt:8) and is NOT installed in this image,
 * so these functions restate OpenCV's *published* 8-bit algorithms
 * (modules/imgproc/src/color_hsv.simd.hpp RGB2HSV_b, color_yuv.simd.hpp
 * RGB2YCrCb_i, histogram.cpp calcHist_8u/compareHist, core convert_scale /
 * norm, canny.cpp, morph, resize.cpp) and anchor on the reference call sites:
 *
 *   cv2.cvtColor(BGR2HSV) + split   scenedetect/detectors/content_detector.py:155
 *   cv2.Canny / cv2.dilate          scenedetect/detectors/content_detector.py:238-239
 *   cv2.cvtColor(BGR2YUV) + split   scenedetect/detectors/histogram_detector.py:156
 *   cv2.calcHist                    scenedetect/detectors/histogram_detector.py:159
 *   cv2.normalize                   scenedetect/detectors/histogram_detector.py:163
 *   cv2.compareHist(CORREL)         scenedetect/detectors/histogram_detector.py:98
 *   cv2.resize(INTER_LINEAR | INTER_NEAREST | INTER_AREA | INTER_CUBIC | INTER_LANCZOS4)   scenedetect/scene_manager.py:670-678 (Interpolation, common.py:148-160)
 *   cv2.cvtColor(BGR2GRAY) / cv2.resize(INTER_AREA) / cv2.dct   scenedetect/detectors/hash_detector.py:125,129,139
 *
 * PARITY UNPINNED at the cv2 boundary: no real cv2 build and none of the
 * reference's video fixtures exist in this environment, so
 * this restatement cannot be checked against OpenCV output here.  It is
 * pinned only by construction (integer formulas) and by self-consistency
 * tests (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define HSV_SHIFT 12

static int g_sdiv[256];
static int g_hdiv180[256];
static int g_tables_ready = 0;

/* cv::saturate_cast<int>(double) == cvRound == lrint (round-half-even). */
static void init_tables(void)
{
    if (g_tables_ready) return;
    g_sdiv[0] = g_hdiv180[0] = 0;
    for (int i = 1; i < 256; i++) {
        g_sdiv[i] = (int)lrint((255 << HSV_SHIFT) / (1. * i));
        g_hdiv180[i] = (int)lrint((180 << HSV_SHIFT) / (6. * i));
    }
    g_tables_ready = 1;
}

/* Export the fixed-point tables (used by tests to check the device LUTs). */
void orc_hsv_tables(int32_t* sdiv, int32_t* hdiv180)
{
    init_tables();
    for (int i = 0; i < 256; i++) { sdiv[i] = g_sdiv[i]; hdiv180[i] = g_hdiv180[i]; }
}

static inline void bgr2hsv_px(int b, int g, int r, uint8_t* h_out, uint8_t* s_out, uint8_t* v_out)
{
    int v = b, vmin = b;
    if (g > v) v = g;
    if (r > v) v = r;
    if (g < vmin) vmin = g;
    if (r < vmin) vmin = r;
    int diff = v - vmin;
    int vr = (v == r) ? -1 : 0;
    int vg = (v == g) ? -1 : 0;
    int s = (diff * g_sdiv[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    /* arithmetic shift of a possibly negative value, as in OpenCV */
    h = (h * g_hdiv180[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT;
    if (h < 0) h += 180;
    *h_out = (uint8_t)(h < 0 ? 0 : (h > 255 ? 255 : h));
    *s_out = (uint8_t)s;
    *v_out = (uint8_t)v;
}

/* cv2.cvtColor(src, COLOR_BGR2HSV): interleaved HSV, H in [0,180). */
void orc_bgr2hsv(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int h, int w)
{
    init_tables();
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * src_step;
        uint8_t* d = dst + (size_t)y * dst_step;
        for (int x = 0; x < w; x++, s += 3, d += 3)
            bgr2hsv_px(s[0], s[1], s[2], d + 0, d + 1, d + 2);
    }
}

/* Planar variant (what cv2.split gives content_detector.py:155). */
void orc_bgr2hsv_planes(const uint8_t* src, size_t src_step, uint8_t* hp, uint8_t* sp, uint8_t* vp,
                        int h, int w)
{
    init_tables();
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * src_step;
        size_t o = (size_t)y * w;
        for (int x = 0; x < w; x++, s += 3, o++)
            bgr2hsv_px(s[0], s[1], s[2], hp + o, sp + o, vp + o);
    }
}

/* BT.601 fixed-point constants of OpenCV's RGB2YCrCb_i (yuv_shift = 14). */
#define YUV_SHIFT 14
#define R2Y 4899
#define G2Y 9617
#define B2Y 1868
#define R2VI 14369
#define B2UI 8061
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* cv2.cvtColor(src, COLOR_BGR2YUV): interleaved Y,U,V. */
void orc_bgr2yuv(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int h, int w)
{
    const int delta = 128 * (1 << YUV_SHIFT);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * src_step;
        uint8_t* d = dst + (size_t)y * dst_step;
        for (int x = 0; x < w; x++, s += 3, d += 3) {
            int b = s[0], g = s[1], r = s[2];
            int Y = DESCALE(b * B2Y + g * G2Y + r * R2Y, YUV_SHIFT);
            int V = DESCALE((r - Y) * R2VI + delta, YUV_SHIFT); /* "Cr" slot */
            int U = DESCALE((b - Y) * B2UI + delta, YUV_SHIFT); /* "Cb" slot */
            d[0] = sat_u8(Y);
            d[1] = sat_u8(U);
            d[2] = sat_u8(V);
        }
    }
}

void orc_bgr2y(const uint8_t* src, size_t src_step, uint8_t* yp, int h, int w)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * src_step;
        uint8_t* d = yp + (size_t)y * w;
        for (int x = 0; x < w; x++, s += 3)
            d[x] = sat_u8(DESCALE(s[0] * B2Y + s[1] * G2Y + s[2] * R2Y, YUV_SHIFT));
    }
}

/* cv2.calcHist([plane],[0],None,[bins],[lo,hi]) for an 8-bit plane, uniform
 * ranges.  OpenCV builds a 256-entry LUT: idx = cvFloor(j*a + b) with
 * a = bins/(hi-lo), b = -a*lo, clamped to [0,bins-1] inside [lo,hi), values
 * outside the range are skipped.  Output: float32 counts. */
void orc_calc_hist_u8(const uint8_t* plane, size_t step, int h, int w, int bins, double lo, double hi,
                      float* hist)
{
    int lut[256];
    double a = bins / (hi - lo), b = -a * lo;
    for (int j = 0; j < 256; j++) {
        int idx = (int)floor(j * a + b);
        if (j >= lo && j < hi) {
            if (idx < 0) idx = 0;
            if (idx > bins - 1) idx = bins - 1;
        } else
            idx = -1;
        lut[j] = idx;
    }
    uint64_t* cnt = (uint64_t*)calloc((size_t)bins, sizeof(uint64_t));
    for (int y = 0; y < h; y++) {
        const uint8_t* p = plane + (size_t)y * step;
        for (int x = 0; x < w; x++) {
            int idx = lut[p[x]];
            if (idx >= 0) cnt[idx]++;
        }
    }
    /* OpenCV accumulates in int and converts; counts < 2^24 are exact in f32,
     * larger counts round once (same as (float)int). */
    for (int i = 0; i < bins; i++) hist[i] += (float)cnt[i];
    free(cnt);
}

/* cv2.normalize(hist, hist) defaults: NORM_L2, alpha=1, beta=0.
 * norm accumulates squares in double; scale = alpha/norm (0 if norm<=eps);
 * convertTo 32f->32f multiplies in float: dst = src*(float)scale. */
void orc_normalize_l2_f32(float* v, int n)
{
    double ss = 0.0;
    for (int i = 0; i < n; i++) ss += (double)v[i] * (double)v[i];
    double nrm = sqrt(ss);
    double scale = nrm > DBL_EPSILON ? 1.0 / nrm : 0.0;
    float fs = (float)scale;
    for (int i = 0; i < n; i++) v[i] = v[i] * fs;
}

/* cv2.compareHist(h1,h2,HISTCMP_CORREL) on float32 histograms.  All sums are
 * double.  x86-64 OpenCV builds accumulate with 2-lane f64 vectors
 * (CV_SIMD_64F, baseline SSE2): lane k sums elements j with j%2==k, lanes are
 * added at the end, the tail (n%4) is added sequentially.  Products of two
 * f32 values are exact in f64, so only this summation order matters. */
double orc_compare_hist_correl(const float* h1, const float* h2, int n)
{
    double s1[2] = {0, 0}, s2[2] = {0, 0}, s11[2] = {0, 0}, s12[2] = {0, 0}, s22[2] = {0, 0};
    int j = 0;
    for (; j <= n - 4; j += 4) {
        for (int k = 0; k < 4; k++) {
            double a = h1[j + k], b = h2[j + k];
            int l = k & 1;
            s12[l] += a * b;
            s11[l] += a * a;
            s22[l] += b * b;
            s1[l] += a;
            s2[l] += b;
        }
    }
    double S1 = s1[0] + s1[1], S2 = s2[0] + s2[1], S11 = s11[0] + s11[1], S12 = s12[0] + s12[1],
           S22 = s22[0] + s22[1];
    for (; j < n; j++) {
        double a = h1[j], b = h2[j];
        S12 += a * b;
        S1 += a;
        S11 += a * a;
        S2 += b;
        S22 += b * b;
    }
    double scale = 1. / n;
    double num = S12 - S1 * S2 * scale;
    double denom2 = (S11 - S1 * S1 * scale) * (S22 - S2 * S2 * scale);
    return fabs(denom2) > DBL_EPSILON ? num / sqrt(denom2) : 1.;
}

/* ---------------------------------------------------------------------------
 * cv2.Canny(img8u, low, high)  (apertureSize=3, L2gradient=False)
 * Sobel 3x3 with BORDER_REPLICATE -> s16 dx,dy; mag = |dx|+|dy|; magnitudes
 * outside the image are 0; non-maximum suppression with TG22 fixed point;
 * hysteresis = 8-connected reachability from strong pixels.
 * ------------------------------------------------------------------------- */
#define CANNY_SHIFT 15
#define TG22 13573 /* (int)(0.4142135623730950488016887242097*(1<<15) + 0.5) */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_sobel3(const uint8_t* src, size_t step, int h, int w, int16_t* dx, int16_t* dy)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = src + (size_t)clampi(y - 1, 0, h - 1) * step;
        const uint8_t* r1 = src + (size_t)y * step;
        const uint8_t* r2 = src + (size_t)clampi(y + 1, 0, h - 1) * step;
        for (int x = 0; x < w; x++) {
            int xl = clampi(x - 1, 0, w - 1), xr = clampi(x + 1, 0, w - 1);
            int gx = (r0[xr] - r0[xl]) + 2 * (r1[xr] - r1[xl]) + (r2[xr] - r2[xl]);
            int gy = (r2[xl] - r0[xl]) + 2 * (r2[x] - r0[x]) + (r2[xr] - r0[xr]);
            dx[(size_t)y * w + x] = (int16_t)gx;
            dy[(size_t)y * w + x] = (int16_t)gy;
        }
    }
}

void orc_canny(const uint8_t* src, size_t step, int h, int w, double low_thresh, double high_thresh,
               uint8_t* dst)
{
    if (low_thresh > high_thresh) { double t = low_thresh; low_thresh = high_thresh; high_thresh = t; }
    if (low_thresh > 32767.0) low_thresh = 32767.0;
    if (high_thresh > 32767.0) high_thresh = 32767.0;
    int low = (int)floor(low_thresh), high = (int)floor(high_thresh);

    size_t n = (size_t)h * w;
    int16_t* dx = (int16_t*)malloc(n * sizeof(int16_t));
    int16_t* dy = (int16_t*)malloc(n * sizeof(int16_t));
    orc_sobel3(src, step, h, w, dx, dy);

    /* magnitude with a zero border of 1 px */
    int mw = w + 2;
    int* mag = (int*)calloc((size_t)(h + 2) * mw, sizeof(int));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            mag[(size_t)(y + 1) * mw + x + 1] = abs(dx[(size_t)y * w + x]) + abs(dy[(size_t)y * w + x]);

    /* map: 1 = cannot be an edge, 0 = weak candidate, 2 = edge */
    uint8_t* map = (uint8_t*)malloc((size_t)(h + 2) * mw);
    memset(map, 1, (size_t)(h + 2) * mw);
    size_t* stack = (size_t*)malloc(n * sizeof(size_t) + sizeof(size_t));
    size_t sp = 0;

    for (int y = 0; y < h; y++) {
        const int* mp = mag + (size_t)y * mw + 1;       /* previous row */
        const int* ma = mag + (size_t)(y + 1) * mw + 1; /* current row  */
        const int* mn = mag + (size_t)(y + 2) * mw + 1; /* next row     */
        uint8_t* pm = map + (size_t)(y + 1) * mw + 1;
        for (int x = 0; x < w; x++) {
            int m = ma[x];
            if (m > low) {
                int xs = dx[(size_t)y * w + x], ys = dy[(size_t)y * w + x];
                int ax = abs(xs), ay = abs(ys) << CANNY_SHIFT;
                int tg22x = ax * TG22;
                int is_max = 0;
                if (ay < tg22x) {
                    is_max = (m > ma[x - 1] && m >= ma[x + 1]);
                } else {
                    int tg67x = tg22x + (ax << (CANNY_SHIFT + 1));
                    if (ay > tg67x) {
                        is_max = (m > mp[x] && m >= mn[x]);
                    } else {
                        int s = (xs ^ ys) < 0 ? -1 : 1;
                        is_max = (m > mp[x - s] && m > mn[x + s]);
                    }
                }
                if (is_max) {
                    if (m > high) {
                        pm[x] = 2;
                        stack[sp++] = (size_t)(y + 1) * mw + x + 1;
                    } else
                        pm[x] = 0;
                    continue;
                }
            }
            pm[x] = 1;
        }
    }
    /* hysteresis */
    while (sp) {
        size_t p = stack[--sp];
        const long off[8] = {-mw - 1, -mw, -mw + 1, -1, 1, mw - 1, mw, mw + 1};
        for (int k = 0; k < 8; k++) {
            size_t q = (size_t)((long)p + off[k]);
            if (map[q] == 0) { map[q] = 2; stack[sp++] = q; }
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[(size_t)y * w + x] = (uint8_t)(map[(size_t)(y + 1) * mw + x + 1] == 2 ? 255 : 0);
    free(stack); free(map); free(mag); free(dx); free(dy);
}

/* cv2.dilate(src, ones(kh,kw)): rectangular window max, anchor at the centre
 * (k/2), iterations=1, default border (BORDER_CONSTANT with the morphology
 * default value = no contribution from outside the image). */
void orc_dilate_rect(const uint8_t* src, size_t step, int h, int w, int kh, int kw, uint8_t* dst)
{
    int ay = kh / 2, ax = kw / 2;
    uint8_t* tmp = (uint8_t*)malloc((size_t)h * w);
    for (int y = 0; y < h; y++) { /* horizontal pass */
        const uint8_t* s = src + (size_t)y * step;
        for (int x = 0; x < w; x++) {
            int x0 = x - ax, x1 = x - ax + kw - 1;
            if (x0 < 0) x0 = 0;
            if (x1 > w - 1) x1 = w - 1;
            uint8_t m = 0;
            for (int i = x0; i <= x1; i++) if (s[i] > m) m = s[i];
            tmp[(size_t)y * w + x] = m;
        }
    }
    for (int y = 0; y < h; y++) { /* vertical pass */
        int y0 = y - ay, y1 = y - ay + kh - 1;
        if (y0 < 0) y0 = 0;
        if (y1 > h - 1) y1 = h - 1;
        for (int x = 0; x < w; x++) {
            uint8_t m = 0;
            for (int i = y0; i <= y1; i++) {
                uint8_t v = tmp[(size_t)i * w + x];
                if (v > m) m = v;
            }
            dst[(size_t)y * w + x] = m;
        }
    }
    free(tmp);
}

/* ---------------------------------------------------------------------------
 * cv2.resize(src, (dw,dh), interpolation=INTER_LINEAR) for 8-bit, cn channels.
 * Classic fixed-point path of resize.cpp: 11-bit coefficients
 * (INTER_RESIZE_COEF_SCALE = 2048), horizontal pass into int rows, vertical
 * pass  dst = (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
 * An exact 2x2 decimation is routed to INTER_AREA by OpenCV
 * (interpolation == INTER_LINEAR && is_area_fast && iscale == 2), which for
 * 8u is the rounded box mean (sum + 2) >> 2.
 * ------------------------------------------------------------------------- */
static short sat_s16_round(float v)
{
    long r = lrintf(v);
    if (r > 32767) r = 32767;
    if (r < -32768) r = -32768;
    return (short)r;
}

/* area_mode != 0: cv2.resize(INTER_AREA) when the image is NOT shrunk along both axes.  resize.cpp: "true area
 * interpolation is only implemented for the case (scale_x >= 1 && scale_y >= 1); in other cases it is emulated using some
 * variant of bilinear interpolation" -- the bilinear passes below with the coefficients of area_mode:
 *     sx = cvFloor(dx * scale_x);  fx = (float)((dx + 1) - (sx + 1) * inv_scale_x);  fx = fx <= 0 ? 0.f : fx - cvFloor(fx) */
static void resize_bilinear_u8(const uint8_t* src, size_t sstep, int sh, int sw, int cn, uint8_t* dst,
                               size_t dstep, int dh, int dw, int area_mode)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (!area_mode && is_area_fast && iscale_x == 2 && iscale_y == 2) {
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    const uint8_t* p = src + (size_t)(2 * y) * sstep + (size_t)(2 * x) * cn + c;
                    dst[(size_t)y * dstep + (size_t)x * cn + c] =
                        (uint8_t)((p[0] + p[cn] + p[sstep] + p[sstep + cn] + 2) >> 2);
                }
        return;
    }
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * dw * 2);
    int* yofs = (int*)malloc(sizeof(int) * dh);
    short* ibeta = (short*)malloc(sizeof(short) * dh * 2);
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
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_s16_round((1.f - fx) * 2048);
        ialpha[dx * 2 + 1] = sat_s16_round(fx * 2048);
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
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_s16_round((1.f - fy) * 2048);
        ibeta[dy * 2 + 1] = sat_s16_round(fy * 2048);
    }
    int* row0 = (int*)malloc(sizeof(int) * dw * cn);
    int* row1 = (int*)malloc(sizeof(int) * dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clampi(yofs[dy], 0, sh - 1), sy1 = clampi(yofs[dy] + 1, 0, sh - 1);
        const uint8_t* S0 = src + (size_t)sy0 * sstep;
        const uint8_t* S1 = src + (size_t)sy1 * sstep;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
            for (int c = 0; c < cn; c++) {
                row0[dx * cn + c] = S0[sx * cn + c] * a0 + S0[sx1 * cn + c] * a1;
                row1[dx * cn + c] = S1[sx * cn + c] * a0 + S1[sx1 * cn + c] * a1;
            }
        }
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = dst + (size_t)dy * dstep;
        for (int x = 0; x < dw * cn; x++)
            D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(row0); free(row1); free(xofs); free(ialpha); free(yofs); free(ibeta);
}

void orc_resize_linear_u8(const uint8_t* src, size_t sstep, int sh, int sw, int cn, uint8_t* dst,
                          size_t dstep, int dh, int dw)
{
    resize_bilinear_u8(src, sstep, sh, sw, cn, dst, dstep, dh, dw, 0);
}

/* cv2.resize(INTER_AREA) with dw > sw or dh > sh (see resize_bilinear_u8) */
void orc_resize_area_upscale_u8(const uint8_t* src, size_t sstep, int sh, int sw, int cn, uint8_t* dst,
                                size_t dstep, int dh, int dw)
{
    resize_bilinear_u8(src, sstep, sh, sw, cn, dst, dstep, dh, dw, 1);
}

/* ---------------------------------------------------------------------------
 * HashDetector primitives (scenedetect/detectors/hash_detector.py:117-151):
 *   cv2.cvtColor(BGR2GRAY)  -> orc_bgr2gray      (color_rgb.simd.hpp RGB2Gray<uchar>, 15-bit coefficients)
 *   cv2.resize(INTER_AREA)  -> orc_resize_area_u8 (resize.cpp: ResizeAreaFast for integer scales,
 *                                                   computeResizeAreaTab + ResizeArea_ float path otherwise)
 *   cv2.dct                 -> orc_dct2d_f32     (SEE NOTE)
 * NOTE on cv2.dct: OpenCV computes the 2-D DCT-II in float32 through its DFT
 * machinery (and through IPP in the official x86 wheels), so the last bits
 * of its output depend on the build.  That operation order is not
 * restatable; this oracle evaluates the orthonormal DCT-II definition in
 * float64 in a fixed loop order and rounds once to float32.  Hash bits can
 * therefore differ from a given cv2 build only where a coefficient lies
 * within float32 rounding noise of the median.
 * ------------------------------------------------------------------------- */
void orc_bgr2gray(const uint8_t* src, size_t src_step, uint8_t* dst, size_t dst_step, int h, int w)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * src_step;
        uint8_t* d = dst + (size_t)y * dst_step;
        for (int x = 0; x < w; x++)
            d[x] = (uint8_t)((s[3 * x] * 3735 + s[3 * x + 1] * 19235 + s[3 * x + 2] * 9798 + (1 << 14)) >> 15);
    }
}

/* One axis of the INTER_AREA decimation table in run-length form.  For every destination index the
 * source cells are consecutive: an optional fractional head cell, whole cells, an optional fractional
 * tail cell -- the order computeResizeAreaTab emits them in. */
typedef struct {
    int32_t first;      /* source index of the first contributing cell */
    int32_t count;      /* number of contributing cells */
    int32_t has_head, has_tail;
    float a_head, a_mid, a_tail;
} orc_area_run;

void orc_area_table(int ssize, int dsize, orc_area_run* tab)
{
    double scale = 1. / ((double)dsize / ssize);
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        orc_area_run r;
        memset(&r, 0, sizeof r);
        r.first = sx1;
        if (sx1 - fsx1 > 1e-3) {
            r.has_head = 1;
            r.first = sx1 - 1;
            r.a_head = (float)((sx1 - fsx1) / cell);
            r.count++;
        }
        r.a_mid = (float)(1.0 / cell);
        r.count += sx2 - sx1;
        if (fsx2 - sx2 > 1e-3) {
            double t = fsx2 - sx2 < 1. ? fsx2 - sx2 : 1.;
            if (t > cell) t = cell;
            r.has_tail = 1;
            r.a_tail = (float)(t / cell);
            r.count++;
        }
        tab[dx] = r;
    }
}

static inline float area_weight(const orc_area_run* r, int k)
{
    if (k == 0 && r->has_head) return r->a_head;
    if (k == r->count - 1 && r->has_tail) return r->a_tail;
    return r->a_mid;
}

/* 8-bit, cn interleaved channels, shrinking in both directions (HashDetector uses cn = 1; SceneManager's
 * Interpolation.AREA downscale uses cn = 3).  Returns 0, or -1 if the request is not a pure decimation. */
int orc_resize_area_u8_cn(const uint8_t* src, size_t sstep, int sh, int sw, int cn, uint8_t* dst, size_t dstep, int dh, int dw)
{
    if (dw > sw || dh > sh || dw <= 0 || dh <= 0 || cn <= 0) return -1;
    double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (is_area_fast) {
        /* integer box: 2x2 uses the rounding shift of the SIMD path, other sizes saturate_cast(sum * (1.f/area)) */
        int area = iscale_x * iscale_y;
        float scale = 1.f / area;
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    int sum = 0;
                    for (int j = 0; j < iscale_y; j++)
                        for (int i = 0; i < iscale_x; i++)
                            sum += src[(size_t)(y * iscale_y + j) * sstep + (size_t)(x * iscale_x + i) * cn + c];
                    uint8_t* d = dst + (size_t)y * dstep + (size_t)x * cn + c;
                    if (iscale_x == 2 && iscale_y == 2) *d = (uint8_t)((sum + 2) >> 2);
                    else {
                        long v = lrintf((float)sum * scale);
                        *d = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
                    }
                }
        return 0;
    }
    orc_area_run* xt = (orc_area_run*)malloc(sizeof(orc_area_run) * dw);
    orc_area_run* yt = (orc_area_run*)malloc(sizeof(orc_area_run) * dh);
    orc_area_table(sw, dw, xt);
    orc_area_table(sh, dh, yt);
    float* buf = (float*)malloc(sizeof(float) * dw * cn);
    float* sum = (float*)malloc(sizeof(float) * dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        for (int j = 0; j < yt[dy].count; j++) {
            const uint8_t* S = src + (size_t)(yt[dy].first + j) * sstep;
            float beta = area_weight(&yt[dy], j);
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    float acc = 0.f;
                    for (int k = 0; k < xt[dx].count; k++) {
                        float term = (float)S[(size_t)(xt[dx].first + k) * cn + c] * area_weight(&xt[dx], k); /* no fused multiply-add */
                        acc += term;
                    }
                    buf[dx * cn + c] = acc;
                }
            for (int i = 0; i < dw * cn; i++) {
                float term = beta * buf[i];
                sum[i] = j == 0 ? term : sum[i] + term;
            }
        }
        for (int i = 0; i < dw * cn; i++) {
            long v = lrintf(sum[i]);
            dst[(size_t)dy * dstep + i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(xt); free(yt); free(buf); free(sum);
    return 0;
}

int orc_resize_area_u8(const uint8_t* src, size_t sstep, int sh, int sw, uint8_t* dst, size_t dstep, int dh, int dw)
{
    return orc_resize_area_u8_cn(src, sstep, sh, sw, 1, dst, dstep, dh, dw);
}

/* cv2.resize(INTER_NEAREST), 8-bit, cn channels (resize.cpp resizeNN): source index = min(floor(d * scale), size-1)
 * with scale = 1 / (dst/src) in double. */
void orc_resize_nearest_u8(const uint8_t* src, size_t sstep, int sh, int sw, int cn, uint8_t* dst, size_t dstep, int dh, int dw)
{
    double ifx = 1. / ((double)dw / sw), ify = 1. / ((double)dh / sh);
    for (int y = 0; y < dh; y++) {
        int sy = (int)floor(y * ify);
        if (sy > sh - 1) sy = sh - 1;
        for (int x = 0; x < dw; x++) {
            int sx = (int)floor(x * ifx);
            if (sx > sw - 1) sx = sw - 1;
            for (int c = 0; c < cn; c++) dst[(size_t)y * dstep + (size_t)x * cn + c] = src[(size_t)sy * sstep + (size_t)sx * cn + c];
        }
    }
}

/* ---------------------------------------------------------------------------
 * cv2.resize(INTER_LANCZOS4), 8-bit, cn channels (scene_manager.py:670-678 with Interpolation.LANCZOS4, common.py:148-160).
 * resize.cpp, the generic path (lanczos4_tab[CV_8U] = resizeGeneric_<HResizeLanczos4<uchar, int, short>,
 * VResizeLanczos4<uchar, int, short, FixedPtCast<int, uchar, INTER_RESIZE_COEF_BITS * 2>, VResizeNoVec>>): everything behind the
 * coefficients is INTEGER arithmetic -- no SIMD vertical pass (VResizeNoVec; unlike INTER_CUBIC, whose 8-bit vertical pass is
 * float32 SIMD and which x86 wheels hand to IPP) and no IPP branch (ipp_resize takes NEAREST / LINEAR / CUBIC / AREA only):
 *   fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = cvFloor(fx);  fx -= sx;          (the taps are source columns sx-3 .. sx+4)
 *   interpolateLanczos4(fx, cbuf);  ialpha[k] = saturate_cast<short>(cbuf[k] * 2048)  (cvRound: to nearest even)
 *   rows likewise from fy;  a tap outside the image is the nearest pixel inside (HResizeLanczos4's while loops, clip() on rows);
 *   D = sum_j S[sx - 3 + j] * alpha[j]  (int);   dst = saturate_cast<uchar>((sum_k D_k * beta[k] + (1 << 21)) >> 22).
 * interpolateLanczos4 (imgproc/src/resize.cpp / imgwarp.cpp, OpenCV 4.x): the eight weights are sin(y) sin(y / 4) / y^2-shaped
 * values at the taps' distances, y = -(x + 3 - i) pi / 4, evaluated through the angle-sum table cs[][] from ONE sin / cos pair in
 * double, cast to float, a tap at distance < 1e-6 takes the weight 1e30 (so it becomes 1 after the normalisation), normalised by
 * their float sum.  (OpenCV 3.x returned (0,0,0,1,0,0,0,0) for x < FLT_EPSILON instead: the same shorts.)
 * ------------------------------------------------------------------------- */
static void lanczos4_coeffs(float x, float* coeffs)
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    const double pi = 3.1415926535897932384626433832795;
    float sum = 0;
    double y0 = -(x + 3) * pi * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        float y0_ = (x + 3 - i);
        if (fabsf(y0_) >= 1e-6f) {
            double y = -y0_ * pi * 0.25;
            coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        } else {
            coeffs[i] = 1e30f;
        }
        sum += coeffs[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

/* the taps of one axis: first[d] = s - 3 (not clamped), coef[8 d .. 8 d + 7] */
void orc_lanczos4_taps(int ssize, int dsize, int* first, short* coef)
{
    double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        float cbuf[8];
        lanczos4_coeffs(f, cbuf);
        first[d] = s - 3;
        for (int k = 0; k < 8; k++) coef[8 * d + k] = sat_s16_round(cbuf[k] * 2048);
    }
}

void orc_resize_lanczos4_u8(const uint8_t* src, size_t sstep, int sh, int sw, int cn, uint8_t* dst, size_t dstep, int dh, int dw)
{
    int* xf = (int*)malloc(sizeof(int) * dw);
    int* yf = (int*)malloc(sizeof(int) * dh);
    short* xa = (short*)malloc(sizeof(short) * 8 * dw);
    short* yb = (short*)malloc(sizeof(short) * 8 * dh);
    orc_lanczos4_taps(sw, dw, xf, xa);
    orc_lanczos4_taps(sh, dh, yf, yb);
    int* rows = (int*)malloc(sizeof(int) * 8 * (size_t)dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        for (int k = 0; k < 8; k++) {
            const uint8_t* S = src + (size_t)clampi(yf[dy] + k, 0, sh - 1) * sstep;
            int* D = rows + (size_t)k * dw * cn;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    int v = 0;
                    for (int j = 0; j < 8; j++) v += S[(size_t)clampi(xf[dx] + j, 0, sw - 1) * cn + c] * xa[8 * dx + j];
                    D[dx * cn + c] = v;
                }
        }
        uint8_t* D = dst + (size_t)dy * dstep;
        for (int x = 0; x < dw * cn; x++) {
            /* (unsigned: the sum of the eight products may pass 2^31 on the way; what OpenCV's int arithmetic leaves is the low 32 bits) */
            uint32_t v = 0;
            for (int k = 0; k < 8; k++) v += (uint32_t)rows[(size_t)k * dw * cn + x] * (uint32_t)(int32_t)yb[8 * dy + k];
            int r = ((int32_t)(v + (1u << 21))) >> 22;
            D[x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
    free(rows); free(xf); free(yf); free(xa); free(yb);
}

/* ---------------------------------------------------------------------------
 * cv2.resize(INTER_CUBIC), 8-bit, cn channels (scene_manager.py:670-678 with Interpolation.CUBIC, common.py:148-160).
 * resize.cpp, the generic path: cubic_tab[CV_8U] = resizeGeneric_<HResizeCubic<uchar, int, short>, VResizeCubic<uchar, int,
 * short, FixedPtCast<int, uchar, INTER_RESIZE_COEF_BITS * 2>, VResizeCubicVec_32s8u>>.
 *   fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = cvFloor(fx);  fx -= sx;          (the taps are source columns sx-1 .. sx+2)
 *   interpolateCubic(fx, cbuf)  (A = -0.75, float32);  ialpha[k] = saturate_cast<short>(cbuf[k] * 2048)
 *   rows likewise;  a tap outside the image is the nearest pixel inside;  D = sum_j S[sx - 1 + j] * alpha[j]  (int)
 * The VERTICAL pass is where OpenCV builds differ, and `form` says which one is restated:
 *   form 0 ("sse": a build whose baseline is SSE2/SSE3 and that has no IPP -- distribution packages, conda-forge, source builds
 *           with WITH_IPP=OFF): VResizeCubicVec_32s8u takes the row in whole groups of v_int16's 8 lanes,
 *             dst = pack_u8(pack_s16(cvtps2dq( S0*b0 + (S1*b1 + (S2*b2 + S3*b3)) )))   with b_k = beta[k] * 2^-22 in float32,
 *           every product and every sum ROUNDED to float32 (v_muladd is mul + add without FMA), cvtps2dq to nearest even; the
 *           width % 8 elements left over go through the scalar FixedPtCast: (S0*b0 + S1*b1 + S2*b2 + S3*b3 + 2^21) >> 22;
 *   form 1 ("fma": a baseline with fused multiply-add -- aarch64 NEON wheels, x86 builds with an AVX2 baseline and 8 lanes kept
 *           here; v_muladd = one rounding per step), same groups of 8, same tail;
 *   form 2 ("fixed": CV_SIMD off, or OpenCV before the vector pass existed): the scalar FixedPtCast everywhere.
 * The x86-64 PyPI wheels do neither: they carry IPP and ipp_resize takes 8-bit CUBIC (closed source; "results differ from the
 * OpenCV implementation" is OpenCV's own comment on it), so no form here claims to be theirs.
 * ------------------------------------------------------------------------- */
static void cubic_coeffs(float x, float* coeffs)
{
    const float A = -0.75f;
    coeffs[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    coeffs[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    coeffs[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    coeffs[3] = 1.f - coeffs[0] - coeffs[1] - coeffs[2];
}

/* the taps of one axis: first[d] = s - 1 (not clamped), coef[4 d .. 4 d + 3] */
void orc_cubic_taps(int ssize, int dsize, int* first, short* coef)
{
    double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        float cbuf[4];
        cubic_coeffs(f, cbuf);
        first[d] = s - 1;
        for (int k = 0; k < 4; k++) coef[4 * d + k] = sat_s16_round(cbuf[k] * 2048);
    }
}

void orc_resize_cubic_u8(const uint8_t* src, size_t sstep, int sh, int sw, int cn, uint8_t* dst, size_t dstep, int dh, int dw, int form)
{
    int* xf = (int*)malloc(sizeof(int) * dw);
    int* yf = (int*)malloc(sizeof(int) * dh);
    short* xa = (short*)malloc(sizeof(short) * 4 * dw);
    short* yb = (short*)malloc(sizeof(short) * 4 * dh);
    orc_cubic_taps(sw, dw, xf, xa);
    orc_cubic_taps(sh, dh, yf, yb);
    const int width = dw * cn;
    const int vec_end = form == 2 ? 0 : width - width % 8;
    const float scale = 1.f / (2048 * 2048);
    int* rows = (int*)malloc(sizeof(int) * 4 * (size_t)width);
    for (int dy = 0; dy < dh; dy++) {
        for (int k = 0; k < 4; k++) {
            const uint8_t* S = src + (size_t)clampi(yf[dy] + k, 0, sh - 1) * sstep;
            int* D = rows + (size_t)k * width;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    int v = 0;
                    for (int j = 0; j < 4; j++) v += S[(size_t)clampi(xf[dx] + j, 0, sw - 1) * cn + c] * xa[4 * dx + j];
                    D[dx * cn + c] = v;
                }
        }
        const int *S0 = rows, *S1 = rows + width, *S2 = rows + 2 * (size_t)width, *S3 = rows + 3 * (size_t)width;
        const short* beta = yb + 4 * dy;
        const float b0 = beta[0] * scale, b1 = beta[1] * scale, b2 = beta[2] * scale, b3 = beta[3] * scale;
        uint8_t* D = dst + (size_t)dy * dstep;
        int x = 0;
        for (; x < vec_end; x++) {
            float t;
            if (form == 1) {
                t = (float)S3[x] * b3;
                t = fmaf((float)S2[x], b2, t);
                t = fmaf((float)S1[x], b1, t);
                t = fmaf((float)S0[x], b0, t);
            } else {
                /* (volatile: each product and each sum is a float32 of its own whatever the compiler would like to contract) */
                volatile float p3 = (float)S3[x] * b3, p2 = (float)S2[x] * b2, p1 = (float)S1[x] * b1, p0 = (float)S0[x] * b0;
                volatile float s2 = p2 + p3;
                volatile float s1 = p1 + s2;
                t = p0 + s1;
            }
            long r = lrintf(t);                      /* cvtps2dq: to nearest even */
            if (r > 32767) r = 32767;                /* v_pack: int32 -> int16, saturating */
            if (r < -32768) r = -32768;
            D[x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);   /* v_pack_u: int16 -> uint8, saturating */
        }
        for (; x < width; x++) {
            uint32_t v = (uint32_t)S0[x] * (uint32_t)(int32_t)beta[0] + (uint32_t)S1[x] * (uint32_t)(int32_t)beta[1] +
                         (uint32_t)S2[x] * (uint32_t)(int32_t)beta[2] + (uint32_t)S3[x] * (uint32_t)(int32_t)beta[3];
            int r = ((int32_t)(v + (1u << 21))) >> 22;
            D[x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
    free(rows); free(xf); free(yf); free(xa); free(yb);
}

/* Orthonormal 2-D DCT-II of an n x n float32 block: out = C * in * C^T,
 * C[0][j] = sqrt(1/n), C[k][j] = sqrt(2/n) * cos(pi*(2j+1)*k/(2n)).  Only the top-left keep x keep block is produced. */
void orc_dct2d_f32(const float* in, int n, int keep, float* out)
{
    double* c = (double*)malloc(sizeof(double) * keep * n);
    double* t = (double*)malloc(sizeof(double) * keep * n);
    for (int k = 0; k < keep; k++)
        for (int j = 0; j < n; j++)
            c[k * n + j] = k == 0 ? sqrt(1.0 / n) : sqrt(2.0 / n) * cos(3.14159265358979323846 * (2 * j + 1) * k / (2.0 * n));
    /* t[u][x] = sum_y C[u][y] * in[y][x] */
    for (int u = 0; u < keep; u++)
        for (int x = 0; x < n; x++) {
            double acc = 0.0;
            for (int y = 0; y < n; y++) acc += c[u * n + y] * (double)in[y * n + x];
            t[u * n + x] = acc;
        }
    for (int u = 0; u < keep; u++)
        for (int v = 0; v < keep; v++) {
            double acc = 0.0;
            for (int x = 0; x < n; x++) acc += t[u * n + x] * c[v * n + x];
            out[u * keep + v] = (float)acc;
        }
    free(c); free(t);
}

/* ---------------------------------------------------------------------------
 * Per-frame integer score records: the CPU restatement of exactly what the
 * device kernels emit (include/psd_engine.h psd_frame_scores), composed from
 * the primitives above the way the reference detectors compose them:
 *   sad_{h,s,v}: sum |cur - prev| of the HSV planes  (content_detector.py:29-36,166-169)
 *   byte_sum   : sum of all BGR bytes                (threshold_detector.py:127)
 *   hist[256]  : 256-bin histogram of Y              (histogram_detector.py:156-159)
 * ------------------------------------------------------------------------- */
typedef struct {
    uint64_t sad_h, sad_s, sad_v, edge_xor, byte_sum;
    uint32_t hist[256];
} orc_frame_scores;

/* flags: 1 = HSV SADs, 2|4 = luma histogram + byte sum (as PSD_SCORE_* in include/psd_engine.h). */
void orc_score_batch_flags(const uint8_t* frames, int n, int h, int w, size_t row_stride, size_t frame_stride,
                           const uint8_t* prev, orc_frame_scores* out, unsigned flags);

void orc_score_batch(const uint8_t* frames, int n, int h, int w, size_t row_stride, size_t frame_stride,
                     const uint8_t* prev, orc_frame_scores* out)
{
    orc_score_batch_flags(frames, n, h, w, row_stride, frame_stride, prev, out, 7u);
}

void orc_score_batch_flags(const uint8_t* frames, int n, int h, int w, size_t row_stride, size_t frame_stride,
                           const uint8_t* prev, orc_frame_scores* out, unsigned flags)
{
    const int do_hsv = (flags & 1u) != 0, do_luma = (flags & 6u) != 0;
    init_tables();
    size_t np = (size_t)h * w;
    uint8_t* cur = (uint8_t*)malloc(np * 3);
    uint8_t* last = (uint8_t*)malloc(np * 3);
    int have_last = 0;
    if (prev && do_hsv) {
        orc_bgr2hsv_planes(prev, row_stride, last, last + np, last + 2 * np, h, w);
        have_last = 1;
    }
    for (int t = 0; t < n; t++) {
        const uint8_t* f = frames + (size_t)t * frame_stride;
        orc_frame_scores* o = out + t;
        memset(o, 0, sizeof(*o));
        if (do_hsv) orc_bgr2hsv_planes(f, row_stride, cur, cur + np, cur + 2 * np, h, w);
        if (do_hsv && have_last) {
            uint64_t a = 0, b = 0, c = 0;
            for (size_t i = 0; i < np; i++) {
                a += (uint64_t)abs((int)cur[i] - (int)last[i]);
                b += (uint64_t)abs((int)cur[np + i] - (int)last[np + i]);
                c += (uint64_t)abs((int)cur[2 * np + i] - (int)last[2 * np + i]);
            }
            o->sad_h = a; o->sad_s = b; o->sad_v = c;
        }
        uint64_t bs = 0;
        for (int y = 0; do_luma && y < h; y++) {
            const uint8_t* s = f + (size_t)y * row_stride;
            for (int x = 0; x < w; x++, s += 3) {
                bs += (uint64_t)s[0] + s[1] + s[2];
                o->hist[sat_u8(DESCALE(s[0] * B2Y + s[1] * G2Y + s[2] * R2Y, YUV_SHIFT))]++;
            }
        }
        o->byte_sum = bs;
        uint8_t* tsw = cur; cur = last; last = tsw;
        have_last = do_hsv;
    }
    free(cur); free(last);
}
