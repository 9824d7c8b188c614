// cv.resize(src, dsize, interpolation=INTER_CUBIC) on gfx950 (reference: Image.to_resized_image element/image.py:836-852,
// Mask.to_resized_mask element/mask.py:454-479, ScoreMap.to_resized_score_map element/score_map.py:616-640; first
// user on the path: the bottom layer of fill_page_inactive_region, pipeline/text_detection/page_distortion.py:146-161).
//
// Separable Keys cubic (A = -0.75), taps s-1 .. s+2 with border replication.  The per-column / per-row tap offsets and
// coefficients depend on one index only, so the host evaluates them once (float32 arithmetic in OpenCV's order, 11-bit
// fixed point for uint8) and stages two small tables; the kernel is a pure gather: one lane per destination pixel,
// 4 x 4 taps per channel, int32 accumulation with the (sum + 2^21) >> 22 rounding for uint8, float32 left-to-right
// sums for float32.  Bound by HBM/L2 reads of the source (each source row is reused by ~4/scale destination rows).
#include "vkx_internal.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

namespace {

// cvRound of a host float: ties to even, "integer indefinite" (INT_MIN) for NaN and out-of-range values -- a LANCZOS4
// coefficient can be NaN (fraction rounding up to exactly 1.0f makes one tap 0 / 0), and saturate_cast<short> of that
// is -32768 in cv2, not whatever a plain (int) cast of NaN yields.
int cv_round_host(float v)
{
    if (!(v >= -2147483648.f && v < 2147483648.f)) return INT_MIN;
    return (int)std::nearbyint((double)v);
}

struct AxisTable {
    std::vector<int> ofs;      // floor of the source coordinate
    std::vector<float> coef;   // [n][4]
    std::vector<short> icoef;  // [n][4], cvRound(coef * 2048)
};

void cubic_coeffs(float x, float c[4])
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

void build_axis(int ssize, int dsize, AxisTable *t)
{
    t->ofs.resize(dsize); t->coef.resize((size_t)dsize * 4); t->icoef.resize((size_t)dsize * 4);
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s0 = (int)std::floor(f);
        f -= s0;
        t->ofs[d] = s0;
        cubic_coeffs(f, &t->coef[(size_t)d * 4]);
        for (int k = 0; k < 4; k++) {
            const int r = cv_round_host(t->coef[(size_t)d * 4 + k] * 2048.f);
            t->icoef[(size_t)d * 4 + k] = (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
        }
    }
}

__device__ __forceinline__ int clip_index(int x, int n) { return x < 0 ? 0 : (x >= n ? n - 1 : x); }

template <int CN>
__global__ void __launch_bounds__(256) k_resize_cubic_u8(const uint8_t *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                         uint8_t *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                         const int *__restrict__ xofs, const short *__restrict__ xa,
                                                         const int *__restrict__ yofs, const short *__restrict__ yb)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int x0 = xofs[dx], y0 = yofs[dy];
    int sx[4], ax[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { sx[j] = clip_index(x0 - 1 + j, sw) * CN; ax[j] = xa[dx * 4 + j]; }
    unsigned acc[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) acc[c] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint8_t *row = src + (ptrdiff_t)clip_index(y0 - 1 + k, sh) * sstride;
        const int b = yb[dy * 4 + k];
#pragma unroll
        for (int c = 0; c < CN; c++) {
            unsigned hsum = 0; // int32 with wrap, like the int accumulators of the reference implementation
#pragma unroll
            for (int j = 0; j < 4; j++) hsum += (unsigned)((int)row[sx[j] + c] * ax[j]);
            acc[c] += (unsigned)__mul24((int)hsum, b);      // |hsum| < 2^20: same low 32 bits as the 32-bit product
        }
    }
    uint8_t *out = dst + (ptrdiff_t)dy * dstride + (ptrdiff_t)dx * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const int r = ((int)(acc[c] + (1u << 21))) >> 22;
        out[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

__global__ void __launch_bounds__(256) k_resize_cubic_f32(const float *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                          float *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                          const int *__restrict__ xofs, const float *__restrict__ xc,
                                                          const int *__restrict__ yofs, const float *__restrict__ yc)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int x0 = xofs[dx], y0 = yofs[dy];
    int sx[4];
    float ax[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { sx[j] = clip_index(x0 - 1 + j, sw); ax[j] = xc[dx * 4 + j]; }
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float *row = src + (ptrdiff_t)clip_index(y0 - 1 + k, sh) * sstride;
        const float t0 = row[sx[0]] * ax[0], t1 = row[sx[1]] * ax[1], t2 = row[sx[2]] * ax[2], t3 = row[sx[3]] * ax[3];
        float hsum = t0 + t1;
        hsum = hsum + t2;
        hsum = hsum + t3;
        const float term = hsum * yc[dy * 4 + k];
        v = k == 0 ? term : v + term;
    }
    dst[(ptrdiff_t)dy * dstride + dx] = v;
}

// INTER_LINEAR (uint8): 2 x 2 taps, horizontal pass in int32 with 11-bit coefficients, OpenCV's vertical rounding
// uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2); tables as for the cubic kernel (2 entries).
template <int CN>
__global__ void __launch_bounds__(256) k_resize_linear_u8(const uint8_t *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                          uint8_t *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                          const int *__restrict__ xofs, const short *__restrict__ xa,
                                                          const int *__restrict__ yofs, const short *__restrict__ yb)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int sx0 = xofs[dx] * CN, sx1 = clip_index(xofs[dx] + 1, sw) * CN;
    const int a0 = xa[dx * 2], a1 = xa[dx * 2 + 1];
    const int b0 = yb[dy * 2], b1 = yb[dy * 2 + 1];
    const uint8_t *r0 = src + (ptrdiff_t)clip_index(yofs[dy], sh) * sstride;
    const uint8_t *r1 = src + (ptrdiff_t)clip_index(yofs[dy] + 1, sh) * sstride;
    uint8_t *out = dst + (ptrdiff_t)dy * dstride + (ptrdiff_t)dx * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const int h0 = r0[sx0 + c] * a0 + r0[sx1 + c] * a1;
        const int h1 = r1[sx0 + c] * a0 + r1[sx1 + c] * a1;
        out[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

// the exact 2 x 2 shrink cv.resize routes to INTER_AREA
template <int CN>
__global__ void __launch_bounds__(256) k_resize_half_u8(const uint8_t *__restrict__ src, ptrdiff_t sstride,
                                                        uint8_t *__restrict__ dst, int dh, int dw, ptrdiff_t dstride)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const uint8_t *p = src + (ptrdiff_t)(2 * dy) * sstride + (ptrdiff_t)(2 * dx) * CN;
#pragma unroll
    for (int c = 0; c < CN; c++)
        dst[(ptrdiff_t)dy * dstride + (ptrdiff_t)dx * CN + c] = (uint8_t)((p[c] + p[CN + c] + p[sstride + c] + p[sstride + CN + c] + 2) >> 2);
}

// INTER_NEAREST: source index min(floor(d * scale), size - 1), scale in double
template <int CN>
__global__ void __launch_bounds__(256) k_resize_nearest_u8(const uint8_t *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                           uint8_t *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                           double ifx, double ify)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int sx = min((int)floor(dx * ifx), sw - 1), sy = min((int)floor(dy * ify), sh - 1);
    const uint8_t *p = src + (ptrdiff_t)sy * sstride + (ptrdiff_t)sx * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) dst[(ptrdiff_t)dy * dstride + (ptrdiff_t)dx * CN + c] = p[c];
}

void build_linear_axis(int ssize, int dsize, bool horizontal, std::vector<int> *ofs, std::vector<short> *coef)
{
    ofs->resize(dsize); coef->resize((size_t)dsize * 2);
    const double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s0 = (int)std::floor(f);
        f -= s0;
        if (horizontal) {
            if (s0 < 0) { f = 0; s0 = 0; }
            if (s0 >= ssize - 1) { f = 0; s0 = ssize - 1; }
        }
        (*ofs)[d] = s0;
        const float c[2] = {1.f - f, f};
        for (int k = 0; k < 2; k++) {
            const int r = cv_round_host(c[k] * 2048.f);
            (*coef)[(size_t)d * 2 + k] = (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
        }
    }
}

// zoom_in_blur (photometric/blur.py:264-316): uint16 accumulation of centred crops of enlarged copies, then
// uint8(clip((1 - alpha) * px + alpha * rint(acc / count))) in float64.
__global__ void __launch_bounds__(256) k_accumulate_crop(const uint8_t *__restrict__ src, ptrdiff_t sstride, int up, int left,
                                                         uint16_t *__restrict__ acc, int h, int wc, int cn, int init)
{
    const int xe = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xe >= wc || y >= h) return;
    const uint16_t v = src[(ptrdiff_t)(up + y) * sstride + (ptrdiff_t)left * cn + xe];
    uint16_t *a = acc + (size_t)y * wc + xe;
    *a = init ? v : (uint16_t)(*a + v);      // numpy uint16 arithmetic wraps
}

__global__ void __launch_bounds__(256) k_zoom_finish(const uint8_t *__restrict__ src, ptrdiff_t sstride,
                                                     const uint16_t *__restrict__ acc, int h, int wc, int count, double w0,
                                                     double w1, uint8_t *__restrict__ dst, ptrdiff_t dstride)
{
    const int xe = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xe >= wc || y >= h) return;
    const double t0 = w0 * (double)src[(ptrdiff_t)y * sstride + xe];
    const double mean = rint((double)acc[(size_t)y * wc + xe] / (double)count);   // np.round: half to even
    const double t1 = w1 * mean;
    double v = t0 + t1;
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    dst[(ptrdiff_t)y * dstride + xe] = (uint8_t)v;
}


// ---- the other interpolations PageResizingStep samples (pipeline/text_detection/page_resizing.py:110-181 via
// utility/opt.py:125-148); arithmetic as restated in oracle/vkx_oracle.c.

// interpolateLanczos4 (imgproc): 8 taps s-3 .. s+4
void lanczos4_coeffs(float x, float c[8])
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    const double pi = 3.1415926535897932384626433832795;
    if (x < FLT_EPSILON) {
        for (int i = 0; i < 8; i++) c[i] = 0;
        c[3] = 1;
        return;
    }
    float sum = 0;
    const double y0 = -(x + 3) * pi * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
    for (int i = 0; i < 8; i++) {
        const double y = -(x + 3 - i) * pi * 0.25;
        c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) c[i] *= sum;
}

struct AxisTable8 {
    std::vector<int> ofs;
    std::vector<float> coef;   // [n][8]
    std::vector<short> icoef;
};

void build_axis8(int ssize, int dsize, AxisTable8 *t)
{
    t->ofs.resize(dsize); t->coef.resize((size_t)dsize * 8); t->icoef.resize((size_t)dsize * 8);
    const double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s0 = (int)std::floor(f);
        f -= s0;
        t->ofs[d] = s0;
        lanczos4_coeffs(f, &t->coef[(size_t)d * 8]);
        for (int k = 0; k < 8; k++) {
            const int r = cv_round_host(t->coef[(size_t)d * 8 + k] * 2048.f);
            t->icoef[(size_t)d * 8 + k] = (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
        }
    }
}

template <int CN>
__global__ void __launch_bounds__(256) k_resize_lanczos4_u8(const uint8_t *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                            uint8_t *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                            const int *__restrict__ xofs, const short *__restrict__ xa,
                                                            const int *__restrict__ yofs, const short *__restrict__ yb)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int x0 = xofs[dx], y0 = yofs[dy];
    int sx[8], ax[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { sx[j] = clip_index(x0 - 3 + j, sw) * CN; ax[j] = xa[dx * 8 + j]; }
    unsigned acc[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) acc[c] = 0;
    for (int k = 0; k < 8; k++) {
        const uint8_t *row = src + (ptrdiff_t)clip_index(y0 - 3 + k, sh) * sstride;
        const int b = yb[dy * 8 + k];
#pragma unroll
        for (int c = 0; c < CN; c++) {
            unsigned hsum = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) hsum += (unsigned)((int)row[sx[j] + c] * ax[j]);
            acc[c] += (unsigned)__mul24((int)hsum, b);      // |hsum| < 2^20: same low 32 bits as the 32-bit product
        }
    }
    uint8_t *out = dst + (ptrdiff_t)dy * dstride + (ptrdiff_t)dx * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const int r = ((int)(acc[c] + (1u << 21))) >> 22;
        out[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

__global__ void __launch_bounds__(256) k_resize_lanczos4_f32(const float *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                             float *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                             const int *__restrict__ xofs, const float *__restrict__ xc,
                                                             const int *__restrict__ yofs, const float *__restrict__ yc)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int x0 = xofs[dx], y0 = yofs[dy];
    int sx[8];
    float ax[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { sx[j] = clip_index(x0 - 3 + j, sw); ax[j] = xc[dx * 8 + j]; }
    float v = 0.f;
    for (int k = 0; k < 8; k++) {
        const float *row = src + (ptrdiff_t)clip_index(y0 - 3 + k, sh) * sstride;
        float hsum = row[sx[0]] * ax[0];
#pragma unroll
        for (int j = 1; j < 8; j++) { const float t = row[sx[j]] * ax[j]; hsum = hsum + t; }
        const float term = hsum * yc[dy * 8 + k];
        v = k == 0 ? term : v + term;
    }
    dst[(ptrdiff_t)dy * dstride + dx] = v;
}

// INTER_NEAREST_EXACT (resizeNN_bitexact): 16.16 index arithmetic; CN = bytes per element (4 = one float32)
template <int CN>
__global__ void __launch_bounds__(256) k_resize_nearest_exact(const uint8_t *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                              uint8_t *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                              int ifx, int ifx0, int ify, int ify0)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int sx = min((int)(((long long)ifx * dx + ifx0) >> 16), sw - 1);
    const int sy = min((int)(((long long)ify * dy + ify0) >> 16), sh - 1);
    const uint8_t *p = src + (ptrdiff_t)sy * sstride + (ptrdiff_t)sx * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) dst[(ptrdiff_t)dy * dstride + (ptrdiff_t)dx * CN + c] = p[c];
}

// INTER_LINEAR_EXACT on uint8 (resize_bitExact): per axis (offset, 8.8 weight of the second sample) and the range
// [mn, mx) of destination indices that interpolate; outside it the first / last source sample is copied
void build_linear_exact_axis(int ssize, int dsize, std::vector<int> *ofs, std::vector<int> *w1, int *dmin, int *dmax)
{
    ofs->resize(dsize); w1->resize(dsize);
    const double inv_scale = (double)dsize / ssize, scale = 1.0 / inv_scale;
    int mn = 0, mx = dsize;
    for (int d = 0; d < dsize; d++) {
        const double fval = scale * ((double)d + 0.5) - 0.5;
        int ival = (int)std::floor(fval);
        (*w1)[d] = 0;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) (*w1)[d] = (int)std::nearbyint((fval - (double)ival) * 256.0);
            else { ival = ssize - 1; mx = std::min(mx, d); }
        } else { mn = std::max(mn, d + 1); ival = 0; }
        (*ofs)[d] = ival;
    }
    if (mx < mn) mx = mn;
    *dmin = mn; *dmax = mx;
}

template <int CN>
__global__ void __launch_bounds__(256) k_resize_linear_exact_u8(const uint8_t *__restrict__ src, ptrdiff_t sstride,
                                                                uint8_t *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                                const int *__restrict__ xofs, const int *__restrict__ xw,
                                                                const int *__restrict__ yofs, const int *__restrict__ yw,
                                                                int xmin, int xmax, int ymin, int ymax)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const bool two = dy >= ymin && dy < ymax;
    const int r0 = dy < ymin ? 0 : (dy >= ymax ? yofs[dh - 1] : yofs[dy]);
    const uint8_t *S0 = src + (ptrdiff_t)r0 * sstride, *S1 = S0 + (two ? sstride : 0);
    // horizontal taps: both on the first / last sample outside [xmin, xmax)
    const int xa = dx < xmin ? 0 : (dx >= xmax ? xofs[dw - 1] : xofs[dx]);
    const bool xin = dx >= xmin && dx < xmax;
    const unsigned w1 = xin ? (unsigned)xw[dx] : 0u, w0 = 256u - w1;
    const int xb = xin ? xa + 1 : xa;
    const unsigned b1 = two ? (unsigned)yw[dy] : 0u, b0 = 256u - b1;
    uint8_t *out = dst + (ptrdiff_t)dy * dstride + (ptrdiff_t)dx * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const unsigned h0 = w0 * S0[xa * CN + c] + w1 * S0[xb * CN + c];
        unsigned r;
        if (two) {
            const unsigned h1 = w0 * S1[xa * CN + c] + w1 * S1[xb * CN + c];
            r = (h0 * b0 + h1 * b1 + (1u << 15)) >> 16;
        } else {
            r = (h0 + 128u) >> 8;
        }
        out[c] = (uint8_t)(r > 255u ? 255u : r);
    }
}

// INTER_LINEAR on float32 (what INTER_LINEAR_EXACT falls back to for a ScoreMap)
__global__ void __launch_bounds__(256) k_resize_linear_f32(const float *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                           float *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                           double scale_x, double scale_y)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int y0 = (int)floorf(fy);
    fy -= y0;
    if (y0 < 0) { y0 = 0; fy = 0; }
    if (y0 >= sh - 1) { y0 = sh - 1; fy = 0; }
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int x0 = (int)floorf(fx);
    fx -= x0;
    if (x0 < 0) { x0 = 0; fx = 0; }
    if (x0 >= sw - 1) { x0 = sw - 1; fx = 0; }
    const int x1 = clip_index(x0 + 1, sw);
    const float *S0 = src + (ptrdiff_t)y0 * sstride, *S1 = src + (ptrdiff_t)clip_index(y0 + 1, sh) * sstride;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const float p0 = S0[x0] * a0, p1 = S0[x1] * a1, q0 = S1[x0] * a0, q1 = S1[x1] * a1;
    const float h0 = p0 + p1, h1 = q0 + q1;
    const float t0 = h0 * b0, t1 = h1 * b1;
    dst[(ptrdiff_t)dy * dstride + dx] = t0 + t1;
}

// INTER_AREA, integer scale factors (ResizeAreaFast): box sums, then * (1.f / area)
template <int CN, bool F32>
__global__ void __launch_bounds__(256) k_resize_area_fast(const void *__restrict__ src_, ptrdiff_t sstride, void *__restrict__ dst_,
                                                          int dh, int dw, ptrdiff_t dstride, int isx, int isy)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int area = isx * isy;
    const float scale = 1.f / area;
    if (F32) {
        const float *S = (const float *)src_ + (ptrdiff_t)(dy * isy) * sstride + (ptrdiff_t)(dx * isx);
        if (isx == 2 && isy == 2) {          // the vector body of the 2 x 2 case pairs the rows
            const float top = S[0] + S[1], bottom = S[sstride] + S[sstride + 1];
            const float s4 = top + bottom;
            ((float *)dst_)[(ptrdiff_t)dy * dstride + dx] = s4 * 0.25f;
            return;
        }
        float sum = 0;
        int k = 0;
        for (; k <= area - 4; k += 4) {      // the reference sums the row-major box four samples at a time
            const float a0 = S[(k / isx) * sstride + (k % isx)], a1 = S[((k + 1) / isx) * sstride + ((k + 1) % isx)];
            const float a2 = S[((k + 2) / isx) * sstride + ((k + 2) % isx)], a3 = S[((k + 3) / isx) * sstride + ((k + 3) % isx)];
            float g = a0 + a1;
            g = g + a2;
            g = g + a3;
            sum = sum + g;
        }
        for (; k < area; k++) sum = sum + S[(k / isx) * sstride + (k % isx)];
        ((float *)dst_)[(ptrdiff_t)dy * dstride + dx] = sum * scale;
    } else {
        const uint8_t *S = (const uint8_t *)src_ + (ptrdiff_t)(dy * isy) * sstride + (ptrdiff_t)(dx * isx) * CN;
        uint8_t *out = (uint8_t *)dst_ + (ptrdiff_t)dy * dstride + (ptrdiff_t)dx * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) {
            int sum = 0;
            for (int y = 0; y < isy; y++)
                for (int x = 0; x < isx; x++) sum += S[(ptrdiff_t)y * sstride + x * CN + c];
            const int r = (isx == 2 && isy == 2) ? (sum + 2) >> 2 : vkd::cv_round((float)sum * scale);
            out[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
}

// INTER_AREA, fractional scale (ResizeArea): computeResizeAreaTab's (source index, weight) runs per destination index
struct AreaTab {
    std::vector<int> start;    // [dsize + 1] first entry of every destination index
    std::vector<int> si;
    std::vector<float> alpha;
};

void build_area_tab(int ssize, int dsize, double scale, AreaTab *t)
{
    t->start.assign(dsize + 1, 0); t->si.clear(); t->alpha.clear();
    for (int dx = 0; dx < dsize; dx++) {
        t->start[dx] = (int)t->si.size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) { t->si.push_back(sx1 - 1); t->alpha.push_back((float)((sx1 - fsx1) / cell)); }
        for (int sx = sx1; sx < sx2; sx++) { t->si.push_back(sx); t->alpha.push_back((float)(1.0 / cell)); }
        if (fsx2 - sx2 > 1e-3) { t->si.push_back(sx2); t->alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)); }
    }
    t->start[dsize] = (int)t->si.size();
}

template <int CN, bool F32>
__global__ void __launch_bounds__(256) k_resize_area(const void *__restrict__ src_, ptrdiff_t sstride, void *__restrict__ dst_, int dh,
                                                     int dw, ptrdiff_t dstride, const int *__restrict__ xstart,
                                                     const int *__restrict__ xsi, const float *__restrict__ xal,
                                                     const int *__restrict__ ystart, const int *__restrict__ ysi,
                                                     const float *__restrict__ yal)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const int x0 = xstart[dx], x1 = xstart[dx + 1], y0 = ystart[dy], y1 = ystart[dy + 1];
    float sum[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) sum[c] = 0.f;
    for (int j = y0; j < y1; j++) {
        const float beta = yal[j];
        float buf[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) buf[c] = 0.f;
        for (int k = x0; k < x1; k++) {
            const float alpha = xal[k];
#pragma unroll
            for (int c = 0; c < CN; c++) {
                const float v = F32 ? ((const float *)src_)[(ptrdiff_t)ysi[j] * sstride + xsi[k]]
                                    : (float)((const uint8_t *)src_)[(ptrdiff_t)ysi[j] * sstride + xsi[k] * CN + c];
                const float t = v * alpha;
                buf[c] = buf[c] + t;
            }
        }
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const float t = beta * buf[c];
            sum[c] = j == y0 ? t : sum[c] + t;
        }
    }
#pragma unroll
    for (int c = 0; c < CN; c++) {
        if (F32) ((float *)dst_)[(ptrdiff_t)dy * dstride + dx] = sum[c];
        else {
            const int r = vkd::cv_round(sum[c]);
            ((uint8_t *)dst_)[(ptrdiff_t)dy * dstride + (ptrdiff_t)dx * CN + c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
}

// Host-built tables of one resize geometry in device memory, from the ctx cache when the geometry was seen recently.
// `build` fills the host arrays (kept alive until the upload has completed) and the metadata on a miss only.
template <class Build>
int cached_tables(vkx_ctx *ctx, const int key[6], Build build, std::vector<const void *> *ptrs, const std::vector<int> **meta)
{
    vkx_ctx::ResizeTabs *slot = nullptr;
    for (auto &t : ctx->resize_tabs)
        if (std::equal(key, key + 6, t.key)) slot = &t;
    int n = 0;
    if (!slot) {
        slot = &ctx->resize_tabs[0];
        for (auto &t : ctx->resize_tabs)
            if (t.stamp < slot->stamp) slot = &t;
        std::vector<std::pair<const void *, size_t>> arrays;
        std::vector<int> m;
        build(&arrays, &m);
        if (arrays.size() > 7) return VKX_ERR_INVALID;
        size_t total = 0;
        for (size_t i = 0; i < arrays.size(); i++) { slot->off[i] = total; total += (arrays[i].second + 255) & ~(size_t)255; }
        slot->off[7] = arrays.size();
        slot->key[0] = -1;
        int rc = vkx_scratch_reserve(ctx, &slot->buf, total ? total : 256);
        if (rc) return rc;
        // the tables travel as ONE copy out of the page-locked ring (which keeps them alive): no copy per table, no stream
        // synchronisation per cache miss -- every page resizes to a geometry of its own
        if (total) {
            void *ring = nullptr;
            if ((rc = vkx_desc_ring_take(ctx, total, &ring))) return rc;
            for (size_t i = 0; i < arrays.size(); i++)
                if (arrays[i].second) memcpy((unsigned char *)ring + slot->off[i], arrays[i].first, arrays[i].second);
            VKX_HIP(hipMemcpyAsync(slot->buf.ptr, ring, total, hipMemcpyHostToDevice, ctx->stream));
        }
        slot->yofs.swap(m);
        std::copy(key, key + 6, slot->key);
    }
    n = (int)slot->off[7];
    slot->stamp = ++ctx->resize_clock;
    ptrs->clear();
    for (int i = 0; i < n; i++) ptrs->push_back((unsigned char *)slot->buf.ptr + slot->off[i]);
    *meta = &slot->yofs;
    return VKX_OK;
}

#define VKX_CN_SWITCH(cn, KERNEL, ...)                                      \
    switch (cn) {                                                           \
    case 1: KERNEL<1><<<grid, 256, 0, ctx->stream>>>(__VA_ARGS__); break;   \
    case 3: KERNEL<3><<<grid, 256, 0, ctx->stream>>>(__VA_ARGS__); break;   \
    default: KERNEL<4><<<grid, 256, 0, ctx->stream>>>(__VA_ARGS__); break;  \
    }

// interpolations shared by the uint8 and float32 entry points; elem = bytes per element for the nearest kernels
int resize_nearest_exact(vkx_ctx *ctx, const void *src, int sh, int sw, int elem, ptrdiff_t sstride_b, void *dst, int dh, int dw,
                         ptrdiff_t dstride_b)
{
    const int ifx = (int)((((long long)sw << 16) + dw / 2) / dw), ifx0 = ifx / 2 - sw % 2;
    const int ify = (int)((((long long)sh << 16) + dh / 2) / dh), ify0 = ify / 2 - sh % 2;
    dim3 grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    VKX_TIMED(ctx, "k_resize_nearest_exact");
    VKX_CN_SWITCH(elem, k_resize_nearest_exact, (const uint8_t *)src, sh, sw, sstride_b, (uint8_t *)dst, dh, dw, dstride_b, ifx, ifx0, ify, ify0)
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

template <bool F32>
int resize_area(vkx_ctx *ctx, const void *src, int sh, int sw, int cn, ptrdiff_t sstride, void *dst, int dh, int dw, ptrdiff_t dstride)
{
    if (dw > sw || dh > sh) {
        vkx_set_error("INTER_AREA is implemented for shrinking only (the reference samples it only then)");
        return VKX_ERR_UNSUPPORTED;
    }
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    const int isx = (int)std::nearbyint(scale_x), isy = (int)std::nearbyint(scale_y);
    dim3 grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    if (std::fabs(scale_x - isx) < DBL_EPSILON && std::fabs(scale_y - isy) < DBL_EPSILON) {
        VKX_TIMED(ctx, "k_resize_area_fast");
        switch (cn) {
        case 1: k_resize_area_fast<1, F32><<<grid, 256, 0, ctx->stream>>>(src, sstride, dst, dh, dw, dstride, isx, isy); break;
        case 3: k_resize_area_fast<3, F32><<<grid, 256, 0, ctx->stream>>>(src, sstride, dst, dh, dw, dstride, isx, isy); break;
        default: k_resize_area_fast<4, F32><<<grid, 256, 0, ctx->stream>>>(src, sstride, dst, dh, dw, dstride, isx, isy); break;
        }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    AreaTab tx, ty;     // filled on a cache miss only
    std::vector<const void *> p;
    const std::vector<int> *meta;
    const int key[6] = {103, 0, sh, sw, dh, dw};
    int rc = cached_tables(ctx, key, [&](std::vector<std::pair<const void *, size_t>> *arrays, std::vector<int> *) {
        build_area_tab(sw, dw, scale_x, &tx);
        build_area_tab(sh, dh, scale_y, &ty);
        *arrays = {{tx.start.data(), sizeof(int) * tx.start.size()}, {tx.si.data(), sizeof(int) * tx.si.size()},
                   {tx.alpha.data(), sizeof(float) * tx.alpha.size()}, {ty.start.data(), sizeof(int) * ty.start.size()},
                   {ty.si.data(), sizeof(int) * ty.si.size()}, {ty.alpha.data(), sizeof(float) * ty.alpha.size()}};
    }, &p, &meta);
    if (rc) return rc;
    VKX_TIMED(ctx, "k_resize_area");
#define VKX_AREA_ARGS src, sstride, dst, dh, dw, dstride, (const int *)p[0], (const int *)p[1], (const float *)p[2], (const int *)p[3], (const int *)p[4], (const float *)p[5]
    switch (cn) {
    case 1: k_resize_area<1, F32><<<grid, 256, 0, ctx->stream>>>(VKX_AREA_ARGS); break;
    case 3: k_resize_area<3, F32><<<grid, 256, 0, ctx->stream>>>(VKX_AREA_ARGS); break;
    default: k_resize_area<4, F32><<<grid, 256, 0, ctx->stream>>>(VKX_AREA_ARGS); break;
    }
#undef VKX_AREA_ARGS
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}


// ---- separable form of the CUBIC / LANCZOS4 gathers ------------------------------------------------------------------
// One workgroup = a 64 x 16 destination tile.  The source rows the tile's 16 destination rows reach (clipped to the
// image like the taps themselves) get their horizontal pass once, into LDS; the vertical pass reads them back.  The
// sums are the ones of the direct kernels above -- the horizontal sum of a source row does not depend on the
// destination row it is used for -- with K (rows / 16 + 1) multiply-adds per sample instead of K^2.  A tile that would
// need more than kSepRows source rows (a shrink by more than ~2x) is left to the direct kernels.
constexpr int kSepTileW = 64, kSepTileH = 16, kSepRows = 40;

template <typename T, typename CT, typename AT, int CN, int KS, int ROWS = kSepRows>
__global__ void __launch_bounds__(256) k_resize_sep(const T *__restrict__ src, int sh, int sw, ptrdiff_t sstride,
                                                    T *__restrict__ dst, int dh, int dw, ptrdiff_t dstride,
                                                    const int *__restrict__ xofs, const CT *__restrict__ xa,
                                                    const int *__restrict__ yofs, const CT *__restrict__ yb)
{
    __shared__ AT hbuf[ROWS * kSepTileW * CN];      // ROWS: 40, or 24 when no tile needs more (twice the workgroups per CU)
    constexpr int LEFT = KS / 2 - 1;          // taps s - LEFT .. s + KS / 2
    const int x0 = blockIdx.x * kSepTileW, y0 = blockIdx.y * kSepTileH;
    const int ylast = min(y0 + kSepTileH, dh) - 1;
    const int rmin = clip_index(yofs[y0] - LEFT, sh), rmax = clip_index(yofs[ylast] + KS / 2, sh);
    const int nrows = rmax - rmin + 1;
    // the vertical pass's row offsets and coefficients (wave-uniform) are fetched now: their latency hides under the horizontal pass
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int vt0[kSepTileH / 4];
    CT vb[kSepTileH / 4][KS];
#pragma unroll
    for (int i = 0; i < kSepTileH / 4; i++) {
        const int y = min(y0 + wave0 + 4 * i, dh - 1);
        vt0[i] = yofs[y] - LEFT;
#pragma unroll
        for (int k = 0; k < KS; k++) vb[i][k] = yb[y * KS + k];
    }
    if constexpr (sizeof(T) == 1) {
        // uint8: a thread keeps its column -- tap offset and the KS coefficients are loaded once, not once per source row --
        // and, when no tap is clipped, fetches the KS x CN consecutive source bytes of a row as whole dwords
        const int hx = threadIdx.x & 63, hdx = x0 + hx;
        if (hdx < dw) {
            const int s0 = xofs[hdx] - LEFT;
            int a[KS];
#pragma unroll
            for (int j = 0; j < KS; j++) a[j] = (int)xa[hdx * KS + j];
            const bool whole = s0 >= 0 && s0 + KS <= sw;
            constexpr int NB = KS * CN, ND = (NB + 3) / 4;
            typedef uint32_t u32_unaligned __attribute__((aligned(1)));
            // (the last dword may reach up to 3 bytes past the taps: only inside the row, never past the plane)
            const bool fast = whole && (ptrdiff_t)(s0 * CN + ND * 4) <= (ptrdiff_t)sw * CN;
            // The rows of a wavefront in batches of kBatch: all tap loads of a batch are issued before the first sum -- the tile used
            // to pay one memory round trip per source row (five in a row for a 1.05 x cubic: the kernel ran at the latency of its
            // loads, 13 % of the HBM roofline).
            constexpr int kBatch = 5;
            for (int rb = threadIdx.x >> 6; rb < nrows; rb += 4 * kBatch) {
                uint32_t w[kBatch][ND];
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    const int r = min(rb + 4 * u, nrows - 1);
                    const uint8_t *row = (const uint8_t *)src + (ptrdiff_t)(rmin + r) * sstride;
#pragma unroll
                    for (int q = 0; q < ND; q++) w[u][q] = fast ? *(const u32_unaligned *)(row + (ptrdiff_t)s0 * CN + 4 * q) : 0u;
                }
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    const int r = rb + 4 * u;
                    if (r >= nrows) break;
                    int hsum[CN];
#pragma unroll
                    for (int c = 0; c < CN; c++) hsum[c] = 0;
                    if (fast) {
#pragma unroll
                        for (int j = 0; j < KS; j++)
#pragma unroll
                            for (int c = 0; c < CN; c++) {
                                const int bb = j * CN + c;
                                hsum[c] += (int)((w[u][bb >> 2] >> (8 * (bb & 3))) & 0xffu) * a[j];
                            }
                    } else {
                        const uint8_t *row = (const uint8_t *)src + (ptrdiff_t)(rmin + r) * sstride;
#pragma unroll
                        for (int j = 0; j < KS; j++) {
                            const int sx = clip_index(s0 + j, sw) * CN;
#pragma unroll
                            for (int c = 0; c < CN; c++) hsum[c] += (int)row[sx + c] * a[j];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CN; c++) hbuf[(r * kSepTileW + hx) * CN + c] = (AT)hsum[c];
                }
            }
        }
    } else {
    for (int idx = threadIdx.x; idx < nrows * kSepTileW; idx += 256) {
        const int r = idx / kSepTileW, lx = idx - r * kSepTileW, dx = x0 + lx;
        if (dx >= dw) continue;
        const T *row = src + (ptrdiff_t)(rmin + r) * sstride;
        const int s0 = xofs[dx] - LEFT;
        AT hsum[CN];
#pragma unroll
        for (int j = 0; j < KS; j++) {
            const int sx = clip_index(s0 + j, sw) * CN;
            const CT a = xa[dx * KS + j];
#pragma unroll
            for (int c = 0; c < CN; c++) {
                const AT term = row[sx + c] * a;
                hsum[c] = j == 0 ? term : hsum[c] + term;
            }
        }
#pragma unroll
        for (int c = 0; c < CN; c++) hbuf[(r * kSepTileW + lx) * CN + c] = hsum[c];
    }
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, dx = x0 + lx;
    // the destination row is the same for the 64 lanes of a wavefront: row offsets and vertical coefficients are scalar
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (sizeof(T) == 1 && CN == 3) {
        // RGB: four neighbouring lanes store their 12 bytes as three dwords (the group store of the fused chain kernel)
        typedef uint32_t u32_unaligned __attribute__((aligned(1)));
        const int tw = min(kSepTileW, dw - x0), full4 = (tw >> 2) << 2;
        const int right4 = min(lx + 1, 63) << 2;
#pragma unroll
        for (int i = 0; i < kSepTileH / 4; i++) {
            const int y = y0 + wave + 4 * i;
            if (y > ylast) break;
            const int t0 = vt0[i];
            int acc[3] = {0, 0, 0};
#pragma unroll
            for (int k = 0; k < KS; k++) {
                const AT *h = hbuf + ((clip_index(t0 + k, sh) - rmin) * kSepTileW + lx) * 3;
                const int b = (int)vb[i][k];
#pragma unroll
                for (int c = 0; c < 3; c++) acc[c] += __mul24((int)h[c], b);      // |h| < 2^20 (255 x the taps' |coefficients|): the 24-bit multiply is the 32-bit one, at full rate
            }
            uint32_t P = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int r = (acc[c] + (1 << 21)) >> 22;
                P |= (uint32_t)(r < 0 ? 0 : (r > 255 ? 255 : r)) << (8 * c);
            }
            const uint32_t Pn = (uint32_t)__builtin_amdgcn_ds_bpermute(right4, (int)P);
            if (dx >= dw) continue;
            uint8_t *orow = (uint8_t *)dst + (ptrdiff_t)y * dstride;
            const int m = lx & 3;
            if (lx < full4) {
                if (m < 3) *(u32_unaligned *)(orow + (ptrdiff_t)x0 * 3 + (lx >> 2) * 12 + m * 4) = (P >> (8 * m)) | (Pn << (24 - 8 * m));
            } else {
                uint8_t *o = orow + (ptrdiff_t)dx * 3;
                o[0] = (uint8_t)P; o[1] = (uint8_t)(P >> 8); o[2] = (uint8_t)(P >> 16);
            }
        }
        return;
    }
    if (dx >= dw) return;
#pragma unroll
    for (int i = 0; i < kSepTileH / 4; i++) {
        const int y = y0 + wave + 4 * i;
        if (y > ylast) break;
        const int t0 = vt0[i];
        AT acc[CN];
#pragma unroll
        for (int k = 0; k < KS; k++) {
            const AT *h = hbuf + ((clip_index(t0 + k, sh) - rmin) * kSepTileW + lx) * CN;
            const CT b = vb[i][k];
#pragma unroll
            for (int c = 0; c < CN; c++) {
                if constexpr (sizeof(T) == 1) {
                    const AT term = (AT)__mul24((int)h[c], (int)b);
                    acc[c] = k == 0 ? term : acc[c] + term;
                } else {
                    const AT term = h[c] * b;
                    acc[c] = k == 0 ? term : acc[c] + term;
                }
            }
        }
        T *out = dst + (ptrdiff_t)y * dstride + (ptrdiff_t)dx * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) {
            if constexpr (sizeof(T) == 1) {
                const int r = ((int)(acc[c] + (1u << 21))) >> 22;
                out[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
            } else {
                out[c] = acc[c];
            }
        }
    }
}

// can every 16-row destination tile keep its source rows in LDS?
// the most source rows a tile of the separable form needs; 0: more than kSepRows somewhere (the direct kernels take the call)
int separable_rows(const std::vector<int> &yofs, int sh, int dh, int ks)
{
    static const bool force_direct = getenv("VKX_RESIZE_DIRECT") != nullptr;   // parity aid: the two forms must agree
    if (force_direct) return 0;
    const int left = ks / 2 - 1;
    auto clip = [sh](int v) { return v < 0 ? 0 : (v >= sh ? sh - 1 : v); };
    int most = 1;
    for (int y0 = 0; y0 < dh; y0 += kSepTileH) {
        const int ylast = std::min(y0 + kSepTileH, dh) - 1;
        most = std::max(most, clip(yofs[ylast] + ks / 2) - clip(yofs[y0] - left) + 1);
    }
    return most > kSepRows ? 0 : most;
}
bool separable_fits(const std::vector<int> &yofs, int sh, int dh, int ks) { return separable_rows(yofs, sh, dh, ks) != 0; }
constexpr int kSepRowsSmall = 24;

template <int KS>
void launch_sep_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t sstride, uint8_t *dst, int dh, int dw,
                   ptrdiff_t dstride, const int *xofs, const short *xa, const int *yofs, const short *yb, int rows = kSepRows)
{
    dim3 grid(vkx_blocks(dw, kSepTileW), vkx_blocks(dh, kSepTileH));
    if (rows <= kSepRowsSmall) {
        switch (cn) {
        case 1: k_resize_sep<uint8_t, short, unsigned, 1, KS, kSepRowsSmall><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, xofs, xa, yofs, yb); break;
        case 3: k_resize_sep<uint8_t, short, unsigned, 3, KS, kSepRowsSmall><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, xofs, xa, yofs, yb); break;
        default: k_resize_sep<uint8_t, short, unsigned, 4, KS, kSepRowsSmall><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, xofs, xa, yofs, yb); break;
        }
        return;
    }
    switch (cn) {
    case 1: k_resize_sep<uint8_t, short, unsigned, 1, KS><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, xofs, xa, yofs, yb); break;
    case 3: k_resize_sep<uint8_t, short, unsigned, 3, KS><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, xofs, xa, yofs, yb); break;
    default: k_resize_sep<uint8_t, short, unsigned, 4, KS><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, sstride, dst, dh, dw, dstride, xofs, xa, yofs, yb); break;
    }
}

// The four tables (column offsets + coefficients, row offsets + coefficients) of a CUBIC (taps = 4) or LANCZOS4
// (taps = 8) resize in device memory, from the ctx cache when the geometry was seen recently.
int resize_tables(vkx_ctx *ctx, int taps, bool fixed, int sh, int sw, int dh, int dw, const int **xofs, const void **xcoef,
                  const int **yofs, const void **ycoef, const std::vector<int> **yofs_host)
{
    const int key[6] = {taps, fixed ? 1 : 0, sh, sw, dh, dw};
    vkx_ctx::ResizeTabs *slot = nullptr;
    for (auto &t : ctx->resize_tabs)
        if (std::equal(key, key + 6, t.key)) slot = &t;
    if (!slot) {
        slot = &ctx->resize_tabs[0];
        for (auto &t : ctx->resize_tabs)
            if (t.stamp < slot->stamp) slot = &t;
        std::vector<int> xo, yo;
        std::vector<float> xc, yc;
        std::vector<short> xi, yi;
        if (taps == 4) {
            AxisTable tx, ty;
            build_axis(sw, dw, &tx);
            build_axis(sh, dh, &ty);
            xo.swap(tx.ofs); yo.swap(ty.ofs); xc.swap(tx.coef); yc.swap(ty.coef); xi.swap(tx.icoef); yi.swap(ty.icoef);
        } else {
            AxisTable8 tx, ty;
            build_axis8(sw, dw, &tx);
            build_axis8(sh, dh, &ty);
            xo.swap(tx.ofs); yo.swap(ty.ofs); xc.swap(tx.coef); yc.swap(ty.coef); xi.swap(tx.icoef); yi.swap(ty.icoef);
        }
        const size_t csz = fixed ? sizeof(short) : sizeof(float);
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        slot->off[0] = 0;
        slot->off[1] = slot->off[0] + up(sizeof(int) * dw);
        slot->off[2] = slot->off[1] + up(csz * taps * dw);
        slot->off[3] = slot->off[2] + up(sizeof(int) * dh);
        slot->key[0] = -1;                        // invalid until the upload below has succeeded
        int rc = vkx_scratch_reserve(ctx, &slot->buf, slot->off[3] + up(csz * taps * dh));
        if (rc) return rc;
        const void *xsrc = fixed ? (const void *)xi.data() : (const void *)xc.data();
        const void *ysrc = fixed ? (const void *)yi.data() : (const void *)yc.data();
        // one copy out of the page-locked ring for the four tables (the ring keeps them alive: no synchronisation)
        const size_t total = slot->off[3] + up(csz * taps * dh);
        void *ring = nullptr;
        if ((rc = vkx_desc_ring_take(ctx, total, &ring))) return rc;
        unsigned char *stage = (unsigned char *)ring;
        memcpy(stage + slot->off[0], xo.data(), sizeof(int) * dw);
        memcpy(stage + slot->off[2], yo.data(), sizeof(int) * dh);
        memcpy(stage + slot->off[1], xsrc, csz * taps * dw);
        memcpy(stage + slot->off[3], ysrc, csz * taps * dh);
        VKX_HIP(hipMemcpyAsync(slot->buf.ptr, stage, total, hipMemcpyHostToDevice, ctx->stream));
        slot->yofs.swap(yo);
        std::copy(key, key + 6, slot->key);
    }
    slot->stamp = ++ctx->resize_clock;
    unsigned char *base = (unsigned char *)slot->buf.ptr;
    *xofs = (const int *)(base + slot->off[0]); *xcoef = base + slot->off[1];
    *yofs = (const int *)(base + slot->off[2]); *ycoef = base + slot->off[3];
    *yofs_host = &slot->yofs;
    return VKX_OK;
}

} // namespace

VKX_EXPORT int vkx_resize_cubic_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                                       uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0, "bad shape");
    VKX_REQUIRE(cn == 1 || cn == 3 || cn == 4, "1, 3 or 4 channels");
    const int *xofs, *yofs;
    const void *xa, *yb;
    const std::vector<int> *yh;
    int rc = resize_tables(ctx, 4, true, sh, sw, dh, dw, &xofs, &xa, &yofs, &yb, &yh);
    if (rc) return rc;
    dim3 grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    VKX_TIMED(ctx, "k_resize_cubic");
    if (const int rows = separable_rows(*yh, sh, dh, 4)) {
        launch_sep_u8<4>(ctx, src, sh, sw, cn, src_stride, dst, dh, dw, dst_stride, xofs, (const short *)xa, yofs, (const short *)yb, rows);
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    switch (cn) {
    case 1: k_resize_cubic_u8<1><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride, dst, dh, dw, dst_stride, xofs, (const short *)xa, yofs, (const short *)yb); break;
    case 3: k_resize_cubic_u8<3><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride, dst, dh, dw, dst_stride, xofs, (const short *)xa, yofs, (const short *)yb); break;
    default: k_resize_cubic_u8<4><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride, dst, dh, dw, dst_stride, xofs, (const short *)xa, yofs, (const short *)yb); break;
    }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_resize_cubic_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                                        float *dst, int dh, int dw, ptrdiff_t dst_stride_el)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0, "bad shape");
    const int *xofs, *yofs;
    const void *xc, *yc;
    const std::vector<int> *yh;
    int rc = resize_tables(ctx, 4, false, sh, sw, dh, dw, &xofs, &xc, &yofs, &yc, &yh);
    if (rc) return rc;
    dim3 grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    VKX_TIMED(ctx, "k_resize_cubic");
    if (separable_fits(*yh, sh, dh, 4)) {
        dim3 sgrid(vkx_blocks(dw, kSepTileW), vkx_blocks(dh, kSepTileH));
        k_resize_sep<float, float, float, 1, 4><<<sgrid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride_el, dst, dh, dw, dst_stride_el,
                                                                                 xofs, (const float *)xc, yofs, (const float *)yc);
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    k_resize_cubic_f32<<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride_el, dst, dh, dw, dst_stride_el, xofs,
                                                      (const float *)xc, yofs, (const float *)yc);
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_resize_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride, uint8_t *dst,
                                 int dh, int dw, ptrdiff_t dst_stride, int interpolation)
{
    if (interpolation == VKX_INTER_CUBIC) return vkx_resize_cubic_u8_dev(ctx, src, sh, sw, cn, src_stride, dst, dh, dw, dst_stride);
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0, "bad shape");
    VKX_REQUIRE(cn == 1 || cn == 3 || cn == 4, "1, 3 or 4 channels");
    dim3 grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    if (interpolation == VKX_INTER_NEAREST) {
        const double ifx = 1. / ((double)dw / sw), ify = 1. / ((double)dh / sh);
        VKX_TIMED(ctx, "k_resize_nearest");
        switch (cn) {
        case 1: k_resize_nearest_u8<1><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride, dst, dh, dw, dst_stride, ifx, ify); break;
        case 3: k_resize_nearest_u8<3><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride, dst, dh, dw, dst_stride, ifx, ify); break;
        default: k_resize_nearest_u8<4><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride, dst, dh, dw, dst_stride, ifx, ify); break;
        }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    if (interpolation == VKX_INTER_NEAREST_EXACT)
        return resize_nearest_exact(ctx, src, sh, sw, cn, src_stride, dst, dh, dw, dst_stride);
    if (interpolation == VKX_INTER_AREA) return resize_area<false>(ctx, src, sh, sw, cn, src_stride, dst, dh, dw, dst_stride);
    if (interpolation == VKX_INTER_LANCZOS4) {
        const int *xofs, *yofs;
        const void *xa, *yb;
        const std::vector<int> *yh;
        int rc = resize_tables(ctx, 8, true, sh, sw, dh, dw, &xofs, &xa, &yofs, &yb, &yh);
        if (rc) return rc;
        VKX_TIMED(ctx, "k_resize_lanczos4");
        if (const int rows = separable_rows(*yh, sh, dh, 8)) {
            launch_sep_u8<8>(ctx, src, sh, sw, cn, src_stride, dst, dh, dw, dst_stride, xofs, (const short *)xa, yofs, (const short *)yb, rows);
            VKX_LAUNCH_CHECK();
            return VKX_OK;
        }
        VKX_CN_SWITCH(cn, k_resize_lanczos4_u8, src, sh, sw, src_stride, dst, dh, dw, dst_stride, xofs, (const short *)xa, yofs, (const short *)yb)
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    if (interpolation == VKX_INTER_LINEAR_EXACT && !(sw == 2 * dw && sh == 2 * dh)) {
        std::vector<int> xo, xw, yo, yw;     // filled on a cache miss only
        std::vector<const void *> p;
        const std::vector<int> *meta;
        const int key[6] = {105, 0, sh, sw, dh, dw};
        int rc = cached_tables(ctx, key, [&](std::vector<std::pair<const void *, size_t>> *arrays, std::vector<int> *m) {
            int xmin_, xmax_, ymin_, ymax_;
            build_linear_exact_axis(sw, dw, &xo, &xw, &xmin_, &xmax_);
            build_linear_exact_axis(sh, dh, &yo, &yw, &ymin_, &ymax_);
            *arrays = {{xo.data(), sizeof(int) * dw}, {xw.data(), sizeof(int) * dw}, {yo.data(), sizeof(int) * dh},
                       {yw.data(), sizeof(int) * dh}};
            *m = {xmin_, xmax_, ymin_, ymax_};
        }, &p, &meta);
        if (rc) return rc;
        const int xmin = (*meta)[0], xmax = (*meta)[1], ymin = (*meta)[2], ymax = (*meta)[3];
        VKX_TIMED(ctx, "k_resize_linear_exact");
        VKX_CN_SWITCH(cn, k_resize_linear_exact_u8, src, src_stride, dst, dh, dw, dst_stride, (const int *)p[0], (const int *)p[1], (const int *)p[2], (const int *)p[3], xmin, xmax, ymin, ymax)
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    if (interpolation != VKX_INTER_LINEAR && interpolation != VKX_INTER_LINEAR_EXACT) {
        vkx_set_error("unknown interpolation code %d", interpolation);
        return VKX_ERR_UNSUPPORTED;
    }
    if (sw == 2 * dw && sh == 2 * dh) {
        VKX_TIMED(ctx, "k_resize_half");
        switch (cn) {
        case 1: k_resize_half_u8<1><<<grid, 256, 0, ctx->stream>>>(src, src_stride, dst, dh, dw, dst_stride); break;
        case 3: k_resize_half_u8<3><<<grid, 256, 0, ctx->stream>>>(src, src_stride, dst, dh, dw, dst_stride); break;
        default: k_resize_half_u8<4><<<grid, 256, 0, ctx->stream>>>(src, src_stride, dst, dh, dw, dst_stride); break;
        }
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    std::vector<int> xo, yo;
    std::vector<short> xa, yb;
    build_linear_axis(sw, dw, true, &xo, &xa);
    build_linear_axis(sh, dh, false, &yo, &yb);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o0 = 0, o1 = o0 + up(sizeof(int) * dw), o2 = o1 + up(sizeof(short) * 2 * dw), o3 = o2 + up(sizeof(int) * dh);
    const size_t total = o3 + up(sizeof(short) * 2 * dh);
    int rc = vkx_scratch_reserve(ctx, &ctx->misc, total);
    if (rc) return rc;
    unsigned char *base = (unsigned char *)ctx->misc.ptr;
    void *ring = nullptr;
    if ((rc = vkx_desc_ring_take(ctx, total, &ring))) return rc;      // one copy for the four tables, no synchronisation
    unsigned char *stage = (unsigned char *)ring;
    memcpy(stage + o0, xo.data(), sizeof(int) * dw);
    memcpy(stage + o1, xa.data(), sizeof(short) * 2 * dw);
    memcpy(stage + o2, yo.data(), sizeof(int) * dh);
    memcpy(stage + o3, yb.data(), sizeof(short) * 2 * dh);
    VKX_HIP(hipMemcpyAsync(base, stage, total, hipMemcpyHostToDevice, ctx->stream));
    const int *dxo = (const int *)(base + o0), *dyo = (const int *)(base + o2);
    const short *dxa = (const short *)(base + o1), *dyb = (const short *)(base + o3);
    VKX_TIMED(ctx, "k_resize_linear");
    switch (cn) {
    case 1: k_resize_linear_u8<1><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride, dst, dh, dw, dst_stride, dxo, dxa, dyo, dyb); break;
    case 3: k_resize_linear_u8<3><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride, dst, dh, dw, dst_stride, dxo, dxa, dyo, dyb); break;
    default: k_resize_linear_u8<4><<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride, dst, dh, dw, dst_stride, dxo, dxa, dyo, dyb); break;
    }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_resize_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el, float *dst, int dh,
                                  int dw, ptrdiff_t dst_stride_el, int interpolation)
{
    if (interpolation == VKX_INTER_CUBIC) return vkx_resize_cubic_f32_dev(ctx, src, sh, sw, src_stride_el, dst, dh, dw, dst_stride_el);
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0, "bad shape");
    dim3 grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    switch (interpolation) {
    case VKX_INTER_NEAREST: {
        const double ifx = 1. / ((double)dw / sw), ify = 1. / ((double)dh / sh);
        VKX_TIMED(ctx, "k_resize_nearest");
        k_resize_nearest_u8<4><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)src, sh, sw, src_stride_el * 4, (uint8_t *)dst, dh, dw,
                                                              dst_stride_el * 4, ifx, ify);
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    case VKX_INTER_NEAREST_EXACT:
        return resize_nearest_exact(ctx, src, sh, sw, 4, src_stride_el * 4, dst, dh, dw, dst_stride_el * 4);
    case VKX_INTER_AREA:
        return resize_area<true>(ctx, src, sh, sw, 1, src_stride_el, dst, dh, dw, dst_stride_el);
    case VKX_INTER_LINEAR:
    case VKX_INTER_LINEAR_EXACT: {     // no bit-exact float32 path in cv.resize: INTER_LINEAR_EXACT falls back to INTER_LINEAR
        if (sw == 2 * dw && sh == 2 * dh) return resize_area<true>(ctx, src, sh, sw, 1, src_stride_el, dst, dh, dw, dst_stride_el);
        VKX_TIMED(ctx, "k_resize_linear");
        k_resize_linear_f32<<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride_el, dst, dh, dw, dst_stride_el,
                                                           1. / ((double)dw / sw), 1. / ((double)dh / sh));
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    case VKX_INTER_LANCZOS4: {
        const int *xofs, *yofs;
        const void *xc, *yc;
        const std::vector<int> *yh;
        int rc = resize_tables(ctx, 8, false, sh, sw, dh, dw, &xofs, &xc, &yofs, &yc, &yh);
        if (rc) return rc;
        VKX_TIMED(ctx, "k_resize_lanczos4");
        if (separable_fits(*yh, sh, dh, 8)) {
            dim3 sgrid(vkx_blocks(dw, kSepTileW), vkx_blocks(dh, kSepTileH));
            k_resize_sep<float, float, float, 1, 8><<<sgrid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride_el, dst, dh, dw, dst_stride_el,
                                                                                     xofs, (const float *)xc, yofs, (const float *)yc);
            VKX_LAUNCH_CHECK();
            return VKX_OK;
        }
        k_resize_lanczos4_f32<<<grid, 256, 0, ctx->stream>>>(src, sh, sw, src_stride_el, dst, dh, dw, dst_stride_el, xofs,
                                                             (const float *)xc, yofs, (const float *)yc);
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    default:
        vkx_set_error("unknown interpolation code %d", interpolation);
        return VKX_ERR_UNSUPPORTED;
    }
}

VKX_EXPORT int vkx_zoom_in_blur_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                       const int32_t *sizes_hw_host, int n_sizes, double alpha, uint8_t *dst,
                                       ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst && (n_sizes == 0 || sizes_hw_host), "NULL argument");
    VKX_REQUIRE(h > 0 && w > 0 && n_sizes >= 0, "bad shape");
    VKX_REQUIRE(cn == 1 || cn == 3 || cn == 4, "1, 3 or 4 channels");
    VKX_REQUIRE(n_sizes < 256, "too many zoom steps for a uint16 accumulator");
    size_t max_plane = 0;
    for (int i = 0; i < n_sizes; i++) {
        const int rh = sizes_hw_host[2 * i], rw = sizes_hw_host[2 * i + 1];
        VKX_REQUIRE(rh >= h && rw >= w, "zoom steps must not shrink the image");
        max_plane = std::max(max_plane, (size_t)rh * rw * cn);
    }
    const int wc = w * cn;
    int rc = vkx_scratch_reserve(ctx, &ctx->chain[0], max_plane + 256);
    if (rc) return rc;
    rc = vkx_scratch_reserve(ctx, &ctx->chain[1], (size_t)h * wc * sizeof(uint16_t));
    if (rc) return rc;
    uint8_t *big = (uint8_t *)ctx->chain[0].ptr;
    uint16_t *acc = (uint16_t *)ctx->chain[1].ptr;
    dim3 grid(vkx_blocks(wc, 64), vkx_blocks(h, 4));
    { VKX_TIMED(ctx, "k_accumulate_crop"); k_accumulate_crop<<<grid, 256, 0, ctx->stream>>>(src, src_stride, 0, 0, acc, h, wc, cn, 1); }
    VKX_LAUNCH_CHECK();
    for (int i = 0; i < n_sizes; i++) {
        const int rh = sizes_hw_host[2 * i], rw = sizes_hw_host[2 * i + 1];
        rc = vkx_resize_cubic_u8_dev(ctx, src, h, w, cn, src_stride, big, rh, rw, (ptrdiff_t)rw * cn);
        if (rc) return rc;
        { VKX_TIMED(ctx, "k_accumulate_crop"); k_accumulate_crop<<<grid, 256, 0, ctx->stream>>>(big, (ptrdiff_t)rw * cn, (rh - h) / 2, (rw - w) / 2, acc, h, wc, cn, 0); }
        VKX_LAUNCH_CHECK();
    }
    { VKX_TIMED(ctx, "k_zoom_finish"); k_zoom_finish<<<grid, 256, 0, ctx->stream>>>(src, src_stride, acc, h, wc, n_sizes + 1, 1 - alpha, alpha, dst, dst_stride); }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

