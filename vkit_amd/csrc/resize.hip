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
#include <cmath>

namespace {

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
            const int r = (int)std::nearbyint((double)(t->coef[(size_t)d * 4 + k] * 2048.f)); // cvRound: ties to even
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
            acc[c] += hsum * (unsigned)b;
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
            const int r = (int)std::nearbyint((double)(c[k] * 2048.f));
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

// Stages the four tables in ctx->misc; returns device pointers.
int stage_tables(vkx_ctx *ctx, int sh, int sw, int dh, int dw, bool fixed, const int **xofs, const void **xcoef,
                 const int **yofs, const void **ycoef)
{
    AxisTable tx, ty;
    build_axis(sw, dw, &tx);
    build_axis(sh, dh, &ty);
    const size_t csz = fixed ? sizeof(short) : sizeof(float);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o0 = 0, o1 = o0 + up(sizeof(int) * dw), o2 = o1 + up(csz * 4 * dw), o3 = o2 + up(sizeof(int) * dh);
    const size_t total = o3 + up(csz * 4 * dh);
    int rc = vkx_scratch_reserve(ctx, &ctx->misc, total);
    if (rc) return rc;
    unsigned char *base = (unsigned char *)ctx->misc.ptr;
    VKX_HIP(hipMemcpyAsync(base + o0, tx.ofs.data(), sizeof(int) * dw, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + o2, ty.ofs.data(), sizeof(int) * dh, hipMemcpyHostToDevice, ctx->stream));
    if (fixed) {
        VKX_HIP(hipMemcpyAsync(base + o1, tx.icoef.data(), csz * 4 * dw, hipMemcpyHostToDevice, ctx->stream));
        VKX_HIP(hipMemcpyAsync(base + o3, ty.icoef.data(), csz * 4 * dh, hipMemcpyHostToDevice, ctx->stream));
    } else {
        VKX_HIP(hipMemcpyAsync(base + o1, tx.coef.data(), csz * 4 * dw, hipMemcpyHostToDevice, ctx->stream));
        VKX_HIP(hipMemcpyAsync(base + o3, ty.coef.data(), csz * 4 * dh, hipMemcpyHostToDevice, ctx->stream));
    }
    VKX_HIP(hipStreamSynchronize(ctx->stream)); // the tables live on this frame
    *xofs = (const int *)(base + o0); *xcoef = base + o1; *yofs = (const int *)(base + o2); *ycoef = base + o3;
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
    int rc = stage_tables(ctx, sh, sw, dh, dw, true, &xofs, &xa, &yofs, &yb);
    if (rc) return rc;
    dim3 grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    VKX_TIMED(ctx, "k_resize_cubic");
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
    int rc = stage_tables(ctx, sh, sw, dh, dw, false, &xofs, &xc, &yofs, &yc);
    if (rc) return rc;
    dim3 grid(vkx_blocks(dw, 64), vkx_blocks(dh, 4));
    VKX_TIMED(ctx, "k_resize_cubic");
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
    if (interpolation != VKX_INTER_LINEAR) {
        vkx_set_error("interpolation %d is not implemented (NEAREST, LINEAR, CUBIC are)", interpolation);
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
    int rc = vkx_scratch_reserve(ctx, &ctx->misc, o3 + up(sizeof(short) * 2 * dh));
    if (rc) return rc;
    unsigned char *base = (unsigned char *)ctx->misc.ptr;
    VKX_HIP(hipMemcpyAsync(base + o0, xo.data(), sizeof(int) * dw, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + o1, xa.data(), sizeof(short) * 2 * dw, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + o2, yo.data(), sizeof(int) * dh, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(base + o3, yb.data(), sizeof(short) * 2 * dh, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
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

