// ellipse_streak's mask: the outlines `cv.ellipse(mask, center, axes, 0, 0, 360, 1, thickness)` draws for a set of
// concentric boxes (reference: vkit/mechanism/distortion/photometric/streak.py:296-326; OpenCV 4.5.x drawing.cpp:
// ellipse -> EllipseEx -> ellipse2Poly -> PolyLine -> ThickLine -> Line2 / FillConvexPoly / Circle).
//
// Split the way the work is shaped: the polygonal arc of an ellipse is at most 73 vertices of double arithmetic -- the
// host walks it (a few microseconds for a whole page) and hands the device one record per segment; the pixels are the
// device's: one wavefront per segment, the lanes sharing the DDA steps of a thin line, or the scanlines of the offset
// quad and its end caps of a thick one.  Every write stores the constant 1, so segments and lanes never need ordering.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

#include "vkx_internal.h"

namespace {

constexpr int kShift = 16;
constexpr int64_t kOne = 1 << kShift, kHalf = kOne >> 1;

struct Segment {
    int32_t x0, y0, x1, y1;   // 16.16 end points (|v| < 2^31: centre and axes are below 2^14 pixels)
    int32_t caps;             // bit 0: cap at (x0, y0), bit 1: cap at (x1, y1)  (PolyLine's flags)
};

struct P2 {
    int64_t x, y;
};

__device__ inline void put(uint8_t *mask, ptrdiff_t stride, int h, int w, int64_t x, int64_t y)
{
    if (x >= 0 && x < w && y >= 0 && y < h) mask[y * stride + x] = 1;
}

// clipLine on 64-bit points: Cohen-Sutherland style, intersections through double products truncated toward zero
__device__ bool clip_segment(int64_t width, int64_t height, P2 &a, P2 &b)
{
    const int64_t right = width - 1, bottom = height - 1;
    auto code = [&](const P2 &p) { return (int)(p.x < 0) + (int)(p.x > right) * 2 + (int)(p.y < 0) * 4 + (int)(p.y > bottom) * 8; };
    int ca = code(a), cb = code(b);
    if ((ca & cb) == 0 && (ca | cb) != 0) {
        if (ca & 12) {
            const int64_t edge = ca < 8 ? 0 : bottom;
            a.x += (int64_t)((double)(edge - a.y) * (double)(b.x - a.x) / (double)(b.y - a.y));
            a.y = edge;
            ca = (int)(a.x < 0) + (int)(a.x > right) * 2;
        }
        if (cb & 12) {
            const int64_t edge = cb < 8 ? 0 : bottom;
            b.x += (int64_t)((double)(edge - b.y) * (double)(b.x - a.x) / (double)(b.y - a.y));
            b.y = edge;
            cb = (int)(b.x < 0) + (int)(b.x > right) * 2;
        }
        if ((ca & cb) == 0 && (ca | cb) != 0) {
            if (ca) {
                const int64_t edge = ca == 1 ? 0 : right;
                a.y += (int64_t)((double)(edge - a.x) * (double)(b.y - a.y) / (double)(b.x - a.x));
                a.x = edge;
                ca = 0;
            }
            if (cb) {
                const int64_t edge = cb == 1 ? 0 : right;
                b.y += (int64_t)((double)(edge - b.x) * (double)(b.y - a.y) / (double)(b.x - a.x));
                b.x = edge;
                cb = 0;
            }
        }
    }
    return (ca | cb) == 0;
}

// Line2: the 16.16 DDA.  Step k of the major axis is a closed form of k, so the lanes of the wavefront take the steps.
__device__ void thin_line(uint8_t *mask, ptrdiff_t stride, int h, int w, P2 a, P2 b, int lane)
{
    if (!clip_segment((int64_t)w << kShift, (int64_t)h << kShift, a, b)) return;
    const int64_t dx = b.x - a.x, dy = b.y - a.y;
    const int64_t adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    const bool xmajor = adx > ady;
    // walk from the end with the smaller major coordinate
    if (xmajor ? dx < 0 : dy < 0) { const P2 t = a; a = b; b = t; }
    const int64_t major = xmajor ? adx : ady;
    const int64_t minor_signed = xmajor ? b.y - a.y : b.x - a.x;
    const int64_t step = (minor_signed * kOne) / (major | 1);            // C division (toward zero)
    const int count = (int)(((xmajor ? b.x - a.x : b.y - a.y)) >> kShift);
    if (lane == 0) put(mask, stride, h, w, (b.x + kHalf) >> kShift, (b.y + kHalf) >> kShift);
    const int64_t m0 = ((xmajor ? a.x : a.y) + kHalf) >> kShift, n0 = (xmajor ? a.y : a.x) + kHalf;
    for (int k = lane; k <= count; k += 64) {
        const int64_t m = m0 + k, n = (n0 + step * k) >> kShift;
        put(mask, stride, h, w, xmajor ? m : n, xmajor ? n : m);
    }
}

// FillConvexPoly of a quad in 16.16: outline through thin_line, then the scan conversion between the two edge chains that
// start at the top vertex.  The chains advance row by row (each new edge restarts from its exact start x), so one lane
// walks them and broadcasts the row's extent; the other lanes share the pixels of the row.
__device__ void fill_quad(uint8_t *mask, ptrdiff_t stride, int h, int w, const P2 (&v)[4], int lane)
{
    int top = 0;
    int64_t xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
    for (int i = 0; i < 4; i++) {
        thin_line(mask, stride, h, w, v[(i + 3) & 3], v[i], lane);
        if (v[i].y < ymin) { ymin = v[i].y; top = i; }
        ymax = max(ymax, v[i].y);
        xmax = max(xmax, v[i].x);
        xmin = min(xmin, v[i].x);
    }
    xmin = (xmin + kHalf) >> kShift; xmax = (xmax + kHalf) >> kShift;
    ymin = (ymin + kHalf) >> kShift; ymax = (ymax + kHalf) >> kShift;
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= w || (int)ymin >= h) return;
    ymax = min(ymax, (int64_t)h - 1);
    struct Chain { int at, dir, until; int64_t x, dx; } chain[2];
    int y = (int)ymin, budget = 4;
    for (int c = 0; c < 2; c++) chain[c] = {top, c == 0 ? 1 : 3, y, -kOne, 0};
    do {
        for (int c = 0; c < 2; c++) {
            if (y < chain[c].until) continue;
            int from = chain[c].at, to = (from + chain[c].dir) & 3;
            while (budget-- > 0) {
                const int ty = (int)((v[to].y + kHalf) >> kShift);
                if (ty > y) {
                    chain[c].until = ty;
                    chain[c].dx = ((v[to].x - v[from].x) * 2 + (ty - y)) / (2 * (ty - y));
                    chain[c].x = v[from].x;
                    chain[c].at = to;
                    break;
                }
                from = to;
                to = (to + chain[c].dir) & 3;
            }
        }
        if (budget < 0) break;
        if (y >= 0) {
            const int64_t xl = min(chain[0].x, chain[1].x), xr = max(chain[0].x, chain[1].x);
            int xa = (int)((xl + kHalf) >> kShift), xb = (int)((xr + kHalf) >> kShift);
            if (xb >= 0 && xa < w) {
                xa = max(xa, 0);
                xb = min(xb, w - 1);
                for (int x = xa + lane; x <= xb; x += 64) mask[(ptrdiff_t)y * stride + x] = 1;
            }
        }
        chain[0].x += chain[0].dx;
        chain[1].x += chain[1].dx;
    } while (++y <= (int)ymax);
}

// Circle(..., fill): the midpoint circle's row pairs; radii here are 1 .. thickness / 2 + 1, lane 0 writes them
__device__ void disc(uint8_t *mask, ptrdiff_t stride, int h, int w, int cx, int cy, int radius)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    auto row = [&](int y, int xa, int xb) {
        if ((unsigned)y >= (unsigned)h) return;
        xa = max(xa, 0);
        xb = min(xb, w - 1);
        for (int x = xa; x <= xb; x++) mask[(ptrdiff_t)y * stride + x] = 1;
    };
    while (dx >= dy) {
        if (cx - dx < w && cx + dx >= 0 && cy - dx < h && cy + dx >= 0) {
            row(cy - dy, cx - dx, cx + dx);
            row(cy + dy, cx - dx, cx + dx);
            if (cx - dy < w && cx + dy >= 0) {
                row(cy - dx, cx - dy, cx + dy);
                row(cy + dx, cx - dy, cx + dy);
            }
        }
        dy++;
        err += plus;
        plus += 2;
        const int m = (err <= 0) - 1;
        err -= minus & m;
        dx += m;
        minus -= m & 2;
    }
}

__global__ __launch_bounds__(256) void k_ellipse_segments(const Segment *__restrict__ segs, int n, int thickness,
                                                          uint8_t *__restrict__ mask, ptrdiff_t stride, int h, int w)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n) return;
    const Segment g = segs[s];
    const P2 p0 = {g.x0, g.y0}, p1 = {g.x1, g.y1};
    if (thickness <= 1) {
        thin_line(mask, stride, h, w, p0, p1, lane);
        return;
    }
    // ThickLine: the segment offset by half the thickness (+ half a pixel for odd ones) to both sides
    const double dx = (double)(p0.x - p1.x) * (1.0 / kOne), dy = (double)(p1.y - p0.y) * (1.0 / kOne);
    double r = dx * dx + dy * dy;
    const int half_thick = thickness << (kShift - 1);
    if (fabs(r) > 2.220446049250313e-16) {
        r = ((double)half_thick + (double)((thickness & 1) * (int)kOne) * 0.5) / sqrt(r);
        const int64_t ox = (int64_t)rint(dy * r), oy = (int64_t)rint(dx * r);
        const P2 quad[4] = {{p0.x + ox, p0.y + oy}, {p0.x - ox, p0.y - oy}, {p1.x - ox, p1.y - oy}, {p1.x + ox, p1.y + oy}};
        fill_quad(mask, stride, h, w, quad, lane);
    }
    if (lane < 2 && (g.caps >> lane & 1)) {
        const P2 c = lane == 0 ? p0 : p1;
        disc(mask, stride, h, w, (int)((c.x + kHalf) >> kShift), (int)((c.y + kHalf) >> kShift), (half_thick + (int)kHalf) >> kShift);
    }
}

// OpenCV's SinTable: sin of whole degrees 0 .. 450 as float literals with seven decimals
const float *sin_table()
{
    static float table[451];
    static bool ready = false;
    if (!ready) {
        for (int d = 0; d <= 450; d++) {
            char text[32];
            snprintf(text, sizeof text, "%.7f", std::sin((double)d * 3.14159265358979323846 / 180.0));
            table[d] = strtof(text, nullptr);
        }
        ready = true;
    }
    return table;
}

int64_t round_vertex(double v)
{
    // EllipseEx rounds in two steps so that cvRound never sees more than 32 bits
    int64_t q = (int64_t)(int)std::nearbyint(v / (double)kOne) * kOne;
    return q + (int64_t)(int)std::nearbyint(v - (double)q);
}

// The open polyline cv.ellipse draws for one axis-aligned, full ellipse, appended to `out` as segments
void ellipse_segments(int cx, int cy, int ax, int ay, std::vector<Segment> &out)
{
    const float *sn = sin_table();
    const int64_t ccx = (int64_t)cx * kOne, ccy = (int64_t)cy * kOne;
    const int64_t aw = std::llabs((int64_t)ax * kOne), ah = std::llabs((int64_t)ay * kOne);
    const int size = (int)((std::max(aw, ah) + kHalf) >> kShift);
    const int step = size < 3 ? 90 : size < 10 ? 30 : size < 15 ? 18 : 5;
    const float ca = sn[450], sa = sn[0];           // the rotation by `angle` = 0 stays in the formula: it is float x double
    std::vector<P2> poly;
    for (int i = 0; i < 360 + step; i += step) {
        const int deg = std::min(i, 360);
        const double x = (double)aw * sn[450 - deg], y = (double)ah * sn[deg];
        const P2 p = {round_vertex((double)ccx + x * ca - y * sa), round_vertex((double)ccy + x * sa + y * ca)};
        if (poly.empty() || p.x != poly.back().x || p.y != poly.back().y) poly.push_back(p);
    }
    if (poly.size() == 1) poly.assign(2, P2{ccx, ccy});
    for (size_t i = 1; i < poly.size(); i++)
        out.push_back({(int32_t)poly[i - 1].x, (int32_t)poly[i - 1].y, (int32_t)poly[i].x, (int32_t)poly[i].y, i == 1 ? 3 : 2});
}

} // namespace

VKX_EXPORT int vkx_ellipse_mask_u8_dev(vkx_ctx *ctx, uint8_t *mask, ptrdiff_t stride, int h, int w, int cx, int cy,
                                       const int32_t *axes_host, int n_ellipses, int thickness)
{
    VKX_REQUIRE(ctx && mask && (axes_host || n_ellipses == 0), "NULL argument");
    VKX_REQUIRE(h > 0 && w > 0 && n_ellipses >= 0, "bad shape");
    VKX_REQUIRE(thickness >= 1 && thickness <= 32767, "thickness must be in 1 .. 32767 (cv.ellipse's MAX_THICKNESS; filled ellipses are not on the path)");
    VKX_REQUIRE(cx > -16384 && cx < 16384 && cy > -16384 && cy < 16384, "centre beyond 2^14 pixels");
    std::vector<Segment> segs;
    for (int e = 0; e < n_ellipses; e++) {
        const int ax = axes_host[2 * e], ay = axes_host[2 * e + 1];
        VKX_REQUIRE(ax >= 0 && ay >= 0 && ax < 16384 && ay < 16384, "axes must be in 0 .. 2^14 pixels");
        ellipse_segments(cx, cy, ax, ay, segs);
    }
    if (segs.empty()) return VKX_OK;
    vkx_device_guard guard(ctx);
    const size_t bytes = segs.size() * sizeof(Segment);
    void *staged = nullptr;
    int rc = vkx_desc_ring_take(ctx, bytes, &staged);
    if (rc) return rc;
    memcpy(staged, segs.data(), bytes);
    rc = vkx_scratch_reserve(ctx, &ctx->misc, bytes);
    if (rc) return rc;
    VKX_HIP(hipMemcpyAsync(ctx->misc.ptr, staged, bytes, hipMemcpyHostToDevice, ctx->stream));
    {
        VKX_TIMED(ctx, "k_ellipse_segments");
        k_ellipse_segments<<<vkx_blocks(segs.size(), 4), 256, 0, ctx->stream>>>((const Segment *)ctx->misc.ptr, (int)segs.size(),
                                                                               thickness, mask, stride, h, w);
    }
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_ellipse_mask_u8(vkx_ctx *ctx, uint8_t *mask, ptrdiff_t stride, int h, int w, int cx, int cy,
                                   const int32_t *axes_host, int n_ellipses, int thickness)
{
    VKX_REQUIRE(ctx && mask, "NULL argument");
    VKX_REQUIRE(h > 0 && w > 0 && stride >= w, "bad shape");
    vkx_device_guard guard(ctx);
    const size_t bytes = (size_t)h * w;
    int rc = vkx_scratch_reserve(ctx, &ctx->stage[1], bytes);
    if (rc) return rc;
    uint8_t *d = (uint8_t *)ctx->stage[1].ptr;
    // the caller's mask is drawn onto, as cv.ellipse does
    VKX_HIP(vkx_copy_plane(d, (size_t)w, mask, (size_t)stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, ctx->stream));
    rc = vkx_ellipse_mask_u8_dev(ctx, d, w, h, w, cx, cy, axes_host, n_ellipses, thickness);
    if (rc) return rc;
    VKX_HIP(vkx_copy_plane(mask, (size_t)stride, d, (size_t)w, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}

// ellipse_streak_image in one call: mask of all outlines, then Mask.fill_image(image, color, alpha) (element/mask.py:601-612
// -> fill_np_array, element/opt.py:118-209) as one composite layer.  In place on a device image.
VKX_EXPORT int vkx_ellipse_streak_u8_dev(vkx_ctx *ctx, uint8_t *img, int h, int w, int cn, ptrdiff_t stride, int cx, int cy,
                                         const int32_t *axes_host, int n_ellipses, int thickness, const uint8_t color[4],
                                         double alpha)
{
    VKX_REQUIRE(ctx && img && color, "NULL argument");
    VKX_REQUIRE(h > 0 && w > 0 && (cn == 1 || cn == 3 || cn == 4), "bad shape");
    VKX_REQUIRE(alpha >= 0.0 && alpha <= 1.0, "alpha must be in [0, 1]");
    vkx_device_guard guard(ctx);
    const size_t bytes = (size_t)h * w;
    int rc = vkx_scratch_reserve(ctx, &ctx->owner, bytes);
    if (rc) return rc;
    uint8_t *mask = (uint8_t *)ctx->owner.ptr;
    VKX_HIP(hipMemsetAsync(mask, 0, bytes, ctx->stream));
    rc = vkx_ellipse_mask_u8_dev(ctx, mask, w, h, w, cx, cy, axes_host, n_ellipses, thickness);
    if (rc) return rc;
    vkx_layer layer = {};
    layer.up = 0; layer.left = 0; layer.height = h; layer.width = w;
    layer.mask = mask; layer.mask_stride = w;
    layer.alpha_scalar = alpha;
    for (int c = 0; c < 4; c++) layer.value_const[c] = color[c];
    layer.mode = VKX_FILL_PLAIN;
    return vkx_fill_u8_dev(ctx, img, h, w, cn, stride, &layer, 1);
}

VKX_EXPORT int vkx_ellipse_streak_u8(vkx_ctx *ctx, uint8_t *img, int h, int w, int cn, ptrdiff_t stride, int cx, int cy,
                                     const int32_t *axes_host, int n_ellipses, int thickness, const uint8_t color[4],
                                     double alpha)
{
    VKX_REQUIRE(ctx && img, "NULL argument");
    VKX_REQUIRE(h > 0 && w > 0 && cn >= 1 && stride >= (ptrdiff_t)w * cn, "bad shape");
    vkx_device_guard guard(ctx);
    const size_t row = (size_t)w * cn, bytes = row * h;
    int rc = vkx_scratch_reserve(ctx, &ctx->stage[0], bytes);
    if (rc) return rc;
    uint8_t *d = (uint8_t *)ctx->stage[0].ptr;
    if ((size_t)stride == row) VKX_HIP(hipMemcpyAsync(d, img, bytes, hipMemcpyHostToDevice, ctx->stream));
    else VKX_HIP(vkx_copy_plane(d, row, img, (size_t)stride, row, (size_t)h, hipMemcpyHostToDevice, ctx->stream));
    rc = vkx_ellipse_streak_u8_dev(ctx, d, h, w, cn, (ptrdiff_t)row, cx, cy, axes_host, n_ellipses, thickness, color, alpha);
    if (rc) return rc;
    if ((size_t)stride == row) VKX_HIP(hipMemcpyAsync(img, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    else VKX_HIP(vkx_copy_plane(img, (size_t)stride, d, row, row, (size_t)h, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}
