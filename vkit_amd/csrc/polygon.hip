// cv.fillPoly(mask, [pts], 1) for one polygon of any vertex count on gfx950
// (reference: PolygonInternals.np_mask, element/polygon.py:70-77; used by Polygon.fill_* and by the active mask
// of image-grid distortions, grid_rendering/interface.py:177-192).
//
//   k_poly_outline  one lane per (edge, major step): pixel of the 8-connected Bresenham line walked from the
//                   edge's left end (cv::LineIterator), closed form of the error recurrence.
//   k_poly_spans    one workgroup per scanline: lanes collect the 16.16 fixed-point crossings of the half-open
//                   edges (y0 <= y < y1) into LDS, sort them, and fill the even-odd spans
//                   [ceil(xa), floor(xb)] cooperatively.
// Vertices are host data (a few to a few thousand points); the edge table is built on the host and staged.
#include "vkx_internal.h"
#include "vkx_cell.h"

#include <algorithm>

namespace {

struct PolyEdge {
    int xa, ya, xb, yb;        // contour order
    int lx, ly, dmaj, dmin;    // Bresenham from the left end
    int sy, ymajor;
    int step_base;             // prefix of (dmaj + 1) over the edges
    int y0, y1;                // scanline range of the edge (y0 == y1: horizontal, not in the edge table)
    long long x0_fix, dx_fix;  // 16.16 x at y0, dx per scanline
};

__global__ void __launch_bounds__(256) k_poly_outline(const PolyEdge *__restrict__ edges, int nedges, int total_steps,
                                                      uint8_t *__restrict__ mask, int h, int w, ptrdiff_t stride)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total_steps) return;
    int lo = 0, hi = nedges - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (edges[mid].step_base <= t) lo = mid; else hi = mid - 1;
    }
    const PolyEdge e = edges[lo];
    const int k = t - e.step_base;
    const int m = vkc::bres_minor(k, e.dmaj, e.dmin);
    const int x = e.ymajor ? e.lx + m : e.lx + k;
    const int y = e.ymajor ? e.ly + e.sy * k : e.ly + e.sy * m;
    if ((unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h) mask[(ptrdiff_t)y * stride + x] = 1;
}

constexpr int kMaxCross = 512;

__global__ void __launch_bounds__(256) k_poly_spans(const PolyEdge *__restrict__ edges, int nedges, int ymin,
                                                    uint8_t *__restrict__ mask, int h, int w, ptrdiff_t stride,
                                                    int *__restrict__ overflow)
{
    __shared__ long long xs[kMaxCross];
    __shared__ int count;
    const int y = ymin + blockIdx.x;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nedges; i += 256) {
        const PolyEdge &e = edges[i];
        if (e.y0 != e.y1 && e.y0 <= y && y < e.y1) {
            const int slot = atomicAdd(&count, 1);
            if (slot < kMaxCross) xs[slot] = e.x0_fix + (long long)(y - e.y0) * e.dx_fix;
        }
    }
    __syncthreads();
    int n = count;
    if (n > kMaxCross) {
        if (threadIdx.x == 0) atomicExch(overflow, 1);
        n = kMaxCross;
    }
    if (threadIdx.x == 0) {
        for (int a = 1; a < n; a++) {
            const long long v = xs[a];
            int b = a - 1;
            while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; b--; }
            xs[b + 1] = v;
        }
    }
    __syncthreads();
    if ((unsigned)y >= (unsigned)h) return;
    uint8_t *row = mask + (ptrdiff_t)y * stride;
    for (int a = 0; a + 1 < n; a += 2) {
        long long x1 = (xs[a] + 65535) >> 16, x2 = xs[a + 1] >> 16;
        if (x1 < 0) x1 = 0;
        if (x2 >= w) x2 = w - 1;
        for (long long x = x1 + threadIdx.x; x <= x2; x += 256) row[x] = 1;
    }
}

} // namespace

VKX_EXPORT int vkx_fill_poly_mask_u8_dev(vkx_ctx *ctx, const int32_t *pts_host, int npts, uint8_t *mask, int h, int w,
                                         ptrdiff_t stride)
{
    VKX_REQUIRE(ctx && pts_host && mask, "NULL argument");
    VKX_REQUIRE(npts > 0 && h > 0 && w > 0, "bad shape");
    std::vector<PolyEdge> edges((size_t)npts);
    long long steps = 0;
    int ymin = INT_MAX, ymax = INT_MIN;
    for (int i = 0; i < npts; i++) {
        const int a = (i + npts - 1) % npts;
        PolyEdge &e = edges[i];
        e.xa = pts_host[2 * a]; e.ya = pts_host[2 * a + 1];
        e.xb = pts_host[2 * i]; e.yb = pts_host[2 * i + 1];
        VKX_REQUIRE(e.xb >= 0 && e.xb < w && e.yb >= 0 && e.yb < h, "polygon vertex outside the mask");
        int lx = e.xa, ly = e.ya, rx = e.xb, ry = e.yb;
        if (rx < lx) { std::swap(lx, rx); std::swap(ly, ry); }
        const int dx = rx - lx, dy = ry - ly, ady = dy < 0 ? -dy : dy;
        e.lx = lx; e.ly = ly; e.sy = dy < 0 ? -1 : 1;
        e.ymajor = ady > dx;
        e.dmaj = e.ymajor ? ady : dx;
        e.dmin = e.ymajor ? dx : ady;
        e.step_base = (int)steps;
        steps += e.dmaj + 1;
        VKX_REQUIRE(steps < 0x7fffffff, "polygon outline too long");
        e.y0 = std::min(e.ya, e.yb); e.y1 = std::max(e.ya, e.yb);
        if (e.ya != e.yb) {
            const long long xa = (long long)e.xa << 16, xb = (long long)e.xb << 16;
            e.dx_fix = (xb - xa) / (long long)(e.yb - e.ya);
            e.x0_fix = e.ya < e.yb ? xa : xb;
            ymin = std::min(ymin, e.y0); ymax = std::max(ymax, e.y1);
        } else {
            e.dx_fix = 0; e.x0_fix = 0;
        }
    }
    const size_t ebytes = sizeof(PolyEdge) * edges.size();
    int rc = vkx_scratch_reserve(ctx, &ctx->misc, ebytes + 256);
    if (rc) return rc;
    unsigned char *misc = (unsigned char *)ctx->misc.ptr;
    int *overflow = (int *)misc;
    PolyEdge *d_edges = (PolyEdge *)(misc + 256);
    VKX_HIP(hipMemsetAsync(overflow, 0, sizeof(int), ctx->stream));
    VKX_HIP(hipMemcpyAsync(d_edges, edges.data(), ebytes, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream)); // `edges` lives on this frame
    { VKX_TIMED(ctx, "k_poly_outline"); k_poly_outline<<<vkx_blocks((size_t)steps, 256), 256, 0, ctx->stream>>>(d_edges, npts, (int)steps, mask, h, w, stride); }
    VKX_LAUNCH_CHECK();
    if (ymin < ymax) {
        { VKX_TIMED(ctx, "k_poly_spans"); k_poly_spans<<<ymax - ymin, 256, 0, ctx->stream>>>(d_edges, npts, ymin, mask, h, w, stride, overflow); }
        VKX_LAUNCH_CHECK();
        int flag = 0;
        VKX_HIP(hipMemcpyAsync(&flag, overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        VKX_HIP(hipStreamSynchronize(ctx->stream));
        if (flag) {
            vkx_set_error("polygon has more than %d edge crossings on one scanline", kMaxCross);
            return VKX_ERR_UNSUPPORTED;
        }
    }
    return VKX_OK;
}

// Host-memory variant: zero-initialised mask of shape [h, w] with the polygon set to 1.
VKX_EXPORT int vkx_fill_poly_mask_u8(vkx_ctx *ctx, const int32_t *pts_host, int npts, uint8_t *mask, int h, int w,
                                     ptrdiff_t stride)
{
    VKX_REQUIRE(ctx && pts_host && mask, "NULL argument");
    VKX_REQUIRE(npts > 0 && h > 0 && w > 0, "bad shape");
    const size_t bytes = (size_t)h * w;
    int rc = vkx_scratch_reserve(ctx, &ctx->stage[1], bytes);
    if (rc) return rc;
    uint8_t *d = (uint8_t *)ctx->stage[1].ptr;
    VKX_HIP(hipMemsetAsync(d, 0, bytes, ctx->stream));
    rc = vkx_fill_poly_mask_u8_dev(ctx, pts_host, npts, d, h, w, w);
    if (rc) return rc;
    VKX_HIP(hipMemcpy2DAsync(mask, (size_t)stride, d, (size_t)w, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}
